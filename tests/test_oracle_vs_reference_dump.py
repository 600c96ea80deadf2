"""Oracle vs the REAL reference arithmetic (brisk submodule of an OKVIS2 checkout).

Skipped unless a dump produced by tools/ref_compare (see its CMakeLists.txt for the four commands)
is present: OKVFE_REF_DUMP=<dir>, or tests/golden/ref_dump/.  The dump cannot be produced in the
build container (no brisk, no OpenCV); this test is the hook that turns "parity unpinned" into a
one-command check wherever an OKVIS2 checkout with submodules exists.  Every difference it reports
is a difference between this repo's restatement (oracle/orc_detect.c, orc_describe.c) and BRISK2."""
import os

import numpy as np
import pytest

DUMP = os.environ.get("OKVFE_REF_DUMP") or os.path.join(os.path.dirname(__file__), "golden", "ref_dump")
needs_dump = pytest.mark.skipif(not os.path.exists(os.path.join(DUMP, "dump_done.txt")),
                                reason="no reference dump (tools/ref_compare needs an OKVIS2 checkout "
                                       "with the brisk submodule + OpenCV)")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
               ("octave", "<i4"), ("class_id", "<i4")])


def _cases(DUMP=DUMP):
    out = []
    with open(os.path.join(DUMP, "manifest.txt")) as f:
        for line in f:
            if line.strip() and not line.startswith("#"):
                t = line.split()
                out.append(dict(name=t[0], image=t[1], w=int(t[2]), h=int(t[3]), radius=float(t[4]),
                                octaves=int(t[5]), thr=int(t[6]), maxk=int(t[7]), rot=int(t[8]),
                                scale=int(t[9]), mode=int(t[10]), fu=float(t[11]),
                                grav=tuple(float(v) for v in t[12:15]), rays=t[15], jac=t[16]))
    return out


def _pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = (int(v) for v in f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w)


def compare_detector(oracle, DUMP):
    for c in _cases(DUMP):
        img = _pgm(os.path.join(DUMP, c["image"]))
        ref = np.fromfile(os.path.join(DUMP, c["name"] + ".kps.bin"), dtype=KP)
        got = oracle.detect(img, c["radius"], c["octaves"], c["thr"], c["maxk"])
        assert len(got) == len(ref), (c["name"], len(got), len(ref))
        for f in ("x", "y", "size", "response"):
            assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), (c["name"], f)
        assert np.array_equal(got["octave"], ref["octave"]), c["name"]


def compare_extractor(oracle, DUMP):
    for c in _cases(DUMP):
        if c["scale"]:
            continue
        img = _pgm(os.path.join(DUMP, c["image"]))
        kin = np.fromfile(os.path.join(DUMP, c["name"] + ".kps.bin"), dtype=KP)
        kref = np.fromfile(os.path.join(DUMP, c["name"] + ".kps_desc.bin"), dtype=KP)
        dref = np.fromfile(os.path.join(DUMP, c["name"] + ".desc.bin"), dtype=np.uint8).reshape(-1, 48)
        if c["mode"] == 2:
            rays = np.fromfile(os.path.join(DUMP, c["rays"]), dtype=np.float32).reshape(c["h"], c["w"], 3)
            jac = np.fromfile(os.path.join(DUMP, c["jac"]), dtype=np.float32).reshape(c["h"], c["w"], 6)
            k, d = oracle.describe(img, kin, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(c["fu"]),
                                   c["grav"])
        else:
            mode = oracle.MODE_GRADIENT if c["rot"] else oracle.MODE_UPRIGHT
            k, d = oracle.describe(img, kin, mode)
        assert len(k) == len(kref), (c["name"], "keypoints removed by the extractor differ")
        assert np.array_equal(k["x"].view(np.uint32), kref["x"].view(np.uint32)), c["name"]
        assert np.array_equal(d, dref), (c["name"], int((d != dref).any(axis=1).sum()), "rows differ")


def compare_hamming(oracle, DUMP):
    c = _cases(DUMP)[0]
    dref = np.fromfile(os.path.join(DUMP, c["name"] + ".desc.bin"), dtype=np.uint8).reshape(-1, 48)
    ham = np.fromfile(os.path.join(DUMP, "hamming.bin"), dtype=np.uint32)
    assert len(ham) == len(dref)
    for r in range(len(dref)):
        assert oracle.popcnt_xor(dref[0], dref[r]) == ham[r]


def _rounded(k):
    return np.stack([np.floor(k["x"] + 0.5).astype(np.int32), np.floor(k["y"] + 0.5).astype(np.int32)], 1)


def stage_report(oracle, DUMP):
    """Walks the pipeline stage by stage on every case of a dump and returns [(stage, ok, detail)] in
    pipeline order; the first entry with ok == False is where this repo's restatement and the dumped
    arithmetic part ways.  Stages whose files are absent (a dump made with
    -DOKVFE_REF_DUMP_NO_INTERNALS has no score map / raw maxima) are reported as skipped."""
    rep = []

    def add(stage, ok, detail=""):
        rep.append((stage, ok, detail))

    # 0. popcount
    try:
        compare_hamming(oracle, DUMP)
        add("hamming (PopcntofXORed)", True)
    except AssertionError as e:
        add("hamming (PopcntofXORed)", False, str(e))
    cases = _cases(DUMP)
    single = [c for c in cases if c["octaves"] == 0]
    # 1. score map
    bad, seen = None, 0
    for c in single:
        f = os.path.join(DUMP, c["name"] + ".score.i32")
        if not os.path.exists(f):
            continue
        seen += 1
        img = _pgm(os.path.join(DUMP, c["image"]))
        ref = np.fromfile(f, dtype=np.int32).reshape(c["h"], c["w"])
        got = oracle.harris_score(img)
        inner = (slice(2, c["h"] - 2), slice(2, c["w"] - 2))
        if not np.array_equal(got[inner], ref[inner]):
            d = np.argwhere(got[inner] != ref[inner])
            y, x = d[0] + 2
            ratio = np.median(ref[inner][ref[inner] != 0] / np.maximum(got[inner][ref[inner] != 0], 1))
            bad = (f"{c['name']}: {len(d)} of {got[inner].size} scores differ, first at (x={x}, y={y}): "
                   f"ours {got[y, x]} reference {ref[y, x]}; median reference/ours ratio {ratio:.4g} "
                   "(a constant ratio = a different score scale: shift / normalisation of the binomial)")
            break
    add("score map (HarrisScoreCalculator)", None if seen == 0 else bad is None, bad or ("not in the dump" if seen == 0 else ""))
    # 2. raw 2-D maxima (NMS rule, threshold meaning, tie handling)
    bad, seen = None, 0
    for c in single:
        f = os.path.join(DUMP, c["name"] + ".maxima.bin")
        if not os.path.exists(f):
            continue
        seen += 1
        img = _pgm(os.path.join(DUMP, c["image"]))
        ref = np.fromfile(f, dtype=np.int32).reshape(-1, 3)
        got = oracle.nms(oracle.harris_score(img), c["thr"])
        g = {(int(p["x"]), int(p["y"])): int(p["score"]) for p in got}
        r = {(int(x), int(y)): int(sc) for x, y, sc in ref}
        if g != r:
            only_g, only_r = sorted(set(g) - set(r)), sorted(set(r) - set(g))
            bad = (f"{c['name']}: {len(g)} maxima here, {len(r)} in the reference; only here {only_g[:3]}, only "
                   f"there {only_r[:3]}" + ("" if only_g or only_r else "; same positions, scores differ"))
            break
    add("2-D maxima (Get2dMaxima)", None if seen == 0 else bad is None, bad or ("not in the dump" if seen == 0 else ""))
    # 3. maxima + sub-pixel without uniformity (public API: radius 0, no cap)
    bad, seen = None, 0
    for c in single:
        f = os.path.join(DUMP, c["name"] + ".kps_nouniform.bin")
        if not os.path.exists(f):
            continue
        seen += 1
        img = _pgm(os.path.join(DUMP, c["image"]))
        ref = np.fromfile(f, dtype=KP)
        got = oracle.detect(img, 0.0, 0, c["thr"], 100000000)
        gs = set(map(tuple, _rounded(got)))
        rs = set(map(tuple, _rounded(ref)))
        if gs != rs:
            bad = f"{c['name']}: maxima sets differ ({len(gs)} vs {len(rs)}; only here {sorted(gs - rs)[:3]}, only there {sorted(rs - gs)[:3]})"
            break
        go, ro = np.lexsort((got["x"], got["y"])), np.lexsort((ref["x"], ref["y"]))
        gx, rx = got[go], ref[ro]
        if not (np.array_equal(gx["x"].view(np.uint32), rx["x"].view(np.uint32)) and
                np.array_equal(gx["y"].view(np.uint32), rx["y"].view(np.uint32))):
            i = int(np.flatnonzero((gx["x"] != rx["x"]) | (gx["y"] != rx["y"]))[0])
            bad = (f"{c['name']}: same maxima, sub-pixel positions differ, first ({gx['x'][i]:.6f}, {gx['y'][i]:.6f}) "
                   f"vs ({rx['x'][i]:.6f}, {rx['y'][i]:.6f})")
            break
        if not np.array_equal(gx["response"].view(np.uint32), rx["response"].view(np.uint32)):
            bad = f"{c['name']}: responses differ"
            break
    add("maxima + sub-pixel, no uniformity", None if seen == 0 else bad is None, bad or ("not in the dump" if seen == 0 else ""))
    # 4. uniformity: the accepted set and its ORDER, then the final records
    bad = None
    for c in cases:
        img = _pgm(os.path.join(DUMP, c["image"]))
        ref = np.fromfile(os.path.join(DUMP, c["name"] + ".kps.bin"), dtype=KP)
        got = oracle.detect(img, c["radius"], c["octaves"], c["thr"], c["maxk"])
        gr, rr = _rounded(got), _rounded(ref)
        if set(map(tuple, gr)) != set(map(tuple, rr)):
            bad = f"{c['name']}: accepted sets differ ({len(got)} vs {len(ref)} keypoints)"
            break
        if not np.array_equal(gr, rr):
            i = int(np.flatnonzero((gr != rr).any(axis=1))[0])
            bad = f"{c['name']}: same accepted set, ORDER differs from position {i} on (tie order of the sort / raster order)"
            break
        for f in ("x", "y", "size", "response"):
            if not np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)):
                bad = f"{c['name']}: field {f} differs"
                break
        if bad is None and not np.array_equal(got["octave"], ref["octave"]):
            bad = f"{c['name']}: octave differs"
        if bad:
            break
    add("uniformity, cap, keypoint records", bad is None, bad or "")
    # 5. extractor probes: pattern positions / half-widths / pairs / bit order
    bad, seen = None, 0
    pf = os.path.join(DUMP, "probes.txt")
    if os.path.exists(pf):
        for line in open(pf):
            if not line.strip() or line.startswith("#"):
                continue
            t = line.split()
            stem = t[0][:-4]
            f = os.path.join(DUMP, stem + ".desc.bin")
            if not os.path.exists(f):
                continue
            seen += 1
            img = _pgm(os.path.join(DUMP, t[0]))
            kp = np.zeros(1, dtype=KP)
            kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"], kp["class_id"] = float(t[3]), float(t[4]), 12.0, -1.0, 1000.0, -1
            k, d = oracle.describe(img, kp, oracle.MODE_UPRIGHT)
            ref = np.fromfile(f, dtype=np.uint8)
            if len(ref) == 0 or len(k) == 0:
                if (len(ref) == 0) != (len(k) == 0):
                    bad = f"{stem}: the keypoint is {'kept' if len(k) else 'removed'} here, {'kept' if len(ref) else 'removed'} there (border rule)"
                    break
                continue
            if not np.array_equal(d[0], ref):
                bits = np.flatnonzero(np.unpackbits(d[0] ^ ref, bitorder="little"))
                bad = (f"{stem}: {len(bits)} of 384 bits differ, first bits {bits[:8].tolist()} (few bits near one sample = "
                       "a half-width / position; many = pair table or bit order)")
                break
    add("extractor probes (pattern, pairs, bit order)", None if seen == 0 else bad is None, bad or ("not in the dump" if seen == 0 else ""))
    # 6. descriptors of the real cases (orientation / camera-aware warp on top of the pattern)
    try:
        compare_extractor(oracle, DUMP)
        add("descriptors of the cases", True)
    except AssertionError as e:
        add("descriptors of the cases", False, str(e))
    return rep


def _fp64_expected(oracle):
    """What the oracle computes for the inputs of tools/ref_compare/ref_dump_fp64.cpp, in the dump's layout:
    (tri [n, 6], bp {'rt': [n, 4], 'eq': [n, 4]}, stereo [n0, 7])."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "ref_compare"))
    import make_inputs
    _, P = make_inputs.fp64_inputs(oracle)
    tri = np.zeros((len(P["tri"]), 6))
    for i, t in enumerate(P["tri"]):
        hp, valid, par = oracle.triangulate_fast(t[0:3], t[3:6], t[6:9], t[9:12], t[12])
        tri[i] = [*hp, float(valid), float(par)]
    bp = {}
    for key in ("rt", "eq"):
        out = np.zeros((len(P["pts"][key]), 4))
        for i, pt in enumerate(P["pts"][key]):
            ok, ray = oracle.cam_backproject(P["cams"][key], pt)
            out[i] = [*(ray if ok else (0.0, 0.0, 0.0)), float(ok)]
        bp[key] = out
    cfg = P["cfg"]
    (k0, d0), (k1, d1) = P["sides"]
    b0, v0 = oracle.backproject_keypoints(cfg.cams[0], k0)
    b1, v1 = oracle.backproject_keypoints(cfg.cams[1], k1)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    m = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, P["poses"][0], P["poses"][1], f0, f1, cfg.match_threshold)
    st = np.zeros((len(k0), 7))
    for i, r in enumerate(m):
        if r["k1"] >= 0:
            st[i] = [float(r["k1"]), float(r["dist"]), float(r["initialisable"]), *r["hp_W"]]
        else:
            st[i] = [-1.0, float(cfg.match_threshold), 0.0, 0.0, 0.0, 0.0, 0.0]
    return tri, bp, st


def _fp64_read(DUMP):
    f = os.path.join(DUMP, "fp64_dump.bin")
    if not os.path.exists(f):
        return None
    raw = open(f, "rb").read()
    n_tri, n_bp, n0, n1 = np.frombuffer(raw[:16], dtype="<i4")
    v = np.frombuffer(raw[16:], dtype="<f8")
    a = 6 * n_tri
    tri = v[:a].reshape(n_tri, 6)
    rt = v[a:a + 4 * n_bp].reshape(n_bp, 4)
    eq = v[a + 4 * n_bp:a + 8 * n_bp].reshape(n_bp, 4)
    st = v[a + 8 * n_bp:a + 8 * n_bp + 7 * n0].reshape(n0, 7)
    return tri, {"rt": rt, "eq": eq}, st


def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def fp64_stage_report(oracle, DUMP):
    """The FP64 chain of the matchers against real Eigen (tools/ref_compare/ref_dump_fp64.cpp): doubles are
    compared as bit patterns; a difference is reported with its size in ulp and where it first shows."""
    rep = []
    got = _fp64_read(DUMP)
    if got is None:
        return [("triangulateFast (Eigen)", None, "no fp64_dump.bin"), ("backProject radial-tangential", None, ""),
                ("backProject equidistant", None, ""), ("matchStereo rows", None, "")]
    tri_r, bp_r, st_r = got
    tri, bp, st = _fp64_expected(oracle)

    def cmp(name, mine, ref, flags):
        mine, ref = np.ascontiguousarray(mine), np.ascontiguousarray(ref)
        if mine.shape != ref.shape:
            return rep.append((name, False, f"shape {mine.shape} here, {ref.shape} in the dump"))
        fl = np.flatnonzero((mine[:, flags] != ref[:, flags]).any(axis=1))
        if len(fl):
            i = int(fl[0])
            return rep.append((name, False, f"{len(fl)} rows with different decisions, first row {i}: "
                                            f"{mine[i, flags].tolist()} here, {ref[i, flags].tolist()} there"))
        vals = [c for c in range(mine.shape[1]) if c not in flags]
        u = _ulps(mine[:, vals], ref[:, vals])
        if u.max() > 0:
            i, j = np.unravel_index(int(u.argmax()), u.shape)
            first = int(np.flatnonzero(u.any(axis=1))[0])
            return rep.append((name, False, f"{int((u > 0).any(axis=1).sum())} of {len(u)} rows differ in some bit "
                                            f"(max {int(u.max())} ulp at row {i}, first row {first}): the evaluation "
                                            "order of a sum / product differs from Eigen's"))
        rep.append((name, True, ""))

    cmp("triangulateFast (Eigen)", tri, tri_r, [4, 5])
    cmp("backProject radial-tangential", bp["rt"], bp_r["rt"], [3])
    cmp("backProject equidistant", bp["eq"], bp_r["eq"], [3])
    cmp("matchStereo rows", st, st_r, [0, 1, 2])
    return rep


def first_divergence(rep):
    for stage, ok, detail in rep:
        if ok is False:
            return f"FIRST DIVERGING STAGE: {stage} -- {detail}"
    return None


@needs_dump
def test_stage_by_stage_against_the_reference_dump(oracle):
    """THE pin: every stage of detector and extractor against what the real brisk produced; the
    failure message names the first stage that differs and how."""
    rep = stage_report(oracle, DUMP) + fp64_stage_report(oracle, DUMP)
    for stage, ok, detail in rep:
        print(f"{'ok  ' if ok else ('skip' if ok is None else 'DIFF')} {stage} {detail}")
    msg = first_divergence(rep)
    assert msg is None, msg


@needs_dump
def test_detector_keypoints_match_reference(oracle):
    compare_detector(oracle, DUMP)


@needs_dump
def test_extractor_descriptors_match_reference(oracle):
    compare_extractor(oracle, DUMP)


@needs_dump
def test_hamming_matches_reference(oracle):
    compare_hamming(oracle, DUMP)


def test_compare_harness_on_a_self_dump(oracle, tmp_path):
    """The three comparisons above run here against a dump in the SAME file format written by the
    oracle itself (two small cases): proves the reader / comparison code, not parity."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "ref_compare"))
    import make_inputs
    from okvis2_amd import capi, synth
    d = str(tmp_path)
    cam = synth.Camera(256, 192, 150.0, 151.0, 127.0, 95.0, 1, (-0.1, 0.01, 0.0005, -0.0003))
    rays, jac = capi.build_awareness_maps(cam)
    rays.tofile(os.path.join(d, "m.rays.f32"))
    jac.tofile(os.path.join(d, "m.jac.f32"))
    lines = []
    first = True
    for name, img, mode, rot in (("a", synth.corners_image(256, 192, 3), 0, 1),
                                 ("b", synth.corners_image(256, 192, 4), 2, 1)):
        make_inputs.write_pgm(os.path.join(d, name + ".pgm"), img)
        lines.append(f"{name} {name}.pgm 256 192 20 0 50 300 {rot} 0 {mode} {float(np.float32(cam.fu)):.9g} "
                     f"0.1 0.98 -0.05 {'m.rays.f32' if mode == 2 else '-'} {'m.jac.f32' if mode == 2 else '-'}")
        k = oracle.detect(img, 20.0, 0, 50, 300)
        k.tofile(os.path.join(d, name + ".kps.bin"))
        sc = oracle.harris_score(img)
        sc.astype(np.int32).tofile(os.path.join(d, name + ".score.i32"))
        mx = oracle.nms(sc, 50)
        np.stack([mx["x"], mx["y"], mx["score"]], 1).astype(np.int32).tofile(os.path.join(d, name + ".maxima.bin"))
        oracle.detect(img, 0.0, 0, 50, 100000000).tofile(os.path.join(d, name + ".kps_nouniform.bin"))
        if mode == 2:
            kk, dd = oracle.describe(img, k, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                                     (0.1, 0.98, -0.05))
        else:
            kk, dd = oracle.describe(img, k, oracle.MODE_GRADIENT)
        kk.tofile(os.path.join(d, name + ".kps_desc.bin"))
        dd.tofile(os.path.join(d, name + ".desc.bin"))
        if first:
            np.array([oracle.popcnt_xor(dd[0], r) for r in dd], dtype=np.uint32).tofile(
                os.path.join(d, "hamming.bin"))
            first = False
        assert len(kk) > 20
    open(os.path.join(d, "manifest.txt"), "w").write("# self dump\n" + "\n".join(lines) + "\n")
    with open(os.path.join(d, "probes.txt"), "w") as pf:  # a few of the extractor probes
        for pname, pimg in make_inputs.probes()[::9]:
            make_inputs.write_pgm(os.path.join(d, pname + ".pgm"), pimg)
            pf.write(f"{pname}.pgm {make_inputs.PROBE_SIZE} {make_inputs.PROBE_SIZE} {make_inputs.PROBE_KP[0]:.1f} "
                     f"{make_inputs.PROBE_KP[1]:.1f}\n")
            kp = np.zeros(1, dtype=KP)
            kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"], kp["class_id"] = (*make_inputs.PROBE_KP, 12.0, -1.0, 1000.0, -1)
            _, pd = oracle.describe(pimg, kp, oracle.MODE_UPRIGHT)
            pd.tofile(os.path.join(d, pname + ".desc.bin"))
    open(os.path.join(d, "dump_done.txt"), "w").write("2 cases\n")
    compare_detector(oracle, d)
    compare_extractor(oracle, d)
    compare_hamming(oracle, d)
    rep = stage_report(oracle, d)
    assert [ok for _, ok, _ in rep] == [True] * 7, rep
    # ... and the report NAMES the stage when one is made to differ: a score map at another scale
    # (the un-normalised binomial sum is the restatement's most likely deviation), a flipped pair
    sc2 = np.fromfile(os.path.join(d, "a.score.i32"), dtype=np.int32) // 16
    sc2.tofile(os.path.join(d, "a.score.i32"))
    msg = first_divergence(stage_report(oracle, d))
    assert msg and msg.startswith("FIRST DIVERGING STAGE: score map") and "ratio" in msg, msg
    oracle.harris_score(_pgm(os.path.join(d, "a.pgm"))).astype(np.int32).tofile(os.path.join(d, "a.score.i32"))
    first_probe = sorted(f for f in os.listdir(d) if f.startswith("probe_") and f.endswith(".desc.bin"))[0]
    pb = np.fromfile(os.path.join(d, first_probe), dtype=np.uint8)
    pb[3] ^= 0x10
    pb.tofile(os.path.join(d, first_probe))
    msg = first_divergence(stage_report(oracle, d))
    assert msg and msg.startswith("FIRST DIVERGING STAGE: extractor probes") and "[28]" in msg, msg


def test_fp64_harness_on_a_self_dump(oracle, tmp_path):
    """The FP64 stages (triangulateFast, backProject x 2, matchStereo rows) against a dump in the format of
    tools/ref_compare/ref_dump_fp64.cpp written by the oracle itself: proves the reader and the comparison,
    not parity -- and that ONE flipped ulp in one triangulated point, or one different match decision, is
    named with its stage."""
    import struct
    d = str(tmp_path)
    tri, bp, st = _fp64_expected(oracle)
    assert (tri[:, 4] == 1).sum() > 100 and (tri[:, 5] == 1).sum() > 50 and (tri[:, 4] == 0).sum() > 20
    assert (bp["rt"][:, 3] == 1).sum() > 400 and (bp["eq"][:, 3] == 1).sum() > 400
    assert (st[:, 0] >= 0).sum() > 50

    def write(tri_, bp_, st_):
        with open(os.path.join(d, "fp64_dump.bin"), "wb") as f:
            f.write(struct.pack("<4i", len(tri_), len(bp_["rt"]), len(st_), 0))
            for a in (tri_, bp_["rt"], bp_["eq"], st_):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())

    write(tri, bp, st)
    rep = fp64_stage_report(oracle, d)
    assert [ok for _, ok, _ in rep] == [True] * 4, rep
    t2 = tri.copy()
    row = int(np.flatnonzero(t2[:, 4] == 1)[3])
    t2[row, 1] = np.nextafter(t2[row, 1], np.inf)
    write(t2, bp, st)
    msg = first_divergence(fp64_stage_report(oracle, d))
    assert msg and msg.startswith("FIRST DIVERGING STAGE: triangulateFast") and "max 1 ulp" in msg and f"row {row}" in msg, msg
    s2 = st.copy()
    row = int(np.flatnonzero(s2[:, 0] >= 0)[0])
    s2[row, 0] += 1.0
    write(tri, bp, s2)
    msg = first_divergence(fp64_stage_report(oracle, d))
    assert msg and msg.startswith("FIRST DIVERGING STAGE: matchStereo rows") and "different decisions" in msg, msg
    b2 = {"rt": bp["rt"], "eq": bp["eq"].copy()}
    row = int(np.flatnonzero(b2["eq"][:, 3] == 1)[5])
    b2["eq"][row, 2] = np.nextafter(b2["eq"][row, 2], 0.0)
    write(tri, b2, st)
    msg = first_divergence(fp64_stage_report(oracle, d))
    assert msg and msg.startswith("FIRST DIVERGING STAGE: backProject equidistant"), msg
