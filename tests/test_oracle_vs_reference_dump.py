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
    open(os.path.join(d, "dump_done.txt"), "w").write("2 cases\n")
    compare_detector(oracle, d)
    compare_extractor(oracle, d)
    compare_hamming(oracle, d)
