"""CPU tests of the oracle's detector / extractor / matcher: committed golden vectors
(tests/golden/frontend_golden.npz, made by tools/make_golden.py), the contract invariants of
SURVEY.md §8 C4 (iv), and independent numpy re-derivations of the integer stages."""
import os

import numpy as np
import pytest

from okvis2_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz")


@pytest.fixture(scope="module")
def gold():
    g = np.load(GOLDEN)
    W, H, radius, thr, maxk, mthr = g["params"]
    cams = [synth.Camera(int(W), int(H), c[0], c[1], c[2], c[3], int(c[4]), tuple(c[5:9]))
            for c in g["cams"]]
    return g, int(W), int(H), float(radius), int(thr), int(maxk), int(mthr), cams


def harris_numpy(img):
    """Independent restatement with numpy slicing (int64 everywhere, floor shifts)."""
    p = img.astype(np.int64)
    h, w = p.shape
    gx = np.zeros((h, w), np.int64)
    gy = np.zeros((h, w), np.int64)
    gx[1:-1, 1:-1] = (3 * (p[:-2, 2:] - p[:-2, :-2]) + 10 * (p[1:-1, 2:] - p[1:-1, :-2]) +
                      3 * (p[2:, 2:] - p[2:, :-2]))
    gy[1:-1, 1:-1] = (3 * (p[2:, :-2] - p[:-2, :-2]) + 10 * (p[2:, 1:-1] - p[:-2, 1:-1]) +
                      3 * (p[2:, 2:] - p[:-2, 2:]))
    ent = [(gx * gx) >> 14, (gy * gy) >> 14, (gx * gy) >> 14]
    for e in ent:
        e[0, :] = e[-1, :] = 0
        e[:, 0] = e[:, -1] = 0
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]])
    sm = []
    for e in ent:
        s = np.zeros_like(e)
        for dy in range(3):
            for dx in range(3):
                s[1:-1, 1:-1] += k[dy, dx] * e[dy:h - 2 + dy, dx:w - 2 + dx]
        sm.append(s)
    A, B, Cc = sm
    tq = ((A >> 1) + (B >> 1)) >> 1
    sc = A * B - Cc * Cc - tq * tq
    sc[0, :] = sc[-1, :] = 0
    sc[:, 0] = sc[:, -1] = 0
    return sc.astype(np.int32)


def test_harris_against_numpy_restatement(oracle):
    for img in (synth.corners_image(200, 150, 3), synth.noise_image(131, 77, 4),
                np.zeros((64, 64), np.uint8), np.full((64, 80), 255, np.uint8)):
        assert np.array_equal(oracle.harris_score(img), harris_numpy(img))


def test_golden_score_nms_detect(oracle, gold):
    g, W, H, radius, thr, maxk, _, _ = gold
    score = oracle.harris_score(g["left"])
    assert np.array_equal(score, g["score_left"])
    assert np.array_equal(oracle.nms(score, thr), g["nms_left"])
    for ci, key in enumerate(("left", "right")):
        k = oracle.detect(g[key], radius, 0, thr, maxk)
        assert np.array_equal(k.view(np.uint8), g[f"kp_detect_{ci}"].view(np.uint8))
    k = oracle.detect(g["noise"], 34.0, 0, 800, 450)
    assert np.array_equal(k.view(np.uint8), g["kp_noise"].view(np.uint8))


def test_golden_describe_and_match(oracle, gold):
    g, W, H, radius, thr, maxk, mthr, cams = gold
    res = []
    for ci, key in enumerate(("left", "right")):
        kd = g[f"kp_detect_{ci}"]
        for mode, name in ((oracle.MODE_UPRIGHT, "upright"), (oracle.MODE_GRADIENT, "gradient")):
            k, d = oracle.describe(g[key], kd, mode)
            assert np.array_equal(k.view(np.uint8), g[f"kp_{name}_{ci}"].view(np.uint8))
            assert np.array_equal(d, g[f"desc_{name}_{ci}"])
        rays, jac = oracle.awareness_maps(cams[ci])
        k, d = oracle.describe(g[key], kd, oracle.MODE_CAMERA_AWARE, rays, jac,
                               np.float32(cams[ci].fu), (0.1, 0.98, -0.05))
        assert np.array_equal(k.view(np.uint8), g[f"kp_aware_{ci}"].view(np.uint8))
        assert np.array_equal(d, g[f"desc_aware_{ci}"])
        bp, bv = oracle.backproject_keypoints(cams[ci], k)
        assert np.array_equal(bp.view(np.uint64), g[f"bp_{ci}"].view(np.uint64))
        assert np.array_equal(bv, g[f"bpv_{ci}"])
        res.append((k, d, bp, bv))
    T0, T1 = synth.stereo_poses(0.11)
    f0, f1 = 0.5 * (cams[0].fu + cams[0].fv), 0.5 * (cams[1].fu + cams[1].fv)
    (k0, d0, b0, v0), (k1, d1, b1, v1) = res
    m = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, mthr)
    assert np.array_equal(m.view(np.uint8), g["match_stereo"].view(np.uint8))
    assert (m["k1"] >= 0).sum() > 50


def test_detector_contract_invariants(oracle):
    for cfg in (synth.euroc_config(), synth.mono640_config()):
        img = synth.corners_image(cfg.w, cfg.h, 11)
        kps, score = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                   want_score=True)
        assert 0 < len(kps) <= cfg.max_kpts                      # n_kpts <= max_num_keypoints
        assert np.all(kps["size"] == 12.0) and np.all(kps["octave"] == 0)  # kp.size == 12*scale
        assert np.all(kps["angle"] == -1.0) and np.all(kps["class_id"] == -1)
        assert np.all(kps["response"] >= cfg.abs_threshold)
        assert np.all(np.diff(kps["response"]) <= 0)             # strongest first
        # every keypoint is a 3x3 maximum of the score map, refined by less than a pixel
        xi, yi = np.rint(kps["x"]).astype(int), np.rint(kps["y"]).astype(int)
        assert xi.min() >= 1 and yi.min() >= 1 and xi.max() <= cfg.w - 2 and yi.max() <= cfg.h - 2
        # score map rim is zero
        assert not score[0].any() and not score[-1].any() and not score[:, 0].any() \
            and not score[:, -1].any()


def test_nms_properties(oracle):
    img = synth.corners_image(320, 240, 8)
    score = oracle.harris_score(img)
    pts = oracle.nms(score, 50)
    assert len(pts) > 100
    for p in pts[::7]:
        x, y, s = int(p["x"]), int(p["y"]), int(p["score"])
        assert 2 <= x < 320 - 2 and 2 <= y < 240 - 2 and s >= 50
        assert score[y, x] == s and score[y - 1:y + 2, x - 1:x + 2].max() == s
    # raster order
    key = pts["y"].astype(np.int64) * 4096 + pts["x"]
    assert np.all(np.diff(key) > 0)
    # plateau rule: of equal horizontal neighbours only the left one is kept
    flat = np.zeros((32, 32), np.int32)
    flat[10, 10] = flat[10, 11] = flat[10, 12] = 500
    got = oracle.nms(flat, 100)
    assert [(int(p["x"]), int(p["y"])) for p in got] == [(10, 10), (12, 10)]


def test_uniformity_properties(oracle):
    rng = np.random.default_rng(3)
    w, h = 400, 300
    n = 3000
    xy = rng.permutation(w * h)[:n]
    pts = np.zeros(n, dtype=oracle.POINT_DTYPE)
    pts["x"], pts["y"] = 2 + xy % (w - 4), 2 + (xy // w) % (h - 4)
    pts["score"] = rng.integers(100, 10_000_000, n)
    sel = oracle.uniformity_select(pts, w, h, 25.0, 150)
    assert 0 < len(sel) <= 150
    assert np.all(np.diff(sel["score"]) <= 0)
    assert sel["score"][0] == pts["score"].max()       # the strongest point always survives
    # cap respected, radius <= 0 disables the stage
    assert len(oracle.uniformity_select(pts, w, h, 25.0, 10)) == 10
    assert len(oracle.uniformity_select(pts, w, h, 0.0, 10)) == n
    # a larger radius never keeps more points
    assert len(oracle.uniformity_select(pts, w, h, 50.0, 3000)) <= \
        len(oracle.uniformity_select(pts, w, h, 25.0, 3000))
    # equal scores: the total order (y, x) decides, independent of the input order
    tie = pts.copy()
    tie["score"] = 5000
    a = oracle.uniformity_select(tie, w, h, 25.0, 100)
    b = oracle.uniformity_select(tie[::-1].copy(), w, h, 25.0, 100)
    assert np.array_equal(a, b)


def test_subpixel_refinement(oracle):
    # symmetric peak -> no shift; tilted peak -> shift towards the larger neighbour, within 1 px
    assert oracle.subpixel2d([[1, 2, 1], [2, 9, 2], [1, 2, 1]]) == (0.0, 0.0)
    dx, dy = oracle.subpixel2d([[10, 20, 30], [20, 90, 60], [10, 20, 30]])
    assert 0 < dx <= 1 and abs(dy) < 1e-6
    dx, dy = oracle.subpixel2d([[100000000, 200000000, 100000000], [200000000, 260000000, 250000000],
                                [100000000, 200000000, 100000000]])
    assert 0 < dx <= 1 and -1 <= dy <= 1   # Harris-sized scores: no 32-bit overflow artefacts
    assert oracle.subpixel2d(np.zeros((3, 3))) == (0.0, 0.0)


def test_descriptor_contract(oracle):
    cfg = synth.euroc_config()
    img = synth.corners_image(cfg.w, cfg.h, 21)
    kps = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    pat = oracle.pattern()
    assert pat.n_points == 66 and pat.n_short == 384 and pat.n_long == 968 and pat.border == 29
    pub = oracle.pattern_published()
    assert pub.n_points == 60 and pub.n_short == 383 and pub.n_long == 870 and pub.border == 29
    for mode in (oracle.MODE_UPRIGHT, oracle.MODE_GRADIENT):
        k, d = oracle.describe(img, kps, mode)
        assert d.shape == (len(k), 48) and len(k) <= len(kps)        # desc.cols == 48
        assert (d[:, 47] & 0x80).any() and not (d[:, 47] & 0x80).all()  # bit 383 live
        b = pat.border
        assert np.all((k["x"] >= b) & (k["x"] < cfg.w - b) & (k["y"] >= b) & (k["y"] < cfg.h - b))
        assert 0.35 < np.unpackbits(d).mean() < 0.65
    # upright descriptors are translation-covariant: shift the image, same bits
    sh = np.zeros_like(img)
    sh[:, 5:] = img[:, :-5]
    k0, d0 = oracle.describe(img, kps, oracle.MODE_UPRIGHT)
    moved = k0.copy()
    moved["x"] += 5
    keep = moved["x"] < cfg.w - pat.border
    k1, d1 = oracle.describe(sh, moved[keep], oracle.MODE_UPRIGHT)
    assert np.array_equal(d1, d0[keep])
    # gradient mode: angles are multiples of 360/1024
    kg, _ = oracle.describe(img, kps, oracle.MODE_GRADIENT)
    assert np.all(np.mod(kg["angle"] / 0.3515625, 1.0) == 0) and np.all(kg["angle"] < 360)


def test_integral_image(oracle):
    img = synth.noise_image(97, 61, 5)
    I = oracle.integral(img)
    ref = np.zeros((62, 98), np.int64)
    ref[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(I, ref)


def test_camera_aware_reduces_to_upright_for_ideal_camera(oracle):
    """Undistorted camera, fu == fv, gravity along +y: M is the identity up to rounding, so the
    descriptors agree with the upright ones on almost every bit."""
    cam = synth.Camera(320, 240, 200.0, 200.0, 160.0, 120.0, 0, (0, 0, 0, 0))
    img = synth.corners_image(320, 240, 31)
    kps = oracle.detect(img, 15.0, 0, 50, 300)
    rays, jac = oracle.awareness_maps(cam)
    ku, du = oracle.describe(img, kps, oracle.MODE_UPRIGHT)
    ka, da = oracle.describe(img, kps, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                             (0.0, 1.0, 0.0))
    # off-centre the tangent-plane pattern is stretched by perspective (1/cos, 1/cos^2), so the
    # camera-aware extractor removes more rim keypoints; compare the ones close to the centre
    assert len(ka) <= len(ku)
    near = np.flatnonzero((np.abs(ka["x"] - 160) < 30) & (np.abs(ka["y"] - 120) < 30))
    assert len(near) > 3
    iu = {(float(k["x"]), float(k["y"])): i for i, k in enumerate(ku)}
    sel_u = np.array([iu[(float(ka["x"][i]), float(ka["y"][i]))] for i in near])
    diff = np.unpackbits(du[sel_u] ^ da[near], axis=1).sum(1)
    assert diff.max() <= 40 and diff.mean() < 20
    # gravity along -y flips the pattern: descriptors change substantially
    kf, df = oracle.describe(img, kps, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                             (0.0, -1.0, 0.0))
    assert np.array_equal(kf["x"], ka["x"])
    assert np.unpackbits(df[near] ^ da[near], axis=1).sum(1).mean() > 60


def test_match_stereo_semantics(oracle):
    """Running minimum with strict '<' and first-lowest wins; gate failures fall through to the
    next candidate; invalid back-projections never match (Frontend.cpp:2016-2076)."""
    rng = np.random.default_rng(9)
    n = 40
    d0 = rng.integers(0, 256, (n, 48), dtype=np.uint8)
    d1 = d0.copy()
    kp = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
    kp["size"] = 12.0
    X = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.5, 0.5, n), rng.uniform(2, 4, n)], 1)
    b = 0.11
    bp0 = X / X[:, 2:3]
    X1 = X - np.array([b, 0, 0])
    bp1 = X1 / X1[:, 2:3]
    v = np.ones(n, np.uint8)
    T0, T1 = synth.stereo_poses(b)
    m = oracle.match_stereo(d0, kp, bp0, v, d1, kp, bp1, v, T0, T1, 458.0, 458.0, 60)
    assert np.array_equal(m["k1"], np.arange(n)) and np.all(m["dist"] == 0)
    assert np.all(m["initialisable"] == 1)
    assert np.allclose(m["hp_W"][:, :3], X, atol=1e-9) and np.all(m["hp_W"][:, 3] == 1.0)
    # duplicate descriptor earlier in image 1 with an inconsistent ray: gate rejects it, the true
    # one still wins; with a consistent ray the FIRST of two equal distances wins
    d1b = np.concatenate([d1[5:6], d1])
    bp1b = np.concatenate([np.array([[5.0, 5.0, 1.0]]), bp1])
    vb = np.ones(n + 1, np.uint8)
    kpb = np.zeros(n + 1, dtype=oracle.KEYPOINT_DTYPE)
    kpb["size"] = 12.0
    m2 = oracle.match_stereo(d0, kp, bp0, v, d1b, kpb, bp1b, vb, T0, T1, 458.0, 458.0, 60)
    assert m2["k1"][5] == 6
    bp1c = bp1b.copy()
    bp1c[0] = bp1[5]
    m3 = oracle.match_stereo(d0, kp, bp0, v, d1b, kpb, bp1c, vb, T0, T1, 458.0, 458.0, 60)
    assert m3["k1"][5] == 0
    # invalid back-projection on either side: no match
    v0 = v.copy()
    v0[3] = 0
    v1 = v.copy()
    v1[7] = 0
    m4 = oracle.match_stereo(d0, kp, bp0, v0, d1, kp, bp1, v1, T0, T1, 458.0, 458.0, 60)
    assert m4["k1"][3] == -1 and m4["k1"][7] == -1 and m4["dist"][3] == 60
    # threshold is strict: distance == threshold does not match
    d1t = d1.copy()
    flip = np.zeros(48, np.uint8)
    flip[:7] = 0xFF
    flip[7] = 0x0F  # 60 bits
    d1t[0] ^= flip
    m5 = oracle.match_stereo(d0, kp, bp0, v, d1t, kp, bp1, v, T0, T1, 458.0, 458.0, 60)
    assert m5["k1"][0] == -1
    # empty inputs
    assert len(oracle.match_stereo(d0[:0], kp[:0], bp0[:0], v[:0], d1, kp, bp1, v, T0, T1, 458.0,
                                   458.0, 60)) == 0
    m6 = oracle.match_stereo(d0, kp, bp0, v, d1[:0], kp[:0], bp1[:0], v[:0], T0, T1, 458.0, 458.0, 60)
    assert np.all(m6["k1"] == -1)


def test_match_motion_stereo_semantics(oracle):
    """matchMotionStereo (Frontend.cpp:1812-1905): same camera, two poses; already-matched
    current keypoints are skipped; the winner must re-project within 4 px."""
    rng = np.random.default_rng(10)
    cam = synth.euroc_config().cams[0]
    n = 30
    X = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(3, 8, n)], 1)
    T0 = (np.eye(3).reshape(-1), np.zeros(3))
    T1 = (np.eye(3).reshape(-1), np.array([0.3, 0.05, 0.0]))

    def observe(T):
        Xc = X - T[1]
        kp = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
        kp["size"] = 12.0
        for i in range(n):
            st, pt, _ = oracle.cam_project(cam, Xc[i])
            assert st == 0
            kp["x"][i], kp["y"][i] = pt
        bp, bv = oracle.backproject_keypoints(cam, kp)
        return kp, bp, bv

    kp0, bp0, bv0 = observe(T0)
    kp1, bp1, bv1 = observe(T1)
    d = rng.integers(0, 256, (n, 48), dtype=np.uint8)
    matched1 = np.zeros(n, np.uint8)
    matched1[4] = 1
    skip0 = np.zeros(n, np.uint8)
    skip0[9] = 1
    m = oracle.match_motion_stereo(d, kp0, bp0, bv0, skip0, d, kp1, bp1, bv1, matched1, T0, T1, cam,
                                   60)
    want = np.arange(n)
    want[4] = -1
    want[9] = -1
    assert np.array_equal(m["k1"], want)
    ok = want >= 0
    assert np.all(m["accepted"][ok] == 1) and np.all(m["accepted"][~ok] == 0)
    assert np.allclose(m["hp_W"][ok, :3], X[ok], atol=1e-3)
    assert np.all(m["quality"][ok] > 0)


def test_scale_space_layers_and_samplers(oracle):
    """octaves > 0 (oracle/orc_detect.c detect_scale_space): samplers against numpy, layer sizes,
    keypoint sizes 12 * scale and octave = layer, layers in ascending order, per-layer cap."""
    img = synth.corners_image(300, 210, 3)
    a = img.astype(np.int32)
    assert np.array_equal(oracle.halfsample(img),
                          ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2)[:105, :150])
    t = oracle.twothirdsample(img)
    assert t.shape == (140, 200)
    wl, wr = np.array([2, 1, 0]), np.array([0, 1, 2])
    blk = a.reshape(70, 3, 100, 3).transpose(0, 2, 1, 3)  # [by, bx, j, i]
    for q, wy in enumerate((wl, wr)):
        for p, wx in enumerate((wl, wr)):
            want = ((blk * wy[None, None, :, None] * wx[None, None, None, :]).sum(axis=(2, 3)) + 4) // 9
            assert np.array_equal(t[q::2, p::2], want)
    assert [oracle.layer_size(752, 480, l) for l in range(4)] == [(752, 480), (500, 320), (376, 240), (250, 160)]
    k = oracle.detect(synth.corners_image(752, 480, 11), 38.0, 2, 150, 700)
    assert np.all(np.diff(k["octave"]) >= 0) and set(np.unique(k["octave"])) == {0, 1, 2, 3}
    scale = np.array([1.0, 1.5, 2.0, 3.0], dtype=np.float32)
    assert np.array_equal(k["size"], 12.0 * scale[k["octave"]])
    assert np.bincount(k["octave"]).max() <= 700
    # a maximum of layer 0 that survives is not dominated by layer 1 around the same spot, so the
    # single-scale detector finds at least as many layer-0 points
    k0 = oracle.detect(synth.corners_image(752, 480, 11), 38.0, 0, 150, 700)
    assert (k["octave"] == 0).sum() <= len(k0) + 5
    # coordinates of every layer stay inside the image
    assert k["x"].min() >= 0 and k["x"].max() < 752 and k["y"].min() >= 0 and k["y"].max() < 480


CIRCLE16 = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
            (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_agast_score_known_answers(oracle):
    """The published FAST / AGAST 9-16 predicate on hand-made circles: the score is the largest
    threshold t at which 9 contiguous circle pixels are all > p + t or all < p - t."""
    def scene(centre, values):
        img = np.full((16, 16), centre, np.uint8)
        for (dx, dy), v in zip(CIRCLE16, values):
            img[8 + dy, 8 + dx] = v
        return img

    # 9 contiguous brighter by 50, start anywhere on the circle (wrap-around included)
    for start in (0, 5, 11, 15):
        vals = [150 if (i - start) % 16 < 9 else 100 for i in range(16)]
        assert oracle.agast_score(scene(100, vals))[8, 8] == 49
    # only 8 contiguous: not a corner at any threshold
    vals = [150 if i < 8 else 100 for i in range(16)]
    assert oracle.agast_score(scene(100, vals))[8, 8] == 0
    # darker arc; the weakest pixel of the arc sets the score
    vals = [60 if 3 <= i < 12 else 100 for i in range(16)]
    vals[7] = 90
    assert oracle.agast_score(scene(100, vals))[8, 8] == 9
    # a longer arc does not raise the score; bright and dark arcs: the better one wins
    vals = [200] * 12 + [0] * 4
    assert oracle.agast_score(scene(100, vals))[8, 8] == 99
    vals = [130] * 9 + [100] * 7
    vals2 = [20] * 9 + [100] * 7
    assert oracle.agast_score(scene(100, vals))[8, 8] == 29 and oracle.agast_score(scene(100, vals2))[8, 8] == 79
    # difference of exactly 1: corner only at t = 0 -> score 0 (thresholds start at 1)
    assert oracle.agast_score(scene(100, [101] * 16))[8, 8] == 0
    assert oracle.agast_score(scene(100, [102] * 16))[8, 8] == 1
    # saturated point and the 3-pixel border rule
    img = np.zeros((16, 16), np.uint8)
    img[8, 8] = 255
    img[2, 2] = 255
    s = oracle.agast_score(img)
    assert s[8, 8] == 254 and s[2, 2] == 0 and s[:3].max() == 0 and s[:, -3:].max() == 0
