"""GPU: the sampling pattern is data (okvfe_get_pattern / okvfe_set_pattern): a pattern with other
sample offsets, half-widths, pair tables and bit order than the built-in restatement is installed on
the GPU and in the oracle, and detect + describe stay byte-equal in all three extraction modes -- the
route by which a pattern confirmed against a real `brisk` build (tools/ref_compare) replaces the
restatement without touching a kernel.  Also: the default pattern round-trips, bad patterns are
rejected."""
import ctypes as C

import numpy as np
import pytest

import gpu_common as G
from okvis2_amd import capi, synth

pytestmark = pytest.mark.gpu


def _to_orc(oracle, p):
    """orc_pattern with the fields of an okvfe_pattern (rotation tables from the oracle's own build)"""
    q = type(oracle.pattern())()
    C.memmove(C.byref(q), C.byref(oracle.pattern()), C.sizeof(q))
    for f in ("n_points", "n_short", "n_long", "border"):
        setattr(q, f, getattr(p, f))
    for f in ("px", "py", "sigma_half", "short_i", "short_j", "long_i", "long_j", "long_wdx", "long_wdy"):
        C.memmove(getattr(q, f), getattr(p, f), C.sizeof(getattr(p, f)))
    return q


def test_default_pattern_round_trips_and_equals_the_oracles(oracle):
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    p, o = fe.get_pattern(), oracle.pattern()
    assert p.n_points == o.n_points == 66 and p.n_short == o.n_short == 384 and p.border == o.border
    for f in ("px", "py", "sigma_half", "short_i", "short_j", "long_wdx", "long_wdy"):
        assert bytes(getattr(p, f)) == bytes(getattr(o, f)), f
    fe.set_pattern(p)  # installing what is there changes nothing
    img = synth.corners_image(cfg.w, cfg.h, 31)
    k, d, _, _ = fe.detect_describe(img)
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_GRADIENT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd)


@pytest.mark.parametrize("widen,n,fat_first", [(1.05, 52, False), (0.97, 52, False), (1.05, 66, False),
                                                (0.97, 66, False), (0.97, 66, True), (0.97, 70, False)])
def test_replaced_pattern_stays_bit_exact_in_every_mode(oracle, monkeypatch, widen, n, fat_first):
    # widen 1.05: some boxes exceed 11 x 11 -> the all-modes descriptor kernel (plain-loop box sums); 0.97: every box
    # fits -> the camera-aware-only kernel, which carries the fixed-trip box sum alone (round 4).
    # n > 64 (round 5: the built-in pattern has 66 points): points 0..n-65 are a second pass of lanes 0..n-65 --
    # 5 x 5 slots in the camera-aware-only kernel, which fat_first (a second-pass box wider than that) must leave.
    # n = 70: four synthetic extra points appended (six second-pass samples)
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    rng = np.random.default_rng(5)
    base = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts).get_pattern()
    p = capi.PatternData()
    C.memmove(C.byref(p), C.byref(base), C.sizeof(p))
    # n samples (52: the last 14 of the outer ring dropped), offsets shrunk and rotated by 7 degrees,
    # half-widths changed, pairs re-derived from the kept samples in a permuted bit order with some
    # pairs flipped, at most 300 of them; long pairs re-weighted
    a = np.deg2rad(7.0)
    bx = list(base.px[:66]) + [3.1, -4.2, 0.7, -1.3][:max(n - 66, 0)]
    by = list(base.py[:66]) + [-2.2, 1.9, 5.1, -6.4][:max(n - 66, 0)]
    bs = list(base.sigma_half[:66]) + [2.5, 3.0, 2.2, 1.7][:max(n - 66, 0)]
    px, py = np.array(bx[:n]) * 0.93, np.array(by[:n]) * 0.93
    p.n_points = n
    for i in range(n):
        p.px[i] = np.float32(np.cos(a) * px[i] - np.sin(a) * py[i])
        p.py[i] = np.float32(np.sin(a) * px[i] + np.cos(a) * py[i])
        p.sigma_half[i] = np.float32(bs[i] * (0.9 if i % 3 else widen))
    if fat_first:
        p.sigma_half[1] = np.float32(2.6)
    pairs = [(base.short_i[b], base.short_j[b]) for b in range(base.n_short)
             if base.short_i[b] < n and base.short_j[b] < n]
    pairs += [(i, i - 5) for i in range(66, n)] + [(i - 60, i) for i in range(66, n)]
    order = rng.permutation(len(pairs))[:300]
    p.n_short = len(order)
    for b, o in enumerate(order):
        i, j = pairs[o]
        if b % 5 == 0:
            i, j = j, i
        p.short_i[b], p.short_j[b] = i, j
    for b in range(p.n_short, 384):
        p.short_i[b] = p.short_j[b] = 0
    longs = [(base.long_i[l], base.long_j[l], base.long_wdx[l], base.long_wdy[l]) for l in range(base.n_long)
             if base.long_i[l] < n and base.long_j[l] < n]
    p.n_long = len(longs)
    for l, (i, j, wx, wy) in enumerate(longs):
        p.long_i[l], p.long_j[l], p.long_wdx[l], p.long_wdy[l] = i, j, wx + (l % 3) - 1, wy - (l % 2)
    p.border = base.border
    assert (max(p.sigma_half[:n]) > 4.75) == (widen > 1.0)
    q = _to_orc(oracle, p)
    monkeypatch.setattr(oracle, "pattern", lambda: q)  # oracle.describe / detect_describe read it per call
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       rotation_invariant=True)
    fe.set_camera(0, cam)
    fe.set_pattern(p)
    rays, jac = oracle.awareness_maps(cam)
    img = synth.corners_image(cfg.w, cfg.h, 32)
    k, d, _, _ = fe.detect_describe(img, cam=0, gravity=(0.1, 0.97, -0.1))
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.1, 0.97, -0.1))
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd) and len(k) > 100
    assert np.all(np.unpackbits(d, axis=1, bitorder="little")[:, p.n_short:] == 0)
    k, d, _, _ = fe.detect_describe(img)  # gradient orientation through the new long pairs
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_GRADIENT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd)
    fu = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       rotation_invariant=False)
    fu.set_pattern(p)
    k, d, _, _ = fu.detect_describe(img)
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_UPRIGHT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd)
    # the built-in pattern gives other descriptors: the data really is what the kernel reads
    kb, db, _, _ = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                 rotation_invariant=False).detect_describe(img)
    assert len(kb) == len(k) and not np.array_equal(db, d)


def test_bad_patterns_are_rejected():
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    for edit in ("points", "pair", "border", "sigma"):
        p = fe.get_pattern()
        if edit == "points":
            p.n_points = 61
        elif edit == "pair":
            p.short_i[5] = 70
        elif edit == "border":
            p.border = 5
        else:
            p.sigma_half[3] = 0.0
        with pytest.raises(capi.OkvfeError):
            fe.set_pattern(p)


@pytest.mark.parametrize("factor", [1.3, 1.73, 2.0, 2.3])
def test_widened_builtin_pattern_stays_bit_exact(oracle, monkeypatch, factor):
    """The built-in pattern with every box wider (the vocabulary's statistics favour
    1.7 - 2 x): boxes up to 21 x 21 take the WIDE instantiations of the descriptor kernel (21 x 21 / 10 x 10 row
    slots, six-dword masks), 2.3 x is beyond them (plain box loops); larger patches, a wider rim."""
    import math
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, rotation_invariant=True)
    fe.set_camera(0, cam)
    p = fe.get_pattern()
    reach = 0.0
    for i in range(p.n_points):
        p.sigma_half[i] = np.float32(p.sigma_half[i] * factor)
        reach = max(reach, math.hypot(p.px[i], p.py[i]) + p.sigma_half[i])
    p.border = int(math.ceil(reach)) + 1
    assert p.border > 29 and max(p.sigma_half[:p.n_points]) > 4.75
    fe.set_pattern(p)
    q = _to_orc(oracle, p)
    monkeypatch.setattr(oracle, "pattern", lambda: q)
    rays, jac = oracle.awareness_maps(cam)
    for seed in (41, 42):
        img = synth.corners_image(cfg.w, cfg.h, seed)
        k, d, _, _ = fe.detect_describe(img, cam=0, gravity=(0.05, 0.99, -0.1))
        rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                        oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.05, 0.99, -0.1))
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd) and len(k) > 100
        k, d, _, _ = fe.detect_describe(img)  # gradient orientation
        rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, oracle.MODE_GRADIENT)
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd)


@pytest.mark.parametrize("which", ["all", "mixed", "second_pass_only"])
def test_point_sample_pattern_stays_bit_exact(oracle, monkeypatch, which):
    """Half-widths below 0.5 are bilinear point samples (the published smoothedIntensity's first branch).  The fast
    camera-aware kernels carry the box sum alone, so such a pattern has to take the all-modes kernel, whose bilinear
    read must come AFTER the patch has landed in LDS (the late-wait form read it early).  Twice per image:
    a stale patch would differ between runs."""
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, rotation_invariant=True)
    fe.set_camera(0, cam)
    p = fe.get_pattern()
    extra = p.n_points - 64
    for i in range(p.n_points):
        if which == "all" or (which == "mixed" and i % 3 == 0) or (which == "second_pass_only" and i < extra):
            p.sigma_half[i] = np.float32(0.3 + 0.01 * (i % 17))
    fe.set_pattern(p)
    q = _to_orc(oracle, p)
    monkeypatch.setattr(oracle, "pattern", lambda: q)
    rays, jac = oracle.awareness_maps(cam)
    for seed in (51, 52):
        img = synth.corners_image(cfg.w, cfg.h, seed)
        rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                        oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.05, 0.99, -0.1))
        for _ in range(2):
            k, d, _, _ = fe.detect_describe(img, cam=0, gravity=(0.05, 0.99, -0.1))
            G.assert_keypoints_equal(k, rk)
            assert np.array_equal(d, rd) and len(k) > 100
        k, d, _, _ = fe.detect_describe(img)  # gradient orientation
        rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, oracle.MODE_GRADIENT)
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd)


@pytest.mark.parametrize("box_scale,kernel_class", [(1.0, 0), (1.73, 1), (0.8, 0), (2.3, 2)])
def test_box_scale_config_field(oracle, monkeypatch, box_scale, kernel_class):
    """okvfe_config.box_scale (ABI 7): the built-in pattern with every smoothing box `box_scale` times wider, as a named
    parameter -- the one open parity parameter of the descriptor with evidence on both sides (tools/pattern/README.md).
    The expected pattern is computed HERE (half-side in double x the float factor, rounded once; border follows) and must
    equal what the context installed; 1.73 runs on the WIDE instantiations of the fast kernels."""
    import math
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    base = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts).get_pattern()
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, rotation_invariant=True,
                       box_scale=box_scale)
    fe.set_camera(0, cam)
    p = fe.get_pattern()
    assert fe.pattern_kernel_class() == kernel_class
    f = float(np.float32(box_scale))
    reach = 0.0
    for i in range(base.n_points):
        want = np.float32(float(base.sigma_half[i]) * f) if box_scale != 1.0 else np.float32(base.sigma_half[i])
        assert np.float32(p.sigma_half[i]).view(np.uint32) == want.view(np.uint32), i
        reach = max(reach, math.hypot(float(p.px[i]), float(p.py[i])) + float(want))
    assert p.border == (int(math.ceil(reach)) + 1 if box_scale != 1.0 else base.border)
    assert bytes(p.short_i) == bytes(base.short_i) and bytes(p.px) == bytes(base.px)
    q = _to_orc(oracle, p)
    monkeypatch.setattr(oracle, "pattern", lambda: q)
    rays, jac = oracle.awareness_maps(cam)
    img = synth.corners_image(cfg.w, cfg.h, 61)
    k, d, _, _ = fe.detect_describe(img, cam=0, gravity=(0.05, 0.99, -0.1))
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.05, 0.99, -0.1))
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd) and len(k) > 100
    k, d, _, _ = fe.detect_describe(img)  # gradient orientation
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, oracle.MODE_GRADIENT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd)


def test_box_scale_out_of_range_is_rejected():
    cfg = synth.euroc_config()
    for bad in (0.1, 3.0, -1.0, float("nan")):
        with pytest.raises(capi.OkvfeError):
            capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, box_scale=bad)
