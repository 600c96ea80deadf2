"""GPU parity of scale_invariant = true extraction (brisk::BriskDescriptorExtractor(rotInv, true),
Frontend.cpp:2410-2412) against the oracle's scale ladder: detect + describe at octaves 0 and 2,
and okvfe_compute on keypoints of arbitrary diameters (every path of the kernel: fixed-trip boxes,
boxes above 11 px, banded patches, direct reads)."""
import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu


def _mode(oracle, name):
    return {"upright": oracle.MODE_UPRIGHT, "gradient": oracle.MODE_GRADIENT, "aware": oracle.MODE_CAMERA_AWARE}[name]


@pytest.mark.parametrize("mode", ["upright", "gradient", "aware"])
@pytest.mark.parametrize("octaves", [0, 2])
def test_detect_describe_scale_invariant(oracle, mode, octaves):
    cfg = synth.euroc_config()
    cfg.octaves = octaves
    fe = G.make_frontend(cfg, rotation_invariant=(mode != "upright"), scale_invariant=True)
    cam = cfg.cams[0]
    rays = jac = None
    kw = {}
    if mode == "aware":
        fe.set_camera(0, cam)
        rays, jac = oracle.awareness_maps(cam)
        kw = dict(cam=0, gravity=(0.2, 0.95, -0.1))
    for seed in (3, 4):
        img = G.image_for(cfg, seed)
        rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, octaves, cfg.abs_threshold, cfg.max_kpts,
                                        _mode(oracle, mode), rays, jac, np.float32(cam.fu),
                                        kw.get("gravity", (0.0, 1.0, 0.0)), scale_invariant=True)
        kps, desc, bp, bpv = fe.detect_describe(img, **kw)
        G.assert_keypoints_equal(kps, rk)
        assert np.array_equal(desc, rd)
        assert len(kps) > 50
        if octaves:
            assert len(np.unique(kps["size"])) > 1


@pytest.mark.parametrize("mode", ["upright", "gradient", "aware"])
def test_compute_arbitrary_sizes(oracle, mode):
    cfg = synth.mono640_config()
    fe = G.make_frontend(cfg, rotation_invariant=(mode != "upright"), scale_invariant=True)
    cam = cfg.cams[0]
    rays = jac = None
    kw = {}
    if mode == "aware":
        fe.set_camera(0, cam)
        rays, jac = oracle.awareness_maps(cam)
        kw = dict(cam=0, gravity=(0.0, 1.0, 0.0))
    img = G.image_for(cfg, 8)
    kps = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    sizes = np.array([5.0, 7.2, 9.0, 12.0, 17.4, 18.0, 24.0, 31.0, 36.0, 48.0, 72.0, 96.0, 150.0, 400.0], dtype=np.float32)
    kps["size"] = sizes[np.arange(len(kps)) % len(sizes)]
    rk, rd = oracle.describe(img, kps, _mode(oracle, mode), rays, jac, np.float32(cam.fu),
                             kw.get("gravity", (0.0, 1.0, 0.0)), scale_invariant=True)
    gk, gd, bp, bpv = fe.compute(img, kps, **kw)
    G.assert_keypoints_equal(gk, rk)
    assert np.array_equal(gd, rd)
    used = {capi.scale_index(s) for s in gk["size"]}
    assert len(used) >= 8 and len(gk) < len(kps)  # large patterns are removed near the rim


def test_basic_size_equals_fixed_scale(oracle):
    cfg = synth.euroc_config()
    fixed = G.make_frontend(cfg)
    si = G.make_frontend(cfg, scale_invariant=True)
    img = G.image_for(cfg, 12)
    kps = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    k0, d0, _, _ = fixed.compute(img, kps)
    kps17 = kps.copy()
    kps17["size"] = np.float32(1.45 * 12.0)
    k1, d1, _, _ = si.compute(img, kps17)
    assert len(k0) == len(k1) and np.array_equal(d0, d1)


def test_installed_pattern_is_the_base_of_the_scale_ladder(oracle, monkeypatch):
    """okvfe_set_pattern on a scale-invariant context: the installed pattern is index 17 of the ladder,
    the other 63 scales are rebuilt from it (PatternScales), exactly as orc_pattern_scaled does."""
    import ctypes as C
    from test_gpu_pattern import _to_orc
    cfg = synth.euroc_config()
    cfg.octaves = 2
    fe = G.make_frontend(cfg, rotation_invariant=True, scale_invariant=True)
    base = fe.get_pattern()
    p = capi.PatternData()
    C.memmove(C.byref(p), C.byref(base), C.sizeof(p))
    for i in range(p.n_points):
        p.px[i] = np.float32(base.px[i] * 0.9)
        p.py[i] = np.float32(base.py[i] * 0.9)
        p.sigma_half[i] = np.float32(base.sigma_half[i] * (1.1 if i % 2 else 0.95))
    fe.set_pattern(p)
    q = _to_orc(oracle, p)
    monkeypatch.setattr(oracle, "pattern", lambda: q)
    img = G.image_for(cfg, 14)
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 2, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_GRADIENT, scale_invariant=True)
    kps, desc, _, _ = fe.detect_describe(img)
    G.assert_keypoints_equal(kps, rk)
    assert np.array_equal(desc, rd) and len(kps) > 50 and len(np.unique(kps["size"])) > 1
