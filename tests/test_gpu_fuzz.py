"""GPU: seeded random configurations (image size, content, threshold, radius, cap, extraction mode,
scale invariance) through detect + describe against the oracle -- the shapes no hand-written case
names: partial strips and tiles of the fused score kernel, overflowing hit stacks (noise at low
thresholds), flagged plateaus, caps inside a selection window, patterns leaving the image."""
import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu


def _image(rng, w, h, kind):
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "corners":
        return synth.corners_image(w, h, int(rng.integers(1, 10000)), cell=int(rng.choice([8, 12, 20])))
    # blocks: every pixel doubled -> equal-score neighbours (fix-up pass), plus a flat band
    small = synth.corners_image((w + 1) // 2, (h + 1) // 2, int(rng.integers(1, 10000)), cell=6)
    img = np.repeat(np.repeat(small, 2, axis=0), 2, axis=1)[:h, :w].copy()
    img[h // 3:h // 3 + 9, :] = 77
    return img


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.integers(20, 230)) * 4
    h = int(rng.integers(70, 420))
    kind = ["noise", "corners", "blocks"][seed % 3]
    radius = float(rng.choice([6.0, 10.0, 17.5, 26.0, 38.0]))
    thr = int(rng.choice([1, 5, 40, 150, 400]))
    maxk = int(rng.choice([50, 300, 700, 2000]))
    mode = ["upright", "gradient", "upright", "gradient"][seed % 4]
    si = bool(seed % 5 == 0)
    img = _image(rng, w, h, kind)
    fe = capi.Frontend(w, h, radius, 0, thr, maxk, rotation_invariant=(mode == "gradient"), scale_invariant=si,
                       max_candidates=1 << 16)
    omode = oracle.MODE_GRADIENT if mode == "gradient" else oracle.MODE_UPRIGHT
    rk, rd = oracle.detect_describe(img, radius, 0, thr, maxk, omode, scale_invariant=si)
    kps, desc, _, _ = fe.detect_describe(img)
    G.assert_keypoints_equal(kps, rk)
    assert np.array_equal(desc, rd), (w, h, kind, radius, thr, maxk, mode, si)
    det = fe.detect(img)
    ref = oracle.detect(img, radius, 0, thr, maxk)
    G.assert_keypoints_equal(det, ref)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_camera_aware_configuration(oracle, seed):
    """The production extraction mode on random cameras (radial-tangential / equidistant, focal length 0.45 .. 1.3
    image widths, off-centre principal point), random extraction directions, sizes, content, thresholds, radii and
    caps: the camera-aware-only descriptor kernel, map-free detection and the self-ordering selection against the
    oracle.  (tools/fuzz_aware.py runs the same generator over thousands of seeds: 3150 configurations, 0 mismatches.)"""
    rng = np.random.default_rng(5000 + seed)
    w = int(rng.integers(24, 260)) * 4
    h = int(rng.integers(80, 500))
    kind = ["noise", "corners", "blocks"][seed % 3]
    radius = float(rng.choice([10.0, 17.5, 26.0, 38.0, 50.0]))
    thr = int(rng.choice([5, 40, 150, 400]))
    maxk = int(rng.choice([50, 300, 700, 1500]))
    img = _image(rng, w, h, kind)
    dist = int(rng.choice([1, 2]))
    f = float(rng.uniform(0.45, 1.3)) * w
    d = (tuple(rng.uniform(-0.3, 0.1, 1)) + tuple(rng.uniform(-0.05, 0.1, 1)) + tuple(rng.uniform(-2e-3, 2e-3, 2))) \
        if dist == 1 else tuple(rng.uniform(-0.02, 0.02, 4))
    cam = synth.Camera(w, h, f, f * float(rng.uniform(0.97, 1.03)), w / 2 + float(rng.uniform(-8, 8)),
                       h / 2 + float(rng.uniform(-8, 8)), dist, tuple(float(x) for x in d))
    g = rng.normal(0, 1, 3)
    g[1] += 2.0
    g = tuple(float(x) for x in (g / np.linalg.norm(g)).astype(np.float32))
    fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 16)
    fe.set_camera(0, cam)
    rays, jac = oracle.awareness_maps(cam)
    rk, rd = oracle.detect_describe(img, radius, 0, thr, maxk, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), g)
    kps, desc, _, _ = fe.detect_describe(img, cam=0, gravity=g)
    G.assert_keypoints_equal(kps, rk)
    assert np.array_equal(desc, rd), (w, h, kind, radius, thr, maxk, dist)
