"""GPU: the B = 1 seams (cv::FeatureDetector::detect / cv::DescriptorExtractor::compute,
Frame.hpp:152,167; Frontend::detectAndDescribe, Frontend.cpp:221-269) after the round-4 latency
work: results leave through ONE pinned block and one synchronisation, the image moves by a copy
kernel, and okvfe_detect_ahead answers the okvfe_compute that follows on the same image without GPU
work.  Every variant must stay bit-exact -- including every way the pairing can MISS."""
import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ref(oracle, cfg, cam, img, grav, kps=None):
    rays, jac = oracle.awareness_maps(cam)
    if kps is None:
        kps = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    k, d = oracle.describe(img, kps, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), grav)
    bp, bv = oracle.backproject_keypoints(cam, k)
    return kps, k, d, bp, bv


def _check(got, want):
    k, d, bp, bv = got
    G.assert_keypoints_equal(k, want[1])
    assert np.array_equal(d, want[2])
    assert np.array_equal(bp.view(np.uint64), want[3].view(np.uint64))
    assert np.array_equal(bv, want[4])


@pytest.mark.parametrize("mk", [synth.euroc_config, synth.hilti_config])
def test_detect_ahead_then_compute(oracle, mk):
    cfg = mk()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    fe.set_camera(0, cam)
    g0, g1 = (0.0, 1.0, 0.0), (0.3, 0.9, -0.2)
    img = G.image_for(cfg, 41)
    want = _ref(oracle, cfg, cam, img, g0)
    # hit: same buffer, same set-up, keypoints untouched
    kd = fe.detect_ahead(img, cam=0, gravity=g0)
    G.assert_keypoints_equal(kd, want[0])
    _check(fe.compute(fe._ahead_image, kd, cam=0, gravity=g0), want)
    _check(fe.compute(fe._ahead_image, kd, cam=0, gravity=g0), want)  # idempotent
    # miss 1: another extraction direction
    _check(fe.compute(fe._ahead_image, kd, cam=0, gravity=g1), _ref(oracle, cfg, cam, img, g1, kd))
    # miss 2: a subset of the keypoints (a caller filtering between detect and compute)
    kd = fe.detect_ahead(img, cam=0, gravity=g0)
    sub = kd[::2].copy()
    _check(fe.compute(fe._ahead_image, sub, cam=0, gravity=g0), _ref(oracle, cfg, cam, img, g0, sub))
    # miss 3: the pixels changed under the same pointer
    kd = fe.detect_ahead(img, cam=0, gravity=g0)
    buf = fe._ahead_image
    other = G.image_for(cfg, 42)
    buf[...] = other
    _check(fe.compute(buf, kd, cam=0, gravity=g0), _ref(oracle, cfg, cam, other, g0, kd))
    # miss 4: another buffer with the same content is still answered correctly (full path)
    kd = fe.detect_ahead(other, cam=0, gravity=g0)
    _check(fe.compute(other.copy(), kd, cam=0, gravity=g0), _ref(oracle, cfg, cam, other, g0, kd))
    # not camera-aware: gradient-oriented descriptors, no back-projection
    kd = fe.detect_ahead(img)
    k, d, bp, bv = fe.compute(fe._ahead_image, kd)
    rk, rd = oracle.describe(img, kd, oracle.MODE_GRADIENT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd) and not bv.any()
    # plain calls after a pairing do not see stale state
    G.assert_keypoints_equal(fe.detect(other), oracle.detect(other, cfg.uniformity_radius, 0, cfg.abs_threshold,
                                                             cfg.max_kpts))
    _check(fe.detect_describe(img, cam=0, gravity=g1), _ref(oracle, cfg, cam, img, g1))


def test_strided_images_and_empty_results(oracle):
    """cv::Mat rows may be padded (step > cols); an image without corners gives n = 0 everywhere."""
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    fe.set_camera(0, cam)
    img = G.image_for(cfg, 43)
    wide = np.zeros((cfg.h, cfg.w + 24), dtype=np.uint8)
    wide[:, :cfg.w] = img
    view = wide[:, :cfg.w]  # strides[0] = w + 24
    assert view.strides[0] == cfg.w + 24
    import ctypes as C
    cap = fe.max_keypoints
    kps = np.zeros(cap, dtype=capi.KEYPOINT_DTYPE)
    desc = np.zeros((cap, 48), np.uint8)
    bp = np.zeros((cap, 3))
    bv = np.zeros(cap, np.uint8)
    n = C.c_int32()
    g = (C.c_float * 3)(0.0, 1.0, 0.0)
    st = capi.lib().okvfe_detect_describe(fe._h, C.c_void_p(wide.ctypes.data), C.c_size_t(cfg.w + 24), 0, g,
                                          C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data),
                                          C.c_void_p(bp.ctypes.data), C.c_void_p(bv.ctypes.data), cap, C.byref(n))
    assert st == 0
    want = _ref(oracle, cfg, cam, img, (0.0, 1.0, 0.0))
    _check((kps[:n.value], desc[:n.value], bp[:n.value], bv[:n.value]), want)
    flat = np.full((cfg.h, cfg.w), 90, np.uint8)
    assert len(fe.detect(flat)) == 0
    k, d, _, _ = fe.detect_describe(flat, cam=0, gravity=(0.0, 1.0, 0.0))
    assert len(k) == 0 and len(d) == 0
    kd = fe.detect_ahead(flat, cam=0, gravity=(0.0, 1.0, 0.0))
    assert len(kd) == 0
    k, d, _, _ = fe.compute(fe._ahead_image, kd, cam=0, gravity=(0.0, 1.0, 0.0))
    assert len(k) == 0
