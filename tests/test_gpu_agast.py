"""GPU: the AGAST 9-16 score calculator (okvfe_config.score_type = OKVFE_SCORE_AGAST_9_16, the detector
score of the reference's ARM branch, okvis_cv/test/TestFrame.cpp:71-72) against the oracle's
restatement of the published predicate: score maps byte-equal, then the whole detector (single scale
and the BriskFeatureDetector(34, 2) scale space), descriptors and a batch."""
import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("w,h,kind", [(752, 480, "corners"), (640, 480, "noise"), (333, 97, "noise"),
                                      (64, 64, "noise"), (1024, 70, "corners"), (70, 66, "noise")])
def test_agast_score_map(oracle, w, h, kind):
    n = 3
    fe = capi.Frontend(w, h, 20.0, 0, 34, 500, max_batch=n, score_type=capi.SCORE_AGAST_9_16)
    imgs = np.stack([synth.noise_image(w, h, 21 + i) if kind == "noise" else synth.corners_image(w, h, 21 + i)
                     for i in range(n)])
    d_img = torch.from_numpy(imgs).cuda()
    d_sc = torch.full((n, h, w), -7, dtype=torch.int32, device="cuda")
    fe.harris_score_device(d_img.data_ptr(), n, d_sc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_sc.cpu().numpy()
    for i in range(n):
        ref = oracle.agast_score(imgs[i])
        assert np.array_equal(got[i], ref), np.argwhere(got[i] != ref)[:5]
    assert got.max() > 34


@pytest.mark.parametrize("w,h,n", [(752, 480, 96), (640, 483, 64), (330, 250, 200)])
def test_agast_score_map_column_walk(oracle, w, h, n):
    """Launches large enough that a workgroup walks several vertically adjacent tiles (double-buffered staging, the
    next tile's loads in flight under the current one's scoring), incl. a partial last tile and the byte-staged
    form (330 px): every replica of the 8 distinct images equals the oracle's map."""
    fe = capi.Frontend(w, h, 20.0, 0, 34, 500, max_batch=n, score_type=capi.SCORE_AGAST_9_16)
    base = np.stack([synth.noise_image(w, h, 61 + i) if i % 2 else synth.corners_image(w, h, 61 + i) for i in range(8)])
    d_img = torch.from_numpy(np.concatenate([base] * (n // 8))).cuda()
    d_sc = torch.full((n, h, w), -7, dtype=torch.int32, device="cuda")
    fe.harris_score_device(d_img.data_ptr(), n, d_sc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.from_numpy(np.stack([oracle.agast_score(base[i]) for i in range(8)])).cuda()
    got = d_sc.view(n // 8, 8, h, w)
    assert bool((got == ref[None]).all()), torch.nonzero(got != ref[None])[:5]


def test_agast_extremes(oracle):
    w, h = 128, 64
    fe = capi.Frontend(w, h, 20.0, 0, 34, 500, max_batch=4, score_type=capi.SCORE_AGAST_9_16)
    chk = (np.indices((h, w)).sum(0) % 2 * 255).astype(np.uint8)
    dot = np.zeros((h, w), np.uint8)
    dot[20, 30] = 255          # bright point: every circle pixel darker by 255 -> score 254
    dot[40, 90] = 7
    imgs = np.stack([np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8), chk, dot])
    d_img = torch.from_numpy(imgs).cuda()
    d_sc = torch.empty((4, h, w), dtype=torch.int32, device="cuda")
    fe.harris_score_device(d_img.data_ptr(), 4, d_sc.data_ptr(), None)
    torch.cuda.synchronize()
    got = d_sc.cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], oracle.agast_score(imgs[i]))
    assert got[0].max() == 0 and got[1].max() == 0
    assert got[3][20, 30] == 254 and got[3][40, 90] == 6


@pytest.mark.parametrize("octaves,radius,maxk", [(0, 20.0, 600), (2, 34.0, 450), (1, 10.0, 300)])
def test_agast_detector_and_descriptors(oracle, octaves, radius, maxk):
    """BriskFeatureDetector(34, 2)-shaped call and two more: threshold 34 on the AGAST score through
    the shared NMS / scale-space / uniformity / sub-pixel pipeline, then descriptors."""
    w, h = 752, 480
    total = 0
    for kind, seed in (("corners", 5), ("noise", 6)):
        img = synth.noise_image(w, h, seed) if kind == "noise" else synth.corners_image(w, h, seed)
        fe = capi.Frontend(w, h, radius, octaves, 34, maxk, rotation_invariant=False,
                           score_type=capi.SCORE_AGAST_9_16, max_candidates=1 << 16)
        ref = oracle.detect(img, radius, octaves, 34, maxk, score_type=oracle.SCORE_AGAST)
        got = fe.detect(img)
        G.assert_keypoints_equal(got, ref)
        k, d = oracle.detect_describe(img, radius, octaves, 34, maxk, oracle.MODE_UPRIGHT, None, None,
                                      np.float32(1.0), (0.0, 1.0, 0.0), score_type=oracle.SCORE_AGAST)
        gk, gd = fe.detect_describe(img)[:2]
        G.assert_keypoints_equal(gk, k)
        assert np.array_equal(gd, d)
        total += len(ref)
        if octaves:
            assert len(np.unique(ref["octave"])) >= 2
    assert total > 100


@pytest.mark.parametrize("octaves,maxk", [(2, 450), (1, 120), (3, 2000)])
def test_brisk_scale_space_detector(oracle, octaves, maxk):
    """score_type OKVFE_SCORE_BRISK_SCALESPACE = brisk::BriskFeatureDetector(34, octaves), the call of
    okvis_cv/test/TestFrame.cpp:71-72, as the published method: AGAST 9-16 on octaves and intra-octaves,
    FAST 5-8 below c0, scale-space maxima, sub-pixel, parabola over the three layers' scores ->
    continuous size.  Keypoints (positions, sizes, responses as float bit patterns, layer) against the
    oracle; descriptors on top (the extractor is not scale invariant: base pattern at the keypoint)."""
    w, h = 752, 480
    total, refined = 0, 0
    for kind, seed in (("corners", 5), ("noise", 6)):
        img = synth.noise_image(w, h, seed) if kind == "noise" else synth.corners_image(w, h, seed)
        fe = capi.Frontend(w, h, 0.0, octaves, 34, maxk, rotation_invariant=False,
                           score_type=capi.SCORE_BRISK_SCALESPACE, max_candidates=1 << 16)
        ref = oracle.detect(img, 0.0, octaves, 34, maxk, score_type=oracle.SCORE_BRISK_SCALESPACE)
        got = fe.detect(img)
        G.assert_keypoints_equal(got, ref)
        assert np.array_equal(got["size"].view(np.uint32), ref["size"].view(np.uint32))
        assert np.array_equal(got["response"].view(np.uint32), ref["response"].view(np.uint32))
        k, d = oracle.detect_describe(img, 0.0, octaves, 34, maxk, oracle.MODE_UPRIGHT, None, None,
                                      np.float32(1.0), (0.0, 1.0, 0.0), score_type=oracle.SCORE_BRISK_SCALESPACE)
        gk, gd = fe.detect_describe(img)[:2]
        G.assert_keypoints_equal(gk, k)
        assert np.array_equal(gd, d)
        total += len(ref)
        layers = np.unique(ref["octave"])
        assert len(layers) >= 2 and layers.max() < 2 * octaves
        for l in layers:  # sizes are continuous: not just 12 x the layer scale
            sc = 12.0 * ((1.5 if l & 1 else 1.0) * 2 ** (l // 2))
            sz = ref["size"][ref["octave"] == l]
            refined += int((np.abs(sz - sc) > 1e-3).sum())
            lo, hi = (2.0 / 3.0, 4.0 / 3.0) if l & 1 else (0.75, 1.5)
            assert np.all(sz >= sc * lo - 1e-3) and np.all(sz <= sc * hi + 1e-3)
            assert np.sum(ref["octave"] == l) <= maxk
    assert total > 200 and refined > 50


def test_fast58_and_scale_refine_oracle_known_answers(oracle):
    """Known answers of the two new oracle pieces: FAST 5-8 (ring of 8, arcs of 5) and the parabola."""
    import ctypes as C
    img = np.full((9, 9), 50, np.uint8)
    img[4, 4] = 120  # isolated bright pixel: all 8 ring pixels darker by 70 -> score 69
    out = np.zeros((9, 9), np.int32)
    oracle.lib().orc_fast58_score(img.ctypes.data_as(C.c_void_p), 9, 9, 9, out.ctypes.data_as(C.c_void_p))
    assert out[4, 4] == 69 and out[0, 0] == 0 and out[4, 5] == 0
    img[:] = 50
    img[:, 5:] = 90  # straight edge: only 3 contiguous ring pixels differ -> no corner on either side
    oracle.lib().orc_fast58_score(img.ctypes.data_as(C.c_void_p), 9, 9, 9, out.ctypes.data_as(C.c_void_p))
    assert out[4, 4] == 0 and out[4, 5] == 0
    img[:] = 50
    img[5:, 5:] = 90  # a bright quadrant: its corner pixel sees 5 contiguous ring pixels darker by 40
    oracle.lib().orc_fast58_score(img.ctypes.data_as(C.c_void_p), 9, 9, 9, out.ctypes.data_as(C.c_void_p))
    assert out[5, 5] == 39 and out[4, 4] == 0
    rel, sc = C.c_float(), C.c_float()
    f = oracle.lib().orc_scale_refine
    f.argtypes = [C.c_double, C.c_int, C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_int32, C.c_double,
                  C.POINTER(C.c_float), C.POINTER(C.c_float)]
    f(0.75, 1, 40, 50, 1.5, 1, 40, 0.75, C.byref(rel), C.byref(sc))  # symmetric in value, not in abscissa
    assert 1.0 < rel.value < 1.5 and sc.value >= 50.0
    f(0.75, 1, 60, 50, 1.5, 1, 20, 0.75, C.byref(rel), C.byref(sc))  # rising towards the layer below: clamped there
    assert rel.value == pytest.approx(0.75) and sc.value == pytest.approx(60.0)
    f(0.75, 1, 10, 50, 1.5, 0, 0, 0.75, C.byref(rel), C.byref(sc))   # top layer: no refinement
    assert rel.value == 1.0 and sc.value == 50.0
    f(2.0 / 3.0, 1, 60, 50, 1.5, 1, 20, 0.7, C.byref(rel), C.byref(sc))  # layer 0: node at 2/3, clamp at 0.7
    assert rel.value == pytest.approx(0.7) and 50.0 < sc.value < 60.0


def test_agast_batch(oracle):
    cfg = synth.mono640_config()
    n = 9
    fe = capi.Frontend(cfg.w, cfg.h, 15.0, 0, 34, 500, rotation_invariant=False, max_batch=n,
                       score_type=capi.SCORE_AGAST_9_16)
    imgs = np.stack([np.full((cfg.h, cfg.w), 40, np.uint8) if i == 3 else
                     (synth.noise_image(cfg.w, cfg.h, 70 + i) if i % 2 else synth.corners_image(cfg.w, cfg.h, 70 + i))
                     for i in range(n)])
    d_img = torch.from_numpy(imgs).cuda()
    fe.detect_describe_batch_device(d_img.data_ptr(), n, None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fe.check_capacity(n)
    for i in range(n):
        k, d = oracle.detect_describe(imgs[i], 15.0, 0, 34, 500, oracle.MODE_UPRIGHT, None, None,
                                      np.float32(1.0), (0.0, 1.0, 0.0), score_type=oracle.SCORE_AGAST)
        g = fe.download(i)
        G.assert_keypoints_equal(g[0], k)
        assert np.array_equal(g[1], d)
    assert len(fe.download(3)[0]) == 0
