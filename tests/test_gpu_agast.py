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
