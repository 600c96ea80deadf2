"""GPU: the scale space of the detector (octaves > 0) against the oracle (oracle/orc_detect.c,
detect_scale_space).  The reference's own detector call uses octaves = 2:
brisk::ScaleSpaceFeatureDetector<HarrisScoreCalculator>(34, 2, 800, 450) on a 752x480 iid-uniform
image (okvis_cv/test/TestFrame.cpp:75-85) -- reproduced here with the extractor of the same test,
cv::BriskDescriptorExtractor(true, false)."""
import numpy as np
import pytest

import gpu_common as G
from okvis2_amd import capi, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_reference_smoke_test_call_octaves2(oracle):
    img = synth.noise_image(752, 480, 0x0C0FFEE0)
    fe = capi.Frontend(752, 480, 34.0, 2, 800, 450, rotation_invariant=True, scale_invariant=False)
    assert fe.max_keypoints == 4 * 450
    ref = oracle.detect(img, 34.0, 2, 800, 450)
    got = fe.detect(img)
    G.assert_keypoints_equal(got, ref)
    assert set(np.unique(ref["octave"])) == {0, 1, 2, 3}
    assert np.array_equal(np.unique(ref["size"]), [12.0, 18.0, 24.0, 36.0])
    k, d, _, _ = fe.detect_describe(img)
    rk, rd = oracle.describe(img, ref, oracle.MODE_GRADIENT)
    G.assert_keypoints_equal(k, rk)
    assert np.array_equal(d, rd) and len(k) > 300


@pytest.mark.parametrize("w,h,octaves,seed", [(752, 480, 1, 5), (640, 480, 2, 6), (1024, 1024, 3, 7),
                                              (333, 201, 2, 8)])
def test_scale_space_sizes_and_batches(oracle, w, h, octaves, seed):
    """Corner images at several sizes (incl. layer widths that are not multiples of 4, which take
    the generic score / NMS kernels) through the device-resident batch path."""
    B = 3
    fe = capi.Frontend(w, h, 30.0, octaves, 100, 300, max_batch=B, max_candidates=0)
    imgs = np.stack([synth.corners_image(w, h, seed + 10 * i) for i in range(B)])
    d_img = torch.from_numpy(imgs).cuda()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), B, None, None, st)
    st.synchronize()
    fe.check_capacity(B)
    for i in range(B):
        rk, rd = oracle.detect_describe(imgs[i], 30.0, octaves, 100, 300, oracle.MODE_GRADIENT)
        k, d, _, _ = fe.download(i)
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd)
        assert len(np.unique(rk["octave"])) >= 2


def test_scale_space_stereo_match_uses_size_classes(oracle):
    """matchStereo's triangulation sigma = max(size0/f0, size1/f1) * 0.125 (Frontend.cpp:2031-2035)
    depends on the keypoint sizes, which differ per layer: device-resident batch matcher and the
    host-buffer matcher against the oracle."""
    cfg = synth.euroc_config()
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 2, cfg.abs_threshold, 300,
                       match_threshold=cfg.match_threshold, num_cameras=2, max_batch=2)
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 4242)
    ref = []
    for ci, img in enumerate((L, R)):
        cam = cfg.cams[ci]
        fe.set_camera(ci, cam)
        rays, jac = oracle.awareness_maps(cam)
        k, d = oracle.detect_describe(img, cfg.uniformity_radius, 2, cfg.abs_threshold, 300,
                                      oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.0, 1.0, 0.0))
        bp, bv = oracle.backproject_keypoints(cam, k)
        ref.append((k, d, bp, bv))
    d_img = torch.from_numpy(np.stack([L, R])).cuda()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), 2, np.array([0, 1], dtype=np.int32),
                                    np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (2, 1)), st)
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    sp = capi.StereoPair()
    sp.image0, sp.image1 = 0, 1
    sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
    sp.f0, sp.f1 = f0, f1
    d_m = torch.zeros((fe.max_keypoints, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe.match_stereo_batch_device([sp], d_m.data_ptr(), st)
    st.synchronize()
    (k0, d0, b0, v0), (k1, d1, b1, v1) = ref
    for ci in range(2):
        g = fe.download(ci)
        G.assert_keypoints_equal(g[0], ref[ci][0])
        assert np.array_equal(g[1], ref[ci][1])
        assert np.array_equal(g[2].view(np.uint64), ref[ci][2].view(np.uint64))
    want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, cfg.match_threshold)
    m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(-1)[:len(k0)]
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8))
    got = fe.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    matched = want["k1"] >= 0
    assert matched.sum() > 30 and len(np.unique(k0["octave"][matched])) >= 2


def test_scale_space_layers_side_by_side_stay_exact(oracle):
    """The layers of a scale-space call run on streams of their own (round 6), so the map-writing score kernel of one layer
    shares the GPU with the kernels of the others.  That exposed a store-data hazard of the score kernel (a 16-byte store at
    the end of a basic block, its first data register overwritten by the next block: LAB_NOTES "Round 6") which made about
    every second call of this configuration store a few wrong score-map entries -- and move sub-pixel positions in layer 2.
    Ten calls on fresh contexts, each against the oracle."""
    w, h, octaves, B = 1024, 1024, 3, 3
    imgs = np.stack([synth.corners_image(w, h, 7 + 10 * i) for i in range(B)])
    want = [oracle.detect_describe(imgs[i], 30.0, octaves, 100, 300, oracle.MODE_GRADIENT) for i in range(B)]
    d_img = torch.from_numpy(imgs).cuda()
    for _ in range(10):
        fe = capi.Frontend(w, h, 30.0, octaves, 100, 300, max_batch=B, max_candidates=0)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        fe.detect_describe_batch_device(d_img.data_ptr(), B, None, None, st)
        st.synchronize()
        fe.check_capacity(B)
        for i in range(B):
            k, d, _, _ = fe.download(i)
            G.assert_keypoints_equal(k, want[i][0])
            assert np.array_equal(d, want[i][1])
        fe.close()
