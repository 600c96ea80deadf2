"""GPU parity on REAL content: crops of the reference's only real camera image
(okvis_multisensor_processing/test/testImage.jpg, decoded into tests/golden/real_image.npz by
tools/make_real_image_fixture.py): 28 % saturated pixels, JPEG 8x8 blocks, a checkerboard whose
inner corners look alike.  The HIP path (through the C ABI) must equal BOTH the committed vectors
and the oracle run live, in all three extractor modes, at every BASELINE shape, on the full
1280x960 frame and on an odd 1024x960 one; matchStereo on a shifted pair of the same scene."""
import hashlib

import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G
import real_image_cases as RC

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _sha(k, d):
    return np.frombuffer(hashlib.sha256(k.tobytes() + d.tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("case", RC.CASES, ids=[c.name for c in RC.CASES])
def test_real_crop_all_modes(oracle, case):
    fx = RC.load()
    img = RC.crop(fx["image"], case)
    common = dict(width=case.w, height=case.h, uniformity_radius=case.radius, octaves=0,
                  absolute_threshold=case.thr, max_keypoints=case.max_kpts)
    # score map (dense view) and detector
    fe = capi.Frontend(**common)
    d_img = torch.from_numpy(img).cuda()
    d_sc = torch.empty((1, case.h, case.w), dtype=torch.int32, device="cuda")
    fe.harris_score_device(d_img.data_ptr(), 1, d_sc.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(d_sc.cpu().numpy()[0], oracle.harris_score(img))
    kd = fe.detect(img)
    G.assert_keypoints_equal(kd, fx[f"{case.name}/kp_detect"])
    G.assert_keypoints_equal(kd, oracle.detect(img, case.radius, 0, case.thr, case.max_kpts))
    # camera-aware (production mode) + FP64 back-projection, fused detect+describe call
    fe.set_camera(0, case.cam)
    k, d, bp, bv = fe.detect_describe(img, cam=0, gravity=RC.GRAVITY)
    G.assert_keypoints_equal(k, fx[f"{case.name}/kp_aware"])
    assert np.array_equal(d, fx[f"{case.name}/desc_aware"])
    assert np.array_equal(bp.view(np.uint64), fx[f"{case.name}/bp"].view(np.uint64))
    assert np.array_equal(bv, fx[f"{case.name}/bpv"])
    # the same through cv::DescriptorExtractor::compute's twin (separate detect, then compute)
    k2, d2, _, _ = fe.compute(img, kd, cam=0, gravity=RC.GRAVITY)
    G.assert_keypoints_equal(k2, k)
    assert np.array_equal(d2, d)
    # gradient-oriented and upright extraction
    for rot, mode, name in ((True, oracle.MODE_GRADIENT, "gradient"), (False, oracle.MODE_UPRIGHT, "upright")):
        fr = capi.Frontend(rotation_invariant=rot, **common)
        k, d, _, _ = fr.detect_describe(img)
        rk, rd = oracle.describe(img, kd, mode)
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd)
        assert np.array_equal(_sha(k, d), fx[f"{case.name}/sha_{name}"])
        assert len(k) == int(fx[f"{case.name}/n_{name}"])
        fr.close()
    fe.close()


def test_real_stereo_pair(oracle):
    fx = RC.load()
    cfg = synth.euroc_config()
    L, R = RC.stereo_images(fx["image"])
    fe = capi.Frontend(cfg.w, cfg.h, 20.0, 0, 20, cfg.max_kpts, match_threshold=cfg.match_threshold,
                       max_batch=2, num_cameras=2)
    for ci, cam in enumerate(cfg.cams):
        fe.set_camera(ci, cam)
    sides = []
    for ci, img in enumerate((L, R)):
        sides.append(fe.detect_describe(img, cam=ci, gravity=(0.0, 1.0, 0.0)))
    ref = RC.stereo_sides(oracle, fx["image"])
    for got, want in zip(sides, ref):
        G.assert_keypoints_equal(got[0], want[0])
        assert np.array_equal(got[1], want[1])
        assert np.array_equal(got[2].view(np.uint64), want[2].view(np.uint64))
        assert np.array_equal(got[3], want[3])
    G.assert_keypoints_equal(sides[1][0], fx["stereo/kp1"])
    assert np.array_equal(sides[1][1], fx["stereo/desc1"])
    T0, T1, f0, f1, _ = RC.stereo_geometry()
    (k0, d0, b0, v0), (k1, d1, b1, v1) = sides
    m = fe.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1)
    want = fx["stereo/match"]
    for f in ("k1", "dist", "initialisable"):
        assert np.array_equal(m[f], want[f]), f
    assert np.array_equal(m["hp_W"].view(np.uint64), want["hp_W"].view(np.uint64))
    # and the device-resident form: both images in one batch, matcher on the batch's outputs
    d_img = torch.from_numpy(np.stack([L, R])).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    fe.detect_describe_batch_device(d_img.data_ptr(), 2, np.array([0, 1], np.int32),
                                    np.tile(np.array([0.0, 1.0, 0.0], np.float32), (2, 1)), stream)
    sp = capi.StereoPair()
    sp.image0, sp.image1 = 0, 1
    sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
    sp.f0, sp.f1 = f0, f1
    d_m = torch.zeros((1, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    fe.match_stereo_batch_device([sp], d_m.data_ptr(), stream)
    torch.cuda.synchronize()
    got = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(cfg.max_kpts)[:len(k0)]
    for f in ("k1", "dist", "initialisable"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["hp_W"].view(np.uint64), want["hp_W"].view(np.uint64))
