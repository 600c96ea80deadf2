"""GPU: the C++ host mirror (okvis2_amd/host/okvfe_frontend.hpp: HipBriskDetector /
HipBriskExtractor / HipFrontend::detectAndDescribe / matchStereo) driven from a C++ program with
one thread per camera, compared byte for byte with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from okvis2_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tests", "cpp", "frontend_cli")


def test_cpp_frontend_matches_oracle(oracle, tmp_path):
    assert os.path.exists(CLI), "run __graft_entry__.build() first"
    cfg = synth.euroc_config()
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 77)
    T0, T1 = synth.stereo_poses(cfg.baseline)
    req = tmp_path / "req.bin"
    with open(req, "wb") as f:
        f.write(struct.pack("<iii", cfg.w, cfg.h, 2))
        f.write(struct.pack("<f", cfg.uniformity_radius))
        f.write(struct.pack("<iii", cfg.abs_threshold, cfg.match_threshold, cfg.max_kpts))
        for cam, T, img in ((cfg.cams[0], T0, L), (cfg.cams[1], T1, R)):
            f.write(struct.pack("<4d", cam.fu, cam.fv, cam.cu, cam.cv))
            f.write(struct.pack("<i", cam.dist_type))
            f.write(struct.pack("<4d", *cam.d))
            f.write(np.asarray(T[0], dtype=np.float64).tobytes())
            f.write(np.asarray(T[1], dtype=np.float64).tobytes())
            f.write(img.tobytes())
    resp = tmp_path / "resp.bin"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "okvis2_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    subprocess.check_call([CLI, str(req), str(resp)], env=env)
    buf = open(resp, "rb").read()
    off = 0
    got = []
    for _ in range(2):
        n = struct.unpack_from("<i", buf, off)[0]
        off += 4
        k = np.frombuffer(buf, capi.KEYPOINT_DTYPE, n, off); off += 28 * n
        d = np.frombuffer(buf, np.uint8, 48 * n, off).reshape(n, 48); off += 48 * n
        bp = np.frombuffer(buf, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
        bv = np.frombuffer(buf, np.uint8, n, off); off += n
        got.append((k, d, bp, bv))
    n0 = struct.unpack_from("<i", buf, off)[0]
    off += 4
    m = np.frombuffer(buf, capi.STEREO_MATCH_DTYPE, n0, off)
    # identity pose: gravity in the camera frame = C^T (0,0,-1) = (0,0,-1)
    ref = []
    for ci, img in enumerate((L, R)):
        cam = cfg.cams[ci]
        rays, jac = oracle.awareness_maps(cam)
        k, d = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                      oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                                      (0.0, 0.0, -1.0))
        bp, bv = oracle.backproject_keypoints(cam, k)
        ref.append((k, d, bp, bv))
        assert np.array_equal(got[ci][0].view(np.uint8), k.view(np.uint8))
        assert np.array_equal(got[ci][1], d)
        assert np.array_equal(got[ci][2].view(np.uint64), bp.view(np.uint64))
        assert np.array_equal(got[ci][3], bv)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    (k0, d0, b0, v0), (k1, d1, b1, v1) = ref
    want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, cfg.match_threshold)
    assert np.array_equal(m.view(np.uint8), want.view(np.uint8))


def test_detect_then_compute_equals_detect_describe(oracle):
    cfg = synth.euroc_config()
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    fe.set_camera(0, cfg.cams[0])
    img = synth.corners_image(cfg.w, cfg.h, 12)
    a = fe.detect_describe(img, cam=0, gravity=(0.2, 0.9, 0.1))
    kd = fe.detect(img)
    b = fe.compute(img, kd, cam=0, gravity=(0.2, 0.9, 0.1))
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    # compute() on zero keypoints and on keypoints that all sit on the rim
    assert len(fe.compute(img, kd[:0])[0]) == 0
    rim = kd[:5].copy()
    rim["x"] = 3.0
    assert len(fe.compute(img, rim)[0]) == 0


def test_cv_adapters_and_vi_frontend_interface_run():
    """tests/cpp/adapters_check: okvfe::cv_adapters::HipDetector / HipExtractor through their
    cv::Feature2D base pointers and okvfe::HipViFrontend through okvis::ViFrontendInterface (both
    against the compile-check declarations of tests/mock/), on the GPU."""
    exe = os.path.join(ROOT, "tests", "cpp", "adapters_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "okvis2_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapters ok" in out.stdout
