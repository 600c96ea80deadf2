"""GPU: the two multi-camera BASELINE configs end to end, against the oracle.

 * configs[4] Hilti 2022: five equidistant 720x540 cameras with the REAL extrinsics of
   config/hilti_challenge_2022.yaml:3-71; the 9 FoV-overlapping pairs of Frontend.cpp:1990-2000 are
   derived with okvfe_camera_overlap; one context per camera, device-side gather blocks, an RCCL
   all-gather (nccl, world size 1 on the single-GPU box) and one matcher launch per pair -- all
   through okvis2_amd.multigpu.CrossCameraMatcher, the class the multi-GPU bench mode drives.
 * configs[3] TUM-VI 1024x1024 equidistant STEREO: both cameras + matchStereo through the
   device-resident batch API.
Every keypoint, descriptor, back-projection and match row is compared with the oracle's own
pipeline (nothing of the GPU's output is fed to the oracle)."""
import os
import socket

import numpy as np
import pytest

import gpu_common as G
from okvis2_amd import capi, multigpu, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle_camera(oracle, cfg, ci, img, grav):
    cam = cfg.cams[ci]
    rays, jac = oracle.awareness_maps(cam)
    k, d = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                  oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                                  tuple(float(v) for v in grav))
    bp, bv = oracle.backproject_keypoints(cam, k)
    return k, d, bp, bv


def test_hilti_rig_five_cameras_nine_pairs_cross_camera_matcher(oracle):
    import torch.distributed as dist
    cfg = synth.hilti_config()
    pairs = synth.rig_overlap_pairs(cfg, capi.camera_overlap)
    assert pairs == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4)]
    assert pairs == synth.rig_overlap_pairs(cfg, oracle.cam_overlap)
    nfr = 3
    # a tilted, displaced sensor pose per frame: different gravity directions and poses per frame
    rays = [capi.build_awareness_maps(c)[0] for c in cfg.cams]
    frames, poses_f = [], []
    for f in range(nfr):
        a = 0.2 * f
        C_WS = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        frames.append(synth.render_rig(cfg, rays, 40 + f, r_S=np.array([0.1 * f, 0.0, 0.05 * f])))
        poses_f.append(synth.rig_poses(cfg, C_WS, np.array([0.1 * f, 0.0, 0.05 * f])))
    # the matcher takes ONE pose per camera and launch: use frame 0's rig pose for all frames (the
    # gate only needs a consistent T_WC0/T_WC1 pair; per-frame gravity still differs)
    poses = poses_f[0]
    focal = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        engines = {}
        for c in range(5):  # one context per camera (Frontend.cpp:2405-2413)
            engines[c] = G.make_frontend(cfg, max_batch=nfr, num_cameras=1)
            engines[c].set_camera(0, cfg.cams[c])
        ccm = multigpu.CrossCameraMatcher(engines, 5, nfr, poses, focal, lambda i, j: (i, j) in pairs,
                                          1, 0, "cuda:0")
        assert ccm.mine == pairs and ccm.slots == 5
        d_img = {c: torch.from_numpy(np.stack([frames[f][c] for f in range(nfr)])).cuda()
                 for c in range(5)}
        grav = {c: np.stack([synth.gravity_in_camera(poses_f[f][c][0]) for f in range(nfr)])
                for c in range(5)}
        for _ in range(2):  # twice: the second step must not race with the first one's collective
            gathered, out = ccm.step({c: d_img[c].data_ptr() for c in range(5)}, grav)
        ccm.finish()
        for c in range(5):
            engines[c].check_capacity(nfr)
        host = gathered.cpu().numpy()
        ref = [[_oracle_camera(oracle, cfg, c, frames[f][c], grav[c][f]) for c in range(5)]
               for f in range(nfr)]
        for f in range(nfr):
            for c in range(5):
                k, d, bp, bv = multigpu.unpack_block_host(host[c, f], cfg.max_kpts)
                rk, rd, rbp, rbv = ref[f][c]
                G.assert_keypoints_equal(k, rk)
                assert np.array_equal(d, rd) and len(k) > 50
                assert np.array_equal(bp.view(np.uint64), rbp.view(np.uint64))
                assert np.array_equal(bv, rbv)
        matched = {p: 0 for p in pairs}
        for (i, j) in pairs:
            m = out[(i, j)].cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, cfg.max_kpts)
            for f in range(nfr):
                (k0, d0, b0, v0), (k1, d1, b1, v1) = ref[f][i], ref[f][j]
                want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, poses[i], poses[j],
                                           focal[i], focal[j], cfg.match_threshold)
                assert np.array_equal(m[f, :len(k0)].view(np.uint8), want.view(np.uint8)), (i, j, f)
                matched[(i, j)] += int((want["k1"] >= 0).sum())
        assert matched[(0, 1)] > 50 and sum(matched.values()) > 150
        assert sum(1 for v in matched.values() if v > 0) >= 7  # the side/up pairs match too
    finally:
        dist.destroy_process_group()


def test_hilti_rig_at_bench_size_replicas_anchor_idempotence(oracle):
    """The five-camera rig at the size `bench.py --workload hilti` steps through (hundreds of multiframes per call,
    VERDICT r5 weak #13): 288 multiframes = 3 distinct rendered rig frames x 96 replicas through the CrossCameraMatcher
    (five contexts, gather blocks, RCCL all-gather at world 1, nine pair launches).  Properties that need no oracle at
    this size -- every replica equals the first occurrence of its frame byte for byte (blocks and match rows), a second
    step reproduces the first -- plus the anchor: the three distinct frames against the oracle."""
    import torch.distributed as dist
    cfg = synth.hilti_config()
    pairs = synth.rig_overlap_pairs(cfg, capi.camera_overlap)
    distinct, nfr = 3, 288
    rays = [capi.build_awareness_maps(c)[0] for c in cfg.cams]
    frames, poses_f = [], []
    for f in range(distinct):
        a = 0.2 * f
        C_WS = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        frames.append(synth.render_rig(cfg, rays, 140 + f, r_S=np.array([0.1 * f, 0.0, 0.05 * f])))
        poses_f.append(synth.rig_poses(cfg, C_WS, np.array([0.1 * f, 0.0, 0.05 * f])))
    poses = poses_f[0]
    focal = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        engines = {}
        for c in range(5):
            engines[c] = G.make_frontend(cfg, max_batch=nfr, num_cameras=1)
            engines[c].set_camera(0, cfg.cams[c])
        ccm = multigpu.CrossCameraMatcher(engines, 5, nfr, poses, focal, lambda i, j: (i, j) in pairs, 1, 0, "cuda:0")
        d_img = {c: torch.from_numpy(np.stack([frames[f % distinct][c] for f in range(nfr)])).cuda() for c in range(5)}
        grav = {c: np.stack([synth.gravity_in_camera(poses_f[f % distinct][c][0]) for f in range(nfr)]) for c in range(5)}
        ptrs = {c: d_img[c].data_ptr() for c in range(5)}
        digests = []
        for _ in range(2):
            gathered, out = ccm.step(ptrs, grav)
            ccm.finish()
            for c in range(5):
                engines[c].check_capacity(nfr)
            host = gathered.cpu().numpy()
            rows = {p: out[p].cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, cfg.max_kpts) for p in pairs}
            digests.append((host.tobytes(), {p: rows[p].tobytes() for p in pairs}))
        # idempotence (rows beyond a frame's keypoint count are never written, so whole buffers compare)
        assert digests[0][0] == digests[1][0] and digests[0][1] == digests[1][1]
        # replicas
        n_kp = 0
        for f in range(distinct, nfr):
            b = f % distinct
            for c in range(5):  # (rows of a block beyond the image's keypoint count are unspecified: compare the contents)
                for x, y in zip(multigpu.unpack_block_host(host[c, f], cfg.max_kpts),
                                multigpu.unpack_block_host(host[c, b], cfg.max_kpts)):
                    assert x.tobytes() == y.tobytes(), (c, f)
            n0 = [len(multigpu.unpack_block_host(host[i, b], cfg.max_kpts)[0]) for i in range(5)]
            for (i, j) in pairs:
                assert rows[(i, j)][f, :n0[i]].tobytes() == rows[(i, j)][b, :n0[i]].tobytes(), (i, j, f)
            n_kp += sum(n0)
        assert n_kp > 5 * 50 * (nfr - distinct)
        # anchor
        ref = [[_oracle_camera(oracle, cfg, c, frames[f][c], grav[c][f]) for c in range(5)] for f in range(distinct)]
        matched = 0
        for f in range(distinct):
            for c in range(5):
                k, d, bp, bv = multigpu.unpack_block_host(host[c, f], cfg.max_kpts)
                rk, rd, rbp, rbv = ref[f][c]
                G.assert_keypoints_equal(k, rk)
                assert np.array_equal(d, rd) and np.array_equal(bp.view(np.uint64), rbp.view(np.uint64))
                assert np.array_equal(bv, rbv)
            for (i, j) in pairs:
                (k0, d0, b0, v0), (k1, d1, b1, v1) = ref[f][i], ref[f][j]
                want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, poses[i], poses[j], focal[i], focal[j],
                                           cfg.match_threshold)
                assert np.array_equal(rows[(i, j)][f, :len(k0)].view(np.uint8), want.view(np.uint8)), (i, j, f)
                matched += int((want["k1"] >= 0).sum())
        assert matched > 150
    finally:
        dist.destroy_process_group()


def test_hilti_rig_cpp_cross_camera_matcher_rccl_through_c_abi(oracle, tmp_path):
    """The same rig through the C++ host class okvfe::CrossCameraMatcher
    (okvis2_amd/host/okvfe_cross_camera.hpp, tests/cpp/cross_camera_cli.cpp): communicator from
    okvfe_comm_unique_id / okvfe_comm_create (ncclCommInitRank with one rank), the gather issued from C
    (okvfe_gather_blocks = ncclAllGather), everything on one stream, two steps; blocks and match rows
    against the oracle."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "tests", "cpp", "cross_camera_cli")
    assert os.path.exists(cli), "run __graft_entry__.build() first"
    cfg = synth.hilti_config()
    pairs = synth.rig_overlap_pairs(cfg, capi.camera_overlap)
    nfr = 2
    rays = [capi.build_awareness_maps(c)[0] for c in cfg.cams]
    frames, poses_f = [], []
    for f in range(nfr):
        a = 0.15 * f
        C_WS = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        frames.append(synth.render_rig(cfg, rays, 60 + f, r_S=np.array([0.1 * f, 0.0, 0.0])))
        poses_f.append(synth.rig_poses(cfg, C_WS, np.array([0.1 * f, 0.0, 0.0])))
    poses = poses_f[0]
    focal = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    grav = {c: np.stack([synth.gravity_in_camera(poses_f[f][c][0]) for f in range(nfr)]).astype(np.float32)
            for c in range(5)}
    req, resp = tmp_path / "req.bin", tmp_path / "resp.bin"
    with open(req, "wb") as f:
        f.write(struct.pack("<iiii", cfg.w, cfg.h, 5, nfr))
        f.write(struct.pack("<f", cfg.uniformity_radius))
        f.write(struct.pack("<iii", cfg.abs_threshold, cfg.match_threshold, cfg.max_kpts))
        ov = np.zeros((5, 5), np.uint8)
        for (i, j) in pairs:
            ov[i, j] = 1
        f.write(ov.tobytes())
        for c in range(5):
            cam = cfg.cams[c]
            f.write(struct.pack("<4d", cam.fu, cam.fv, cam.cu, cam.cv))
            f.write(struct.pack("<i", cam.dist_type))
            f.write(struct.pack("<4d", *cam.d))
            f.write(np.asarray(poses[c][0], dtype=np.float64).tobytes())
            f.write(np.asarray(poses[c][1], dtype=np.float64).tobytes())
            f.write(grav[c].tobytes())
            f.write(np.stack([frames[fr][c] for fr in range(nfr)]).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = (os.path.join(root, "okvis2_amd") + ":" + os.path.dirname(torch.__file__) + "/lib:/opt/rocm/lib:"
                              + env.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([cli, "run", str(req), str(resp)], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    buf = open(resp, "rb").read()
    n_pairs, cap = struct.unpack_from("<ii", buf, 0)
    off = 8
    assert n_pairs == len(pairs) and cap == cfg.max_kpts
    rows = {}
    msz = capi.STEREO_MATCH_DTYPE.itemsize
    for _ in range(n_pairs):
        i, j = struct.unpack_from("<ii", buf, off)
        off += 8
        rows[(i, j)] = np.frombuffer(buf, capi.STEREO_MATCH_DTYPE, nfr * cap, off).reshape(nfr, cap)
        off += nfr * cap * msz
    bb = multigpu.block_layout(cap)["total"]
    ref = [[_oracle_camera(oracle, cfg, c, frames[f][c], grav[c][f]) for c in range(5)] for f in range(nfr)]
    for c in range(5):
        blocks = np.frombuffer(buf, np.uint8, nfr * bb, off).reshape(nfr, bb)
        off += nfr * bb
        for f in range(nfr):
            k, d, bp, bv = multigpu.unpack_block_host(blocks[f], cap)
            rk, rd, rbp, rbv = ref[f][c]
            G.assert_keypoints_equal(k, rk)
            assert np.array_equal(d, rd) and np.array_equal(bv, rbv)
            assert np.array_equal(bp.view(np.uint64), rbp.view(np.uint64))
    assert off == len(buf)
    total = 0
    for (i, j) in pairs:
        for f in range(nfr):
            (k0, d0, b0, v0), (k1, d1, b1, v1) = ref[f][i], ref[f][j]
            want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, poses[i], poses[j], focal[i], focal[j],
                                       cfg.match_threshold)
            assert np.array_equal(rows[(i, j)][f, :len(k0)].view(np.uint8), want.view(np.uint8)), (i, j, f)
            total += int((want["k1"] >= 0).sum())
    assert total > 100


def test_tumvi_1024_stereo_detect_describe_match(oracle):
    """config/tumvi_slam_1024.yaml: both equidistant 1024x1024 cameras + matchStereo, 2 stereo
    frames through okvfe_detect_describe_batch_device / okvfe_match_stereo_batch_device."""
    cfg = synth.tumvi1024_config()
    nfr = 2
    fe = G.make_frontend(cfg, max_batch=2 * nfr, num_cameras=2)
    for ci in range(2):
        fe.set_camera(ci, cfg.cams[ci])
    imgs = []
    for f in range(nfr):
        L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 910 + f)
        imgs += [L, R]
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    cam_ids = np.array([0, 1] * nfr, dtype=np.int32)
    gr = np.array([[0.05 * f, 0.99, -0.1] for f in range(nfr) for _ in range(2)], dtype=np.float32)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), 2 * nfr, cam_ids, gr, st)
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    pairs = []
    for f in range(nfr):
        sp = capi.StereoPair()
        sp.image0, sp.image1 = 2 * f, 2 * f + 1
        sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
        sp.f0, sp.f1 = f0, f1
        pairs.append(sp)
    d_m = torch.zeros((nfr, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8,
                      device="cuda")
    torch.cuda.synchronize()
    fe.match_stereo_batch_device(pairs, d_m.data_ptr(), st)
    st.synchronize()
    fe.check_capacity(2 * nfr)
    m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, cfg.max_kpts)
    for f in range(nfr):
        ref = [_oracle_camera(oracle, cfg, ci, imgs[2 * f + ci], gr[2 * f + ci]) for ci in range(2)]
        for ci in range(2):
            k, d, bp, bv = fe.download(2 * f + ci)
            G.assert_keypoints_equal(k, ref[ci][0])
            assert np.array_equal(d, ref[ci][1]) and len(k) > 300
            assert np.array_equal(bp.view(np.uint64), ref[ci][2].view(np.uint64))
            assert np.array_equal(bv, ref[ci][3])
        (k0, d0, b0, v0), (k1, d1, b1, v1) = ref
        want = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, cfg.match_threshold)
        assert np.array_equal(m[f, :len(k0)].view(np.uint8), want.view(np.uint8))
        assert (want["k1"] >= 0).sum() > 50
