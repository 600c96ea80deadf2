"""GPU: (1) the HIP path against the COMMITTED golden vectors (tests/golden/frontend_golden.npz),
(2) the cross-camera gather path: device-side block packing, an RCCL all-gather (nccl backend,
world size 1 on the single-GPU box) and matching on gathered blocks, against the oracle."""
import os
import socket

import numpy as np
import pytest

from okvis2_amd import capi, multigpu, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz")


def _gold():
    g = np.load(GOLDEN)
    W, H, radius, thr, maxk, mthr = g["params"]
    cams = [synth.Camera(int(W), int(H), c[0], c[1], c[2], c[3], int(c[4]), tuple(c[5:9]))
            for c in g["cams"]]
    return g, int(W), int(H), float(radius), int(thr), int(maxk), int(mthr), cams


def test_hip_path_reproduces_golden_vectors():
    g, W, H, radius, thr, maxk, mthr, cams = _gold()
    for rot, name in ((False, "upright"), (True, "gradient")):
        fe = capi.Frontend(W, H, radius, 0, thr, maxk, rotation_invariant=rot, match_threshold=mthr)
        for ci, key in enumerate(("left", "right")):
            assert np.array_equal(fe.detect(g[key]).view(np.uint8), g[f"kp_detect_{ci}"].view(np.uint8))
            k, d, _, _ = fe.detect_describe(g[key])
            assert np.array_equal(k.view(np.uint8), g[f"kp_{name}_{ci}"].view(np.uint8))
            assert np.array_equal(d, g[f"desc_{name}_{ci}"])
    fe = capi.Frontend(W, H, radius, 0, thr, maxk, match_threshold=mthr, num_cameras=2)
    res = []
    for ci, key in enumerate(("left", "right")):
        fe.set_camera(ci, cams[ci])
        k, d, bp, bv = fe.detect_describe(g[key], cam=ci, gravity=(0.1, 0.98, -0.05))
        assert np.array_equal(k.view(np.uint8), g[f"kp_aware_{ci}"].view(np.uint8))
        assert np.array_equal(d, g[f"desc_aware_{ci}"])
        assert np.array_equal(bp.view(np.uint64), g[f"bp_{ci}"].view(np.uint64))
        assert np.array_equal(bv, g[f"bpv_{ci}"])
        res.append((k, d, bp, bv))
    T0, T1 = synth.stereo_poses(0.11)
    f0, f1 = 0.5 * (cams[0].fu + cams[0].fv), 0.5 * (cams[1].fu + cams[1].fv)
    (k0, d0, b0, v0), (k1, d1, b1, v1) = res
    m = fe.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1)
    assert np.array_equal(m.view(np.uint8), g["match_stereo"].view(np.uint8))
    # noise frame (the reference's smoke-test INPUT kind, TestFrame.cpp:83-85) with its radius /
    # threshold / count (34, 800, 450) but octaves = 0; the reference's own call passes octaves = 2
    # (TestFrame.cpp:75-77) -- that case is tests/test_gpu_octaves.py
    fe2 = capi.Frontend(W, H, 34.0, 0, 800, 450)
    assert np.array_equal(fe2.detect(g["noise"]).view(np.uint8), g["kp_noise"].view(np.uint8))
    # score map
    d_img = torch.from_numpy(g["left"]).cuda()
    d_sc = torch.empty((H, W), dtype=torch.int32, device="cuda")
    fe.harris_score_device(d_img.data_ptr(), 1, d_sc.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(d_sc.cpu().numpy(), g["score_left"])


def test_gather_blocks_rccl_and_block_matching(oracle):
    import torch.distributed as dist
    g, W, H, radius, thr, maxk, mthr, cams = _gold()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        fe = capi.Frontend(W, H, radius, 0, thr, maxk, match_threshold=mthr, num_cameras=2, max_batch=2)
        for ci in range(2):
            fe.set_camera(ci, cams[ci])
        imgs = torch.from_numpy(np.stack([g["left"], g["right"]])).cuda()
        grav = np.tile(np.array([0.1, 0.98, -0.05], dtype=np.float32), (2, 1))
        stream = torch.cuda.current_stream()  # torch default stream -> OKVFE_STREAM_LEGACY_DEFAULT
        fe.detect_describe_batch_device(imgs.data_ptr(), 2, np.array([0, 1], dtype=np.int32), grav, stream)
        nb = fe.gather_block_bytes()
        assert nb == multigpu.block_layout(maxk)["total"]
        local = torch.zeros((2, nb), dtype=torch.uint8, device="cuda")
        for i in range(2):
            fe.pack_gather_block_device(i, local[i].data_ptr(), stream)
        allb = multigpu.all_gather_blocks(local)  # RCCL all-gather, world size 1
        assert allb.shape == (1, 2, nb)
        host = allb.cpu().numpy()
        for ci in range(2):
            k, d, bp, bv = multigpu.unpack_block_host(host[0, ci], maxk)
            assert np.array_equal(k.view(np.uint8), g[f"kp_aware_{ci}"].view(np.uint8))
            assert np.array_equal(d, g[f"desc_aware_{ci}"])
            assert np.array_equal(bp.view(np.uint64), g[f"bp_{ci}"].view(np.uint64))
            assert np.array_equal(bv, g[f"bpv_{ci}"])
        T0, T1 = synth.stereo_poses(0.11)
        f0, f1 = 0.5 * (cams[0].fu + cams[0].fv), 0.5 * (cams[1].fu + cams[1].fv)
        d_m = torch.zeros((maxk, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        fe.match_stereo_blocks_device(allb[0, 0].data_ptr(), allb[0, 1].data_ptr(), T0, T1, f0, f1,
                                      d_m.data_ptr(), stream)
        torch.cuda.synchronize()
        n0 = len(g["kp_aware_0"])
        m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(-1)[:n0]
        assert np.array_equal(m.view(np.uint8), g["match_stereo"].view(np.uint8))
    finally:
        dist.destroy_process_group()


def test_batched_block_pack_and_match(oracle):
    """okvfe_pack_gather_blocks_device / okvfe_match_stereo_blocks_batch_device: the whole batch is
    packed by one kernel and every frame of a camera pair is matched by one launch (the path
    okvis2_amd.multigpu.CrossCameraMatcher drives on every rank)."""
    cfg = synth.euroc_config()
    nfr = 3
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                       match_threshold=cfg.match_threshold, num_cameras=2, max_batch=2 * nfr)
    for ci in range(2):
        fe.set_camera(ci, cfg.cams[ci])
    pairs = [synth.stereo_pair(cfg.w, cfg.h, 300 + i) for i in range(nfr)]
    # camera-major layout: images 0..nfr-1 = camera 0, nfr..2nfr-1 = camera 1
    imgs = np.stack([p[0] for p in pairs] + [p[1] for p in pairs])
    d_img = torch.from_numpy(imgs).cuda()
    cam_ids = np.array([0] * nfr + [1] * nfr, dtype=np.int32)
    grav = np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (2 * nfr, 1))
    stream = torch.cuda.current_stream()  # torch default stream -> OKVFE_STREAM_LEGACY_DEFAULT
    fe.detect_describe_batch_device(d_img.data_ptr(), 2 * nfr, cam_ids, grav, stream)
    nb = fe.gather_block_bytes()
    blocks = torch.zeros((2, nfr, nb), dtype=torch.uint8, device="cuda")
    fe.pack_gather_blocks_device(0, nfr, blocks[0].data_ptr(), stream)
    fe.pack_gather_blocks_device(nfr, nfr, blocks[1].data_ptr(), stream)
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    d_m = torch.zeros((nfr, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8,
                      device="cuda")
    fe.match_stereo_blocks_batch_device(blocks[0].data_ptr(), blocks[1].data_ptr(), nfr, T0, T1, f0, f1,
                                        d_m.data_ptr(), stream)
    torch.cuda.synchronize()
    host = blocks.cpu().numpy()
    m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, cfg.max_kpts)
    for f in range(nfr):
        k0, d0, b0, v0 = multigpu.unpack_block_host(host[0, f], cfg.max_kpts)
        k1, d1, b1, v1 = multigpu.unpack_block_host(host[1, f], cfg.max_kpts)
        g0 = fe.download(f)
        assert np.array_equal(k0.view(np.uint8), g0[0].view(np.uint8)) and np.array_equal(d0, g0[1])
        ref = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, cfg.match_threshold)
        assert np.array_equal(m[f, :len(k0)].view(np.uint8), ref.view(np.uint8))
        assert (ref["k1"] >= 0).sum() > 10
