"""GPU: the device-resident, batched map matchers (okvfe_match_to_map_blocks_device,
okvfe_match_to_map_uninitialised_blocks_device, okvfe_verify_place_blocks_device): frame f of a batch
is gather block f in device memory, the pooled landmark set is device-resident, one launch serves all
frames.  Every frame's rows against the oracle's single-frame loops (Frontend.cpp:1552-1589,
1616-1719, 330-355); ragged batches incl. an empty frame; rows past a frame's keypoint count stay
untouched."""
import numpy as np
import pytest

import gpu_common as G
from okvis2_amd import capi, multigpu, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _frames(oracle, cfg, rng, sizes):
    cam = cfg.cams[0]
    out = []
    for n in sizes:
        kps = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
        kps["x"] = rng.uniform(30, 720, n)
        kps["y"] = rng.uniform(30, 450, n)
        desc = rng.integers(0, 256, (n, 48), dtype=np.uint8)
        bp, bv = oracle.backproject_keypoints(cam, kps) if n else (np.zeros((0, 3)), np.zeros(0, np.uint8))
        out.append((kps, desc, bp, bv))
    return out


def test_match_to_map_and_verify_place_blocks_device(oracle):
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    K = fe.max_keypoints
    rng = np.random.default_rng(31)
    sizes = [650, 0, 333, K]
    frames = _frames(oracle, cfg, rng, sizes)
    nf, n_lm = len(frames), 1500
    counts = rng.integers(1, 4, n_lm)
    counts[::19] = 0
    desc_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    pool = rng.integers(0, 256, (desc_begin[-1], 48), dtype=np.uint8)
    proj = np.stack([np.stack([rng.uniform(0, 752, n_lm), rng.uniform(0, 480, n_lm)], 1) for _ in range(nf)])
    use = (rng.random((nf, K)) > 0.15).astype(np.uint8)
    for f, (kps, desc, _, _) in enumerate(frames):  # plant matches: landmark l observes keypoint l of frame f
        for l in range(0, min(len(kps), n_lm), 2):
            if counts[l] == 0:
                continue
            proj[f, l] = (kps["x"][l] + rng.normal(0, 3), kps["y"][l] + rng.normal(0, 3))
            if f == 0:
                d = desc_begin[l] + rng.integers(0, counts[l])
                pool[d] = desc[l] ^ ((rng.random(48) < 0.05) * rng.integers(0, 256, 48)).astype(np.uint8)
    for f in range(1, nf):  # the other frames see (noisy copies of) frame 0's descriptors
        n = min(len(frames[f][0]), len(frames[0][0]))
        frames[f][1][:n] = frames[0][1][:n] ^ ((rng.random((n, 48)) < 0.02) * rng.integers(0, 256, (n, 48))).astype(np.uint8)
    blocks = np.stack([multigpu.pack_block_host(K, *fr) for fr in frames])
    assert blocks.shape[1] == fe.gather_block_bytes()
    d_blocks, d_use = _dev(blocks), _dev(use)
    d_begin, d_pool, d_proj = _dev(desc_begin), _dev(pool), _dev(proj)
    md = fe.make_map_device(n_lm, d_begin.data_ptr(), d_pool.data_ptr(), d_proj.data_ptr())
    st = torch.cuda.Stream()
    for thr in (20.0, 150.0):
        d_lm = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
        d_bd = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        fe.match_to_map_blocks_device(d_blocks.data_ptr(), nf, d_use.data_ptr(), md, thr, d_lm.data_ptr(),
                                      d_bd.data_ptr(), st)
        st.synchronize()
        lm, bd = d_lm.cpu().numpy(), d_bd.cpu().numpy()
        hits = 0
        for f, (kps, desc, _, _) in enumerate(frames):
            n = len(kps)
            rl, rd = oracle.match_to_map(desc, kps, use[f, :n], proj[f], desc_begin, pool, thr, cfg.match_threshold)
            assert np.array_equal(lm[f, :n], rl) and np.array_equal(bd[f, :n], rd), (f, thr)
            assert np.all(lm[f, n:] == -7) and np.all(bd[f, n:] == -7)
            hits += int((rl >= 0).sum())
        assert hits > 150
    # use == NULL: every keypoint takes part
    d_lm = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
    d_bd = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fe.match_to_map_blocks_device(d_blocks.data_ptr(), nf, None, md, 20.0, d_lm.data_ptr(), d_bd.data_ptr(), st)
    st.synchronize()
    kps, desc, _, _ = frames[0]
    rl, rd = oracle.match_to_map(desc, kps, np.ones(len(kps), np.uint8), proj[0], desc_begin, pool, 20.0,
                                 cfg.match_threshold)
    assert np.array_equal(d_lm.cpu().numpy()[0, :len(kps)], rl)
    # verifyRecognisedPlace: all landmarks against every frame
    d_k = torch.full((nf, n_lm), -7, dtype=torch.int32, device="cuda")
    d_d = torch.full((nf, n_lm), 7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fe.verify_place_blocks_device(d_blocks.data_ptr(), nf, md, d_k.data_ptr(), d_d.data_ptr(), st)
    st.synchronize()
    gk, gd = d_k.cpu().numpy(), d_d.cpu().numpy().view(np.uint32)
    below = 0
    for f, (kps, desc, _, _) in enumerate(frames):
        rk, rd = oracle.verify_place(pool, desc_begin, desc, cfg.match_threshold)
        assert np.array_equal(gk[f], rk) and np.array_equal(gd[f], rd), f
        below += int((rd < cfg.match_threshold).sum())
    assert below > 200


def test_match_to_map_uninitialised_blocks_device(oracle):
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    K = fe.max_keypoints
    rng = np.random.default_rng(41)
    n_lm, focal = 600, 0.5 * (cam.fu + cam.fv)
    X = np.stack([rng.uniform(-2, 2, n_lm), rng.uniform(-1, 1, n_lm), rng.uniform(2.5, 10, n_lm)], 1)
    counts = rng.integers(1, 4, n_lm)
    desc_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    m = desc_begin[-1]
    pool = rng.integers(0, 256, (m, 48), dtype=np.uint8)
    lm_desc = rng.integers(0, 256, (n_lm, 48), dtype=np.uint8)
    r0, e0 = np.zeros((m, 3)), np.zeros((m, 3))
    for l in range(n_lm):
        for d in range(desc_begin[l], desc_begin[l + 1]):
            r0[d] = rng.normal(0, 0.3, 3) + np.array([-0.2, 0, 0])
            ray = X[l] - r0[d] + rng.normal(0, 0.002, 3)
            e0[d] = ray / np.linalg.norm(ray)
            if rng.random() < 0.8:
                pool[d] = lm_desc[l] ^ ((rng.random(48) < 0.04) * rng.integers(0, 256, 48)).astype(np.uint8)
    frames, poses, uses, prevs = [], [], [], []
    for f, n_k in enumerate((400, 0, 250)):
        th = 0.03 + 0.01 * f
        Ry = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T1 = (Ry.reshape(-1), np.array([0.25 + 0.05 * f, -0.03, 0.05]))
        Xc = (X[:n_k] - T1[1]) @ Ry
        kps = np.zeros(n_k, dtype=oracle.KEYPOINT_DTYPE)
        for i in range(n_k):
            s, pt, _ = oracle.cam_project(cam, Xc[i])
            kps["x"][i], kps["y"][i] = pt if s == 0 else (9.0, 9.0)
        kps["x"] += rng.normal(0, 0.4, n_k).astype(np.float32)
        kps["y"] += rng.normal(0, 0.4, n_k).astype(np.float32)
        bp, bv = oracle.backproject_keypoints(cam, kps) if n_k else (np.zeros((0, 3)), np.zeros(0, np.uint8))
        desc = lm_desc[:n_k] ^ ((rng.random((n_k, 48)) < 0.02) * rng.integers(0, 256, (n_k, 48))).astype(np.uint8)
        frames.append((kps, desc, bp, bv))
        poses.append(T1)
        u = np.zeros(K, np.uint8)
        u[:n_k] = (bv != 0) & (rng.random(n_k) > 0.1)
        uses.append(u)
        p = np.full(K, -1, np.int32)
        p[:n_k:9] = np.arange(n_k)[::9]
        p[4:n_k:9] = (np.arange(n_k)[4::9] + 1) % n_lm
        prevs.append(p)
    nf = len(frames)
    d_blocks = _dev(np.stack([multigpu.pack_block_host(K, *fr) for fr in frames]))
    d_use, d_prev = _dev(np.stack(uses)), _dev(np.stack(prevs))
    d_begin, d_pool, d_e0, d_r0 = _dev(desc_begin), _dev(pool), _dev(e0), _dev(r0)
    md = fe.make_map_device(n_lm, d_begin.data_ptr(), d_pool.data_ptr(), None, d_e0.data_ptr(), d_r0.data_ptr())
    d_lm = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
    d_bd = torch.full((nf, K), -7, dtype=torch.int32, device="cuda")
    d_hp = torch.zeros((nf, K, 4), dtype=torch.float64, device="cuda")
    d_hs = torch.full((nf, K), 9, dtype=torch.uint8, device="cuda")
    d_ctr = torch.full((nf,), 123, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(2):  # twice: the pose records travel through the parameter ring
        fe.match_to_map_uninitialised_blocks_device(d_blocks.data_ptr(), nf, d_use.data_ptr(), d_prev.data_ptr(), md,
                                                    poses, focal, d_lm.data_ptr(), d_bd.data_ptr(), d_hp.data_ptr(),
                                                    d_hs.data_ptr(), d_ctr.data_ptr(), st)
    st.synchronize()
    lm, bd, hp, hs, ctr = (t.cpu().numpy() for t in (d_lm, d_bd, d_hp, d_hs, d_ctr))
    total = 0
    for f, (kps, desc, bp, bv) in enumerate(frames):
        n = len(kps)
        ref = oracle.match_to_map_uninit(desc, bp, uses[f][:n], prevs[f][:n], desc_begin, pool, e0, r0, poses[f],
                                         focal, cfg.match_threshold)
        assert np.array_equal(lm[f, :n], ref[0]) and np.array_equal(bd[f, :n], ref[1]), f
        assert np.array_equal(hs[f, :n], ref[3])
        assert np.array_equal(hp[f, :n].view(np.uint64), ref[2].view(np.uint64))
        assert ctr[f] == ref[4]
        assert np.all(lm[f, n:] == -7) and np.all(hs[f, n:] == 9)
        total += int((ref[0] >= 0).sum())
    assert total > 150 and ctr[1] == 0
