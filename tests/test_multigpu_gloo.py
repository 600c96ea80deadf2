"""N > 1 host logic on CPU: world_size-2 gloo processes exercise frame sharding, the gather-block
wire format and the cross-camera all-gather + pair schedule.  The matcher run on
the gathered blocks is the CPU oracle here (checker); on the GPU box the same blocks go through
okvfe_match_stereo_blocks_device (tests/test_gpu_multigpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from okvis2_amd import multigpu, synth

torch = pytest.importorskip("torch")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz")


def test_shard_range_partitions():
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [multigpu.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pair_schedule_covers_overlapping_pairs_once():
    # Hilti rig: 5 cameras, every pair overlaps except (3, 4)
    def overlap(i, j):
        return {i, j} != {3, 4}
    for world in (1, 2, 5, 8):
        sched = multigpu.pair_schedule(5, overlap, world)
        assert sorted((i, j) for i, j, _ in sched) == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3),
                                                       (1, 4), (2, 3), (2, 4)]
        assert all(0 <= o < world and o == (i + j) % world for i, j, o in sched)


def test_gather_block_roundtrip():
    g = np.load(GOLDEN)
    cap = 300
    b = multigpu.pack_block_host(cap, g["kp_aware_0"], g["desc_aware_0"], g["bp_0"], g["bpv_0"])
    assert len(b) == multigpu.block_layout(cap)["total"] and len(b) % 256 == 0
    k, d, bp, bv = multigpu.unpack_block_host(b, cap)
    assert np.array_equal(k, g["kp_aware_0"]) and np.array_equal(d, g["desc_aware_0"])
    assert np.array_equal(bp, g["bp_0"]) and np.array_equal(bv, g["bpv_0"])
    with pytest.raises(ValueError):
        multigpu.pack_block_host(10, g["kp_aware_0"], g["desc_aware_0"], g["bp_0"], g["bpv_0"])
    # empty camera
    e = multigpu.pack_block_host(cap, g["kp_aware_0"][:0], g["desc_aware_0"][:0], g["bp_0"][:0],
                                 g["bpv_0"][:0])
    assert len(multigpu.unpack_block_host(e, cap)[0]) == 0


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(GOLDEN)
        cap, frames = 300, 3
        # rank r owns camera r; every frame carries the same golden content plus a frame tag
        blocks = []
        for f in range(frames):
            kp = g[f"kp_aware_{rank}"].copy()
            kp["class_id"] = f
            blocks.append(multigpu.pack_block_host(cap, kp, g[f"desc_aware_{rank}"], g[f"bp_{rank}"],
                                                   g[f"bpv_{rank}"]))
        local = torch.from_numpy(np.stack(blocks))
        allb = multigpu.all_gather_blocks(local).numpy()
        assert allb.shape == (world, frames, multigpu.block_layout(cap)["total"])
        sched = multigpu.pair_schedule(world, lambda i, j: True, world)
        mine = [(i, j) for i, j, o in sched if o == rank]
        T0, T1 = synth.stereo_poses(0.11)
        cams = g["cams"]
        f0, f1 = 0.5 * (cams[0][0] + cams[0][1]), 0.5 * (cams[1][0] + cams[1][1])
        done = []
        for (i, j) in mine:
            for f in range(frames):
                k0, d0, b0, v0 = multigpu.unpack_block_host(allb[i, f], cap)
                k1, d1, b1, v1 = multigpu.unpack_block_host(allb[j, f], cap)
                assert np.all(k0["class_id"] == f) and np.all(k1["class_id"] == f)
                m = O.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, 60)
                assert np.array_equal(m["k1"], g["match_stereo"]["k1"])
                assert np.array_equal(m["hp_W"].view(np.uint64), g["match_stereo"]["hp_W"].view(np.uint64))
                done.append((i, j, f))
        # weak-scaling shard of independent frames: max-over-ranks timing reduction works
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        dist.barrier()
        q.put((rank, done))
    finally:
        dist.destroy_process_group()


def _rig_reference(cfg, imgs, poses, pairs):
    """Every overlapping pair matched by the oracle directly (no gather, no schedule)."""
    import oracle_lib as O
    res = []
    for ci, cam in enumerate(cfg.cams):
        rays, jac = O.awareness_maps(cam)
        g = synth.gravity_in_camera(poses[ci][0])
        k, d = O.detect_describe(imgs[ci], cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                 O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                                 tuple(float(v) for v in g))
        bp, bv = O.backproject_keypoints(cam, k)
        res.append((k, d, bp, bv))
    f = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    out = {}
    for (i, j) in pairs:
        (k0, d0, b0, v0), (k1, d1, b1, v1) = res[i], res[j]
        out[(i, j)] = O.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, poses[i], poses[j], f[i], f[j],
                                     cfg.match_threshold)
    return res, out


def _hilti_worker(rank, world, port, q):
    """One rank of the Hilti 5-camera rig: okvis2_amd.multigpu.CrossCameraMatcher itself (schedule,
    slots, gather, per-pair launches) driven over gloo, with the oracle-backed host engine."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import oracle_lib as O
    from host_engine import OracleEngine
    from okvis2_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synth.hilti_config()
        cfg.max_kpts = 200
        pairs = synth.rig_overlap_pairs(cfg, O.cam_overlap)
        overlap = lambda i, j: (i, j) in pairs  # noqa: E731
        poses = synth.rig_poses(cfg)
        focal = [0.5 * (c.fu + c.fv) for c in cfg.cams]
        imgs = synth.render_rig(cfg, [O.awareness_maps(c)[0] for c in cfg.cams], 7)
        local = [c for c in range(5) if multigpu.camera_owner(c, world) == rank]
        engines = {c: OracleEngine(cfg, cfg.cams[c]) for c in local}
        ccm = multigpu.CrossCameraMatcher(engines, 5, 1, poses, focal, overlap, world, rank, "cpu")
        assert ccm.slots == 3 and ccm.local_cams == local
        held = {c: torch.from_numpy(imgs[c][None].copy()) for c in local}
        grav = {c: synth.gravity_in_camera(poses[c][0])[None, :] for c in local}
        gathered, out = ccm.step({c: held[c].data_ptr() for c in local}, grav)
        ccm.finish()
        # every engine ran detect+describe and pack exactly once; only engine 0 of the rank matches
        for c in local:
            kinds = [k for k, _ in engines[c].calls]
            assert kinds.count("detect_describe") == 1 and kinds.count("pack") == 1
        # all five cameras' blocks arrived, addressed through block_of
        counts = [int(ccm.block_of(gathered, c)[0, :4].numpy().view(np.int32)[0]) for c in range(5)]
        q.put((rank, counts, {p: out[p].numpy().copy() for p in ccm.mine}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_hilti_rig_cross_camera_matcher_world2(oracle):
    """BASELINE configs[4] on CPU: the real Hilti extrinsics (hilti_challenge_2022.yaml:3-71) give 9
    FoV-overlapping pairs; two gloo ranks own cameras {0,2,4} / {1,3}, gather, and match the pairs
    of their static schedule; the union must equal the oracle matching every pair directly."""
    import torch.multiprocessing as mp
    cfg = synth.hilti_config()
    cfg.max_kpts = 200
    pairs = synth.rig_overlap_pairs(cfg, oracle.cam_overlap)
    assert pairs == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4)]
    poses = synth.rig_poses(cfg)
    imgs = synth.render_rig(cfg, [oracle.awareness_maps(c)[0] for c in cfg.cams], 7)
    res, want = _rig_reference(cfg, imgs, poses, pairs)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hilti_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    done = {}
    for rank, counts, out in got:
        assert counts == [len(r[0]) for r in res]
        for pair, rows in out.items():
            assert pair not in done and (pair[0] + pair[1]) % 2 == rank
            done[pair] = rows
    assert sorted(done) == pairs
    matched = 0
    for pair in pairs:
        m = done[pair].view(capi_dtype()).reshape(-1)[:len(want[pair])]
        assert np.array_equal(m.view(np.uint8), want[pair].view(np.uint8)), pair
        matched += int((want[pair]["k1"] >= 0).sum())
    assert matched > 50


def capi_dtype():
    from okvis2_amd import capi
    return capi.STEREO_MATCH_DTYPE


def test_cross_camera_gather_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pairs_done = sorted(x for _, d in res for x in d)
    # the single pair (0, 1) is owned by rank (0 + 1) % 2 = 1, once per frame
    assert pairs_done == [(0, 1, 0), (0, 1, 1), (0, 1, 2)]
    assert dict(res)[0] == []
