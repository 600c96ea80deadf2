"""GPU, at the size BASELINE.json's metric is quoted on: ONE batch of 768 EuRoC-shaped stereo frames
(1536 images of 752x480) through detect + describe + matchStereo, checked through properties that
do not need the oracle at full size:

* replicas: the batch holds 16 distinct stereo frames, each 48 times; all replicas of a frame
  (same image, same camera, same extraction direction, same poses) must give byte-identical
  keypoints, descriptors, back-projections and match rows, wherever they sit in the batch
  (different workgroups, XCDs, packed groups, sort / select slots);
* anchor: the 16 distinct frames equal the CPU oracle (seconds on the host);
* permutation: the same multiframes in reversed batch order give the reversed results;
* idempotence: a second run over the same device buffers gives the same checksum of checksums.
"""
import hashlib
import os
import sys

import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DISTINCT = 16


def _run(fe, cfg, d_img, n_frames, grav, pairs, d_match):
    cam_ids = np.array([0, 1] * n_frames, dtype=np.int32)
    s = torch.cuda.current_stream().cuda_stream
    fe.detect_describe_batch_device(d_img.data_ptr(), 2 * n_frames, cam_ids, grav, s)
    fe.match_stereo_batch_device(pairs, d_match.data_ptr(), s)
    torch.cuda.synchronize()
    fe.check_capacity(2 * n_frames)
    res = [fe.download(i) for i in range(2 * n_frames)]
    rows = d_match.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(n_frames, -1)
    matches = [rows[f, :len(res[2 * f][0])].copy() for f in range(n_frames)]
    return res, matches


def _digest(res, matches):
    h = hashlib.sha256()
    for (k, d, bp, bv) in res:
        for a in (k, d, bp, bv):
            h.update(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest())
    for m in matches:
        h.update(hashlib.sha256(np.ascontiguousarray(m).tobytes()).digest())
    return h.hexdigest()


@pytest.mark.parametrize("workload,FRAMES,keep_map", [("euroc", 768, False), ("tumvi", 256, False),
                                                      ("euroc", 3072, False), ("euroc", 3072, True)])
def test_full_size_batch_replicas_permutation_idempotence(oracle, workload, FRAMES, keep_map):
    """euroc: BASELINE configs[2], 768 stereo frames of 752x480 and the 3072 of bench.py's default step (image
    buffer 2.2 GB > 2^31 B, 6144-image XCD streams of the score kernel; with keep_map the 9 GB score map of
    okvfe_set_keep_score_map as well); tumvi: configs[3] (256 stereo frames of 1024x1024, equidistant cameras:
    packed last strips of the score kernel, occupancy grid of radius 50, up to 1000 keypoints per image)."""
    import bench
    cfg = synth.euroc_config() if workload == "euroc" else synth.tumvi1024_config()
    imgs, base = bench.make_inputs(cfg, FRAMES, DISTINCT, 4242)
    fe = G.make_frontend(cfg, max_batch=2 * FRAMES, num_cameras=2)
    if keep_map:
        fe.set_keep_score_map(True)
    for ci, cam in enumerate(cfg.cams):
        fe.set_camera(ci, cam)
    # one extraction direction per DISTINCT frame and camera, repeated with the frame
    g16 = np.stack([[0.03 * ((i % 5) - 2), 1.0, 0.02 * ((i % 3) - 1)] for i in range(2 * DISTINCT)])
    g16 = (g16 / np.linalg.norm(g16, axis=1, keepdims=True)).astype(np.float32)
    grav = np.concatenate([g16] * (FRAMES // DISTINCT))
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f = [0.5 * (c.fu + c.fv) for c in cfg.cams]

    def make_pairs(n):
        arr = []
        for i in range(n):
            sp = capi.StereoPair()
            sp.image0, sp.image1 = 2 * i, 2 * i + 1
            sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
            sp.f0, sp.f1 = f[0], f[1]
            arr.append(sp)
        return (capi.StereoPair * n)(*arr)

    pairs = make_pairs(FRAMES)
    d_img = torch.from_numpy(imgs).cuda()
    d_match = torch.zeros((FRAMES, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8,
                          device="cuda")
    res, matches = _run(fe, cfg, d_img, FRAMES, grav, pairs, d_match)

    # ---- replicas
    total_kp = 0
    for fr in range(DISTINCT, FRAMES):
        b = fr % DISTINCT
        for c in range(2):
            for x, y in zip(res[2 * fr + c], res[2 * b + c]):
                assert x.tobytes() == y.tobytes(), (fr, c)
        assert matches[fr].tobytes() == matches[b].tobytes(), fr
        total_kp += len(res[2 * fr][0])
    assert total_kp > 150 * (FRAMES - DISTINCT)

    # ---- anchor: the distinct frames against the oracle
    maps = [oracle.awareness_maps(c) for c in cfg.cams]
    n_match = 0
    for b in range(DISTINCT):
        side = []
        for c in range(2):
            k, d = oracle.detect_describe(base[2 * b + c], cfg.uniformity_radius, 0, cfg.abs_threshold,
                                          cfg.max_kpts, oracle.MODE_CAMERA_AWARE, maps[c][0], maps[c][1],
                                          np.float32(cfg.cams[c].fu), tuple(float(v) for v in g16[2 * b + c]))
            bp, bv = oracle.backproject_keypoints(cfg.cams[c], k)
            gk, gd, gbp, gbv = res[2 * b + c]
            G.assert_keypoints_equal(gk, k)
            assert np.array_equal(gd, d)
            assert np.array_equal(gbp.view(np.uint64), bp.view(np.uint64)) and np.array_equal(gbv, bv)
            side.append((k, d, bp, bv))
        (k0, d0, b0, v0), (k1, d1, b1, v1) = side
        m = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f[0], f[1], cfg.match_threshold)
        assert matches[b].tobytes() == np.ascontiguousarray(m).tobytes(), b
        n_match += int((m["k1"] >= 0).sum())
    assert n_match > 30 * DISTINCT

    # ---- idempotence
    first = _digest(res, matches)
    res2, matches2 = _run(fe, cfg, d_img, FRAMES, grav, pairs, d_match)
    assert _digest(res2, matches2) == first
    del res2, matches2

    # ---- lanes inside the call (okvfe_set_internal_lanes): slices on the context's own streams, same bytes
    if FRAMES == 768:
        for lanes in (4, 3):
            fe.set_internal_lanes(lanes)
            res3, matches3 = _run(fe, cfg, d_img, FRAMES, grav, pairs, d_match)
            assert _digest(res3, matches3) == first, lanes
            del res3, matches3
        fe.set_internal_lanes(0)
        # ---- pipelined lanes (okvfe_set_internal_lanes(-k)): no join onto the caller's stream, the matcher on the lane
        # streams too; three steps queued back to back on alternating content (lane l starts step n + 1 behind its own
        # step n only), then the join through the host-side readers: same bytes as the unsplit call
        d_img_b = torch.flip(d_img, dims=[2]).contiguous()  # other content in the same buffers' slices
        cam_ids = np.array([0, 1] * FRAMES, dtype=np.int32)
        sptr = torch.cuda.current_stream().cuda_stream
        for lanes in (-4, -3):
            fe.set_internal_lanes(lanes)
            for img_t in (d_img, d_img_b, d_img):
                fe.detect_describe_batch_device(img_t.data_ptr(), 2 * FRAMES, cam_ids, grav, sptr)
                fe.match_stereo_batch_device(pairs, d_match.data_ptr(), sptr)
            fe.lanes_join(sptr)
            torch.cuda.synchronize()
            fe.check_capacity(2 * FRAMES)
            res4 = [fe.download(i) for i in range(2 * FRAMES)]
            rows = d_match.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(FRAMES, -1)
            matches4 = [rows[f, :len(res4[2 * f][0])].copy() for f in range(FRAMES)]
            assert _digest(res4, matches4) == first, lanes
            del res4, matches4
        fe.set_internal_lanes(0)

    # ---- permutation: multiframes in reversed order
    order = np.arange(FRAMES)[::-1].copy()
    idx = np.stack([2 * order, 2 * order + 1], axis=1).reshape(-1)
    d_img_r = torch.from_numpy(np.ascontiguousarray(imgs[idx])).cuda()
    res_r, matches_r = _run(fe, cfg, d_img_r, FRAMES, np.ascontiguousarray(grav[idx]), pairs, d_match)
    for j, fr in enumerate(order):
        for c in range(2):
            for x, y in zip(res_r[2 * j + c], res[2 * fr + c]):
                assert x.tobytes() == y.tobytes(), (j, fr, c)
        assert matches_r[j].tobytes() == matches[fr].tobytes(), (j, fr)


def test_concurrent_contexts_equal_one_context_alone():
    """Four contexts on four HIP streams, several steps in flight, each lane on different content at the same time
    (the `four_lanes` leg of bench.py; frames sharded over streams, the sharding the reference does over cameras with
    threads, Frontend.cpp:119-135): every frame of every lane is byte-identical to the same frame through one context
    with the GPU to itself (tools/stress_lanes.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_lanes.py"), "4", "48", "4"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 mismatching frames" in out.stdout


def test_different_workloads_on_streams_of_their_own_equal_each_alone():
    """Six different workloads -- EuRoC map-free, EuRoC with the score map kept, TUM-VI 1024 x 1024 (packed last strips), a
    682-px mono camera (generic score / NMS kernels), a 3-octave Harris scale space, the BRISK scale space -- on six HIP
    streams at once, two steps of each in flight: every context's results equal its own results with the GPU to itself
    (tools/stress_mixed.py).  The net for hazards that only show beside other kernels' memory traffic: the binary before
    round 6's store-data fix fails it on the scale space in most iterations."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_mixed.py"), "8"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
