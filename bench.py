#!/usr/bin/env python3
"""bench.py -- front-end throughput on MI355X: stereo-frames/s (detect + describe + match).

Workload (BASELINE.json metric, configs[2]): synthetic EuRoC-shaped stereo, 752x480 x 2 cameras,
front-end parameters of config/euroc.yaml:63-67 (uniformity radius 38, Harris threshold 150,
<= 700 keypoints, Hamming threshold 60), camera-aware gravity-aligned BRISK2 extraction,
matchStereo with the FP64 triangulation gate.  One "step" = one batch of `--batch` stereo frames
(default 3072 = 6144 images since the end of round 4; rounds 1-3 and most of round 4 quote 768) through the whole hot path (K1 score map with the K2 NMS fused in ->
K3 sort + greedy selection, K4 sub-pixel -> K6 describe -> compaction + back-projection -> K7 gated
stereo match), inputs resident in HBM.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
N > 1 is launched by torch.distributed.run (one rank per GPU); stereo frames are independent
units, so ranks shard batches with no data-path collective ("scaling": "weak").
"""
import argparse
import json
import math
import os
import re
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MIN_TIMED_S = 0.5
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable


def make_inputs(cfg, n_frames, n_distinct, seed0, content="corners", tile=True):
    """Multiframes of C = len(cfg.cams) images: cameras 0/1 are a synthetic stereo pair, further
    cameras (Hilti-shaped rig without extrinsics) look elsewhere and get independent images.
    content: "corners" = jittered random-gray cells of 12 px + noise (about 320 keypoints per 752x480
    image under the EuRoC parameters: the uniformity stage, not the corner supply, sets that number);
    "checker" = exact two-level 12 px checker cells without noise, whose tied scores all pass the
    uniformity stage (about 660 keypoints: the matcher's 700 x 700 regime)."""
    from okvis2_amd import synth
    C = len(cfg.cams)
    kw = dict(cell=12) if content == "corners" else dict(cell=12, levels=(0, 255), noise=0, jitter=0)
    base = []
    for i in range(n_distinct):
        L, R, _ = synth.stereo_pair(cfg.w, cfg.h, seed0 + i, **kw)
        base.append(L)
        base.append(R)
        for c in range(2, C):
            base.append(synth.corners_image(cfg.w, cfg.h, seed0 + 7919 * c + i, **kw))
    base = np.stack(base)  # [C*n_distinct, H, W]
    if not tile:  # (bench: the batch is tiled ON THE DEVICE from the distinct frames -- tile_on_device)
        return None, base
    reps = (n_frames + n_distinct - 1) // n_distinct
    return np.concatenate([base] * reps)[: C * n_frames], base


def tile_on_device(base, n_images, dev):
    """The batch = the distinct frames repeated, built on the GPU: a rank uploads C * distinct images instead of
    tiling gigabytes in host memory first (8 ranks x 2.2 GB at the default step)."""
    import torch
    d_base = torch.from_numpy(base).to(dev)
    reps = (n_images + len(base) - 1) // len(base)
    return d_base.repeat(reps, 1, 1)[:n_images].contiguous()


def real_inputs(cfg, n_frames, n_distinct):
    """Stereo frames cut from the reference's own camera image (okvis_multisensor_processing/test/testImage.jpg,
    decoded into tests/golden/real_image.npz by tools/make_real_image_fixture.py: 1280x960, a checkerboard on a
    carpet, 28 % saturated pixels, JPEG blocks): windows of the configuration's size at `n_distinct` offsets,
    the right image the same window shifted by 23 px (tests/real_image_cases.py: STEREO_DISPARITY).  None when
    the fixture or the shape does not allow it."""
    path = os.path.join(ROOT, "tests", "golden", "real_image.npz")
    if not os.path.exists(path) or len(cfg.cams) != 2:
        return None
    full = np.load(path)["image"]
    H, W = full.shape
    disp = 23
    if cfg.w + disp > W or cfg.h > H:
        return None
    base = []
    for i in range(n_distinct):
        x0 = ((i * 67) % max(1, W - cfg.w - disp)) & ~3  # (dword-aligned windows of a u8 image are still arbitrary content)
        y0 = (i * 41) % max(1, H - cfg.h + 1)
        base.append(full[y0:y0 + cfg.h, x0:x0 + cfg.w])
        base.append(full[y0:y0 + cfg.h, x0 + disp:x0 + disp + cfg.w])
    return np.ascontiguousarray(np.stack(base))  # the distinct frames; the caller tiles them (tile_on_device)


def oracle_follows_pattern(fe):
    """--box-widen: the context was created with okvfe_config.box_scale; the oracle that the cpu_baseline leg checks
    against gets the same pattern (the pattern is data on both sides)"""
    import ctypes as C
    p = fe.get_pattern()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    q = type(O.pattern())()
    C.memmove(C.byref(q), C.byref(O.pattern()), C.sizeof(q))
    for f in ("n_points", "n_short", "n_long", "border"):
        setattr(q, f, getattr(p, f))
    for f in ("px", "py", "sigma_half", "short_i", "short_j", "long_i", "long_j", "long_wdx", "long_wdy"):
        C.memmove(getattr(q, f), getattr(p, f), C.sizeof(getattr(p, f)))
    O._PATTERN = q


N_VARIANTS = 4  # distinct per-step host parameter sets (gravity directions, poses)


def gravity_variant(v, n_images):
    """Extraction directions of step variant v: a few degrees around (0, 1, 0), different per image
    and per variant, so every step uploads fresh parameters like every real frame does (new T_WC =>
    new gravity direction, Frontend.cpp:247-251)."""
    i = np.arange(n_images)
    g = np.stack([0.02 * (((i + v) % 5) - 2), np.ones(n_images), 0.01 * ((i // 3 + v) % 3 - 1)], axis=1)
    return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)


def pose_variant(cfg, v):
    """T_WC0 / T_WC1 of step variant v: the rig translated by a variant-dependent offset."""
    from okvis2_amd import synth
    (C0, r0), (C1, r1) = synth.stereo_poses(cfg.baseline)
    off = np.array([0.01 * v, -0.02 * v, 0.005 * v])
    return (C0, r0 + off), (C1, r1 + off)


def cpu_baseline(cfg, base_imgs, fe, grav, poses, budget_s=12.0):
    """Times the CPU oracle (a port of the algorithm, NOT the reference binary, which cannot be
    built here) with the reference's threading shape: one thread per camera for detect+describe
    (ThreadedSlam.cpp:434-448), matchStereo single-threaded (Frontend.cpp:2016).  Also compares
    the oracle's outputs with the GPU's for the same frames (the oracle acting as checker).
    grav / poses: what the GPU's last step used for these frames."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    maps = [O.awareness_maps(c) for c in cfg.cams]
    T0, T1 = poses
    f = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    C = len(cfg.cams)
    n_distinct = len(base_imgs) // C
    out = [None] * C

    def work(ci, img, g):
        cam = cfg.cams[ci]
        k, d = O.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                 O.MODE_CAMERA_AWARE, maps[ci][0], maps[ci][1], np.float32(cam.fu),
                                 tuple(float(x) for x in g))
        bp, bv = O.backproject_keypoints(cam, k)
        out[ci] = (k, d, bp, bv)

    done, checked, mismatches = 0, 0, 0
    t0 = time.perf_counter()
    while True:
        i = done % n_distinct
        # the batch tiles the distinct frames (each replica with its own extraction direction): frame i is
        # computed -- and, in the checker leg below, compared -- as its replica j from the head, the middle or
        # the tail of the batch in turn, so the checked rows span the whole address range of the step
        reps = max(1, fe._bench_frames // n_distinct) if fe is not None else 1
        j = i + n_distinct * (0, reps // 2, reps - 1)[i % 3]
        ths = [threading.Thread(target=work, args=(c, base_imgs[C * i + c], grav[C * j + c]))
               for c in range(1, C)]
        for th in ths:
            th.start()
        work(0, base_imgs[C * i], grav[C * j])
        for th in ths:
            th.join()
        (k0, d0, b0, v0) = out[0]
        m = None
        if C > 1:
            (k1, d1, b1, v1) = out[1]
            m = O.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f[0], f[1], cfg.match_threshold)
        done += 1
        elapsed = time.perf_counter() - t0
        if done <= n_distinct and fe is not None:  # checker leg, outside the measured work
            t_chk = time.perf_counter()
            ok = m is None or np.array_equal(fe._bench_matches[j, :len(k0)].view(np.uint8),
                                             m.view(np.uint8))
            for c in range(C):
                g = fe.download(C * j + c)
                ok = ok and np.array_equal(g[0].view(np.uint8), out[c][0].view(np.uint8)) \
                    and np.array_equal(g[1], out[c][1]) \
                    and np.array_equal(g[2].view(np.uint64), out[c][2].view(np.uint64))
            checked += 1
            mismatches += 0 if ok else 1
            t0 += time.perf_counter() - t_chk
        if elapsed >= budget_s and done >= 8:
            break
    elapsed = time.perf_counter() - t0
    return {"value": done / elapsed,
            "unit": {1: "frames/s", 2: "stereo-frames/s"}.get(C, "multiframes/s"),
            "cores": C, "kind": "port",
            "sample": f"{done} multiframes of the bench workload ({n_distinct} distinct), "
                      f"{elapsed:.1f} s; 1 thread per camera for detect+describe, match serial; "
                      f"host has {os.cpu_count()} logical cores",
            "parity_checked_frames": checked, "parity_mismatches": mismatches,
            "parity_scope": "keypoints, descriptors, back-projections (u64), match rows incl. hp_W"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-run under torch.distributed.run with one
    rank per GPU on this node and pass its output through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


WORKLOAD_TEXT = {
    "euroc": ("front-end stereo-frames/s (detect+describe+match), 752x480 stereo",
              "EuRoC-shaped 752x480 stereo, euroc.yaml front-end params (radius 38, thr 150, "
              "<=700 kpts, match thr 60)"),
    "tumvi": ("front-end stereo-frames/s (detect+describe+match), 1024x1024 stereo (TUM-VI)",
              "TUM-VI-shaped 1024x1024 equidistant stereo, tumvi_slam_1024.yaml front-end params "
              "(radius 50, thr 5, <=1000 kpts, match thr 60)"),
    "hilti": ("front-end multiframes/s (5 x detect+describe + cross-camera match), 720x540 x 5 "
              "cameras (Hilti 2022)",
              "Hilti 2022 rig, 5 equidistant 720x540 cameras, hilti_challenge_2022.yaml front-end "
              "params (radius 50, thr 20, <=700 kpts, match thr 60)"),
    "mono640": ("front-end frames/s (detect+describe), 640x480 mono",
                "640x480 mono (radius 10, thr 5, <=1000 kpts), detect+describe"),
    # the shipped configurations BASELINE does not name
    "tumvi512": ("front-end stereo-frames/s (detect+describe+match), 512x512 stereo (TUM-VI 512)",
                 "TUM-VI-shaped 512x512 equidistant stereo, tumvi_slam_512.yaml front-end params (radius 40, thr 4, "
                 "<=800 kpts, match thr 55)"),
    "d455": ("front-end stereo-frames/s (detect+describe+match), 640x480 stereo (RealSense D455)",
             "640x480 rectified stereo, realsense_D455.yaml front-end params (radius 30, thr 5, <=2500 kpts, match thr 60)"),
    "d435i": ("front-end stereo-frames/s (detect+describe+match), 640x480 stereo (RealSense D435i)",
              "640x480 rectified stereo, realsense_D435i.yaml front-end params (radius 30, thr 5, <=400 kpts, match thr 60)"),
}


# VALU issue rates measured on this part (tools/ubench/valu_rate.hip, profiles/round2_valu_rate_ubench.txt):
# 2.3 cycles per wave64 instruction for xor / add / shifts, 4.2 for v_bcnt_u32_b32 and the rest of VOP3.
# A 384-bit Hamming distance = 12 x (v_xor + v_bcnt accumulate) per lane = 78 cycles per 64 pairs on
# one of 1024 SIMDs at 2.4 GHz.
VALU_PEAK_HAMMING_PAIRS = 1024 * 64 / (12 * (2.3 + 4.2)) * 2.4e9
# L2 -> LDS direct loads (buffer_load ... lds), the path the descriptor kernel stages its patches
# through: ~12 B per cycle per CU (guides/MI355X_MICROARCH.md, "LDS-DMA prologue ~12-13 B/cyc/CU")
LDS_DMA_PEAK_GBPS = 256 * 12.0 * 2.4


def alu_rooflines(stage_ms, pair_evals_per_launch, kp_described_per_launch, patch_bytes_per_kp, n_img_launch):
    """Roofline blocks of the kernels that are NOT HBM-bound: what bounds them, the
    achieved rate from this run's stage times and the fraction of that bound's peak."""
    out = {}
    if stage_ms.get("match") and pair_evals_per_launch:
        a = pair_evals_per_launch / (stage_ms["match"] * 1e-3)
        out["match_stereo"] = {"kernel": "match_stereo_kernel (K7)", "bound": "valu", "achieved": a / 1e9,
                               "peak": VALU_PEAK_HAMMING_PAIRS / 1e9, "unit": "G Hamming pairs/s (384 bit)",
                               "frac": a / VALU_PEAK_HAMMING_PAIRS,
                               "algorithmic_pairs_per_launch": pair_evals_per_launch,
                               "note": "sum over stereo frames of n0 x n1 descriptor pairs (the reference evaluates "
                                       "every pair, Frontend.cpp:2016-2026); peak = 12 x (v_xor + v_bcnt) at the "
                                       "measured issue rates; the rest of the launch is the FP64 gate rounds"}
    if stage_ms.get("describe") and kp_described_per_launch:
        a = kp_described_per_launch * patch_bytes_per_kp / (stage_ms["describe"] * 1e-3) / 1e9
        out["describe"] = {"kernel": "describe_aware_kernel (K6, camera-aware; describe_kernel for the other modes)",
                           "bound": "l2_to_lds (buffer_load ... lds)",
                           "achieved": a, "peak": LDS_DMA_PEAK_GBPS, "unit": "GB/s staged into LDS",
                           "frac": a / LDS_DMA_PEAK_GBPS,
                           "patch_bytes_per_keypoint": patch_bytes_per_kp,
                           "keypoints_per_launch": kp_described_per_launch,
                           "note": "a keypoint's pattern patch (64 rows x 64..80 B) goes L2 -> LDS once; the box "
                                   "sums (round 6: ~285 VALU + ~63 LDS wave instructions per keypoint, the LDS gather of "
                                   "the row windows is the busiest unit) overlap with it; a bookkeeping ratio, see the "
                                   "valu_issue block for the instruction side"}
    if stage_ms.get("select"):
        out["select"] = {"kernel": "select_lazy_kernel (K3 uniformity + K4 sub-pixel)", "bound": "latency",
                         "achieved": n_img_launch / (stage_ms["select"] * 1e-3), "peak": None, "unit": "images/s",
                         "frac": None,
                         "note": "one workgroup per image, all images of the launch resident at once: the launch "
                                 "lasts as long as ONE image's serial chain (candidates in score order, ~2 k "
                                 "cycles per 64-candidate window); there is no throughput peak to compare with"}
    if stage_ms.get("sort"):
        out["sort"] = {"kernel": "sort_rb_kernel (K3 sort)", "bound": "lds", "achieved": None, "peak": None,
                       "unit": None, "frac": None,
                       "note": "bitonic network on 64-bit keys in LDS, 24 passes for 8192 keys; ~24 G keys/s"}
    return out


def valu_issue_blocks(rooflines, stage_ms, n_img_launch, workload, content):
    """Adds a `valu_issue` block to the non-K1 kernels: wave64 vector-ALU instructions per launch (SQ_INSTS_VALU of the
    newest committed SQ pass, taken on the default EuRoC workload at 1536 images per launch and scaled by the images of
    this launch) over the stage time, against one VALU issue per SIMD per 4 cycles.  Only for the workload the counters
    were taken on."""
    import glob
    if workload != "euroc" or content != "corners":
        return
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_sq.json")) if "withmap" not in f]
    if not files:
        return

    def tag(path):  # (round, collection) in numeric order: round6_v2 after round5_v10 (profiles/INDEX.md names the authoritative one)
        m = re.match(r"round(\d+)(?:_v(\d+))?_", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (0, 0)
    files.sort(key=tag, reverse=True)
    try:
        sq = json.load(open(files[0]))
    except Exception:
        return
    # images per launch of that pass = the bench line of the same collection (1536 before round4_v12)
    sq_images = 1536.0
    try:
        line = json.load(open(files[0].replace("_pmc_sq.json", "_bench.json")))
        sq_images = float(line["config"]["stereo_frames_per_launch"] * line["config"].get("cameras_per_multiframe", 2))
    except Exception:
        pass
    for key, prefix, stage in (("describe", ("describe_aware_kernel", "describe_kernel"), "describe"),
                               ("select", ("select_lazy_kernel",), "select"),
                               ("match_stereo", ("match_stereo_kernel",), "match")):
        hit = [v for pf in prefix for k, v in sq.items() if k.startswith(pf)]
        if key not in rooflines or not hit or not stage_ms.get(stage):
            continue
        insts = hit[0]["mean_per_dispatch"].get("SQ_INSTS_VALU")
        if not insts:
            continue
        insts = insts * n_img_launch / sq_images
        g = insts / (stage_ms[stage] * 1e-3) / 1e9
        rooflines[key]["valu_issue"] = {
            "insts_per_launch": insts, "achieved": g, "peak": VALU_PEAK_GINST, "frac": g / VALU_PEAK_GINST,
            "unit": "G wave64 VALU instructions/s",
            "source": "SQ_INSTS_VALU of profiles/%s (per image, times the images of this launch); peak = 1024 SIMDs x 2.4 GHz / "
                      "4 cycles per wave64 op; simple VOP2 forms issue faster, so 1.0 is not a hard ceiling" %
                      os.path.basename(files[0])}


def run_map_workload(args, torch, capi, synth, dev):
    """SURVEY.md 8 D3 / Frontend.cpp:1515-1589: the map matcher on device-resident data.  One step =
    okvfe_match_to_map_blocks_device over B frames of 700 keypoints against 5000 pooled 3-D landmarks
    (1..3 descriptors each, projections per frame), nothing crossing PCIe.  Reports frames/s and the
    VALU roofline of the Hamming work; rank 0 / one GPU (replicas only: the map needs estimator state)."""
    from okvis2_amd import multigpu
    cfg = synth.euroc_config()
    B = args.batch if args.batch is not None else 256
    K, L = cfg.max_kpts, 5000
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, K,
                       match_threshold=cfg.match_threshold, max_batch=1, num_cameras=1, device=dev.index)
    rng = np.random.default_rng(7)
    counts = rng.integers(1, 4, L)
    begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    pool = rng.integers(0, 256, (begin[-1], 48), dtype=np.uint8)
    lm_xy = np.stack([rng.uniform(-200, 952, L), rng.uniform(-150, 630, L)], 1)  # ~60 % inside the image
    distinct = min(args.distinct, B)
    blocks, proj, useful = [], [], []
    kdt = capi.KEYPOINT_DTYPE
    for f in range(distinct):
        kps = np.zeros(K, dtype=kdt)
        kps["x"] = rng.uniform(30, cfg.w - 30, K)
        kps["y"] = rng.uniform(30, cfg.h - 30, K)
        desc = rng.integers(0, 256, (K, 48), dtype=np.uint8)
        p = lm_xy + rng.normal(0, 2.0, (L, 2))
        vis = np.flatnonzero((p[:, 0] > 0) & (p[:, 0] < cfg.w) & (p[:, 1] > 0) & (p[:, 1] < cfg.h))
        obs = rng.permutation(vis)[:K]  # keypoint i observes landmark obs[i]
        kps["x"][:len(obs)] = p[obs, 0] + rng.normal(0, 1.5, len(obs))
        kps["y"][:len(obs)] = p[obs, 1] + rng.normal(0, 1.5, len(obs))
        for i, l in enumerate(obs[::2]):
            desc[2 * i] = pool[begin[l]] ^ ((rng.random(48) < 0.04) * rng.integers(0, 256, 48)).astype(np.uint8)
        blocks.append(multigpu.pack_block_host(K, kps, desc, np.zeros((K, 3)), np.ones(K, np.uint8)))
        proj.append(p)
        # Hamming evaluations of the reference loop: (keypoint, pooled descriptor) pairs within the radius
        dx = kps["x"][:, None].astype(np.float64) - p[None, :, 0]
        dy = kps["y"][:, None].astype(np.float64) - p[None, :, 1]
        near = (dx * dx + dy * dy) <= args.map_radius ** 2
        useful.append(int((near * counts[None, :]).sum()))
    rep = [i % distinct for i in range(B)]
    d_blocks = torch.from_numpy(np.stack([blocks[i] for i in rep])).to(dev)
    d_proj = torch.from_numpy(np.stack([proj[i] for i in rep])).to(dev)
    d_begin, d_pool = torch.from_numpy(begin).to(dev), torch.from_numpy(pool).to(dev)
    md = fe.make_map_device(L, d_begin.data_ptr(), d_pool.data_ptr(), d_proj.data_ptr())
    d_lm = torch.empty((B, K), dtype=torch.int32, device=dev)
    d_bd = torch.empty((B, K), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    fe.profile_enable(True, stages=("map",))

    def step():
        fe.match_to_map_blocks_device(d_blocks.data_ptr(), B, None, md, args.map_radius, d_lm.data_ptr(),
                                      d_bd.data_ptr(), st)

    for _ in range(args.warmup):
        step()
    st.synchronize()
    torch.cuda.synchronize()
    steps = args.steps
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    st.synchronize()
    elapsed = time.perf_counter() - t0
    ms, cnt = fe.profile_read()["map"]
    launch_ms = ms / cnt
    lm = d_lm.cpu().numpy()
    matched = int((lm[:distinct] >= 0).sum())
    # parity on the first distinct frames against the host-buffer entry point (itself oracle-checked in tests/)
    pairs_launch = sum(useful[i] for i in rep)
    brute = B * K * int(begin[-1])
    a = pairs_launch / (launch_ms * 1e-3)
    res = {
        "metric": "map-matcher frames/s (matchToMapByThread, 5000 pooled landmarks x 700 keypoints)",
        "value": B * steps / elapsed, "unit": "frames/s", "n_gpus": 1, "steps": steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 (Hamming) + f64 (reprojection gate)", "data": "synthetic",
        "config": {"workload": "Frontend::matchToMapByThread (Frontend.cpp:1552-1589) on device-resident data: "
                               f"{B} frames x {K} keypoints (gather blocks) against {L} pooled 3-D landmarks "
                               f"({int(begin[-1])} descriptors), reprojection radius {args.map_radius} px, one "
                               "launch per step (okvfe_match_to_map_blocks_device), nothing crosses PCIe",
                   "frames_per_step": B, "distinct_frames": distinct,
                   "matched_keypoints_per_frame": matched / distinct,
                   "parallelism": "replicas only (the map needs estimator state)"},
        "roofline": {"kernel": "match_to_map_kernel (batched over gather blocks)", "bound": "valu",
                     "achieved": a / 1e9, "peak": VALU_PEAK_HAMMING_PAIRS / 1e9,
                     "unit": "G Hamming pairs/s (384 bit)", "frac": a / VALU_PEAK_HAMMING_PAIRS,
                     "algorithmic_pairs_per_launch": pairs_launch, "avg_launch_ms": launch_ms,
                     "brute_force_pairs_per_launch": brute,
                     "brute_force_equivalent_frac": brute / (launch_ms * 1e-3) / VALU_PEAK_HAMMING_PAIRS,
                     "traffic": None,
                     "note": "algorithmic pairs = (keypoint, pooled descriptor) pairs inside the reprojection "
                             "radius = the Hamming distances the reference loop evaluates; the kernel tests the "
                             "radius per (wave of 64 keypoints, landmark) and skips a landmark no keypoint of the "
                             "wave is near, so most of its time is the FP64 radius test over K x L, not popcounts; "
                             "peak = 12 x (v_xor + v_bcnt) per pair at the measured VALU issue rates"},
    }
    return res



def batch_sweep(cfg, capi, torch, dev, local_rank, base, max_candidates, budget_s=0.35):
    """the hot path at B in {1, 16, 256, 768} stereo frames per call.  The reference's
    seams are B = 1 calls; the batch entry points amortise launches over B frames.  Per B, device-
    and host-fed: `pipelined` = calls enqueued back to back on one stream (throughput), `latency_ms`
    = one call with a host synchronisation after it (what a caller waiting for the result sees)."""
    C = len(cfg.cams)
    out = {}
    n_distinct = len(base) // C
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[min(1, C - 1)].fu + cfg.cams[min(1, C - 1)].fv)
    for B in (1, 16, 256, 768):  # (768: the batch rounds 1-3 quoted `value` on; the default step holds 3072)
        fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold, cfg.max_kpts,
                           match_threshold=cfg.match_threshold, max_batch=C * B, num_cameras=C,
                           device=local_rank, max_candidates=max_candidates)
        for ci, cam in enumerate(cfg.cams):
            fe.set_camera(ci, cam)
        imgs = np.concatenate([base] * ((B + n_distinct - 1) // n_distinct))[:C * B]
        d_img = torch.from_numpy(imgs).to(dev)
        h_img = torch.from_numpy(imgs).pin_memory() if B <= 256 else None  # (the host-fed rate is PCIe-bound from B = 16 on)
        d_match = torch.zeros((B, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        cam_ids = np.array(list(range(C)) * B, dtype=np.int32)
        grav = gravity_variant(0, C * B)
        T0, T1 = pose_variant(cfg, 0)
        pairs = []
        for i in range(B):
            sp = capi.StereoPair()
            sp.image0, sp.image1 = C * i, C * i + (1 if C > 1 else 0)
            sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
            sp.f0, sp.f1 = f0, f1
            pairs.append(sp)
        pairs = (capi.StereoPair * B)(*pairs)
        st = torch.cuda.Stream(device=dev)

        def call(feed):
            if feed == "host":
                fe.detect_describe_batch_host(h_img.data_ptr(), C * B, cam_ids, grav, st)
            else:
                fe.detect_describe_batch_device(d_img.data_ptr(), C * B, cam_ids, grav, st)
            if C > 1:
                fe.match_stereo_batch_device(pairs, d_match.data_ptr(), st)

        row = {}
        for feed in (("device", "host") if h_img is not None else ("device",)):
            for _ in range(3):
                call(feed)
            st.synchronize()
            # pipelined
            n = 4
            while True:
                t0 = time.perf_counter()
                for _ in range(n):
                    call(feed)
                st.synchronize()
                el = time.perf_counter() - t0
                if el >= budget_s or n >= 1 << 14:
                    break
                n *= 4
            # one call at a time
            lat = []
            t_end = time.perf_counter() + budget_s
            while time.perf_counter() < t_end or len(lat) < 5:
                t0 = time.perf_counter()
                call(feed)
                st.synchronize()
                lat.append(time.perf_counter() - t0)
            lat.sort()
            row[feed] = {"pipelined_frames_per_s": B * n / el, "pipelined_ms_per_call": 1e3 * el / n,
                         "latency_ms_median": 1e3 * lat[len(lat) // 2], "latency_ms_p90": 1e3 * lat[(len(lat) * 9) // 10],
                         "calls_timed": n, "latency_calls": len(lat)}
        fe.check_capacity(C * B)
        out[str(B)] = row
        fe.close()
        del d_img, h_img, d_match
    out["note"] = ("B = stereo frames per okvfe_detect_describe_batch_* + okvfe_match_stereo_batch_device call pair; "
                   "pipelined = calls back to back on one stream, latency = host waits for every call; host = images "
                   "cross PCIe from pinned memory in the call")
    return out


def latency_b1(cfg, base, n_frames=8, iters=300, warmup=30):
    """B = 1 latency through the C++ seams (tests/cpp/latency_cli.cpp): okvfe::HipViFrontend::
    detectAndDescribe with one thread per camera as ThreadedSlam.cpp:434-448 runs it, then matchStereo;
    the cv::Feature2D adapters (Frame::detect + Frame::describe); HipFrontend over the C ABI."""
    import struct
    import subprocess
    import tempfile
    from okvis2_amd import synth
    exe = os.path.join(ROOT, "tests", "cpp", "latency_cli")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "tests", "mock"),
                               "-o", exe, exe + ".cpp", "-L" + os.path.join(ROOT, "okvis2_amd"), "-lokvfe",
                               "-Wl,-rpath," + os.path.join(ROOT, "okvis2_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    C = len(cfg.cams)
    n_frames = min(n_frames, len(base) // C)
    T = synth.stereo_poses(cfg.baseline)
    # camera y axis = world -z: the extraction direction T_WC^-1 (0,0,-1) is (0, 1, 0)
    Cm = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])
    with tempfile.NamedTemporaryFile(suffix=".req", delete=False) as f:
        f.write(struct.pack("<5i", cfg.w, cfg.h, n_frames, iters, warmup))
        f.write(struct.pack("<f3i", cfg.uniformity_radius, cfg.abs_threshold, cfg.match_threshold, cfg.max_kpts))
        for c in range(2):
            cam = cfg.cams[c]
            f.write(struct.pack("<4d", cam.fu, cam.fv, cam.cu, cam.cv))
            f.write(struct.pack("<i", cam.dist_type))
            f.write(struct.pack("<4d", *cam.d))
            f.write(struct.pack("<9d", *Cm.reshape(-1)))
            f.write(struct.pack("<3d", *T[c][1]))
        for i in range(n_frames):
            for c in range(2):
                f.write(np.ascontiguousarray(base[C * i + c]).tobytes())
        path = f.name
    res = {}
    try:
        for route in ("vi", "cv", "c"):  # one process per route: each sees the GPU's hardware queues alone
            p = subprocess.run([exe, path, route], capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                return {"error": f"latency_cli {route} rc {p.returncode}: {p.stderr[-300:]}"}
            r = json.loads(p.stdout.strip().splitlines()[-1])
            res.update({k: v for k, v in r.items() if v is not None and (k not in res or route == "vi")})
    finally:
        os.unlink(path)
    res["note"] = ("one stereo frame per iteration, wall clock on the host: vi = HipViFrontend::detectAndDescribe "
                   "(std::thread per camera >= 1 as ThreadedSlam.cpp:434-448, mock OKVIS2/OpenCV containers) + "
                   "HipFrontend::matchStereo on the GPU's outputs; cv = cv::FeatureDetector::detect + "
                   "cv::DescriptorExtractor::compute per camera (Frame.hpp:152,167); hipfrontend = detect+describe "
                   "of both cameras through the C ABI")
    return res


def k1_name(map_free):
    return ("harris_kernel<61, true> (K1 score + fused K2 NMS, candidates only: no score map)" if map_free
            else "harris_kernel<61, true> (K1 score map + fused K2 NMS)")


def mean_candidates(capi, fe, n_images):
    """Mean NMS maxima per image of the last detect call of `fe` (okvfe_device_outputs.candidate_counts)."""
    import ctypes
    out = fe.device_outputs()
    host = np.zeros(n_images, dtype=np.int32)
    st = capi.lib().okvfe_copy_to_host(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(out.candidate_counts),
                                       ctypes.c_size_t(host.nbytes), None)
    return float(host.mean()) if st == 0 and n_images > 0 else 0.0


def k1_map_free(fe):
    """True when the last detect call of `fe` wrote no score map (okvfe_set_keep_score_map)."""
    return fe.device_outputs().scores is None


VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0  # wave-instructions/ns: 1024 SIMDs, one wave64 VALU op per 4 cycles at 2.4 GHz


def k1_pmc_for(w, h, map_free):
    """Newest committed PMC collection of K1 taken on THIS image shape and map mode (None: none)."""
    import glob
    import re

    def tag(path):  # (round, collection) as numbers: round4_v12 is newer than round4_v5
        m = re.match(r"round(\d+)(?:_v(\d+))?_", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (0, 0)

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_k1_pmc*.json")), key=tag, reverse=True)
    cands = [c for c in cands if c.endswith(".json") and "_sq" not in os.path.basename(c)]
    for pmc_path in cands:
        pmc = json.load(open(pmc_path))
        if bool(pmc.get("map_free", False)) != bool(map_free):
            continue
        # (files written before round 4 carry no shape: same kernel and shape <=> same 5 P n)
        same = (pmc.get("width") == w and pmc.get("height") == h) if "width" in pmc else \
            pmc.get("algorithmic_bytes_per_launch") == 5 * w * h * pmc.get("images_per_launch", 0)
        if same:
            return pmc, os.path.basename(pmc_path)
    return None, None


def roofline_block(P, n_img_launch, harris_ms, extra, w=None, h=None, map_free=False, cand_per_image=0.0):
    """K1 moves 5 P bytes per image (1 B in, 4 B out per pixel) when it writes the score map;
    fused with the NMS and the map never materialised (the default since round 4) the algorithmic figure is
    P + 12 C (C = candidate records of 12 B) and the kernel is bound by vector-ALU issue, not by HBM: both
    fractions are reported, `bound` names the memory roofline the contract asks for, `live_bound` the unit
    that is actually saturated."""
    per_image = (P + 12.0 * cand_per_image) if map_free else 5.0 * P
    alg = per_image * n_img_launch
    achieved = alg / (harris_ms * 1e-3) / 1e9
    traffic, traffic_src, valu = None, None, None
    pmc, pmc_name = k1_pmc_for(w, h, map_free) if w else (None, None)
    if pmc:
        traffic = pmc["hbm_bytes_per_image"] * n_img_launch
        traffic_src = ("profiles/%s (FETCH_SIZE/WRITE_SIZE passes at %d images per launch, scaled "
                       "per image; rocprofv3 cannot run inside this process)" % (pmc_name, pmc["images_per_launch"]))
        if pmc.get("valu_insts_per_image"):
            ginst = pmc["valu_insts_per_image"] * n_img_launch / (harris_ms * 1e-3) / 1e9
            valu = {"achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instructions/s",
                    "frac": ginst / VALU_PEAK_GINST,
                    "insts_per_launch": pmc["valu_insts_per_image"] * n_img_launch,
                    "source": "SQ_INSTS_VALU of profiles/%s; peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 op" % pmc_name}
    r = {"kernel": k1_name(map_free), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
         "traffic_over_algorithmic": (traffic / alg) if traffic else None,
         "algorithmic_bytes_per_launch": alg,
         "algorithmic_note": ("P + 12 C per image: 1 B per pixel in, 12 B per candidate out (%.0f candidates per image); the "
                              "score map is not written (SURVEY.md 8 D4, fused form)" % cand_per_image) if map_free
         else "5 P per image: 1 B per pixel in, 4 B per pixel out",
         "avg_launch_ms": harris_ms,
         "live_bound": "valu" if map_free else "hbm",
         "valu": valu}
    r.update(extra)
    return r


def run_hilti_split_cameras(args, torch, dist, capi, synth, world, rank, dev):
    """BASELINE configs[4]: the Hilti rig with its real extrinsics, cameras spread over the ranks
    (camera c on rank c % world), ONE RCCL all-gather of the gather blocks per step and the 9
    FoV-overlapping pairs matched by their owner ranks -- okvis2_amd.multigpu.CrossCameraMatcher."""
    from okvis2_amd import multigpu
    cfg = synth.hilti_config()
    B = args.batch
    pairs = synth.rig_overlap_pairs(cfg, capi.camera_overlap)
    poses = synth.rig_poses(cfg)
    focal = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    local = [c for c in range(5) if multigpu.camera_owner(c, world) == rank]
    if not local:
        raise SystemExit("--split cameras needs at most 5 ranks")
    distinct = min(args.distinct, B, 8)
    rays = {c: capi.build_awareness_maps(cfg.cams[c])[0] for c in range(5)}
    frames = [synth.render_rig(cfg, [rays[c] for c in range(5)], 500 + i) for i in range(distinct)]
    engines, d_img, grav = {}, {}, {}
    for c in local:
        engines[c] = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold,
                                   cfg.max_kpts, match_threshold=cfg.match_threshold, max_batch=B,
                                   num_cameras=1, device=dev.index, max_candidates=args.max_candidates)
        engines[c].set_camera(0, cfg.cams[c])
        imgs = np.stack([frames[i % distinct][c] for i in range(B)])
        d_img[c] = torch.from_numpy(imgs).to(dev)
        grav[c] = np.tile(synth.gravity_in_camera(poses[c][0]), (B, 1))
    ccm = multigpu.CrossCameraMatcher(engines, 5, B, poses, focal, lambda i, j: (i, j) in pairs,
                                      world, rank, dev)
    ptrs = {c: d_img[c].data_ptr() for c in local}
    for c in local:
        engines[c].profile_enable(True, stages=("harris",))

    def step():
        ccm.step(ptrs, grav)

    for _ in range(args.warmup):
        step()
    ccm.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ccm.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for c in local:
        engines[c].check_capacity(B)
    prof = [engines[c].profile_read()["harris"] for c in local]
    harris_ms = sum(p[0] for p in prof) / max(1, sum(p[1] for p in prof))
    host_blocks = ccm.gathered.cpu().numpy()
    n_matches = {}
    for p in ccm.mine:  # rows beyond a frame's keypoint count are not written by the matcher
        rows = ccm.out[p].cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(B, -1)
        blk = host_blocks[multigpu.camera_owner(p[0], world) * ccm.slots + p[0] // world]
        n0 = blk[:, :4].copy().view(np.int32)[:, 0]
        n_matches[p] = int(sum((rows[f, :n0[f]]["k1"] >= 0).sum() for f in range(B)))
    if rank != 0:
        return None
    metric, text = WORKLOAD_TEXT["hilti"]
    res = {
        "metric": metric, "value": B * args.steps / elapsed, "unit": "multiframes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "u8/int32 (detect, describe, Hamming) + f64 (match gate)", "data": "synthetic",
        "config": {"workload": text + ", real T_SC (hilti_challenge_2022.yaml:3-71), rendered "
                               "textured-sphere scene, 9 FoV-overlapping pairs matched",
                   "multiframes_per_step": B, "distinct_frames": distinct,
                   "cameras_per_multiframe": 5, "pairs": [list(p) for p in pairs],
                   "parallelism": f"cameras split over {world} rank(s) (camera c on rank c % world), "
                                  f"one RCCL all-gather of {ccm.slots} x {B} gather blocks "
                                  f"({ccm.block_bytes} B each) per rank per step, pair (i, j) matched "
                                  f"on rank (i + j) % world",
                   "rank0_pairs": [list(p) for p in ccm.mine],
                   "rank0_matches_per_step": sum(n_matches.values())},
        "roofline": roofline_block(cfg.w * cfg.h, B, harris_ms,
                                   {"note": "one launch per local camera; the cameras of a rank run on "
                                            "their own streams and may overlap"}, cfg.w, cfg.h,
                                   k1_map_free(engines[local[0]]), mean_candidates(capi, engines[local[0]], B)),
    }
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="steps of the timed region, timed EXACTLY; without the flag: 340 (about 0.5 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="stereo frames per step per GPU (default: 3072 EuRoC frames; per workload otherwise, see below)")
    ap.add_argument("--distinct", type=int, default=256,
                    help="distinct synthetic stereo pairs in the batch (tiled on the device up to the step's size); rounds 1-5 "
                         "ran 16, which the default line still reports as value_16_distinct")
    ap.add_argument("--max-candidates", type=int, default=16384)
    ap.add_argument("--lanes", type=int, default=1,
                    help="independent contexts/streams the batch is split over on each GPU.  With 3-4 lanes the "
                         "latency-bound kernels of one lane hide behind the VALU-bound score kernel of another "
                         "(+9-11 %% at the same frames per step), but the score kernel then shares the GPU while it "
                         "is being timed; the default run reports that configuration as the `four_lanes` leg")
    ap.add_argument("--stagger", type=int, default=1,
                    help="with --lanes > 1: serialise the score kernels of the lanes (okvfe_set_heavy_kernel_chaining "
                         "mode 1) so that the lanes run out of phase")
    ap.add_argument("--map-radius", type=float, default=20.0,
                    help="--workload map: reprojection threshold in px (20 with IMU, 150 without; Frontend.cpp:1530)")
    ap.add_argument("--workload", choices=("euroc", "tumvi", "hilti", "mono640", "map", "tumvi512", "d455", "d435i"), default="euroc",
                    help="euroc = the BASELINE.json metric (752x480 stereo); tumvi = configs[3]; "
                         "hilti = configs[4] (5 cameras); mono640 = configs[1] (informational; batch "
                         "192 by default for these)")
    ap.add_argument("--split", choices=("frames", "cameras"), default="frames",
                    help="hilti only: frames = every rank runs whole multiframes (forward pair "
                         "matched, no collective); cameras = the rig's cameras are spread over the "
                         "ranks, gathered with one RCCL all-gather per step, 9 pairs matched")
    ap.add_argument("--feed", choices=("device", "host"), default="device",
                    help="device = images resident in HBM (the metric); host = every step copies its "
                         "images from pinned host memory (okvfe_detect_describe_batch_host, copy "
                         "overlapped with the previous step's kernels).  The device-fed run also "
                         "reports a short host-fed measurement as `host_fed`")
    ap.add_argument("--content", choices=("corners", "checker"), default="corners")
    ap.add_argument("--box-widen", type=float, default=1.0,
                    help="informational: install the built-in pattern with every smoothing box widened by this factor "
                         "(1.73 = what the vocabulary's statistics favour, tools/pattern/README.md); the oracle of the "
                         "cpu_baseline leg gets the same pattern")
    ap.add_argument("--pipelined-lanes", type=int, default=4,
                    help="lanes of the `pipelined_lanes` leg (okvfe_set_internal_lanes(-K) on the one context of `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-fed / dense-content legs")
    ap.add_argument("--exact-steps", action="store_true",
                    help="skip the informational `long_region` leg (the timed region is always exactly --steps)")
    args = ap.parse_args()
    steps_flag = args.steps is not None
    if args.steps is None:
        args.steps = 85 if args.workload == "euroc" and args.batch is None else (340 if args.workload == "euroc" else 100)

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...)")
    import torch
    from okvis2_amd import capi, synth
    if args.lanes > 1 and args.stagger:
        capi.set_heavy_kernel_chaining(1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if world > 1 or (args.workload == "hilti" and args.split == "cameras"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload == "map":
        res = run_map_workload(args, torch, capi, synth, dev)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.batch is None:
        # frames per step: large enough that the latency-bound selection has several images per CU in flight and
        # the launch tails are amortised (EuRoC 768 -> 1536 -> 3072 frames: 638 -> 667 -> 685 k stereo-frames/s,
        # TUM-VI 256 -> 1024: +16 %, Hilti 192 -> 960 multiframes: +17 %); `batch_sweep` carries the curve
        args.batch = {"euroc": 3072, "tumvi": 1024, "hilti": 960, "mono640": 3072, "tumvi512": 1536, "d455": 1536,
                      "d435i": 1536}[args.workload]
    if args.workload == "hilti" and args.split == "cameras":
        res = run_hilti_split_cameras(args, torch, dist, capi, synth, world, rank, dev)
        if res is not None:
            print(json.dumps(res), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    cfg = {"euroc": synth.euroc_config, "tumvi": synth.tumvi1024_config,
           "hilti": synth.hilti_config, "mono640": synth.mono640_config, "tumvi512": synth.tumvi512_config,
           "d455": synth.d455_config, "d435i": synth.d435i_config}[args.workload]()
    if args.workload == "hilti":
        # frame-sharded variant: cameras 0/1 form the forward stereo pair (shared intrinsics so that
        # the synthetic disparity is epipolar-consistent), 2..4 get independent images
        cfg.cams = [cfg.cams[0], cfg.cams[0]] + list(cfg.cams[2:])
    if os.environ.get("OKVFE_BENCH_MAXKP"):  # experiment knob: keypoint capacity of the context
        cfg.max_kpts = int(os.environ["OKVFE_BENCH_MAXKP"])
    B = args.batch
    C = len(cfg.cams)  # images per multiframe
    n_img = C * B
    n_content = min(args.distinct, B)  # distinct stereo pairs of the batch's content
    _, base = make_inputs(cfg, B, n_content, 1000 + 977 * rank, args.content, tile=False)
    d_img = tile_on_device(base, n_img, dev)
    distinct = min(n_content, 16)  # frames whose results are downloaded for statistics and the parity legs
    # `--lanes` independent contexts, each with its own HIP stream and B / lanes stereo frames of
    # the batch.  Frames are independent units, so this is the same sharding as across GPUs.
    S = max(1, min(args.lanes, B))
    while B % S or B // S < distinct:  # the parity leg checks the first `distinct` frames of lane 0
        S -= 1
    Bl = B // S
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[min(1, C - 1)].fu + cfg.cams[min(1, C - 1)].fv)
    d_match = torch.zeros((B, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8,
                          device=dev)
    cam_ids = np.array(list(range(C)) * Bl, dtype=np.int32)
    # host parameters differ from step to step (N_VARIANTS sets, cycled): every call goes through
    # the library's parameter upload, as every real frame would
    grav_v = [gravity_variant(v, C * Bl) for v in range(N_VARIANTS)]
    pairs_v = []
    for v in range(N_VARIANTS):
        T0, T1 = pose_variant(cfg, v)
        pairs = []
        for i in range(Bl):
            sp = capi.StereoPair()
            sp.image0, sp.image1 = C * i, C * i + (1 if C > 1 else 0)
            sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
            sp.f0, sp.f1 = f0, f1
            pairs.append(sp)
        pairs_v.append((capi.StereoPair * Bl)(*pairs))
    lanes = []
    for l in range(S):
        lfe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold,
                            cfg.max_kpts, match_threshold=cfg.match_threshold, max_batch=C * Bl,
                            num_cameras=C, device=local_rank, max_candidates=args.max_candidates,
                            box_scale=args.box_widen)
        for ci, cam in enumerate(cfg.cams):
            lfe.set_camera(ci, cam)
        if args.box_widen != 1.0 and l == 0:
            oracle_follows_pattern(lfe)
        st = torch.cuda.Stream(device=dev)
        lanes.append((lfe, st, d_img[C * l * Bl:].data_ptr(), d_match[l * Bl:].data_ptr()))
    fe = lanes[0][0]
    n_lane_img = C * Bl
    h_img = None  # pinned host copy of the images for the host-fed legs
    state = {"step": 0}

    def ensure_pinned():
        nonlocal h_img
        if h_img is None:  # host-fed legs only: the tiled batch in pinned memory, copied back from the device
            h_img = torch.empty(d_img.shape, dtype=torch.uint8, pin_memory=True)
            h_img.copy_(d_img)
        return h_img

    def step(feed="device"):
        v = state["step"] % N_VARIANTS
        state["step"] += 1
        for l, (lfe, st, img_ptr, match_ptr) in enumerate(lanes):
            if feed == "host":
                lfe.detect_describe_batch_host(h_img[C * l * Bl:].data_ptr(), n_lane_img, cam_ids,
                                               grav_v[v], st)
            else:
                lfe.detect_describe_batch_device(img_ptr, n_lane_img, cam_ids, grav_v[v], st)
            if C > 1:
                lfe.match_stereo_batch_device(pairs_v[v], match_ptr, st)
        return v

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(n_steps, feed):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = 0
        for _ in range(n_steps):
            v = step(feed)
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        state["elapsed_local"] = el
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, v

    if args.feed == "host":
        ensure_pinned()
    torch.cuda.synchronize()  # allocations / fills above ran on torch's default stream
    # clock ramp: a box that sat idle while the inputs were generated runs its first tens of
    # milliseconds below its sustained clocks; untimed, before the W warm-up steps of the contract
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        step(args.feed)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(args.feed)
    torch.cuda.synchronize()
    if os.environ.get("OKVFE_PMC_CALIB"):
        # counter calibration for the --pmc passes (tools/collect_profiles.sh): a device-to-device
        # copy of known size in the same process, so FETCH_SIZE / WRITE_SIZE units can be checked
        cal_src = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
        cal_dst = torch.empty_like(cal_src)
        for _ in range(3):
            torch.bitwise_not(cal_src, out=cal_dst)  # a kernel name of its own in the csv
        torch.cuda.synchronize()
    # timed region: only the dominant kernel (score+NMS) carries HIP events, on its launch stream;
    # the full per-stage breakdown is taken in a short extra pass after the timed region
    for lane in lanes:
        lane[0].profile_enable(True, stages=("harris",))
    # the timed region is EXACTLY args.steps steps (the contract).  A region shorter than
    # MIN_TIMED_S cannot resolve a 2 % change, so a second, longer region is timed as well and
    # reported beside it as `long_region` -- `value` always comes from the requested steps.
    elapsed, last_v = timed(args.steps, args.feed)
    elapsed_local = state["elapsed_local"]
    steps_timed = args.steps
    long_region = None
    if elapsed < MIN_TIMED_S and not args.exact_steps and steps_flag:
        reps = int(math.ceil(MIN_TIMED_S / max(elapsed, 1e-6)))
        el_long, last_v = timed(args.steps * reps, args.feed)
        long_region = {"steps": args.steps * reps, "ms_per_step": 1e3 * el_long / (args.steps * reps),
                       "value": world * B * args.steps * reps / el_long,
                       "note": "the same step repeated until the region lasts >= 0.5 s (a %.0f ms region is "
                               "noisier); informational, `value` is the requested region" % (1e3 * elapsed)}
    # capacity check of EVERY image of every lane (outside the timed region): an overflowed NMS
    # candidate list would have left that image without keypoints
    for lane in lanes:
        lane[0].check_capacity(n_lane_img)
    # multi-GPU sanity for the driver's scaling runs: every rank reports in (an all-reduce of ones must
    # give the world size) and its own rate, so a rank that idled or ran elsewhere shows in the line
    ranks_seen, per_rank = 1, None
    if dist is not None:
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        mine = torch.tensor([B * args.steps / elapsed_local], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [float(v.item()) for v in allv]
    kp_total = 0
    kp_counts = []
    for i in range(min(n_img, C * distinct)):
        k, _, _, _ = fe.download(i)
        kp_total += len(k)
        kp_counts.append(len(k))
    m_host = d_match.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(B, cfg.max_kpts).copy()
    map_free = k1_map_free(fe)
    cand_mean = mean_candidates(capi, fe, n_lane_img)

    def read_profiles():
        acc = {}
        for lane in lanes:  # per-launch averages over all lanes
            for k, v in lane[0].profile_read().items():
                a = acc.setdefault(k, [0.0, 0])
                a[0] += v[0]
                a[1] += v[1]
            lane[0].profile_enable(False)
        return acc

    prof = read_profiles()
    # the CPU leg must see exactly what the GPU computed in its LAST step -> run it before any
    # further step overwrites the context's outputs
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        fe._bench_matches = m_host
        fe._bench_frames = B
        cpu = cpu_baseline(cfg, base, fe, grav_v[last_v], pose_variant(cfg, last_v))
    for lane in lanes:
        lane[0].profile_enable(True)
    for _ in range(3):
        step("device")
    torch.cuda.synchronize()
    prof_all = read_profiles()
    # the same kernel with nothing else on the GPU: one lane alone, harris events only
    lanes[0][0].profile_enable(True, stages=("harris",))
    for _ in range(3):
        lanes[0][0].detect_describe_batch_device(lanes[0][2], n_lane_img, cam_ids, grav_v[0], lanes[0][1])
        torch.cuda.synchronize()
    iso = lanes[0][0].profile_read()["harris"]
    # ... and the same launch WITH the score map written (okvfe_set_keep_score_map(1): the form of rounds 1-3,
    # 5 P algorithmic bytes per image, HBM-bound) -- what the byte mover below is the floor of
    with_map = iso
    # (not under a PMC collection: its per-dispatch means must see ONE form of the kernel)
    if map_free and not os.environ.get("OKVFE_PMC_CALIB"):
        lanes[0][0].set_keep_score_map(True)
        lanes[0][0].profile_enable(True, stages=("harris",))
        for _ in range(5):
            lanes[0][0].detect_describe_batch_device(lanes[0][2], n_lane_img, cam_ids, grav_v[0], lanes[0][1])
            torch.cuda.synchronize()
        with_map = lanes[0][0].profile_read()["harris"]
        lanes[0][0].set_keep_score_map(False)
        lanes[0][0].detect_describe_batch_device(lanes[0][2], n_lane_img, cam_ids, grav_v[0], lanes[0][1])
        torch.cuda.synchronize()
    # ... and its byte mover: the same loads and stores on the same layout without the arithmetic
    # (okvfe_harris_byte_mover_device) -- the kernel's own memory floor, same box, same minute
    mover = None
    try:
        lanes[0][0].profile_enable(True, stages=("harris",))
        for _ in range(5):
            lanes[0][0].harris_byte_mover_device(lanes[0][2], n_lane_img, lanes[0][1])
            torch.cuda.synchronize()
        mover = lanes[0][0].profile_read()["harris"]
    except capi.OkvfeError:
        mover = None
    # the score kernel WITHOUT the fused NMS (okvfe_harris_score_device), same images, for reference
    d_sc = torch.empty((n_lane_img, cfg.h, cfg.w), dtype=torch.int32, device=dev)
    lanes[0][0].profile_enable(True, stages=("harris",))
    for _ in range(5):
        lanes[0][0].harris_score_device(lanes[0][2], n_lane_img, d_sc.data_ptr(), lanes[0][1])
    torch.cuda.synchronize()
    solo = lanes[0][0].profile_read()["harris"]
    lanes[0][0].profile_enable(False)
    # what plain streaming kernels reach on this box, same run
    d_cp = torch.empty_like(d_sc)
    d_cp.copy_(d_sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        d_cp.copy_(d_sc)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 2.0 * d_sc.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    e0.record()
    for _ in range(5):
        d_cp.zero_()
    e1.record()
    torch.cuda.synchronize()
    fill_gbps = d_sc.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    # the score kernel's own byte mix through a trivial kernel: read 1 B, write 4 B per pixel
    # (torch's u8 -> int32 conversion of the same batch into the same buffer)
    d_px = d_img.reshape(-1)[:d_cp.numel()].view(d_cp.shape) if d_img.numel() >= d_cp.numel() else None
    widen_ms = None
    if d_px is not None:
        d_cp.copy_(d_px)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            d_cp.copy_(d_px)
        e1.record()
        torch.cuda.synchronize()
        widen_ms = e0.elapsed_time(e1) / 5
    del d_sc, d_cp

    extras = {}
    if not args.no_extras and args.feed == "device":
        # (1) host-fed: the same step with every image crossing PCIe from pinned memory, the copy of
        # step k+1 overlapped with the kernels of step k inside the library
        ensure_pinned()
        for _ in range(2):
            step("host")
        n_h = max(3, min(args.steps, 8))
        el_h, _ = timed(n_h, "host")
        extras["host_fed"] = {
            "value": world * B * n_h / el_h, "unit": "multiframes/s" if C > 2 else
            ("stereo-frames/s" if C == 2 else "frames/s"), "steps": n_h,
            "ms_per_step": 1e3 * el_h / n_h,
            "pcie_GBps_per_gpu": C * B * cfg.w * cfg.h * n_h / el_h / 1e9,
            "note": "okvfe_detect_describe_batch_host: pinned host images, H2D copy on the "
                    "library's copy stream into a double buffer, kernels wait through events; "
                    "PCIe-inclusive, never `value`"}
        # (1b) the whole step with the score map kept (okvfe_set_keep_score_map(1)): the HBM-bound form of K1,
        # what rounds 1-3 measured
        if map_free:
            for lane in lanes:
                lane[0].set_keep_score_map(True)
            for _ in range(2):
                step("device")
            n_m = max(3, min(args.steps, 60))
            el_m, _ = timed(n_m, "device")
            for lane in lanes:
                lane[0].set_keep_score_map(False)
            step("device")
            torch.cuda.synchronize()
            extras["score_map_kept"] = {
                "value": world * B * n_m / el_m, "steps": n_m, "ms_per_step": 1e3 * el_m / n_m,
                "note": "the same step with okvfe_set_keep_score_map(1): K1 writes its 4 B per pixel again "
                        "(roofline.with_score_map is that kernel)"}
        # (1c) the same B frames per step through FOUR contexts on four HIP streams (B / 4 frames each, no
        # chaining between them): the latency-bound selection / descriptor / matcher kernels of one lane run
        # under the VALU-bound score kernel of another.  Same sharding as across GPUs, inside one GPU.
        if S == 1 and C == 2 and B % 4 == 0 and B // 4 >= distinct:
            B4 = B // 4
            n4 = C * B4
            lanes4 = []
            for l in range(4):
                lfe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold,
                                    cfg.max_kpts, match_threshold=cfg.match_threshold, max_batch=n4,
                                    num_cameras=C, device=local_rank, max_candidates=args.max_candidates)
                for ci, cam in enumerate(cfg.cams):
                    lfe.set_camera(ci, cam)
                lanes4.append((lfe, torch.cuda.Stream(device=dev), d_img[C * l * B4:].data_ptr(),
                               d_match[l * B4:].data_ptr()))
            pairs4 = [(capi.StereoPair * B4)(*pv[:B4]) for pv in pairs_v]

            def step4():
                v = state["step"] % N_VARIANTS
                state["step"] += 1
                for lfe, st, img_ptr, match_ptr in lanes4:
                    lfe.detect_describe_batch_device(img_ptr, n4, cam_ids[:n4], grav_v[v][:n4], st)
                    lfe.match_stereo_batch_device(pairs4[v], match_ptr, st)

            for _ in range(3):
                step4()
            n_4 = max(3, min(args.steps, 40))
            barrier()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(n_4):
                step4()
            torch.cuda.synchronize()
            barrier()
            el_4 = time.perf_counter() - t4
            if dist is not None:
                t = torch.tensor([el_4], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el_4 = float(t.item())
            for lane in lanes4:
                lane[0].check_capacity(n4)
            # the lanes computed what one context computes: the same host parameters through the single context,
            # then keypoints, descriptors, back-projections and match rows of the checked frames byte by byte
            m4 = d_match[:distinct].cpu().numpy().copy()
            out4 = [lanes4[0][0].download(i) for i in range(C * distinct)]
            state["step"] -= 1
            step("device")
            torch.cuda.synchronize()
            m1 = d_match[:distinct].cpu().numpy()
            same = all(np.array_equal(p.view(np.uint8), q.view(np.uint8))
                       for i in range(C * distinct) for p, q in zip(out4[i], fe.download(i)))
            same = same and all(np.array_equal(m4[f, :len(out4[C * f][0])], m1[f, :len(out4[C * f][0])])
                                for f in range(distinct))
            extras["four_lanes"] = {
                "value": world * B * n_4 / el_4, "steps": n_4, "ms_per_step": 1e3 * el_4 / n_4,
                "stereo_frames_per_launch": B4, "outputs_equal_single_context": bool(same),
                "note": "the same %d stereo frames per step as `value`, split over 4 contexts / HIP streams per GPU; "
                        "`value` stays on one context so that the K1 roofline is taken with the GPU to itself" % B}
            for lane in lanes4:
                lane[0].close()
            del lanes4
        # (1c') the same on ONE context: okvfe_set_internal_lanes(-4), pipelined lanes -- the call does not join its four
        # slices onto the caller's stream and the matcher runs on the lane streams too, so lane l starts the next step
        # behind its own previous work (what the four contexts above get from being separate); joined at the end
        if S == 1 and C == 2 and B % 4 == 0 and B // 4 >= distinct:
            try:
                fe.set_internal_lanes(-args.pipelined_lanes)
                for _ in range(3):
                    step("device")
                n_p = max(3, min(args.steps, 40))
                el_p, _ = timed(n_p, "device")  # (torch.cuda.synchronize() inside covers the lane streams: device-wide)
                fe.lanes_join(lanes[0][1])
                torch.cuda.synchronize()
                fe.check_capacity(n_lane_img)
                mp_ = d_match[:distinct].cpu().numpy().copy()
                outp = [fe.download(i) for i in range(C * distinct)]
                fe.set_internal_lanes(0)
                state["step"] -= 1
                step("device")
                torch.cuda.synchronize()
                m1 = d_match[:distinct].cpu().numpy()
                same = all(np.array_equal(p.view(np.uint8), q.view(np.uint8))
                           for i in range(C * distinct) for p, q in zip(outp[i], fe.download(i)))
                same = same and all(np.array_equal(mp_[f, :len(outp[C * f][0])], m1[f, :len(outp[C * f][0])])
                                    for f in range(distinct))
                extras["pipelined_lanes"] = {
                    "value": world * B * n_p / el_p, "steps": n_p, "ms_per_step": 1e3 * el_p / n_p, "internal_lanes": -args.pipelined_lanes,
                    "outputs_equal_unsplit_call": bool(same),
                    "note": "ONE context, one caller stream, okvfe_set_internal_lanes(-K): K slices per call on the "
                            "context's own streams without a join per call (include/okvfe.h states the contract)"}
            except Exception as e:  # (an informational leg must not take the line down)
                extras["pipelined_lanes"] = {"error": str(e)[:300]}
            finally:
                try:
                    fe.set_internal_lanes(0)
                except Exception:
                    pass
        # (1d) the content of rounds 1-5: 16 distinct stereo pairs tiled over the batch (selection and matcher then see
        # 32 different images per step instead of 2 x --distinct)
        if args.content == "corners" and n_content > 16:
            d_img.copy_(tile_on_device(base[:C * 16], n_img, dev))
            for _ in range(2):
                step("device")
            n_16 = max(3, min(args.steps, 60))
            el_16, _ = timed(n_16, "device")
            extras["sixteen_distinct"] = {
                "value": world * B * n_16 / el_16, "steps": n_16, "ms_per_step": 1e3 * el_16 / n_16,
                "note": "the same step on the first 16 of the %d distinct stereo pairs, tiled: the content of the "
                        "round 1-5 lines" % n_content}
        # (2) dense content: tied checker corners that all pass the uniformity stage (~700
        # keypoints per image): the matcher's 700 x 700 regime
        if args.content == "corners" and C > 1:
            _, base_d = make_inputs(cfg, B, distinct, 5000 + 977 * rank, "checker", tile=False)
            d_img.copy_(tile_on_device(base_d, n_img, dev))
            for _ in range(4):
                step("device")
            n_d = max(3, min(args.steps, 60))
            el_d, _ = timed(n_d, "device")
            for lane in lanes:
                lane[0].check_capacity(n_lane_img)
            kp_d = sum(len(fe.download(i)[0]) for i in range(min(n_img, C * distinct)))
            extras["dense_content"] = {
                "value": world * B * n_d / el_d, "steps": n_d, "ms_per_step": 1e3 * el_d / n_d,
                "mean_keypoints_per_image": kp_d / max(1, min(n_img, C * distinct)),
                "note": "exact two-level checker cells: tied maxima all pass the uniformity stage, "
                        "the stereo matcher works on ~700 x 700 descriptors per frame"}
        # (2b) real content: windows of the reference's own test image (natural texture: carpet, a
        # checkerboard, saturated regions, JPEG blocks -- candidate density, plateaus and match ambiguity
        # differ from the synthetic cells)
        if args.content == "corners" and C == 2:
            imgs_r = real_inputs(cfg, B, distinct)
            if imgs_r is not None:
                d_img.copy_(tile_on_device(imgs_r, n_img, dev))
                for _ in range(2):
                    step("device")
                n_r = max(3, min(args.steps, 60))
                el_r, _ = timed(n_r, "device")
                for lane in lanes:
                    lane[0].check_capacity(n_lane_img)
                n_r_img = min(n_img, C * distinct)
                kp_each = [len(fe.download(i)[0]) for i in range(n_r_img)]
                kp_r = sum(kp_each)
                m_all = d_match.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(B, cfg.max_kpts)
                n_match = sum(int((m_all[f, :kp_each[2 * f]]["k1"] >= 0).sum()) for f in range(n_r_img // 2))
                extras["real_content"] = {
                    "value": world * B * n_r / el_r, "steps": n_r, "ms_per_step": 1e3 * el_r / n_r,
                    "mean_keypoints_per_image": kp_r / max(1, min(n_img, C * distinct)),
                    "mean_candidates_per_image": mean_candidates(capi, fe, n_lane_img),
                    "mean_matches_per_frame": n_match / max(1, n_r_img // 2),
                    "note": "stereo windows of the reference's testImage.jpg (tests/golden/real_image.npz), right "
                            "image = the window 23 px further right; the same parameters and poses as `value`"}
        # (3) SURVEY.md 8 D2: B in {1, 16, 256} through the batch entry points, and the B = 1 latency of
        # the C++ seams (rank 0 of a single-GPU run only: other ranks would share the host)
        if rank == 0 and world == 1 and C == 2:
            torch.cuda.synchronize()
            extras["batch_sweep"] = batch_sweep(cfg, capi, torch, dev, local_rank, base, args.max_candidates)
            extras["latency_b1"] = latency_b1(cfg, base)

    if rank == 0:
        P = cfg.w * cfg.h
        stage_ms = {k: (v[0] / v[1] if v[1] else None) for k, v in prof_all.items()}
        harris_ms = prof["harris"][0] / prof["harris"][1]
        metric, text = WORKLOAD_TEXT[args.workload]
        if args.workload == "hilti":
            text += ", forward pair matched (frame-sharded variant; --split cameras runs the 9-pair rig)"
        unit = {1: "frames/s", 2: "stereo-frames/s"}.get(C, "multiframes/s")
        result = {
            "metric": metric,
            "value": world * B * steps_timed / elapsed,
            "unit": unit,
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "per_rank_value": per_rank,
            "steps": steps_timed,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps_timed,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32 (detect, describe, Hamming) + f64 (match gate)",
            "data": "synthetic",
            "config": {"workload": text,
                       "stereo_frames_per_step_per_gpu": B, "lanes_per_gpu": S,
                       "score_kernels_serialised_across_lanes": bool(S > 1 and args.stagger),
                       "stereo_frames_per_launch": Bl, "distinct_frames": n_content,
                       "content": args.content, "feed": args.feed,
                       "host_parameter_variants": N_VARIANTS,
                       "mean_keypoints_per_image": kp_total / max(1, min(n_img, C * distinct)),
                       "cameras_per_multiframe": C,
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "roofline": roofline_block(P, n_lane_img, harris_ms, {
                "isolated_launch_ms": iso[0] / iso[1],
                "with_score_map": {
                    "note": "the same launch with okvfe_set_keep_score_map(1): 5 P algorithmic bytes per image, "
                            "HBM-bound (the kernel of rounds 1-3); 5 isolated launches after the timed region",
                    "avg_launch_ms": with_map[0] / with_map[1],
                    "achieved": 5.0 * P * n_lane_img / (with_map[0] / with_map[1] * 1e-3) / 1e9,
                    "frac": 5.0 * P * n_lane_img / (with_map[0] / with_map[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": 5 * P * n_lane_img,
                    "byte_mover_ms": (mover[0] / mover[1]) if mover and mover[1] else None,
                    "frac_of_byte_mover": (mover[0] / mover[1]) / (with_map[0] / with_map[1]) if mover and mover[1] else None,
                    "byte_mover_note": "harris_kernel<61, true, PACK, MEMONLY>: the map-writing kernel's own loads and "
                                       "stores (same tiles, same slotted layout, same launch geometry) with the "
                                       "arithmetic, the NMS and the candidate records removed",
                    "copy_kernel_GBps": copy_gbps, "fill_kernel_GBps": fill_gbps,
                    "widen_kernel_ms": widen_ms,
                    "widen_kernel_note": "torch u8 -> int32 conversion of the same images into the same buffer: 1 B in, "
                                         "4 B out per pixel through a kernel that computes nothing",
                    "traffic_over_algorithmic": (lambda pm: pm["hbm_bytes_per_image"] / (5.0 * P) if pm else None)(
                        k1_pmc_for(cfg.w, cfg.h, False)[0]),
                },
                "frac_if_counted_as_5P": 5.0 * P * n_lane_img / (harris_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_if_counted_as_5P_note": "the launch's duration against the bytes the map-writing form moves "
                                              "(continuity with rounds 1-3; NOT this kernel's algorithmic bytes when "
                                              "live_bound is valu)",
                "score_only_launch_ms": solo[0] / solo[1],
                "score_only_note": "harris_kernel<30, false>: the score map alone (no NMS), 5 launches "
                                   "after the timed region",
                "note": ("one lane: the launch has the GPU to itself in the timed region as well"
                         if S == 1 else
                         "avg_launch_ms is taken while the other lanes' kernels share the GPU; "
                         "isolated_* is the same launch with the GPU to itself")},
                cfg.w, cfg.h, map_free, cand_mean),
            "rooflines_other": alu_rooflines(
                stage_ms,
                (sum(kp_counts[C * i] * kp_counts[C * i + 1] for i in range(len(kp_counts) // C)) *
                 (Bl / max(1, len(kp_counts) // C))) if C > 1 else 0,
                kp_total * (n_lane_img / max(1, len(kp_counts))), 4512, n_lane_img),
            "stage_ms_per_launch": stage_ms,
            "stage_ms_note": "all-stage event pass of 3 steps after the timed region; avg_launch_ms of "
                             "the roofline comes from the timed region itself",
        }
        valu_issue_blocks(result["rooflines_other"], stage_ms, n_lane_img, args.workload, args.content)
        result.update(extras)
        if long_region is not None:
            result["long_region"] = long_region
        # the less favourable legs next to `value`, at the top level
        result["value_16_distinct"] = extras.get("sixteen_distinct", {}).get("value")
        result["value_dense"] = extras.get("dense_content", {}).get("value")
        result["value_real_content"] = extras.get("real_content", {}).get("value")
        result["value_score_map_kept"] = extras.get("score_map_kept", {}).get("value")
        result["value_host_fed"] = extras.get("host_fed", {}).get("value")
        result["value_four_lanes"] = extras.get("four_lanes", {}).get("value")
        result["value_pipelined_lanes"] = extras.get("pipelined_lanes", {}).get("value")
        if cpu is not None:
            result["cpu_baseline"] = cpu
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
