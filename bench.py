#!/usr/bin/env python3
"""bench.py -- front-end throughput on MI355X: stereo-frames/s (detect + describe + match).

Workload (BASELINE.json metric, configs[2]): synthetic EuRoC-shaped stereo, 752x480 x 2 cameras,
front-end parameters of config/euroc.yaml:63-67 (uniformity radius 38, Harris threshold 150,
<= 700 keypoints, Hamming threshold 60), camera-aware gravity-aligned BRISK2 extraction,
matchStereo with the FP64 triangulation gate.  One "step" = one batch of `--batch` stereo frames
(default 768 = 1536 images) through the whole hot path (K1 score map with the K2 NMS fused in ->
K3 sort + greedy selection, K4 sub-pixel -> K6 describe -> compaction + back-projection -> K7 gated
stereo match), inputs resident in HBM.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
N > 1 is launched by torch.distributed.run (one rank per GPU); stereo frames are independent
units, so ranks shard batches with no data-path collective ("scaling": "weak").
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable


def make_inputs(cfg, n_frames, n_distinct, seed0):
    """Multiframes of C = len(cfg.cams) images: cameras 0/1 are a synthetic stereo pair, further
    cameras (Hilti-shaped rig) look elsewhere and get independent images."""
    from okvis2_amd import synth
    C = len(cfg.cams)
    base = []
    for i in range(n_distinct):
        L, R, _ = synth.stereo_pair(cfg.w, cfg.h, seed0 + i)
        base.append(L)
        base.append(R)
        for c in range(2, C):
            base.append(synth.corners_image(cfg.w, cfg.h, seed0 + 7919 * c + i))
    base = np.stack(base)  # [C*n_distinct, H, W]
    reps = (n_frames + n_distinct - 1) // n_distinct
    return np.concatenate([base] * reps)[: C * n_frames], base


def cpu_baseline(cfg, base_imgs, fe, budget_s=12.0):
    """Times the CPU oracle (a port of the algorithm, NOT the reference binary, which cannot be
    built here) with the reference's threading shape: one thread per camera for detect+describe
    (ThreadedSlam.cpp:434-448), matchStereo single-threaded (Frontend.cpp:2016).  Also compares
    the oracle's outputs with the GPU's for the same frames (the oracle acting as checker)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from okvis2_amd import synth
    maps = [O.awareness_maps(c) for c in cfg.cams]
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f = [0.5 * (c.fu + c.fv) for c in cfg.cams]
    grav = (0.0, 1.0, 0.0)
    C = len(cfg.cams)
    n_distinct = len(base_imgs) // C
    out = [None] * C

    def work(ci, img):
        cam = cfg.cams[ci]
        k, d = O.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                 O.MODE_CAMERA_AWARE, maps[ci][0], maps[ci][1], np.float32(cam.fu),
                                 grav)
        bp, bv = O.backproject_keypoints(cam, k)
        out[ci] = (k, d, bp, bv)

    done, checked, mismatches = 0, 0, 0
    t0 = time.perf_counter()
    while True:
        i = done % n_distinct
        ths = [threading.Thread(target=work, args=(c, base_imgs[C * i + c])) for c in range(1, C)]
        for th in ths:
            th.start()
        work(0, base_imgs[C * i])
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
            ok = m is None or np.array_equal(fe._bench_matches[i, :len(k0)]["k1"], m["k1"])
            for c in range(C):
                g = fe.download(C * i + c)
                ok = ok and np.array_equal(g[0].view(np.uint8), out[c][0].view(np.uint8)) \
                    and np.array_equal(g[1], out[c][1])
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
            "parity_checked_frames": checked, "parity_mismatches": mismatches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=768, help="stereo frames per step per GPU")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic stereo pairs")
    ap.add_argument("--max-candidates", type=int, default=16384)
    ap.add_argument("--lanes", type=int, default=1,
                    help="independent contexts/streams the batch is split over on each GPU.  With "
                         "3 lanes and the score kernels serialised across them (--stagger) the "
                         "latency-bound kernels of one lane hide behind the score kernel of "
                         "another: about +10 %% frames/s, but the score kernel then shares the GPU "
                         "while it is being timed (roofline.frac 0.35 instead of 0.44)")
    ap.add_argument("--stagger", type=int, default=1,
                    help="with --lanes > 1: serialise the score kernels of the lanes (library env "
                         "OKVFE_SCORE_TOKEN) so that the lanes run out of phase")
    ap.add_argument("--workload", choices=("euroc", "tumvi", "hilti", "mono640"), default="euroc",
                    help="euroc = the BASELINE.json metric (752x480 stereo); tumvi = configs[3], "
                         "1024x1024 equidistant stereo with config/tumvi_slam_1024.yaml parameters; "
                         "hilti = configs[4] shape, 5 equidistant 720x540 cameras per multiframe "
                         "(hilti_challenge_2022.yaml parameters), the forward pair matched "
                         "; mono640 = configs[1], 640x480 mono, ~1000 keypoints, detect+describe "
                         "only (informational; batch 192 by default for these)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.lanes > 1 and args.stagger:
        os.environ["OKVFE_SCORE_TOKEN"] = "1"  # read by libokvfe.so at its first batch call
    import torch
    from okvis2_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = {"euroc": synth.euroc_config, "tumvi": synth.tumvi1024_config,
           "hilti": synth.hilti_config, "mono640": synth.mono640_config}[args.workload]()
    if args.workload == "hilti":
        # synthetic rig: cameras 0/1 form the forward stereo pair (shared intrinsics so that the
        # synthetic disparity is epipolar-consistent), 2..4 look elsewhere (no FoV overlap)
        cfg.cams = [cfg.cams[0], cfg.cams[0]] + list(cfg.cams[2:])
    if args.workload != "euroc" and args.batch == 768:
        args.batch = 192
    if os.environ.get("OKVFE_BENCH_MAXKP"):  # experiment knob: keypoint capacity of the context
        cfg.max_kpts = int(os.environ["OKVFE_BENCH_MAXKP"])
    B = args.batch
    C = len(cfg.cams)  # images per multiframe
    n_img = C * B
    distinct = min(args.distinct, B)
    imgs, base = make_inputs(cfg, B, distinct, 1000 + 977 * rank)
    d_img = torch.from_numpy(imgs).to(dev)
    # `--lanes` independent contexts, each with its own HIP stream and B / lanes stereo frames of
    # the batch: the latency-bound kernels of one lane (greedy select, sort, gated match) overlap
    # with the throughput-bound ones (score+NMS, describe) of the others.  Frames are independent
    # units, so this is the same sharding as across GPUs, applied within one.
    S = max(1, min(args.lanes, B))
    while B % S or B // S < distinct:  # the parity leg checks the first `distinct` frames of lane 0
        S -= 1
    Bl = B // S
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[min(1, C - 1)].fu + cfg.cams[min(1, C - 1)].fv)
    d_match = torch.zeros((B, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8,
                          device=dev)
    cam_ids = np.array(list(range(C)) * Bl, dtype=np.int32)
    grav = np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (C * Bl, 1))
    pairs = []
    for i in range(Bl):
        sp = capi.StereoPair()
        sp.image0, sp.image1 = C * i, C * i + (1 if C > 1 else 0)
        sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
        sp.f0, sp.f1 = f0, f1
        pairs.append(sp)
    pairs_arr = (capi.StereoPair * Bl)(*pairs)
    lanes = []
    for l in range(S):
        lfe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold,
                            cfg.max_kpts, match_threshold=cfg.match_threshold, max_batch=C * Bl,
                            num_cameras=C, device=local_rank, max_candidates=args.max_candidates)
        for ci, cam in enumerate(cfg.cams):
            lfe.set_camera(ci, cam)
        st = torch.cuda.Stream(device=dev) if S > 1 else torch.cuda.current_stream()
        lanes.append((lfe, st.cuda_stream, d_img[C * l * Bl:].data_ptr(), d_match[l * Bl:].data_ptr(), st))
    fe = lanes[0][0]
    n_lane_img = C * Bl

    def step():
        for lfe, stream, img_ptr, match_ptr, _ in lanes:
            lfe.detect_describe_batch_device(img_ptr, n_lane_img, cam_ids, grav, stream)
            if C > 1:
                lfe.match_stereo_batch_device(pairs_arr, match_ptr, stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
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
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # capacity check (outside the timed region): download() raises OKVFE_ERR_CAPACITY if any NMS
    # candidate list of the checked images overflowed its buffer
    kp_total = 0
    for i in range(min(n_img, C * distinct)):
        k, _, _, _ = fe.download(i)
        kp_total += len(k)

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
    for lane in lanes:
        lane[0].profile_enable(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof_all = read_profiles()
    # the same kernel with nothing else on the GPU: one lane alone, harris events only
    lanes[0][0].profile_enable(True, stages=("harris",))
    for _ in range(3):
        lanes[0][0].detect_describe_batch_device(lanes[0][2], n_lane_img, cam_ids, grav, lanes[0][1])
        torch.cuda.synchronize()
    iso = lanes[0][0].profile_read()["harris"]
    # the score kernel WITHOUT the fused NMS (okvfe_harris_score_device: the kernel behind the
    # stand-alone K1 entry point), same images, for reference next to the fused launch
    d_sc = torch.empty((n_lane_img, cfg.h, cfg.w), dtype=torch.int32, device=dev)
    lanes[0][0].profile_enable(True, stages=("harris",))
    for _ in range(5):
        lanes[0][0].harris_score_device(lanes[0][2], n_lane_img, d_sc.data_ptr(), lanes[0][1])
    torch.cuda.synchronize()
    solo = lanes[0][0].profile_read()["harris"]
    lanes[0][0].profile_enable(False)
    # what a trivial device-to-device copy of the score map reaches on this box, same run
    # (fraction of the nominal AND of the measured attainable bandwidth)
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
    del d_sc, d_cp

    if rank == 0:
        P = cfg.w * cfg.h
        n_img_launch = n_lane_img
        stage_ms = {k: (v[0] / v[1] if v[1] else None) for k, v in prof_all.items()}
        harris_ms = prof["harris"][0] / prof["harris"][1]
        achieved = 5.0 * P * n_img_launch / (harris_ms * 1e-3) / 1e9
        # HBM traffic of the K1 launch from the PMC passes committed under profiles/ (rocprofv3
        # cannot run inside this process); only valid for the launch shape it was measured on
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "round1_v3_k1_pmc.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            # same kernel, same image shape: per-image HBM bytes x the images of one launch here
            if pmc.get("algorithmic_bytes_per_launch") == 5 * P * pmc.get("images_per_launch", 0):
                traffic = pmc["hbm_bytes_per_image"] * n_img_launch
                traffic_src = ("profiles/round1_v3_k1_pmc.json (FETCH_SIZE/WRITE_SIZE passes at %d "
                               "images per launch, scaled per image)" % pmc["images_per_launch"])
        m = d_match.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(B, cfg.max_kpts)
        fe._bench_matches = m
        result = {
            "metric": {"euroc": "front-end stereo-frames/s (detect+describe+match), 752x480 stereo",
                       "tumvi": "front-end stereo-frames/s (detect+describe+match), 1024x1024 stereo "
                                "(TUM-VI)",
                       "hilti": "front-end multiframes/s (5 x detect+describe + forward-pair match), "
                                "720x540 x 5 cameras (Hilti 2022)",
                       "mono640": "front-end frames/s (detect+describe), 640x480 mono"}[args.workload],
            "value": world * B * args.steps / elapsed,
            "unit": {1: "frames/s", 2: "stereo-frames/s"}.get(C, "multiframes/s"),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32 (detect, describe, Hamming) + f64 (match gate)",
            "data": "synthetic",
            "config": {"workload": ("EuRoC-shaped 752x480 stereo, euroc.yaml front-end params "
                                    "(radius 38, thr 150, <=700 kpts, match thr 60)"
                                    if args.workload == "euroc" else
                                    "TUM-VI-shaped 1024x1024 equidistant stereo, tumvi_slam_1024.yaml "
                                    "front-end params (radius 50, thr 5, <=1000 kpts, match thr 60)"
                                    if args.workload == "tumvi" else
                                    "640x480 mono (radius 10, thr 5, <=1000 kpts), detect+describe"
                                    if args.workload == "mono640" else
                                    "Hilti-shaped rig, 5 equidistant 720x540 cameras, "
                                    "hilti_challenge_2022.yaml front-end params (radius 50, thr 20, "
                                    "<=700 kpts, match thr 60), forward pair matched"),
                       "stereo_frames_per_step_per_gpu": B, "lanes_per_gpu": S,
                       "score_kernels_serialised_across_lanes": bool(S > 1 and args.stagger),
                       "stereo_frames_per_launch": Bl, "distinct_frames": distinct,
                       "mean_keypoints_per_image": kp_total / max(1, min(n_img, C * distinct)),
                       "cameras_per_multiframe": C,
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "roofline": {"kernel": "harris_kernel<30, true> (K1 score map + fused K2 NMS)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 5 * P * n_img_launch,
                         "avg_launch_ms": harris_ms,
                         "isolated_launch_ms": iso[0] / iso[1],
                         "isolated_frac": 5.0 * P * n_img_launch / (iso[0] / iso[1] * 1e-3) / 1e9
                                          / HBM_PEAK_GBPS,
                         "copy_kernel_GBps": copy_gbps,
                         "frac_of_copy_kernel": achieved / copy_gbps,
                         "score_only_launch_ms": solo[0] / solo[1],
                         "score_only_frac": 5.0 * P * n_img_launch / (solo[0] / solo[1] * 1e-3) / 1e9
                                            / HBM_PEAK_GBPS,
                         "score_only_note": "harris_kernel<30, false>: the score map alone (no NMS), "
                                            "5 launches after the timed region; the pipeline uses "
                                            "the fused kernel because a separate NMS pass re-reads "
                                            "the whole score map (+0.2 ms per 512 images)",
                         "note": ("one lane: the launch has the GPU to itself in the timed region "
                                  "as well" if S == 1 else
                                  "avg_launch_ms is taken while the other lanes' kernels share the "
                                  "GPU; isolated_* is the same launch with the GPU to itself")},
            "stage_ms_per_launch": stage_ms,
            "stage_ms_note": "all-stage event pass of 3 steps after the timed region; avg_launch_ms of "
                             "the roofline comes from the timed region itself",
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, base, fe)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
