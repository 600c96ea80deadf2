#!/usr/bin/env python3
"""Lab build only: phase times of select_lazy_kernel for image 0 (s_memrealtime, 10-ns ticks), B = 1 and batch."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKVFE_LIB"] = os.path.join(ROOT, "okvis2_amd", "libokvfe_lab.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from okvis2_amd import capi, synth
import bench
cfg = synth.euroc_config()
BS = [int(v) for v in os.environ.get("LAZY_PROF_B", "1,1536").split(",")]
imgs, base = bench.make_inputs(cfg, max(max(BS) // 2, 1), 16, 1000, os.environ.get("LAZY_PROF_CONTENT", "corners"))
def prof(reset=True):
    out = (C.c_ulonglong * 16)()
    assert capi.lib().okvfe_lab_lazy_prof(out, int(reset)) == 0
    v = list(out); n = max(v[0], 1)
    return {"launches": v[0], "init_us": v[1] / n / 100, "blocks_us": v[2] / n / 100, "tail_us": v[3] / n / 100,
            "total_us": v[4] / n / 100, "prefilter_us": v[5] / n / 100, "survivors": v[6] / n, "win_walk_us": v[8] / n / 100, "win_accept_us": v[9] / n / 100, "win_insert_us": v[10] / n / 100, "rounds": v[11] / n, "candidates": v[7] / n, "tables_us": v[12] / n / 100, "count_us": v[13] / n / 100, "sched_us": v[14] / n / 100, "scatter_us": v[15] / n / 100}
for B in BS:
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, max_batch=B, num_cameras=2,
                       max_candidates=16384)
    for ci, cam in enumerate(cfg.cams): fe.set_camera(ci, cam)
    d = torch.from_numpy(imgs[:B]).cuda()
    cam_ids = np.array([0, 1] * (B // 2) if B > 1 else [0], np.int32)
    g = bench.gravity_variant(0, B)
    for _ in range(5): fe.detect_describe_batch_device(d.data_ptr(), B, cam_ids, g, None)
    torch.cuda.synchronize(); prof()
    for _ in range(20):
        fe.detect_describe_batch_device(d.data_ptr(), B, cam_ids, g, None)
        if B == 1: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("B", B, prof())
