#!/usr/bin/env python3
"""K1-only micro-benchmark on the GPU box: harris_kernel over a resident batch, timed with the
library's own HIP-event stage profiling.  Usage: python tools/bench_k1.py [n_images] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis2_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for (w, h) in ((752, 480), (1024, 1024)):
    nn = n if w == 752 else max(1, n * 752 * 480 // (1024 * 1024))
    fe = capi.Frontend(w, h, 38.0, 0, 150, 700, max_batch=1, max_candidates=1024)
    base = np.stack([synth.corners_image(w, h, i) for i in range(8)])
    imgs = torch.from_numpy(np.concatenate([base] * ((nn + 7) // 8))[:nn]).cuda()
    sc = torch.empty((nn, h, w), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        fe.harris_score_device(imgs.data_ptr(), nn, sc.data_ptr(), s)
    torch.cuda.synchronize()
    fe.profile_enable(True)
    for _ in range(reps):
        fe.harris_score_device(imgs.data_ptr(), nn, sc.data_ptr(), s)
    ms, cnt = fe.profile_read()["harris"]
    avg = ms / cnt
    gbps = 5.0 * w * h * nn / (avg * 1e-3) / 1e9
    # copy-kernel reference in the same run: same bytes moved by torch (u8 read + int32 write)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst = torch.empty_like(sc)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(reps):
        dst.copy_(sc)
    t1.record()
    torch.cuda.synchronize()
    copy_gbps = 8.0 * w * h * nn / (t0.elapsed_time(t1) / reps * 1e-3) / 1e9
    torch.cuda.synchronize()
    t0.record()
    for _ in range(reps):
        dst.zero_()
    t1.record()
    torch.cuda.synchronize()
    fill_gbps = 4.0 * w * h * nn / (t0.elapsed_time(t1) / reps * 1e-3) / 1e9
    print(f"   int32 fill kernel (write-only, same bytes as the score map): {fill_gbps:.0f} GB/s = "
          f"{t0.elapsed_time(t1) / reps * 1e3:.1f} us")
    print(f"{w}x{h} x{nn}: harris {avg*1e3:.1f} us  {gbps:.0f} GB/s algorithmic "
          f"({gbps/8000:.1%} of 8 TB/s); int32 copy kernel {copy_gbps:.0f} GB/s")
