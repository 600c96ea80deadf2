#!/usr/bin/env python3
"""GPU box: the published BRISK scale-space detector (score_type 2 = brisk::BriskFeatureDetector(34, 2), the ARM call of
okvis_cv/test/TestFrame.cpp:71-72) + extractor through the batch entry point, on EuRoC-shaped images of the bench content.
usage: python tools/bench_scalespace.py [images] [octaves] [score_type]   (run under rocprofv3 --kernel-trace --stats for the split)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis2_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
octaves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
score_type = int(sys.argv[3]) if len(sys.argv) > 3 else capi.SCORE_BRISK_SCALESPACE
w, h = 752, 480
fe = capi.Frontend(w, h, 0.0 if score_type == capi.SCORE_BRISK_SCALESPACE else 20.0, octaves, 34, 800, max_batch=n,
                   score_type=score_type, max_candidates=1 << 15,
                   rotation_invariant=os.environ.get("SS_UPRIGHT") is None)
base = np.stack([synth.corners_image(w, h, i) for i in range(8)])
imgs = torch.from_numpy(np.concatenate([base] * (n // 8))).cuda()
st = torch.cuda.Stream()
for _ in range(3):
    fe.detect_describe_batch_device(imgs.data_ptr(), n, None, None, st)
st.synchronize()
fe.check_capacity(n)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    fe.detect_describe_batch_device(imgs.data_ptr(), n, None, None, st)
st.synchronize()
dt = (time.perf_counter() - t0) / reps
kp = np.mean([len(fe.download(i)[0]) for i in range(8)])
print(f"score_type {score_type}, octaves {octaves}: {1e3 * dt:.3f} ms per {n} images = {n / dt:.0f} frames/s, {kp:.0f} keypoints per image")
