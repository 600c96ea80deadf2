#!/usr/bin/env python3
"""Single stereo frame latency through the HOST-buffer API (what a live front-end would call per
multiframe): 2 x okvfe_detect_describe + okvfe_match_stereo, PCIe copies and syncs included.
Usage: python tools/latency.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis2_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = synth.euroc_config()
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold, cfg.max_kpts,
                   match_threshold=cfg.match_threshold, num_cameras=2, max_batch=2)
for ci, cam in enumerate(cfg.cams):
    fe.set_camera(ci, cam)
L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 7)
T0, T1 = synth.stereo_poses(cfg.baseline)
f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)


def frame():
    k0, d0, b0, v0 = fe.detect_describe(L, cam=0, gravity=(0.0, 1.0, 0.0))
    k1, d1, b1, v1 = fe.detect_describe(R, cam=1, gravity=(0.0, 1.0, 0.0))
    return fe.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1)


for _ in range(10):
    m = frame()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    frame()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    fe.detect_describe(L, cam=0, gravity=(0.0, 1.0, 0.0))
td = (time.perf_counter() - t0) / reps * 1e3
print(f"stereo frame (2 x detect_describe + match_stereo, host buffers): median {np.median(ts):.3f} ms, "
      f"p95 {np.percentile(ts, 95):.3f} ms, min {ts.min():.3f} ms; one detect_describe {td:.3f} ms; "
      f"{(m['k1'] >= 0).sum()} matches")
