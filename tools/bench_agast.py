#!/usr/bin/env python3
"""GPU box: AGAST 9-16 score kernel timing (okvfe_harris_score_device with score_type 1) on 512
EuRoC-shaped images of the bench content, of smooth content and of noise."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis2_amd import capi, synth
w, h, n = 752, 480, 512
fe = capi.Frontend(w, h, 20.0, 0, 34, 500, max_batch=1, score_type=capi.SCORE_AGAST_9_16)
yy, xx = np.mgrid[0:h, 0:w]
smooth = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 60 + 128).astype(np.uint8)
smooth[100:200, 300:500] = 30  # a few real corners
for name, gen in (("bench corners", lambda i: synth.corners_image(w, h, i)), ("smooth + one box", lambda i: smooth),
                  ("noise", lambda i: synth.noise_image(w, h, i))):
    base = np.stack([gen(i) for i in range(8)])
    imgs = torch.from_numpy(np.concatenate([base] * (n // 8))).cuda()
    sc = torch.empty((n, h, w), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()  # a real stream: its raw handle is what the library launches on
    torch.cuda.synchronize()
    for _ in range(2):
        fe.harris_score_device(imgs.data_ptr(), n, sc.data_ptr(), st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        fe.harris_score_device(imgs.data_ptr(), n, sc.data_ptr(), st)
    e1.record(st)
    st.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms per {n} images, nonzero scores {float((sc[:8] > 0).float().mean()):.3f}")
