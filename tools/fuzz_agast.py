#!/usr/bin/env python3
"""One-off fuzz (GPU): AGAST 9-16 score maps of random shapes / batch sizes against the oracle, incl. launches whose workgroups
walk several tiles, widths that are not multiples of 4 (byte staging) and images of the smallest size the library takes (64 x 64); then the
BriskFeatureDetector scale space (FAST 5-8 virtual layer, AGAST layers) on random shapes.
usage: python tools/fuzz_agast.py [first_seed] [count]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(31000 + seed)
    kind = seed % 4
    if kind == 0:      # many small images: several tiles per workgroup
        w, h = int(rng.integers(16, 40)) * 4, int(rng.integers(64, 130))
        n = int(rng.integers(1000, 4000))
    elif kind == 1:    # unaligned widths
        w, h = int(rng.integers(64, 400)) | 1, int(rng.integers(64, 300))
        n = int(rng.integers(1, 600))
    elif kind == 2:    # tiny
        w, h = int(rng.integers(64, 80)), int(rng.integers(64, 80))
        n = int(rng.integers(1, 50))
    else:
        w, h = int(rng.integers(16, 300)) * 4, int(rng.integers(64, 500))
        n = int(rng.integers(1, 120))
    D = min(n, 6)
    base = np.stack([synth.noise_image(w, h, seed * 7 + i) if (i + seed) % 2 else synth.corners_image(w, h, seed * 7 + i, cell=int(rng.choice([6, 12])))
                     for i in range(D)])
    reps = (n + D - 1) // D
    imgs = np.concatenate([base] * reps)[:n]
    fe = capi.Frontend(w, h, 20.0, 0, 34, 200, max_batch=n, score_type=capi.SCORE_AGAST_9_16)
    d_img = torch.from_numpy(imgs).cuda()
    d_sc = torch.full((n, h, w), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # the upload and the fill ran on torch's stream
    fe.harris_score_device(d_img.data_ptr(), n, d_sc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.from_numpy(np.stack([O.agast_score(base[i]) for i in range(D)])).cuda()
    idx = torch.arange(n, device="cuda") % D
    ok = bool((d_sc == ref[idx]).all())
    if not ok:
        bad += 1
        d = torch.nonzero(d_sc != ref[idx]).cpu().numpy()
        print(f"MISMATCH seed {seed}: {w}x{h} n={n}: {len(d)} pixels, images {np.unique(d[:, 0])[:8]} (of {len(np.unique(d[:, 0]))}), "
              f"rows {np.unique(d[:, 1])[:12]}, cols {d[:, 2].min()}..{d[:, 2].max()}, first got {int(d_sc[tuple(d[0])])} want {int(ref[d[0][0] % D][d[0][1], d[0][2]])}")
    fe.close()
    del d_img, d_sc
print(f"agast score maps: {count} configurations, {bad} mismatching")
bad2 = 0
for seed in range(first, first + max(4, count // 5)):
    rng = np.random.default_rng(41000 + seed)
    w, h = int(rng.integers(40, 200)) * 4, int(rng.integers(120, 400))
    octaves = int(rng.integers(1, 4))
    img = synth.corners_image(w, h, seed, cell=int(rng.choice([8, 12, 16]))) if seed % 2 else synth.noise_image(w, h, seed)
    fe = capi.Frontend(w, h, 0.0, octaves, 34, 3000, max_batch=1, score_type=capi.SCORE_BRISK_SCALESPACE, max_candidates=1 << 16)
    ref = O.detect(img, 0.0, octaves, 34, 3000, score_type=O.SCORE_BRISK_SCALESPACE)
    got = fe.detect(img)
    try:
        G.assert_keypoints_equal(got, ref)
    except AssertionError as e:
        bad2 += 1
        print(f"SCALESPACE MISMATCH seed {seed}: {w}x{h} octaves {octaves}: {len(got)} vs {len(ref)}: {str(e)[:200]}")
    fe.close()
print(f"scale-space detector: {max(4, count // 5)} configurations, {bad2} mismatching")
sys.exit(1 if bad or bad2 else 0)
