#!/usr/bin/env python3
"""Generates tests/golden/frontend_golden.npz: small seeded inputs and the CPU oracle's outputs
for every stage of the path (score map, NMS maxima, selected keypoints, descriptors in the three
extraction modes, back-projections, gated stereo matches).

The reference ships no golden vectors for this path, so these certify
"HIP path == oracle == committed vectors", not "oracle == reference binary".
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from okvis2_amd import synth  # noqa: E402

W, H = 256, 192
RADIUS, THR, MAXK, MTHR = 12.0, 40, 300, 60
cams = [synth.Camera(W, H, 160.0, 161.0, 127.3, 95.6, 1, (-0.2834, 0.0739, 0.00019, 1.76e-05)),
        synth.Camera(W, H, 159.0, 160.5, 130.1, 97.2, 1, (-0.2836, 0.0745, -0.0001, -3.5e-05))]
out = {"params": np.array([W, H, RADIUS, THR, MAXK, MTHR], dtype=np.float64),
       "cams": np.array([[c.fu, c.fv, c.cu, c.cv, c.dist_type, *c.d] for c in cams])}
L, R, disp = synth.stereo_pair(W, H, 4242)
out["left"], out["right"] = L, R
score = O.harris_score(L)
out["score_left"] = score
out["nms_left"] = O.nms(score, THR)
for ci, img in enumerate((L, R)):
    cam = cams[ci]
    rays, jac = O.awareness_maps(cam)
    kd = O.detect(img, RADIUS, 0, THR, MAXK)
    out[f"kp_detect_{ci}"] = kd
    for mode, name in ((O.MODE_UPRIGHT, "upright"), (O.MODE_GRADIENT, "gradient")):
        k, d = O.describe(img, kd, mode)
        out[f"kp_{name}_{ci}"], out[f"desc_{name}_{ci}"] = k, d
    k, d = O.describe(img, kd, O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.1, 0.98, -0.05))
    bp, bv = O.backproject_keypoints(cam, k)
    out[f"kp_aware_{ci}"], out[f"desc_aware_{ci}"], out[f"bp_{ci}"], out[f"bpv_{ci}"] = k, d, bp, bv
T0, T1 = synth.stereo_poses(0.11)
f0, f1 = 0.5 * (cams[0].fu + cams[0].fv), 0.5 * (cams[1].fu + cams[1].fv)
m = O.match_stereo(out["desc_aware_0"], out["kp_aware_0"], out["bp_0"], out["bpv_0"],
                   out["desc_aware_1"], out["kp_aware_1"], out["bp_1"], out["bpv_1"], T0, T1, f0, f1,
                   MTHR)
out["match_stereo"] = m
# noise image (the reference's own smoke-test input shape, TestFrame.cpp:83-85, scaled down)
N = synth.noise_image(W, H, 77)
out["noise"] = N
out["kp_noise"] = O.detect(N, 34.0, 0, 800, 450)
dst = os.path.join(ROOT, "tests", "golden", "frontend_golden.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes;", len(out["kp_detect_0"]), "kps,",
      int((m["k1"] >= 0).sum()), "matches, disparity", disp, "noise kps", len(out["kp_noise"]))
