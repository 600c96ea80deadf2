#!/usr/bin/env python3
"""One-off extended fuzz (GPU): random image sizes / content / thresholds / radii / caps with CAMERA-AWARE extraction on
random radial-tangential and equidistant cameras and random extraction directions, detect + describe against the oracle.
usage: python tools/fuzz_aware.py [first_seed] [count] [all] [wide]   (all: seeds also cycle through upright /
gradient / scale-invariant extraction; wide: every configuration installs the built-in pattern with its boxes widened
by a random factor in [1.05, 2.4] on the GPU and in the oracle -- the WIDE instantiations of the descriptor kernel and
the plain-loop fall-back beyond them)"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ALL_MODES = len(sys.argv) > 3 and sys.argv[3] == "all"
WIDE = "wide" in sys.argv[3:]
import ctypes as C, math
BASE_PATTERN = type(O.pattern())()
C.memmove(C.byref(BASE_PATTERN), C.byref(O.pattern()), C.sizeof(BASE_PATTERN))


def install_wide(fe, factor):
    p = fe.get_pattern()
    reach = 0.0
    for i in range(p.n_points):
        p.sigma_half[i] = np.float32(BASE_PATTERN.sigma_half[i] * factor)
        reach = max(reach, math.hypot(p.px[i], p.py[i]) + p.sigma_half[i])
    p.border = int(math.ceil(reach)) + 1
    fe.set_pattern(p)
    q = type(BASE_PATTERN)()
    C.memmove(C.byref(q), C.byref(BASE_PATTERN), C.sizeof(q))
    q.border = p.border
    C.memmove(q.sigma_half, p.sigma_half, C.sizeof(p.sigma_half))
    O._PATTERN = q

bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    w = int(rng.integers(24, 260)) * 4
    h = int(rng.integers(80, 500))
    kind = ["noise", "corners", "blocks"][seed % 3]
    radius = float(rng.choice([10.0, 17.5, 26.0, 38.0, 50.0]))
    thr = int(rng.choice([5, 40, 150, 400]))
    maxk = int(rng.choice([50, 300, 700, 1500]))
    img = F._image(rng, w, h, kind)
    dist = int(rng.choice([1, 2]))
    f = float(rng.uniform(0.45, 1.3)) * w
    d = (tuple(rng.uniform(-0.3, 0.1, 1)) + tuple(rng.uniform(-0.05, 0.1, 1)) + tuple(rng.uniform(-2e-3, 2e-3, 2))) if dist == 1 \
        else tuple(rng.uniform(-0.02, 0.02, 4))
    cam = synth.Camera(w, h, f, f * float(rng.uniform(0.97, 1.03)), w / 2 + float(rng.uniform(-8, 8)), h / 2 + float(rng.uniform(-8, 8)), dist, tuple(float(x) for x in d))
    g = rng.normal(0, 1, 3); g[1] += 2.0; g = (g / np.linalg.norm(g)).astype(np.float32)
    mode = seed % 4 if ALL_MODES else 0  # 0 camera-aware, 1 upright, 2 gradient, 3 gradient + scale-invariant
    if WIDE and mode == 3:
        mode = 0  # (the scale ladder multiplies the widths again: not what this run is about)
    factor = float(rng.uniform(1.05, 2.4)) if WIDE else 1.0
    if mode == 0:
        fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1 << 16)
        fe.set_camera(0, cam)
        if WIDE:
            install_wide(fe, factor)
        rays, jac = O.awareness_maps(cam)
        rk, rd = O.detect_describe(img, radius, 0, thr, maxk, O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), tuple(float(x) for x in g))
        kps, desc, bp, bpv = fe.detect_describe(img, cam=0, gravity=tuple(float(x) for x in g))
    else:
        si = mode == 3
        fe = capi.Frontend(w, h, radius, 0, thr, maxk, rotation_invariant=(mode >= 2), scale_invariant=si, max_candidates=1 << 16)
        if WIDE:
            install_wide(fe, factor)
        rk, rd = O.detect_describe(img, radius, 0, thr, maxk, O.MODE_GRADIENT if mode >= 2 else O.MODE_UPRIGHT, scale_invariant=si)
        kps, desc, _, _ = fe.detect_describe(img)
    try:
        G.assert_keypoints_equal(kps, rk)
        assert np.array_equal(desc, rd)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, (w, h, kind, radius, thr, maxk, dist, mode, factor), len(rk), str(e)[:200])
print("done", count, "configs,", bad, "mismatches")
