#!/usr/bin/env python3
"""One-off fuzz (GPU): random stereo rigs through the DEVICE-RESIDENT batch path -- okvfe_detect_describe_batch_device on a
batch of random size + okvfe_match_stereo_batch_device -- against the oracle: keypoints, descriptors, back-projections
(u64 patterns) and match rows incl. hp_W.
usage: python tools/fuzz_stereo.py [first_seed] [count]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from okvis2_amd import capi, synth
import oracle_lib as O, gpu_common as G
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    w = int(rng.integers(40, 200)) * 4
    h = int(rng.integers(120, 420))
    radius = float(rng.choice([17.5, 26.0, 38.0, 50.0]))
    thr = int(rng.choice([5, 40, 150]))
    maxk = int(rng.choice([100, 400, 700]))
    mthr = int(rng.choice([40, 60, 80]))
    dist = int(rng.choice([1, 2]))
    cams = []
    for c in range(2):
        f = float(rng.uniform(0.5, 1.2)) * w
        d = (tuple(rng.uniform(-0.3, 0.1, 1)) + tuple(rng.uniform(-0.05, 0.1, 1)) + tuple(rng.uniform(-2e-3, 2e-3, 2))) if dist == 1 \
            else tuple(rng.uniform(-0.02, 0.02, 4))
        cams.append(synth.Camera(w, h, f, f * float(rng.uniform(0.98, 1.02)), w / 2 + float(rng.uniform(-6, 6)),
                                 h / 2 + float(rng.uniform(-6, 6)), dist, tuple(float(x) for x in d)))
    nfr = int(rng.integers(1, 4))
    fe = capi.Frontend(w, h, radius, 0, thr, maxk, match_threshold=mthr, max_batch=2 * nfr, num_cameras=2, max_candidates=1 << 16)
    for ci, cam in enumerate(cams):
        fe.set_camera(ci, cam)
    baseline = float(rng.uniform(0.05, 0.3))
    T0, T1 = synth.stereo_poses(baseline)
    f0, f1 = 0.5 * (cams[0].fu + cams[0].fv), 0.5 * (cams[1].fu + cams[1].fv)
    grav = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (2 * nfr, 1))
    frames, imgs = [], []
    for i in range(nfr):
        L, R, _ = synth.stereo_pair(w, h, int(rng.integers(1, 1 << 20)), cell=int(rng.choice([8, 12, 16])))
        sides = []
        for ci, img in enumerate((L, R)):
            rays, jac = O.awareness_maps(cams[ci])
            k, dd = O.detect_describe(img, radius, 0, thr, maxk, O.MODE_CAMERA_AWARE, rays, jac, np.float32(cams[ci].fu), (0.0, 1.0, 0.0))
            bp, bv = O.backproject_keypoints(cams[ci], k)
            sides.append((k, dd, bp, bv))
            imgs.append(img)
        frames.append(sides)
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    fe.detect_describe_batch_device(d_img.data_ptr(), 2 * nfr, np.array([0, 1] * nfr, np.int32), grav, None)
    pairs = []
    for i in range(nfr):
        sp = capi.StereoPair()
        sp.image0, sp.image1 = 2 * i, 2 * i + 1
        sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
        sp.f0, sp.f1 = f0, f1
        pairs.append(sp)
    d_m = torch.zeros((nfr, maxk, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    fe.match_stereo_batch_device(pairs, d_m.data_ptr(), None)
    torch.cuda.synchronize()
    m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, maxk)
    try:
        for i, ((k0, d0, b0, v0), (k1, d1, b1, v1)) in enumerate(frames):
            g0, g1 = fe.download(2 * i), fe.download(2 * i + 1)
            G.assert_keypoints_equal(g0[0], k0); G.assert_keypoints_equal(g1[0], k1)
            assert np.array_equal(g0[1], d0) and np.array_equal(g1[1], d1)
            assert np.array_equal(g0[2].view(np.uint64), b0.view(np.uint64)) and np.array_equal(g0[3], v0)
            assert np.array_equal(g1[2].view(np.uint64), b1.view(np.uint64)) and np.array_equal(g1[3], v1)
            ref = O.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f0, f1, mthr)
            got = m[i, :len(k0)]
            for fld in ("k1", "dist", "initialisable"):
                assert np.array_equal(got[fld], ref[fld]), fld
            assert np.array_equal(got["hp_W"].view(np.uint64), ref["hp_W"].view(np.uint64))
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, (w, h, radius, thr, maxk, mthr, dist, nfr), str(e)[:200])
print("done", count, "rigs,", bad, "mismatches")
