#!/usr/bin/env python3
"""Stress (GPU): several contexts on several HIP streams at once against one context alone.  Every image of every lane must
come out bit-identical (keypoints, descriptors, back-projections, match rows) to the same frame processed by a single
context with the GPU to itself.
usage: python tools/stress_lanes.py [lanes] [frames_per_lane] [iterations] [workload]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from okvis2_amd import capi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Bl = int(sys.argv[2]) if len(sys.argv) > 2 else 96
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = getattr(synth, (sys.argv[4] if len(sys.argv) > 4 else "euroc") + "_config")()
D = 24  # distinct frames
imgs = []
for i in range(D):
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 4000 + i)
    imgs += [L, R]
imgs = np.stack(imgs)
f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
T0, T1 = synth.stereo_poses(cfg.baseline)


def make(nfr):
    fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold, cfg.max_kpts,
                       match_threshold=cfg.match_threshold, max_batch=2 * nfr, num_cameras=2)
    for ci, cam in enumerate(cfg.cams):
        fe.set_camera(ci, cam)
    pairs = []
    for i in range(nfr):
        sp = capi.StereoPair()
        sp.image0, sp.image1 = 2 * i, 2 * i + 1
        sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
        sp.f0, sp.f1 = f0, f1
        pairs.append(sp)
    return fe, (capi.StereoPair * nfr)(*pairs)


def results(fe, d_m, nfr):
    m = d_m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(nfr, cfg.max_kpts)
    out = []
    for f in range(nfr):
        a, b = fe.download(2 * f), fe.download(2 * f + 1)
        out.append((a, b, m[f, :len(a[0])].copy()))
    return out


def same(x, y):
    return all(np.array_equal(p.view(np.uint8), q.view(np.uint8)) for s in (0, 1) for p, q in zip(x[s], y[s])) and \
        np.array_equal(x[2].view(np.uint8), y[2].view(np.uint8))


grav1 = np.tile(np.array([0.05, 0.99, -0.02], np.float32) / np.linalg.norm([0.05, 0.99, -0.02]), (2 * max(D, Bl), 1)).astype(np.float32)
ids = np.array([0, 1] * max(D, Bl), np.int32)
ref_fe, ref_pairs = make(D)
d_ref = torch.from_numpy(imgs).cuda()
d_mref = torch.zeros((D, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
ref_fe.detect_describe_batch_device(d_ref.data_ptr(), 2 * D, ids[:2 * D], grav1[:2 * D], None)
ref_fe.match_stereo_batch_device(ref_pairs, d_mref.data_ptr(), None)
torch.cuda.synchronize()
ref = results(ref_fe, d_mref, D)
print("reference: mean keypoints", np.mean([len(r[0][0]) for r in ref]), "matches", np.mean([(r[2]["k1"] >= 0).sum() for r in ref]))

lanes = []
for l in range(S):
    order = [(l * 5 + i) % D for i in range(Bl)]  # every lane works on different content at the same time
    li = np.concatenate([imgs[2 * o:2 * o + 2] for o in order])
    fe, pairs = make(Bl)
    lanes.append(dict(fe=fe, pairs=pairs, order=order, img=torch.from_numpy(li).cuda(), st=torch.cuda.Stream(),
                      m=torch.zeros((Bl, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")))
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for _ in range(3):  # several steps in flight before the check
        for ln in lanes:
            ln["fe"].detect_describe_batch_device(ln["img"].data_ptr(), 2 * Bl, ids[:2 * Bl], grav1[:2 * Bl], ln["st"])
            ln["fe"].match_stereo_batch_device(ln["pairs"], ln["m"].data_ptr(), ln["st"])
    torch.cuda.synchronize()
    for l, ln in enumerate(lanes):
        ln["fe"].check_capacity(2 * Bl)
        got = results(ln["fe"], ln["m"], Bl)
        for f, o in enumerate(ln["order"]):
            if not same(got[f], ref[o]):
                bad += 1
                if bad <= 10:
                    print(f"MISMATCH iteration {it} lane {l} frame {f} (content {o}): keypoints {len(got[f][0][0])}/{len(got[f][1][0])} "
                          f"vs {len(ref[o][0][0])}/{len(ref[o][1][0])}")
print(f"{S} lanes x {Bl} frames x {iters} iterations: {bad} mismatching frames")
sys.exit(1 if bad else 0)
