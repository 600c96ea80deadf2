#!/usr/bin/env python3
"""Stress (GPU): DIFFERENT workloads on streams of their own at once -- EuRoC stereo (map-free score kernel), EuRoC with the score
map kept (the map-writing form), TUM-VI 1024 x 1024 stereo (packed last strips, equidistant cameras), a 682-px mono camera
(generic score / NMS kernels), a 3-octave Harris scale space, the BRISK scale space -- every context's results against the
same context's results with the GPU to itself.  Round 6 found a store-data hazard of the map-writing score kernel that
only showed beside other kernels' memory traffic (LAB_NOTES); this is the net for that class.
usage: python tools/stress_mixed.py [iterations]"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from okvis2_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12


class Job:
    def __init__(self, name, cfg, n_frames, stereo=True, keep_map=False, octaves=None, score_type=capi.SCORE_HARRIS, aware=True, seed=0):
        self.name, self.cfg, self.stereo, self.n = name, cfg, stereo, n_frames
        C = 2 if stereo else 1
        self.C = C
        oc = cfg.octaves if octaves is None else octaves
        self.fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, oc, cfg.abs_threshold, cfg.max_kpts,
                                match_threshold=cfg.match_threshold, max_batch=C * n_frames, num_cameras=C,
                                score_type=score_type, max_candidates=1 << 15 if score_type != capi.SCORE_HARRIS else 0)
        if keep_map:
            self.fe.set_keep_score_map(True)
        self.aware = aware and oc == 0
        if self.aware:
            for ci in range(C):
                self.fe.set_camera(ci, cfg.cams[ci])
        imgs = []
        for i in range(n_frames):
            if stereo:
                L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 9000 + seed + i)
                imgs += [L, R]
            else:
                imgs.append(synth.corners_image(cfg.w, cfg.h, 9000 + seed + i))
        self.img = torch.from_numpy(np.stack(imgs)).cuda()
        self.ids = np.array(list(range(C)) * n_frames, np.int32) if self.aware else None
        g = np.array([0.04, 0.99, -0.03], np.float32)
        self.grav = np.tile(g / np.linalg.norm(g), (C * n_frames, 1)).astype(np.float32) if self.aware else None
        self.st = torch.cuda.Stream()
        self.pairs = None
        if stereo and oc == 0:
            T0, T1 = synth.stereo_poses(cfg.baseline)
            f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
            f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
            arr = []
            for i in range(n_frames):
                sp = capi.StereoPair()
                sp.image0, sp.image1 = 2 * i, 2 * i + 1
                sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1)
                sp.f0, sp.f1 = f0, f1
                arr.append(sp)
            self.pairs = (capi.StereoPair * n_frames)(*arr)
            self.m = torch.zeros((n_frames, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")

    def launch(self):
        self.fe.detect_describe_batch_device(self.img.data_ptr(), self.C * self.n, self.ids, self.grav, self.st)
        if self.pairs is not None:
            self.fe.match_stereo_batch_device(self.pairs, self.m.data_ptr(), self.st)

    def digest(self):
        h = hashlib.sha256()
        counts = []
        for i in range(self.C * self.n):
            k, d, bp, bv = self.fe.download(i)
            counts.append(len(k))
            for a in (k, d, bp, bv):
                h.update(np.ascontiguousarray(a).tobytes())
        if self.pairs is not None:
            rows = self.m.cpu().numpy().view(capi.STEREO_MATCH_DTYPE).reshape(self.n, -1)
            for f in range(self.n):
                h.update(np.ascontiguousarray(rows[f, :counts[2 * f]]).tobytes())
        return h.hexdigest(), int(np.mean(counts))


euroc, tumvi, mono = synth.euroc_config(), synth.tumvi1024_config(), synth.mono640_config()
# a mono camera whose width is not a multiple of 4: generic score kernel, generic NMS
c682 = synth.Config("mono682", 682, 682, [mono.cams[0]], 0.0, 30.0, 100, 60, 0, 300)
jobs = [Job("euroc map-free", euroc, 96, seed=0), Job("euroc map kept", euroc, 48, keep_map=True, seed=200),
        Job("tumvi 1024", tumvi, 24, seed=400), Job("mono 682 (generic kernels)", c682, 48, stereo=False, aware=False, seed=600),
        Job("harris scale space 1024^2 x 3 octaves", synth.Config("ss", 1024, 1024, [mono.cams[0]], 0.0, 30.0, 100, 60, 0, 300), 3,
            stereo=False, octaves=3, aware=False, seed=800),
        Job("brisk scale space", synth.Config("bss", 752, 480, [mono.cams[0]], 0.0, 0.0, 34, 60, 0, 800), 16, stereo=False, octaves=2,
            score_type=capi.SCORE_BRISK_SCALESPACE, aware=False, seed=1000)]
ref = []
for j in jobs:  # each with the GPU to itself
    j.launch()
    torch.cuda.synchronize()
    ref.append(j.digest())
    print("alone:", j.name, "mean keypoints", ref[-1][1], flush=True)
bad = 0
for it in range(iters):
    for rep in range(2):  # two steps of every job in flight
        for j in (jobs if (it + rep) % 2 == 0 else jobs[::-1]):
            j.launch()
    torch.cuda.synchronize()
    for j, r in zip(jobs, ref):
        if j.digest()[0] != r[0]:
            bad += 1
            print("iteration", it, "MISMATCH:", j.name, flush=True)
print("done", iters, "iterations x", len(jobs), "workloads,", bad, "mismatches")
sys.exit(1 if bad else 0)
