#!/usr/bin/env python3
"""GPU box: one Hilti camera, a batch of replicas of a few rendered frames -- which images differ from the first
occurrence of their frame, and in what?  usage: dbg_replicas.py [cam] [nfr]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from okvis2_amd import capi, synth
import gpu_common as G
cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 288
cfg = synth.hilti_config()
distinct = 3
rays = [capi.build_awareness_maps(c)[0] for c in cfg.cams]
frames, poses_f = [], []
for f in range(distinct):
    a = 0.2 * f
    C_WS = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    frames.append(synth.render_rig(cfg, rays, 140 + f, r_S=np.array([0.1 * f, 0.0, 0.05 * f])))
    poses_f.append(synth.rig_poses(cfg, C_WS, np.array([0.1 * f, 0.0, 0.05 * f])))
for keep in (False, True):
    fe = G.make_frontend(cfg, max_batch=nfr, num_cameras=1)
    fe.set_camera(0, cfg.cams[cam_i])
    fe.set_keep_score_map(keep)
    d_img = torch.from_numpy(np.stack([frames[f % distinct][cam_i] for f in range(nfr)])).cuda()
    grav = np.stack([synth.gravity_in_camera(poses_f[f % distinct][cam_i][0]) for f in range(nfr)]).astype(np.float32)
    cam_ids = np.zeros(nfr, np.int32)
    for rep in range(2):
        fe.detect_describe_batch_device(d_img.data_ptr(), nfr, cam_ids, grav, None)
        torch.cuda.synchronize()
        fe.check_capacity(nfr)
        res = [fe.download(i) for i in range(nfr)]
        bad = []
        for f in range(distinct, nfr):
            b = f % distinct
            k, d, bp, bv = res[f]; k0, d0, bp0, bv0 = res[b]
            if len(k) != len(k0):
                bad.append((f, "count", len(k), len(k0)))
            elif k.tobytes() != k0.tobytes():
                w = [i for i in range(len(k)) if k[i].tobytes() != k0[i].tobytes()]
                bad.append((f, "kps", len(w), w[:4], [(k[i]["x"], k[i]["y"], k[i]["response"], k0[i]["x"], k0[i]["y"], k0[i]["response"]) for i in w[:2]]))
            elif d.tobytes() != d0.tobytes():
                w = [i for i in range(len(k)) if d[i].tobytes() != d0[i].tobytes()]
                bad.append((f, "desc", len(w), w[:4], [(k[i]["x"], k[i]["y"]) for i in w[:3]]))
            elif bp.tobytes() != bp0.tobytes() or bv.tobytes() != bv0.tobytes():
                bad.append((f, "bp"))
        print("keep_map", keep, "rep", rep, "kp counts", [len(res[i][0]) for i in range(3)], "bad", len(bad), bad[:6])
    fe.close()
