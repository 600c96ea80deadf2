#!/usr/bin/env python3
"""GPU box: does the map-writing PACK form of the fused score kernel (small launch: 25-row tiles) give a different score map
or different keypoints when another context's kernels share the GPU?  512 x 512 x 3 images with okvfe_set_keep_score_map(1)
on stream A, a 682 x 682 context (unaligned width: generic kernels) on stream B, repeated."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from okvis2_amd import capi, synth
import oracle_lib as O
W = int(os.environ.get("PR_W", "512")); B = 3
imgs = np.stack([synth.corners_image(W, W, 7 + 10 * i) for i in range(B)])
fa = capi.Frontend(W, W, 30.0, 0, 100, 300, max_batch=B, max_candidates=0)
fa.set_keep_score_map(True)
fb = capi.Frontend(682, 682, 30.0, 0, 100, 300, max_batch=B, max_candidates=0)
imgs_b = np.stack([synth.corners_image(682, 682, 70 + i) for i in range(B)])
da, db = torch.from_numpy(imgs).cuda(), torch.from_numpy(imgs_b).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref = [O.detect_describe(imgs[i], 30.0, 0, 100, 300, O.MODE_GRADIENT)[0] for i in range(B)]
sc_ref = [O.harris_score(imgs[i]) for i in range(B)]
torch.cuda.synchronize()
bad_kp = bad_map = 0
noisy = os.environ.get("PR_ALONE") is None
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if noisy:
        for _ in range(2): fb.detect_batch_device(db.data_ptr(), B, sb)
    fa.detect_describe_batch_device(da.data_ptr(), B, None, None, sa)
    if noisy:
        for _ in range(2): fb.detect_batch_device(db.data_ptr(), B, sb)
    torch.cuda.synchronize()
    out = fa.device_outputs()
    pitch = out.score_pitch
    host = np.empty(B * W * pitch, dtype=np.int32)
    assert capi.lib().okvfe_copy_to_host(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(out.scores), ctypes.c_size_t(host.nbytes), None) == 0
    cols = np.array([capi.lib().okvfe_score_column(fa._h, int(x)) for x in range(W)])
    for i in range(B):
        m = host.reshape(B, W, pitch)[i][:, cols]
        d = np.argwhere(m[3:-3, 3:-3] != sc_ref[i][3:-3, 3:-3])
        if len(d):
            bad_map += 1
            print("run", run, "image", i, "map differs at", len(d), "pixels, first (y, x):", (d[:6] + 3).tolist(), flush=True)
    for i in range(B):
        k = fa.download(i)[0]
        if len(k) != len(ref[i]) or k.tobytes() != ref[i].tobytes():
            bad_kp += 1
            w = [j for j in range(min(len(k), len(ref[i]))) if k[j].tobytes() != ref[i][j].tobytes()]
            print("run", run, "image", i, "keypoints differ:", len(k), len(ref[i]), [(float(k[j]["x"]), float(k[j]["y"]), float(ref[i][j]["x"]), float(ref[i][j]["y"]), float(k[j]["response"])) for j in w[:3]], flush=True)
print("maps differing:", bad_map, "keypoint sets differing:", bad_kp)
