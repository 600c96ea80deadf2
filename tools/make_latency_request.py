# writes a latency_cli request file from bench helpers
import sys, os
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from okvis2_amd import synth
cfg = synth.euroc_config()
imgs, base = bench.make_inputs(cfg, 8, 8, 1000, "corners")
import struct
T = synth.stereo_poses(cfg.baseline)
Cm = np.array([[1.0,0,0],[0,0,1.0],[0,-1.0,0]])
with open(sys.argv[1], "wb") as f:
    f.write(struct.pack("<5i", cfg.w, cfg.h, 8, int(sys.argv[2]), 20))
    f.write(struct.pack("<f3i", cfg.uniformity_radius, cfg.abs_threshold, cfg.match_threshold, cfg.max_kpts))
    for c in range(2):
        cam = cfg.cams[c]
        f.write(struct.pack("<4d", cam.fu, cam.fv, cam.cu, cam.cv)); f.write(struct.pack("<i", cam.dist_type))
        f.write(struct.pack("<4d", *cam.d)); f.write(struct.pack("<9d", *Cm.reshape(-1))); f.write(struct.pack("<3d", *T[c][1]))
    for i in range(8):
        for c in range(2): f.write(np.ascontiguousarray(base[2*i+c]).tobytes())
