import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib as O
from okvis2_amd import capi, synth
cfg = synth.tumvi1024_config(); cam = cfg.cams[0]
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
fe.set_camera(0, cam)
img = synth.corners_image(cfg.w, cfg.h, 77)
kps, desc, bp, bpv = fe.detect_describe(img, cam=0, gravity=(0.0, 1.0, 0.0))
rbp, rv = O.backproject_keypoints(cam, kps)
d = np.abs(bp - rbp)
i = np.unravel_index(d.argmax(), d.shape)
print('max diff', d.max(), 'at', i, kps[i[0]], bp[i[0]], rbp[i[0]])
print('count > 1e-12', (d.max(1) > 1e-12).sum(), 'of', len(kps), ' > 1e-9', (d.max(1) > 1e-9).sum())
r = np.hypot(kps['x']-cam.cu, kps['y']-cam.cv)
print('radius of worst', r[i[0]], 'max radius', r.max())
