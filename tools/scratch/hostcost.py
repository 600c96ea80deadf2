import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from okvis2_amd import capi, synth
cfg = synth.euroc_config()
B = 64
imgs = np.stack([synth.stereo_pair(cfg.w, cfg.h, 1000 + (i % 8))[j] for i in range(B) for j in (0, 1)])
d_img = torch.from_numpy(imgs).cuda()
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, cfg.octaves, cfg.abs_threshold, cfg.max_kpts,
                   match_threshold=cfg.match_threshold, max_batch=2 * B, num_cameras=2, max_candidates=16384)
for ci, cam in enumerate(cfg.cams): fe.set_camera(ci, cam)
cam_ids = np.array([0, 1] * B, dtype=np.int32)
grav = np.tile(np.array([0.0, 1.0, 0.0], dtype=np.float32), (2 * B, 1))
T0, T1 = synth.stereo_poses(cfg.baseline)
pairs = []
for i in range(B):
    sp = capi.StereoPair(); sp.image0, sp.image1 = 2 * i, 2 * i + 1
    sp.T_WC0, sp.T_WC1 = capi.make_pose(*T0), capi.make_pose(*T1); sp.f0 = sp.f1 = 458.0
    pairs.append(sp)
pa = (capi.StereoPair * B)(*pairs)
dm = torch.zeros((B, cfg.max_kpts, capi.STEREO_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def step():
    fe.detect_describe_batch_device(d_img.data_ptr(), 2 * B, cam_ids, grav, s)
    fe.match_stereo_batch_device(pa, dm.data_ptr(), s)
for prof in (False, True):
    fe.profile_enable(prof)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"profile={prof}: enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step (B={B})")
