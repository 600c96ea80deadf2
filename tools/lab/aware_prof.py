#!/usr/bin/env python3
"""Lab build only: where a wave of describe_aware_kernel spends its time (s_memtime, 100 MHz ticks), EuRoC batch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKVFE_LIB"] = os.path.join(ROOT, "okvis2_amd", "libokvfe_lab.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from okvis2_amd import capi, synth
import bench
cfg = synth.euroc_config()
B = 1536
imgs, base = bench.make_inputs(cfg, B // 2, 16, 1000, "corners")
def prof(reset=True):
    out = (C.c_ulonglong * 16)()
    assert capi.lib().okvfe_lab_aware_prof(out, int(reset)) == 0
    v = list(out); n = max(v[0], 1); k = max(v[6], 1)
    u = 0.01  # us per tick
    return {"waves": v[0], "kp_per_wave": v[6] / n, "prologue_us": v[1] / n * u, "extras_us_per_wave": v[2] / n * u,
            "dma_wait_us_per_kp": v[3] / k * u, "box_us_per_kp": v[4] / k * u, "kp_total_us": v[5] / k * u,
            "wave_us": v[7] / n * u}
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, max_batch=B, num_cameras=2,
                   max_candidates=16384)
for ci, cam in enumerate(cfg.cams): fe.set_camera(ci, cam)
d = torch.from_numpy(imgs[:B]).cuda()
cam_ids = np.array([0, 1] * (B // 2), np.int32)
g = bench.gravity_variant(0, B)
for _ in range(3): fe.detect_describe_batch_device(d.data_ptr(), B, cam_ids, g, None)
torch.cuda.synchronize(); prof()
for _ in range(10): fe.detect_describe_batch_device(d.data_ptr(), B, cam_ids, g, None)
torch.cuda.synchronize()
print(os.environ.get("OKVFE_DESC_TILES", "16"), prof())
