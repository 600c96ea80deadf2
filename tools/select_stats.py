"""GPU box, profiling build only (k_select.hip compiled with -DOKVFE_SELECT_STATS into
okvis2_amd/libokvfe_selstats.so): per-image averages of the greedy selection's rounds, windows and
cycle split on the bench content.   OKVFE_LIB=$PWD/okvis2_amd/libokvfe_selstats.so python tools/select_stats.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from okvis2_amd import capi, synth
cfg = synth.euroc_config()
B = 256
imgs, _ = bench.make_inputs(cfg, B, 16, 1000)
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, max_batch=2 * B, num_cameras=2)
d = torch.from_numpy(imgs).cuda()
L = capi.lib()
L.okvfe_debug_select_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
for it in range(2):
    L.okvfe_debug_select_stats(None, 1)
    fe.detect_batch_device(d.data_ptr(), 2 * B, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    L.okvfe_debug_select_stats(out, 0)
n = out[0]
# the grid kernels (OKVFE_SELECT_GRID=1) report rounds / decide / stamp / init / sub-pixel cycles; the lazy kernel
# (default) reports, from wave 0's view: slot 1 = (passing lanes << 20 | linked lanes), walk, accept, barrier
# waits, and (neighbour test + sub-pixel) cycles -- see the OKVFE_SELECT_STATS blocks in k_select.hip
names = ["blocks", "rounds|pass<<20+linked", "windows", "kept", "cyc_decide|walk", "cyc_stamp|accept",
         "cyc_init|sync", "cyc_subpix(+neighbour test)"]
print({k: round(out[i] / n, 1) for i, k in enumerate(names)})
