"""GPU box: how many candidates the fused kernel flags for the fix-up pass (per workload)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from okvis2_amd import capi, synth
import gpu_common as G
for name, mk in (("euroc", synth.euroc_config), ("mono640", synth.mono640_config), ("tumvi", synth.tumvi1024_config), ("hilti", synth.hilti_config)):
    cfg = mk()
    B = 8
    fe = G.make_frontend(cfg, max_batch=B)
    imgs = np.stack([synth.stereo_pair(cfg.w, cfg.h, 5000 + i, cell=12)[0] for i in range(B)])
    d = torch.from_numpy(imgs).cuda()
    fe.detect_batch_device(d.data_ptr(), B)
    torch.cuda.synchronize()
    out = fe.device_outputs()
    cnt = np.zeros(2 * B, dtype=np.int32)
    capi.lib().okvfe_copy_to_host(C.c_void_p(cnt.ctypes.data), C.c_void_p(out.candidate_counts), C.c_size_t(cnt.nbytes), None)
    print(name, "candidates", cnt[:B].tolist(), "flagged", cnt[B:].tolist())
