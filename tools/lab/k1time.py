"""GPU box: time of the fused score+NMS kernel alone (HIP events of the harris stage) for A/B
variants whose score-map LAYOUT is experimental (the rest of the pipeline would read garbage):
only okvfe_detect_batch_device is called.   OKVFE_LIB=... python tools/lab/k1time.py [n_images]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from okvis2_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
cfg = synth.euroc_config()
imgs, _ = bench.make_inputs(cfg, n // 2, 16, 1000)
fe = capi.Frontend(cfg.w, cfg.h, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts, max_batch=n, num_cameras=2)
d = torch.from_numpy(imgs).cuda()
fe.profile_enable(True, stages=("harris",))
for it in range(8):
    fe.detect_batch_device(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ms, cnt = fe.profile_read()["harris"]
print(os.path.basename(os.environ.get("OKVFE_LIB", "libokvfe.so")), "k1 ms per %d images: %.4f" % (n, ms / cnt))
