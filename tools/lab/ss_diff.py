#!/usr/bin/env python3
"""GPU box: the 6-layer Harris scale space of tests/test_gpu_octaves.py run repeatedly -- which keypoints differ from the oracle
when a run goes wrong (layer, position, fields)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from okvis2_amd import capi, synth
import oracle_lib as O
w, h, octaves, seed, B = 1024, 1024, 3, 7, 3
imgs = np.stack([synth.corners_image(w, h, seed + 10 * i) for i in range(B)])
d_img = torch.from_numpy(imgs).cuda()
ref = [O.detect(imgs[i], 30.0, octaves, 100, 300) for i in range(B)]
bad_runs = 0
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    fe = capi.Frontend(w, h, 30.0, octaves, 100, 300, max_batch=B, max_candidates=0)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), B, None, None, st)
    st.synchronize()
    for i in range(B):
        k = fe.download(i)[0]
        r = ref[i]
        # detect-only reference vs described keypoints: compare the sets per layer by (octave, x, y, response)
        ks = {(int(a["octave"]), float(a["x"]), float(a["y"]), float(a["response"])) for a in k}
        rs = {(int(a["octave"]), float(a["x"]), float(a["y"]), float(a["response"])) for a in r}
        extra, missing = sorted(ks - rs), sorted(rs - ks)
        # (the extractor removes keypoints near the rim: `missing` near the border is expected; report the rest)
        if extra:
            bad_runs += 1
            print("run", run, "image", i, "extra", extra[:4], "missing near them", [m for m in missing if any(abs(m[1]-e[1]) < 40 and abs(m[2]-e[2]) < 40 and m[0] == e[0] for e in extra)][:4], flush=True)
    fe.close()
print("runs with keypoints the oracle does not have:", bad_runs)
