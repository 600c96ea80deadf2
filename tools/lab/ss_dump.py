#!/usr/bin/env python3
"""GPU box, lab library: buffers of every scale-space layer after a call, against a reference dump of the one-stream form.
usage: OKVFE_SS_OWN=0 ss_dump.py ref   (writes /tmp/ss_ref.npz);   ss_dump.py cmp [runs]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OKVFE_LIB", os.path.join(ROOT, "okvis2_amd", "libokvfe_lab.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from okvis2_amd import capi, synth
w, h, octaves, seed, B = 1024, 1024, 3, 7, 3
imgs = np.stack([synth.corners_image(w, h, seed + 10 * i) for i in range(B)])
d_img = torch.from_numpy(imgs).cuda()
lib = capi.lib(); lib.okvfe_lab_dump_layer.restype = C.c_longlong
def run():
    fe = capi.Frontend(w, h, 30.0, octaves, 100, 300, max_batch=B, max_candidates=0)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), B, None, None, st)
    st.synchronize()
    out = {}
    for l in range(2 * octaves):
        for what, name in ((0, "map"), (1, "img"), (3, "cnt")):
            n = lib.okvfe_lab_dump_layer(fe._h, l, what, None, C.c_size_t(0))
            if n <= 0: continue
            buf = np.empty(n, np.uint8)
            assert lib.okvfe_lab_dump_layer(fe._h, l, what, C.c_void_p(buf.ctypes.data), C.c_size_t(n)) == n
            out[f"{name}{l}"] = buf
        n = lib.okvfe_lab_dump_layer(fe._h, l, 2, None, C.c_size_t(0))
        buf = np.empty(n, np.uint8)
        lib.okvfe_lab_dump_layer(fe._h, l, 2, C.c_void_p(buf.ctypes.data), C.c_size_t(n))
        cnt = out[f"cnt{l}"].view(np.int32)
        cap = n // 12 // B
        rec = buf.view(np.int32).reshape(B, cap, 3)
        out[f"cand{l}"] = np.concatenate([np.sort(rec[i, :cnt[i]].view([("x", "<i4"), ("y", "<i4"), ("s", "<i4")]).reshape(-1), order=("y", "x")).view(np.int32) for i in range(B)])
    kp = np.concatenate([fe.download(i)[0].view(np.uint8) for i in range(B)])
    out["kp"] = kp
    fe.close()
    return out
if sys.argv[1] == "ref":
    np.savez("/tmp/ss_ref.npz", **run()); print("reference written")
else:
    ref = np.load("/tmp/ss_ref.npz")
    for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
        o = run()
        diff = [k for k in ref.files if o[k].shape != ref[k].shape or not np.array_equal(o[k], ref[k])]
        if diff:
            msg = []
            for k in diff:
                if o[k].shape == ref[k].shape:
                    idx = np.flatnonzero(o[k] != ref[k])
                    msg.append(f"{k}: {len(idx)} bytes/ints differ, first at {idx[:4].tolist()}")
                else:
                    msg.append(f"{k}: shape {o[k].shape} vs {ref[k].shape}")
            print("run", r, "; ".join(msg), flush=True)
            if "map2" in diff and os.environ.get("SS_DETAIL"):
                a, b = o["map2"].view(np.int32), ref["map2"].view(np.int32)
                pitch = len(a) // (B * 512)
                for i in np.flatnonzero(a != b)[:40]:
                    im, rem = divmod(int(i), pitch * 512); row, col = divmod(rem, pitch)
                    print("   image", im, "row", row, "int", col, "(quad", col // 4, "strip slot", col // 256, ") ref", int(b[i]), "got", int(a[i]))
                break
    print("done")
