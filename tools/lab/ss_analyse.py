#!/usr/bin/env python3
"""GPU box, lab library: take one corrupted layer-2 score map of the concurrent scale space and test what single wrong input byte
reproduces the wrong scores (the left neighbour's last pixel of image row y - 2, as a stale DPP read would give)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OKVFE_LIB", os.path.join(ROOT, "okvis2_amd", "libokvfe_lab.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from okvis2_amd import capi, synth
import oracle_lib as O
w, h, octaves, seed, B = 1024, 1024, 3, 7, 3
imgs = np.stack([synth.corners_image(w, h, seed + 10 * i) for i in range(B)])
d_img = torch.from_numpy(imgs).cuda()
lib = capi.lib(); lib.okvfe_lab_dump_layer.restype = C.c_longlong
def dump(fe, l, what):
    n = lib.okvfe_lab_dump_layer(fe._h, l, what, None, C.c_size_t(0))
    buf = np.empty(n, np.uint8)
    assert lib.okvfe_lab_dump_layer(fe._h, l, what, C.c_void_p(buf.ctypes.data), C.c_size_t(n)) == n
    return buf
for run in range(200):
    fe = capi.Frontend(w, h, 30.0, octaves, 100, 300, max_batch=B, max_candidates=0)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    fe.detect_describe_batch_device(d_img.data_ptr(), B, None, None, st)
    st.synchronize()
    m = dump(fe, 2, 0).view(np.int32).reshape(B, 512, -1)
    im2 = dump(fe, 2, 1).reshape(B, 512, 512)
    pitch = m.shape[2]
    # slotted layout: pixel x -> dword d = x / 4, strip = owner of d (62 per strip, strip 0 owns 0..62), slot quad = d + 2 strip
    def col(x):
        d = x // 4
        s = 0 if d <= 62 else (1 if d <= 124 else 2)
        return (d + 2 * s) * 4 + (x & 3)
    cols = np.array([col(x) for x in range(512)])
    found = False
    for i in range(B):
        ref = O.harris_score(im2[i])
        got = m[i][:, cols]
        bad = np.argwhere(got[3:-3, 3:-3] != ref[3:-3, 3:-3]) + 3
        if len(bad) == 0: continue
        found = True
        print("run", run, "image", i, "bad pixels", len(bad), "rows", sorted(set(bad[:, 0].tolist())), "x mod 64:", sorted(set((bad[:, 1] % 64).tolist())))
        img = im2[i].astype(np.int64)
        def asr(v, k): return np.floor_divide(v, 1 << k)
        gx = np.zeros_like(img); gy = np.zeros_like(img)
        gx[1:-1, 1:-1] = 3 * (img[:-2, 2:] - img[:-2, :-2]) + 10 * (img[1:-1, 2:] - img[1:-1, :-2]) + 3 * (img[2:, 2:] - img[2:, :-2])
        gy[1:-1, 1:-1] = 3 * (img[2:, :-2] - img[:-2, :-2]) + 10 * (img[2:, 1:-1] - img[:-2, 1:-1]) + 3 * (img[2:, 2:] - img[:-2, 2:])
        G = [asr(gx * gx, 14), asr(gy * gy, 14), asr(gx * gy, 14)]
        def score_with(y, x, edit):
            """edit(c, Hrow) may change the horizontally smoothed entries of covariance row y - 1 at column x"""
            S = []
            for c in range(3):
                g = G[c]
                H = {r: g[r, x - 1] + 2 * g[r, x] + g[r, x + 1] for r in (y - 1, y, y + 1)}
                H[y - 1] = edit(c, g, H[y - 1])
                S.append(H[y - 1] + 2 * H[y] + H[y + 1])
            tq = asr(asr(S[0], 1) + asr(S[1], 1), 1)
            return int(S[0] * S[1] - S[2] * S[2] - tq * tq)
        for (y, x) in bad[:8]:
            y, x = int(y), int(x)
            hyp = {
                "none": lambda c, g, H: H,
                "gl=0 (left G of row y-1 missing)": lambda c, g, H: H - g[y - 1, x - 1],
                "gl from row y-2": lambda c, g, H: H - g[y - 1, x - 1] + g[y - 2, x - 1],
                "gl from row y": lambda c, g, H: H - g[y - 1, x - 1] + g[y, x - 1],
                "H[0] = 0": lambda c, g, H: 0,
                "H[0] of row y-2": lambda c, g, H: g[y - 2, x - 1] + 2 * g[y - 2, x] + g[y - 2, x + 1],
                "gl from lane-1's G[2]": lambda c, g, H: H - g[y - 1, x - 1] + g[y - 1, x - 2],
                "gl = own G[3]": lambda c, g, H: H - g[y - 1, x - 1] + g[y - 1, x + 3],
            }
            res = {k: score_with(y, x, f) for k, f in hyp.items()}
            nb = {(dy, dx): int(ref[y + dy, x + dx]) for dy in range(-3, 4) for dx in range(-8, 9)}
            print("   pixel", (y, x), "ref", int(ref[y, x]), "got", int(got[y, x]), "equals the reference score at offset (dy, dx):", [k for k, v in nb.items() if v == int(got[y, x])])
        break
    fe.close()
    if found: break
