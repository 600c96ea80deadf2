"""Score ring-pattern hypotheses against the real vocabulary: a hypothesis is a generator of the ordered pair list;
its score is the number of samples that show a cyclic (impossible) outcome on the triangles the list implies."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from decode import forb_table

def rings(rr, nn, offs=None, center_sigma=0.5):
    pts = []; sig = []; ring = []
    for q, (r, n) in enumerate(zip(rr, nn)):
        for k in range(n):
            a = 2 * np.pi * k / n + (offs[q] if offs is not None else 0.0)
            pts.append((r * np.cos(a), r * np.sin(a)))
            sig.append(center_sigma if r == 0 else r * np.sin(np.pi / n)); ring.append(q)
    return np.array(pts), np.array(sig), np.array(ring)

def pairs_sum_rule(pts, sig, kappa):
    N = len(pts)
    return [(i, j) for i in range(1, N) for j in range(i)
            if np.linalg.norm(pts[i] - pts[j]) < kappa * (sig[i] + sig[j])]

def triangle_score(pairs, F):
    E = {p: a for a, p in enumerate(pairs)}
    adj = {}
    for i, j in pairs: adj.setdefault(i, set()).add(j)
    tot = 0.0; n = 0; worst = []
    for (z, y), c in E.items():          # c=(z,y), need x<y with (y,x) and (z,x)
        for x in adj.get(y, ()):
            b = E.get((z, x))
            if b is not None:
                a = E[(y, x)]
                f = F[a, b, c]; tot += f; n += 1
                if f > 8: worst.append((int(f), a, b, c))
    return tot, n, worst

def load_voc_bits():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    voc = np.fromfile(os.path.join(root, 'tests/golden/small_voc_desc.bin'), dtype=np.uint8).reshape(-1, 48)
    return np.unpackbits(voc, axis=1, bitorder='little')

