"""Run-by-run decode with the vertex identities fixed (66 points) and candidate partners limited to geometric
neighbours (thresholds of assemble.py widened); inside a run every subset of the candidates is scored by the
triangle evidence.  Prints the runs and where they differ from the pure threshold rule."""
import numpy as np, sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from assemble import vertex_table, EARLY_RUNS, THRESH

def decode(F, R=8.0, widen=1.45, verbose=True):
    ring, ang = vertex_table()
    edge = {}; s = 0; runs = {}
    for z in range(1, len(ring)):
        if z in EARLY_RUNS:
            js = EARLY_RUNS[z]
        else:
            cand = []; rule = []
            for j in range(0, z):
                t = THRESH.get((ring[j], ring[z]))
                if t is None: continue
                d = abs((ang[z] - ang[j] + 180) % 360 - 180)
                if d <= t * widen + 1e-6: cand.append(j)
                if d <= t + 1e-6: rule.append(j)
            best = None
            for m in range(max(1, len(rule) - 3), min(len(cand), len(rule) + 2) + 1):
                if s + m > 384: break
                for js in itertools.combinations(cand, m):
                    sc = 0.0
                    for q in range(m):
                        for p in range(q):
                            e = edge.get((js[q], js[p]))
                            if e is not None: sc += F[e, s + p, s + q] - R
                    # tie-break towards the rule
                    sc += 0.5 * len(set(js) ^ set(rule))
                    if best is None or sc < best[0]: best = (sc, js)
            js = list(best[1])
            if verbose and set(js) != set(rule):
                print("v%d %s%+.0f: decoded %s  rule %s  (score %.1f)" % (z, ring[z], ang[z], js, rule, best[0]))
        for p, j in enumerate(js): edge[(z, j)] = s + p
        runs[z] = list(js); s += len(js)
    return runs, s

if __name__ == "__main__":
    from decode import forb_table
    from pattern_hyp import load_voc_bits
    F = forb_table(load_voc_bits())
    runs, total = decode(F)
    print("total bits", total)
    pairs = [(z, j) for z in sorted(runs) for j in runs[z]]
    from pattern_hyp import triangle_score
    t, n, w = triangle_score(pairs[:384], F); print("triangle violations", t, "over", n, "worst", sorted(w)[-5:])
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'brisk2_pairs.npy'), np.array(pairs[:384]))
