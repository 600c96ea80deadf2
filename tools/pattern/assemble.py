"""Assemble the recovered BRISK2 pair list: 66 sample points (centre, hexagon X, rings of 10/14/15/20), bit order =
for i: for j<i.  Bits 0..62 (centre/X/R1 runs) are taken from the triangle decode (tools/pattern/decode*.py,
confirmed by simulation ranking); the R2..R4 runs follow angular thresholds read off the decode."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

N_X, RINGS = 6, (10, 14, 15, 20)
def vertex_table():
    ring = ['c'] + ['X'] * 6; ang = [0.0] + [60.0 * k for k in range(6)]
    for q, n in enumerate(RINGS):
        ring += ['R%d' % (q + 1)] * n; ang += [360.0 * k / n for k in range(n)]
    return ring, np.array(ang)

EARLY_RUNS = {1: [], 2: [1], 3: [1, 2], 4: [0, 1, 2, 3], 5: [1, 2, 3, 4], 6: [1, 2, 3, 4, 5], 7: [1, 2, 6],
              8: [3, 6, 7], 9: [0, 1, 3, 7, 8], 10: [2, 4, 8, 9], 11: [2, 3, 5, 9, 10], 12: [3, 4, 5, 10, 11],
              13: [3, 5, 6, 11, 12], 14: [4, 6, 12, 13], 15: [0, 1, 5, 7, 13, 14], 16: [1, 5, 6, 7, 8, 14, 15]}
THRESH = {('X', 'R2'): 38, ('R1', 'R2'): 59, ('R2', 'R2'): 52, ('R1', 'R3'): 30, ('R2', 'R3'): 50.5, ('R3', 'R3'): 49,
          ('R2', 'R4'): 9, ('R3', 'R4'): 39, ('R4', 'R4'): 37}

def pair_list(thresh=THRESH, early=EARLY_RUNS):
    ring, ang = vertex_table(); pairs = []
    for i in range(1, len(ring)):
        if i in early:
            pairs += [(i, j) for j in early[i]]; continue
        for j in range(1, i):
            t = thresh.get((ring[j], ring[i]))
            if t is not None and abs((ang[i] - ang[j] + 180) % 360 - 180) <= t + 1e-6: pairs.append((i, j))
    return pairs

if __name__ == "__main__":
    from pattern_hyp import triangle_score
    from decode import forb_table
    from pattern_hyp import load_voc_bits
    F = forb_table(load_voc_bits())
    pr = pair_list(); print("pairs", len(pr))
    t, n, w = triangle_score(pr[:384], F); print("triangle violations", t, "over", n, "triangles; worst", sorted(w)[-5:])
