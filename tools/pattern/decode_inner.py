"""Exhaustive decode of the inner runs (hexagon X, ring R1) from a seed every triangle of which is violation-free.
Seed: centre c=0, X0..X5 = 1..6 (complete graph on 1..6: 20 triangles, 0 violations), bit 3 = (X3, c),
R1.0 = vertex 7 with partners X0, X1, X5.  For each further R1 vertex every subset of the lower vertices is scored
(sum over implied triangles of violations - R); the best subsets are printed -- they end exactly at bit 63, where
the first R2 run begins."""
import numpy as np, itertools, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from decode import forb_table
from pattern_hyp import load_voc_bits

def main(R=8.0):
    F = forb_table(load_voc_bits())
    seed = [(1, []), (2, [1]), (3, [1, 2]), (4, [0, 1, 2, 3]), (5, [1, 2, 3, 4]), (6, [1, 2, 3, 4, 5]), (7, [1, 2, 6])]
    edge = {}; a = 0
    for z, js in seed:
        for j in js: edge[(z, j)] = a; a += 1
    s = a
    for z in range(8, 17):
        out = []
        for m in range(1, 8):
            for js in itertools.combinations(range(z), m):
                sc = 0.0; nt = 0
                for q in range(m):
                    for p in range(q):
                        e = edge.get((js[q], js[p]))
                        if e is not None: sc += F[e, s + p, s + q] - R; nt += 1
                out.append((sc, m, js, nt))
        out.sort(key=lambda t: t[0])
        print("vertex", z, "first bit", s)
        for sc, m, js, nt in out[:4]: print("   score %.0f partners %s triangles %d" % (sc, js, nt))
        sc, m, js, nt = out[0]
        for p, j in enumerate(js): edge[(z, j)] = s + p
        s += m
    print("next run starts at bit", s)

if __name__ == "__main__":
    main()
