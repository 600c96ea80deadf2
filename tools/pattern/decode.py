"""Sequential decoder for an i-major pair list (for i in 1..N-1: for j in 0..i-1: if short(i,j): emit bit).
Bits come in runs, one run per sample point i, inside a run j ascends.  Evidence: for x < y < z with all three
comparisons present, the bits a=(y,x) < b=(z,x) < c=(z,y) never show the cyclic outcome a=1,b=0,c=1 / 0,1,0."""
import numpy as np, sys

def forb_table(B):
    """F[a,b,c] = #samples with (a,b,c) = (1,0,1) or (0,1,0)."""
    B = B.astype(np.float32); n = B.shape[1]; Bn = 1 - B
    F = np.empty((n, n, n), np.float32)
    for a in range(n):
        F[a] = (B[:, a][:, None] * Bn).T @ B + (Bn[:, a][:, None] * B).T @ Bn
    return F

def decode(F, n_bits, R=8.0, beam=300, max_run=16, verbose=False, lookahead=True):
    edge = {}            # (i,j) -> bit
    runs = []            # list of (start, [j...])
    def best_run(s, m, z, edge):
        """assign ascending j's (vertices < z) to bits s..s+m-1; returns (score, js). lower score = better."""
        cands = [(0.0, [])]
        for p in range(m):
            b = s + p
            new = []
            for sc, js in cands:
                lo = js[-1] + 1 if js else 0
                for j in range(lo, z - (m - 1 - p)):
                    add = 0.0
                    for q, jq in enumerate(js):      # earlier bit in the run: s+q = (z,jq); closing edge (j,jq)
                        a = edge.get((j, jq))
                        if a is not None:
                            add += F[a, s + q, b] - R
                    new.append((sc + add, js + [j]))
            new.sort(key=lambda t: t[0])
            cands = new[:beam]
            if not cands: return None
        return cands[0]
    s = 0; z = 1
    while s < n_bits:
        opts = []
        for m in range(1, min(max_run, z, n_bits - s) + 1):
            r = best_run(s, m, z, edge)
            if r is None: continue
            sc, js = r
            if lookahead and s + m < n_bits:
                e2 = dict(edge)
                for p, j in enumerate(js): e2[(z, j)] = s + p
                best2 = 0.0
                for m2 in range(1, min(max_run, z + 1, n_bits - s - m) + 1):
                    r2 = best_run(s + m, m2, z + 1, e2)
                    if r2 is not None: best2 = min(best2, r2[0])
                sc2 = sc + best2
            else:
                sc2 = sc
            opts.append((sc2, sc, m, js))
        opts.sort(key=lambda t: t[0])
        sc2, sc, m, js = opts[0]
        for p, j in enumerate(js): edge[(z, j)] = s + p
        runs.append((s, js))
        if verbose: print(f"vertex {z}: bits {s}..{s+m-1} -> j={js} score {sc:.0f} (+next {sc2-sc:.0f})", flush=True)
        s += m; z += 1
    return runs, edge

if __name__ == "__main__":
    src = sys.argv[1]
    if src == 'voc':
        voc = np.fromfile('tests/golden/small_voc_desc.bin', dtype=np.uint8).reshape(-1, 48)
        B = np.unpackbits(voc, axis=1, bitorder='little'); truth = None
    else:
        d = np.load(src); B = d['voc']; truth = d['pairs']
    F = forb_table(B)
    runs, edge = decode(F, B.shape[1], verbose=True)
    pairs = np.zeros((B.shape[1], 2), int)
    for (i, j), a in edge.items(): pairs[a] = (i, j)
    if truth is not None:
        print("exact pair matches", int((pairs == truth).all(1).sum()), "of", len(truth))
    if len(sys.argv) > 2: np.save(sys.argv[2], pairs)
