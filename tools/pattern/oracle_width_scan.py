"""Like with like, exact arithmetic: the oracle extracts with the built-in pattern whose boxes are widened by a factor m,
its descriptors of 1/f images are clustered into a 9^3 k-majority tree, and the bit-correlation matrix of the 819
nodes is compared with the reference vocabulary's.  Measured (round 5): m 1.0: 0.814 (mean |c| 0.131; vocabulary
0.172), 1.3: 0.852, 1.5: 0.865 (0.156), 1.73: 0.874 (0.166), 2.0: 0.878 (0.177), 2.3: 0.863, 2.6: 0.861 (0.202)."""
import sys, os, math, ctypes as C, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, HERE)
import oracle_lib as O, simulate as S
voc = np.fromfile(os.path.join(ROOT, 'tests/golden/small_voc_desc.bin'), dtype=np.uint8).reshape(-1, 48)
bv = np.unpackbits(voc, axis=1, bitorder='little').astype(float); Cv = np.corrcoef(bv.T); iu = np.triu_indices(384, 1)
imgs = [S.synth_image(960, 1280, s) for s in range(4)]
base = O.pattern()
keep = type(base)(); C.memmove(C.byref(keep), C.byref(base), C.sizeof(keep))
for m in [float(a) for a in sys.argv[1:]] or (1.0, 1.3, 1.5, 1.73, 2.0, 2.3, 2.6):
    p = type(base)(); C.memmove(C.byref(p), C.byref(keep), C.sizeof(p))
    reach = 0.0
    for i in range(p.n_points):
        p.sigma_half[i] = np.float32(keep.sigma_half[i] * m)
        reach = max(reach, math.hypot(p.px[i], p.py[i]) + p.sigma_half[i])
    p.border = int(math.ceil(reach)) + 1
    O._PATTERN = p
    B = np.unpackbits(np.concatenate([O.detect_describe(im, 8.0, 0, 5, 6000, O.MODE_UPRIGHT)[1] for im in imgs]), axis=1, bitorder='little')
    Cs = np.corrcoef(S.kmajority_tree(B, seed=1).astype(float).T)
    print("box half-side x %.2f: %d descriptors, clustered correlation %.3f, mean |c| %.3f (vocabulary %.3f)" % (
        m, len(B), np.corrcoef(Cs[iu], Cv[iu])[0, 1], np.abs(Cs[iu]).mean(), np.abs(Cv[iu]).mean()), flush=True)
