import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import oracle_lib as O
from okvis2_amd import capi, synth
import test_gpu_fuzz as F
def run(seed, radius=None, maxk=None, thr=None):
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.integers(20, 230)) * 4; h = int(rng.integers(70, 420))
    kind = ["noise", "corners", "blocks"][seed % 3]
    r = float(rng.choice([6.0, 10.0, 17.5, 26.0, 38.0])); t = int(rng.choice([1, 5, 40, 150, 400])); mk = int(rng.choice([50, 300, 700, 2000]))
    radius = radius or r; thr = thr or t; maxk = maxk or mk
    img = F._image(rng, w, h, kind)
    fe = capi.Frontend(w, h, radius, 0, thr, maxk, max_candidates=1<<16)
    det = fe.detect(img); ref = O.detect(img, radius, 0, thr, maxk)
    nn = len(O.nms(O.harris_score(img), thr))
    a = set(zip(det['x'].tolist(), det['y'].tolist())); b = set(zip(ref['x'].tolist(), ref['y'].tolist()))
    first = next((i for i in range(min(len(det),len(ref))) if det[i]!=ref[i]), None)
    print(seed, w, h, kind, radius, thr, maxk, "cand", nn, "gpu", len(det), "ref", len(ref), "first mismatch", first, "only gpu", len(a-b), "only ref", len(b-a))
    if first is not None:
        print("   gpu", det[first], "\n   ref", ref[first])
run(5); run(5, radius=26.0); run(5, radius=10.0); run(5, maxk=2000); run(2); run(8); run(11); run(14)
