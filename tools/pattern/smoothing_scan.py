"""Which smoothing width do the vocabulary's statistics ask for?  Forward simulation (simulate.py) of the recovered
pattern on 1/f images at the extraction scale; the simulated descriptors are CLUSTERED like a vocabulary (9^3
k-majority tree, 819 nodes) before their bit-correlation matrix is compared with the real one -- cluster centres are
denoised descriptors, and comparing raw descriptors with them favours spuriously wide smoothing.
k = Gaussian std of a sample in units of the published sigma (the oracle's box of half-side sigma: 0.58),
blur = additional Gaussian blur of the image in pixels."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import simulate as S, recovered as R
from scipy import ndimage
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pts, sig = R.points(); pr = R.pairs()
voc = np.fromfile(os.path.join(ROOT, 'tests/golden/small_voc_desc.bin'), dtype=np.uint8).reshape(-1, 48)
bv = np.unpackbits(voc, axis=1, bitorder='little').astype(float)
C = np.corrcoef(bv.T); iu = np.triu_indices(384, 1)
imgs = S.default_images(6)[1:]
kps = [S.keypoints(im) for im in imgs]
scale = 2.468
ring = np.concatenate([[q] * n for q, n in enumerate(R.COUNTS)])
inner = (ring[pr[:, 0]] <= 2) & (ring[pr[:, 1]] <= 2); outer = (ring[pr[:, 0]] >= 4) & (ring[pr[:, 1]] >= 4)
def blockcorr(Cs, m):
    idx = np.where(m)[0]; a = Cs[np.ix_(idx, idx)]; b = C[np.ix_(idx, idx)]; i2 = np.triu_indices(len(idx), 1)
    return np.corrcoef(a[i2], b[i2])[0, 1]
print("vocabulary: mean |c| %.3f" % np.abs(C[iu]).mean())
for k, blur in ((0.58, 0), (0.58, 1.5), (0.58, 3.0), (0.75, 0), (0.9, 0), (1.0, 0), (1.15, 0), (1.3, 0), (1.5, 0), (2.2, 0)):
    Bs = []
    for im, xy in zip(imgs, kps):
        imb = ndimage.gaussian_filter(im.astype(float), blur) if blur > 0 else im
        v = S.sample_values(imb, xy, pts * scale, sig * scale * k)
        Bs.append((v[:, pr[:, 0]] > v[:, pr[:, 1]]).astype(np.uint8))
    B = np.concatenate(Bs)
    nodes = S.kmajority_tree(B, seed=1).astype(float)
    Cs = np.corrcoef(nodes.T)
    print("k %.2f blur %.1f: all %.3f inner %.3f outer %.3f mean |c| %.3f" % (
        k, blur, np.corrcoef(Cs[iu], C[iu])[0, 1], blockcorr(Cs, inner), blockcorr(Cs, outer), np.abs(Cs[iu]).mean()), flush=True)
