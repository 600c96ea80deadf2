"""Forward simulator for the pattern-recovery experiments: a plain numpy BRISK-style descriptor (Gaussian
smoothing per point, bilinear sampling, upright) on Harris keypoints of the real test image and of synthetic
1/f images.  Used only to validate tools/pattern/* on a pattern whose pair table is known."""
import numpy as np, sys, os
from scipy import ndimage
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

def classic_pattern(dmax=5.10, scale=1.0):
    rr = [0, 2.9, 4.9, 7.4, 10.8]; nn = [1, 10, 14, 15, 20]
    pts = []; sig = []
    for r, n in zip(rr, nn):
        for j in range(n):
            a = 2 * np.pi * j / n
            pts.append((0.85 * r * np.cos(a), 0.85 * r * np.sin(a)))
            sig.append(1.3 * 0.5 if r == 0 else 1.3 * 0.85 * r * np.sin(np.pi / n))
    pts = np.array(pts) * scale; sig = np.array(sig) * scale
    pairs = [(i, j) for i in range(1, 60) for j in range(i) if np.hypot(*(pts[i] - pts[j])) < dmax * scale]
    return pts, sig, np.array(pairs)

def synth_image(h, w, seed, beta=2.0):
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(h)[:, None]; fx = np.fft.fftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1
    spec = (rng.standard_normal((h, w)) + 1j * rng.standard_normal((h, w))) / f ** (beta / 2)
    img = np.real(np.fft.ifft2(spec)); img = (img - img.mean()) / img.std()
    return np.clip(128 + 50 * img, 0, 255).astype(np.uint8)

def sample_values(img, kps_xy, pts, sig):
    """values[n_kp, n_pts] of the smoothed image at kp + pts."""
    img = img.astype(np.float64)
    out = np.empty((len(kps_xy), len(pts)))
    levels = {}
    for k, s in enumerate(sig):
        key = round(float(s), 3)
        if key not in levels:
            levels[key] = ndimage.gaussian_filter(img, s, mode='nearest')
        xs = kps_xy[:, 0] + pts[k, 0]; ys = kps_xy[:, 1] + pts[k, 1]
        out[:, k] = ndimage.map_coordinates(levels[key], [ys, xs], order=1, mode='nearest')
    return out

def keypoints(img, radius=8.0, thr=5, maxk=6000, border=30):
    import oracle_lib as O
    O.lib()
    kd = O.detect(img, radius, 0, thr, maxk)
    xy = np.stack([kd['x'], kd['y']], 1).astype(np.float64)
    h, w = img.shape
    ok = (xy[:, 0] > border) & (xy[:, 0] < w - border) & (xy[:, 1] > border) & (xy[:, 1] < h - border)
    return xy[ok]

def descriptors(imgs, pts, sig, pairs):
    bits = []
    for img in imgs:
        xy = keypoints(img)
        v = sample_values(img, xy, pts, sig)
        bits.append((v[:, pairs[:, 0]] > v[:, pairs[:, 1]]).astype(np.uint8))
    return np.concatenate(bits)

def default_images(n_synth=6):
    fx = np.load(os.path.join(ROOT, 'tests/golden/real_image.npz'))
    imgs = [fx['image']]
    for s in range(n_synth):
        imgs.append(synth_image(960, 1280, s))
    return imgs

def kmajority_tree(bits, k=9, levels=3, seed=0, iters=8):
    """hierarchical k-majority clustering: the node descriptors of a DBoW2-style vocabulary."""
    rng = np.random.default_rng(seed)
    nodes = []
    def split(idx, lev):
        if lev == levels or len(idx) < k: return
        X = bits[idx].astype(np.float32)
        cent = X[rng.choice(len(idx), k, replace=False)]
        for _ in range(iters):
            d = X @ (1 - cent).T + (1 - X) @ cent.T
            lab = d.argmin(1)
            for c in range(k):
                if (lab == c).any(): cent[c] = (X[lab == c].mean(0) >= 0.5)
        for c in range(k):
            nodes.append(cent[c].astype(np.uint8))
            split(idx[lab == c], lev + 1)
    split(np.arange(len(bits)), 0)
    return np.array(nodes)

if __name__ == "__main__":
    pts, sig, pairs = classic_pattern()
    bits = descriptors(default_images(), pts, sig, pairs)
    print(bits.shape, bits.mean())
    voc = kmajority_tree(bits)
    print(voc.shape)
    np.savez('/tmp/sim_classic.npz', bits=bits, voc=voc, pts=pts, sig=sig, pairs=pairs)
