"""Localise the two sample points of a descriptor bit from its correlations with bits of known geometry.
Simulated (tools/pattern/simulate.py) sign vectors of candidate point pairs on a polar grid are correlated with the
simulated known bits; the candidate whose correlation profile matches the vocabulary's best wins."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import simulate as S

def build(known_pts, known_sig, known_pairs, grid_pts, grid_sig, n_images=6, max_kp=12000, seed=0):
    imgs = S.default_images(n_images)
    allpts = np.concatenate([known_pts, grid_pts]); allsig = np.concatenate([known_sig, grid_sig])
    vals = []
    for img in imgs:
        xy = S.keypoints(img)
        vals.append(S.sample_values(img, xy, allpts, allsig))
    V = np.concatenate(vals)
    rng = np.random.default_rng(seed)
    if len(V) > max_kp: V = V[rng.choice(len(V), max_kp, replace=False)]
    nk = len(known_pts)
    VK, VG = V[:, :nk], V[:, nk:]
    SK = np.sign(VK[:, known_pairs[:, 0]] - VK[:, known_pairs[:, 1]]).astype(np.float32)
    return SK, VG.astype(np.float32), VK.astype(np.float32)

def pair_profiles(VA, VB, SK, chunk=64):
    """profiles[p, q, k] = corr(sign(VA[:,p] - VB[:,q]), SK[:,k])"""
    n = len(SK); out = np.empty((VA.shape[1], VB.shape[1], SK.shape[1]), np.float32)
    for p in range(VA.shape[1]):
        D = np.sign(VA[:, p][:, None] - VB)          # n x nb
        out[p] = (D.T @ SK) / n
    return out

def best_match(profiles, target):
    """profiles [..., K], target [K] -> Pearson correlation map"""
    P = profiles - profiles.mean(-1, keepdims=True); T = target - target.mean()
    return (P @ T) / (np.linalg.norm(P, axis=-1) * np.linalg.norm(T) + 1e-9)
