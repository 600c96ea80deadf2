"""The recovered BRISK2 pattern as data: 66 points + 384 ordered pairs (bit b = value[i] > value[j])."""
import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
RADII = (0.0, 1.4, 2.9, 4.9, 7.4, 10.8); COUNTS = (1, 6, 10, 14, 15, 20); F = 0.85; SIGMA_SCALE = 1.3
def points():
    pts = []; sig = []
    for r, n in zip(RADII, COUNTS):
        for k in range(n):
            a = 2 * np.pi * k / n
            pts.append((F * r * np.cos(a), F * r * np.sin(a)))
            sig.append(SIGMA_SCALE * 0.5 if r == 0 else SIGMA_SCALE * F * r * np.sin(np.pi / n))
    return np.array(pts), np.array(sig)
def pairs():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'brisk2_pairs.npy'))
