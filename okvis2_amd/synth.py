"""Deterministic synthetic inputs for the front-end.

S0 "noise"   : iid uniform u8 image -- mirrors the reference's own smoke-test input
               (okvis_cv/test/TestFrame.cpp:83-85, Eigen setRandom on 752x480).
S1 "corners" : piecewise-constant random-gray grid (cell 16 px, level U[16,240], cell borders
               jittered by +-3 px) plus iid noise U[-4,4]; the right image of a stereo pair is
               the left one shifted by a per-frame disparity, so true stereo matches exist.

Camera rigs mirror the shipped configs (config/euroc.yaml:3-27 etc.) with an idealised
x-baseline so that the synthetic disparity is epipolar-consistent near the image centre.
Everything is numpy on the host; nothing here touches the oracle or the GPU.
"""
from __future__ import annotations

import dataclasses

import numpy as np


def noise_image(w: int, h: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w), dtype=np.uint8)


def corners_image(w: int, h: int, seed: int, cell: int = 16) -> np.ndarray:
    rng = np.random.default_rng(seed)
    ncx, ncy = w // cell + 3, h // cell + 3
    level = rng.integers(16, 241, size=(ncy + 1, ncx + 1)).astype(np.int32)
    # jittered borders: bx[row band][i] is the x position of vertical border i in that band
    bx = cell * np.arange(1, ncx + 1)[None, :] + rng.integers(-3, 4, size=(ncy, ncx))
    by = cell * np.arange(1, ncy + 1)[None, :] + rng.integers(-3, 4, size=(ncx, ncy))
    xs = np.arange(w)
    ys = np.arange(h)
    col = np.empty((h, w), dtype=np.int32)
    row = np.empty((h, w), dtype=np.int32)
    for j in range(h // cell + 1):
        y0, y1 = j * cell, min((j + 1) * cell, h)
        if y0 >= h:
            break
        col[y0:y1, :] = np.searchsorted(bx[j], xs, side="right")[None, :]
    for i in range(w // cell + 1):
        x0, x1 = i * cell, min((i + 1) * cell, w)
        if x0 >= w:
            break
        row[:, x0:x1] = np.searchsorted(by[i], ys, side="right")[:, None]
    img = level[row, col] + rng.integers(-4, 5, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def stereo_pair(w: int, h: int, seed: int, kind: str = "corners"):
    """Left/right images; right = left content shifted left by `disparity` px (fresh content
    enters on the right edge), plus independent sensor noise."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    disparity = int(rng.integers(4, 41))
    if kind == "noise":
        wide = noise_image(w + 64, h, seed)
    else:
        wide = corners_image(w + 64, h, seed)
    left = wide[:, :w].copy()
    right = wide[:, disparity:disparity + w].astype(np.int32)
    right = np.clip(right + rng.integers(-2, 3, size=right.shape), 0, 255).astype(np.uint8)
    return left, right, disparity


@dataclasses.dataclass
class Camera:
    w: int
    h: int
    fu: float
    fv: float
    cu: float
    cv: float
    dist_type: int  # 0 none, 1 radial-tangential, 2 equidistant
    d: tuple


@dataclasses.dataclass
class Config:
    name: str
    w: int
    h: int
    cams: list
    baseline: float
    uniformity_radius: float
    abs_threshold: int
    match_threshold: int
    octaves: int
    max_kpts: int


def euroc_config() -> Config:
    """config/euroc.yaml:8-26,63-67 (intrinsics of cam0/cam1, front-end parameters)."""
    c0 = Camera(752, 480, 458.654880721, 457.296696463, 367.215803962, 248.37534061, 1,
                (-0.28340811217, 0.0739590738929, 0.000193595028569, 1.76187114545e-05))
    c1 = Camera(752, 480, 457.587426604, 456.13442556, 379.99944652, 255.238185386, 1,
                (-0.283683654496, 0.0745128430929, -0.000104738949098, -3.55590700274e-05))
    return Config("euroc", 752, 480, [c0, c1], 0.11, 38.0, 150, 60, 0, 700)


def mono640_config() -> Config:
    """BASELINE.json configs[1]: synthetic 640x480 mono, ~1000 keypoints."""
    c0 = Camera(640, 480, 350.0, 360.0, 320.0, 240.0, 1, (-0.1, 0.01, 0.0005, -0.0003))
    return Config("mono640", 640, 480, [c0], 0.0, 10.0, 5, 60, 0, 1000)


def tumvi1024_config() -> Config:
    """config/tumvi_slam_1024.yaml:8-12,21-25,64-68 (equidistant, 1024x1024)."""
    c0 = Camera(1024, 1024, 382.3307, 382.3203, 510.3634, 514.2949, 2,
                (0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202,
                 0.00020293673591811182))
    c1 = Camera(1024, 1024, 381.2574, 381.2415, 505.6990, 510.3345, 2,
                (0.0034003170790442797, 0.001766278153469831, -0.00266312569781606,
                 0.0003299517423931039))
    return Config("tumvi1024", 1024, 1024, [c0, c1], 0.101, 50.0, 5, 60, 0, 1000)


def hilti_config() -> Config:
    """config/hilti_challenge_2022.yaml:9-13,23-27,37-41,51-55,65-69,106-110
    (5 equidistant 720x540 cameras)."""
    f = [(351.31400364193297, 351.4911744656785), (352.6489794433894, 352.8586498571586),
         (350.70040966794545, 350.8792449525716), (352.9514843860555, 353.32837903547403),
         (351.5132148653381, 351.7557554938886)]
    c = [(367.8522793375995, 253.84021449809963), (347.8170010310082, 270.5806692485468),
         (375.2977403521422, 268.5927747079796), (363.93345228274336, 266.14511705007413),
         (342.8425988673232, 259.91793254535776)]
    d = [(-0.03696737352869157, -0.008917880497032812, 0.008912969593422046,
          -0.0037685977496087313),
         (-0.039086652082708805, -0.005525347047415151, 0.004398151558986798,
          -0.0019701263170917808),
         (-0.041202246303621064, -0.0012607385825244833, 0.0006712169937177444,
          -0.0006234254968089226),
         (-0.03890973498616883, -0.002604676547864069, 0.0004634700730293949,
          -0.00036698216675371063),
         (-0.03842764034005408, -0.005841411460411122, 0.003451041303088915,
          -0.0011463543672005018)]
    cams = [Camera(720, 540, f[i][0], f[i][1], c[i][0], c[i][1], 2, d[i]) for i in range(5)]
    return Config("hilti", 720, 540, cams, 0.1, 50.0, 20, 60, 0, 700)


def stereo_poses(baseline: float):
    """T_WC0 = identity, T_WC1 = pure x translation by the baseline (row-major C, r)."""
    eye = np.eye(3, dtype=np.float64).reshape(-1)
    return (eye.copy(), np.zeros(3)), (eye.copy(), np.array([baseline, 0.0, 0.0]))
