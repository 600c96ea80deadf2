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


def corners_image(w: int, h: int, seed: int, cell: int = 16, levels=None, noise: int = 4,
                  jitter: int = 3) -> np.ndarray:
    """levels: None = cell grays U[16, 240], else the grays to draw from; noise: +-noise iid;
    jitter: +-jitter px on every cell border (0 = exact grid)."""
    rng = np.random.default_rng(seed)
    ncx, ncy = w // cell + 3, h // cell + 3
    if levels is None:
        level = rng.integers(16, 241, size=(ncy + 1, ncx + 1)).astype(np.int32)
    else:
        level = rng.choice(np.asarray(levels, dtype=np.int32), size=(ncy + 1, ncx + 1))
    # jittered borders: bx[row band][i] is the x position of vertical border i in that band
    bx = cell * np.arange(1, ncx + 1)[None, :] + rng.integers(-jitter, jitter + 1, size=(ncy, ncx))
    by = cell * np.arange(1, ncy + 1)[None, :] + rng.integers(-jitter, jitter + 1, size=(ncx, ncy))
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
    img = level[row, col] + rng.integers(-noise, noise + 1, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def stereo_pair(w: int, h: int, seed: int, kind: str = "corners", **corner_kw):
    """Left/right images; right = left content shifted left by `disparity` px (fresh content
    enters on the right edge), plus independent sensor noise (none when corner_kw asks for
    noise = 0).  corner_kw: passed to corners_image."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    disparity = int(rng.integers(4, 41))
    if kind == "noise":
        wide = noise_image(w + 64, h, seed)
    else:
        wide = corners_image(w + 64, h, seed, **corner_kw)
    left = wide[:, :w].copy()
    right = wide[:, disparity:disparity + w].astype(np.int32)
    sn = 0 if corner_kw.get("noise", 4) == 0 else 2
    right = np.clip(right + rng.integers(-sn, sn + 1, size=right.shape), 0, 255).astype(np.uint8)
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
    T_SC: list = None  # per camera (C_SC 3x3, r_SC 3): sensor-from-camera extrinsics, if the rig has them


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


def tumvi512_config() -> Config:
    """config/tumvi_slam_512.yaml:8-12,21-25,64-68 (equidistant, 512x512)."""
    c0 = Camera(512, 512, 190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504, 2,
                (0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202,
                 0.00020293673591811182))
    c1 = Camera(512, 512, 190.44236969414825, 190.4344384721956, 252.59949716835982, 254.91723064636983, 2,
                (0.0034003170790442797, 0.001766278153469831, -0.00266312569781606,
                 0.0003299517423931039))
    return Config("tumvi512", 512, 512, [c0, c1], 0.101, 40.0, 4, 55, 0, 800)


def d455_config() -> Config:
    """config/realsense_D455.yaml:8-12,21-25,79-83 (rectified 640x480 pair, zero distortion, 2500 keypoints)."""
    c = Camera(640, 480, 390.598938, 390.598938, 320.581665, 237.712845, 1, (0.0, 0.0, 0.0, 0.0))
    return Config("d455", 640, 480, [c, c], 0.095, 30.0, 5, 60, 0, 2500)


def d435i_config() -> Config:
    """config/realsense_D435i.yaml:9-13,22-26,63-67 (rectified 640x480 pair, zero distortion, 400 keypoints)."""
    c = Camera(640, 480, 386.235, 386.235, 323.074, 238.489, 1, (0.0, 0.0, 0.0, 0.0))
    return Config("d435i", 640, 480, [c, c], 0.05, 30.0, 5, 60, 0, 400)


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
    # T_SC rows 0..2 of hilti_challenge_2022.yaml:4-7,18-21,32-35,46-49,60-63 (row-major 3x4)
    T = [
        [0.0067080214518005230, 0.0024256436418034670, 0.9999745590269414341, 0.0512635496824681014,
         0.9999264200621181820, 0.0100911970216882967, -0.0067321767970210744, 0.0453901220122367860,
         -0.0101072701536598954, 0.9999461405473762943, -0.0023577731969992165, -0.0132149126987094190],
        [0.0016556126470598073, 0.0009350089535378861, 0.9999981923508760584, 0.0503920182199237218,
         0.9999840642813063729, 0.0053956906150397291, -0.0016606342845620788, -0.0627831669975007084,
         -0.0053972335694489823, 0.9999850060281121333, -0.0009260608798763888, -0.0131432680426852699],
        [0.9999897552434932058, 0.0042384707008869303, -0.0015889537990238949, 0.0068411376086779559,
         0.0042340516721276989, -0.9999871881804247575, -0.0027742172670246496, -0.0079967741380297819,
         -0.0016006918802386875, 0.0027674611333545632, -0.9999948894591306203, -0.0341138037431789470],
        [-0.9998916894135628786, 0.0127071404640830849, 0.0074254981595154035, -0.0027416993947013656,
         0.0073960416730231545, -0.0023635309353156166, 0.9999698556902045787, 0.0572301653448225520,
         0.0127243078107144667, 0.9999164676625461601, 0.0022692923994765855, -0.0110114836506446101],
        [0.9999880402484466746, 0.0047970088651064294, -0.0009529249803788963, -0.0065676390423793146,
         -0.0009424276629318935, -0.0021900897852223104, -0.9999971576643765792, -0.0748375416968655310,
         -0.0047990822216628587, 0.9999860960096795814, -0.0021855427584385988, -0.0168332753720958905],
    ]
    T_SC = []
    for row in T:
        m = np.array(row, dtype=np.float64).reshape(3, 4)
        T_SC.append((m[:, :3].copy(), m[:, 3].copy()))
    return Config("hilti", 720, 540, cams, 0.1, 50.0, 20, 60, 0, 700, T_SC)


def stereo_poses(baseline: float):
    """T_WC0 = identity, T_WC1 = pure x translation by the baseline (row-major C, r)."""
    eye = np.eye(3, dtype=np.float64).reshape(-1)
    return (eye.copy(), np.zeros(3)), (eye.copy(), np.array([baseline, 0.0, 0.0]))


def rig_poses(cfg: Config, C_WS=None, r_WS=None):
    """T_WC = T_WS * T_SC per camera (Frontend.cpp:2004-2005) as (C row-major flat, r)."""
    C_WS = np.eye(3) if C_WS is None else np.asarray(C_WS, dtype=np.float64)
    r_WS = np.zeros(3) if r_WS is None else np.asarray(r_WS, dtype=np.float64)
    return [((C_WS @ C).reshape(-1).copy(), C_WS @ r + r_WS) for C, r in cfg.T_SC]


def gravity_in_camera(C_WC) -> np.ndarray:
    """Extraction direction of Frontend::detectAndDescribe: T_WC.inverse().C() * (0,0,-1) as
    float32 (Frontend.cpp:247-251)."""
    C_WC = np.asarray(C_WC, dtype=np.float64).reshape(3, 3)
    return (C_WC.T @ np.array([0.0, 0.0, -1.0])).astype(np.float32)


def rig_overlap_pairs(cfg: Config, overlap_fn, subsample: int = 4):
    """Camera pairs im0 < im1 Frontend::matchStereo visits (Frontend.cpp:1990-2000): those with
    MultiFrame::hasOverlap(im0, im1) -- NCameraSystem::computeOverlaps
    (okvis_cv/src/NCameraSystem.cpp:48-119) -- evaluated on `subsample`-times smaller cameras
    (every subsample-th pixel ray; the test is an "any pixel" test).  overlap_fn(camera, other,
    R_other_cam) -> bool is okvfe_camera_overlap (capi.camera_overlap) or the oracle's."""
    q = float(subsample)
    small = [Camera(c.w // subsample, c.h // subsample, c.fu / q, c.fv / q, c.cu / q, c.cv / q,
                    c.dist_type, c.d) for c in cfg.cams]
    n = len(cfg.cams)
    return [(i, j) for i in range(n) for j in range(i + 1, n)
            if overlap_fn(small[j], small[i], cfg.T_SC[i][0].T @ cfg.T_SC[j][0])]


_CUBE_CACHE = {}


def _cube_faces(seed: int, size: int, cell: int):
    key = (seed, size, cell)
    if key not in _CUBE_CACHE:
        _CUBE_CACHE[key] = np.stack([corners_image(size, size, seed + 31 * f, cell=cell)
                                     for f in range(6)]).astype(np.float32)
    return _CUBE_CACHE[key]


def render_rig(cfg: Config, rays_per_cam, seed: int, radius: float = 4.0, face: int = 1536,
               cell: int = 32, r_S=None):
    """One multiframe of a rig with extrinsics: the sensor sits at r_S (default origin) inside a
    textured sphere of `radius` metres centred at the origin; every camera pixel's unit ray
    (rays_per_cam[c]: H x W x 3, e.g. the awareness-map rays of okvfe_build_awareness_maps) is
    intersected with the sphere and the hit point looks up a cube-mapped corner texture.  All
    cameras therefore see ONE consistent 3-D scene with real parallax (baseline / radius), so
    cross-camera matches triangulate."""
    r_S = np.zeros(3) if r_S is None else np.asarray(r_S, dtype=np.float64)
    tex = _cube_faces(seed, face, cell)
    out = []
    for ci, cam in enumerate(cfg.cams):
        C, r = cfg.T_SC[ci]
        rays = np.asarray(rays_per_cam[ci], dtype=np.float64).reshape(-1, 3)
        valid = np.abs(rays).sum(axis=1) > 0
        d = rays @ C.T                       # ray directions in the sensor (= world) frame
        o = r + r_S                          # camera centre
        od = d @ o
        s = -od + np.sqrt(np.maximum(od * od - (o @ o - radius * radius), 0.0))
        p = o[None, :] + s[:, None] * d
        a = np.abs(p)
        ax = np.argmax(a, axis=1)
        m = np.take_along_axis(a, ax[:, None], axis=1)[:, 0] + 1e-30
        sign = np.take_along_axis(p, ax[:, None], axis=1)[:, 0] < 0
        fidx = 2 * ax + sign
        u_ax, v_ax = (ax + 1) % 3, (ax + 2) % 3
        u = np.take_along_axis(p, u_ax[:, None], axis=1)[:, 0] / m
        v = np.take_along_axis(p, v_ax[:, None], axis=1)[:, 0] / m
        fu_ = (u * 0.5 + 0.5) * (face - 1)
        fv_ = (v * 0.5 + 0.5) * (face - 1)
        x0 = np.clip(np.floor(fu_).astype(np.int64), 0, face - 2)
        y0 = np.clip(np.floor(fv_).astype(np.int64), 0, face - 2)
        ax_, ay_ = (fu_ - x0).astype(np.float32), (fv_ - y0).astype(np.float32)
        t = tex[fidx, y0, x0] * (1 - ax_) * (1 - ay_) + tex[fidx, y0, x0 + 1] * ax_ * (1 - ay_) + \
            tex[fidx, y0 + 1, x0] * (1 - ax_) * ay_ + tex[fidx, y0 + 1, x0 + 1] * ax_ * ay_
        rng = np.random.default_rng(seed * 131 + ci)
        img = np.where(valid, t + rng.integers(-2, 3, size=t.shape), 0.0)
        out.append(np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(cam.h, cam.w))
    return out
