"""ctypes binding of the CPU oracle (oracle/libokvfe_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package okvis2_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
POINT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
CAND_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("dist", "<i4")])
STEREO_MATCH_DTYPE = np.dtype([("k1", "<i4"), ("dist", "<i4"), ("initialisable", "<i4"),
                               ("pad", "<i4"), ("hp_W", "<f8", (4,))])
MOTION_MATCH_DTYPE = np.dtype([("k1", "<i4"), ("dist", "<i4"), ("initialisable", "<i4"),
                               ("accepted", "<i4"), ("quality", "<f8"), ("hp_W", "<f8", (4,))])

MODE_UPRIGHT, MODE_GRADIENT, MODE_CAMERA_AWARE = 0, 1, 2


class Pattern(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("px", C.c_float * 72), ("py", C.c_float * 72),
                ("sigma_half", C.c_float * 72), ("n_short", C.c_int32),
                ("short_i", C.c_uint8 * 384), ("short_j", C.c_uint8 * 384),
                ("n_long", C.c_int32), ("long_i", C.c_uint8 * 1100), ("long_j", C.c_uint8 * 1100),
                ("long_wdx", C.c_int32 * 1100), ("long_wdy", C.c_int32 * 1100),
                ("border", C.c_int32), ("rot_cos", C.c_int32 * 1024), ("rot_sin", C.c_int32 * 1024),
                ("rot_cosf", C.c_float * 1024), ("rot_sinf", C.c_float * 1024)]


class Camera(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("fu", C.c_double), ("fv", C.c_double),
                ("cu", C.c_double), ("cv", C.c_double), ("dist_type", C.c_int32),
                ("d", C.c_double * 4)]


class Pose(C.Structure):
    _fields_ = [("C", C.c_double * 9), ("r", C.c_double * 3)]


class FrontendParams(C.Structure):
    _fields_ = [("uniformity_radius", C.c_float), ("octaves", C.c_int32),
                ("abs_threshold", C.c_int32), ("max_kpts", C.c_int32), ("mode", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "libokvfe_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
            if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "CC=gcc"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_popcnt_xor.restype = C.c_uint32
        _LIB.orc_smoothed_intensity.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_camera(cam) -> Camera:
    c = Camera()
    c.w, c.h, c.fu, c.fv, c.cu, c.cv, c.dist_type = cam.w, cam.h, cam.fu, cam.fv, cam.cu, cam.cv, \
        cam.dist_type
    for i in range(4):
        c.d[i] = cam.d[i]
    return c


def make_pose(Cm, r) -> Pose:
    p = Pose()
    for i in range(9):
        p.C[i] = float(np.asarray(Cm).reshape(-1)[i])
    for i in range(3):
        p.r[i] = float(r[i])
    return p


_PATTERN = None


def pattern() -> Pattern:
    global _PATTERN
    if _PATTERN is None:
        _PATTERN = Pattern()
        lib().orc_pattern_build(C.byref(_PATTERN))
    return _PATTERN


def set_reduction(eigen_tree: bool = True):
    """orc_set_reduction: order of the 3-term FP64 sums of the gate chain (default: Eigen's x0 + (x1 + x2))"""
    lib().orc_set_reduction(1 if eigen_tree else 0)


def pattern_published() -> Pattern:
    """the 60-point / 383-pair published-BRISK form (the default until round 5), as a second pattern"""
    p = Pattern()
    lib().orc_pattern_build_published(C.byref(p))
    return p


def harris_score(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.int32)
    lib().orc_harris_score(_p(img), w, h, w, _p(out))
    return out


def nms(score: np.ndarray, thr: int) -> np.ndarray:
    h, w = score.shape
    cap = (w // 2 + 1) * h
    out = np.empty(cap, dtype=POINT_DTYPE)
    n = lib().orc_nms(_p(np.ascontiguousarray(score)), w, h, int(thr), _p(out), cap)
    return out[:n].copy()


def uniformity_select(pts: np.ndarray, w: int, h: int, radius: float, max_kpts: int) -> np.ndarray:
    pts = pts.copy()
    n = lib().orc_uniformity_select(_p(pts), len(pts), w, h, C.c_float(radius), int(max_kpts))
    return pts[:n].copy()


def subpixel2d(patch) -> tuple:
    s = np.ascontiguousarray(patch, dtype=np.int32).reshape(9)
    dx, dy = C.c_float(), C.c_float()
    lib().orc_subpixel2d(_p(s), C.byref(dx), C.byref(dy))
    return dx.value, dy.value


SCORE_HARRIS, SCORE_AGAST, SCORE_BRISK_SCALESPACE = 0, 1, 2


def agast_score(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w), dtype=np.int32)
    lib().orc_agast_score(_p(img), w, h, w, _p(out))
    return out


def detect(img, radius, octaves, thr, max_kpts, want_score=False, score_type=SCORE_HARRIS):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = max(int(max_kpts), 1) * max(1, 2 * int(octaves)) + 8
    if not radius > 0:
        cap = (w // 2 + 1) * h * max(1, 2 * int(octaves))
    kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
    score = np.empty((h, w), dtype=np.int32) if want_score else None
    n = lib().orc_detect_scored(_p(img), w, h, w, C.c_float(radius), int(octaves), int(thr),
                                int(max_kpts), _p(kps), cap, _p(score), int(score_type))
    return (kps[:n].copy(), score) if want_score else kps[:n].copy()


def layer_size(w, h, layer):
    lw, lh = C.c_int(), C.c_int()
    lib().orc_layer_size(int(w), int(h), int(layer), C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def halfsample(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), dtype=np.uint8)
    lib().orc_halfsample(_p(img), w, h, w, _p(out))
    return out


def twothirdsample(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty(((h // 3) * 2, (w // 3) * 2), dtype=np.uint8)
    lib().orc_twothirdsample(_p(img), w, h, w, _p(out))
    return out


def integral(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h + 1, w + 1), dtype=np.int32)
    lib().orc_integral(_p(img), w, h, w, _p(out))
    return out


def describe(img, kps, mode, rays=None, jac=None, fu=1.0, direction=(0.0, 1.0, 0.0), scale_invariant=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    kps = kps.copy()
    desc = np.zeros((max(len(kps), 1), 48), dtype=np.uint8)
    d = (C.c_float * 3)(*[float(v) for v in direction])
    fn = lib().orc_describe_scaled if scale_invariant else lib().orc_describe
    n = fn(_p(img), w, h, w, C.byref(pattern()), int(mode), _p(rays), _p(jac),
           C.c_float(fu), d, _p(kps), len(kps), _p(desc))
    return kps[:n].copy(), desc[:n].copy()


def detect_describe(img, radius, octaves, thr, max_kpts, mode, rays=None, jac=None, fu=1.0,
                    direction=(0.0, 1.0, 0.0), score_type=SCORE_HARRIS, scale_invariant=False):
    kps = detect(img, radius, octaves, thr, max_kpts, score_type=score_type)
    return describe(img, kps, mode, rays, jac, fu, direction, scale_invariant)


def scale_index(size) -> int:
    f = lib().orc_scale_index
    f.restype, f.argtypes = C.c_int, [C.c_float]
    return int(f(float(size)))


def pattern_scaled(index):
    out = type(pattern())()
    lib().orc_pattern_scaled(C.byref(pattern()), int(index), C.byref(out))
    return out


def popcnt_xor(a, b, n128=3) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().orc_popcnt_xor(_p(a), _p(b), int(n128)))


def atan_fixed(x) -> np.ndarray:
    """orc_atan_fixed element-wise (the oracle's fixed-sequence FP64 atan)."""
    f = lib().orc_atan_fixed
    f.restype, f.argtypes = C.c_double, [C.c_double]
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    return np.array([f(float(v)) for v in x], dtype=np.float64)


def awareness_maps(cam):
    c = make_camera(cam)
    rays = np.zeros((cam.h, cam.w, 3), dtype=np.float32)
    jac = np.zeros((cam.h, cam.w, 6), dtype=np.float32)
    lib().orc_cam_awareness_maps(C.byref(c), _p(rays), _p(jac))
    return rays, jac


def cam_overlap(cam, other, R_other_cam, want_mask=False):
    c, o = make_camera(cam), make_camera(other)
    R = (C.c_double * 9)(*[float(v) for v in np.asarray(R_other_cam).reshape(-1)])
    mask = np.zeros((cam.h, cam.w), dtype=np.uint8) if want_mask else None
    has = lib().orc_cam_overlap(C.byref(c), C.byref(o), R, _p(mask))
    return (bool(has), mask) if want_mask else bool(has)


def backproject_keypoints(cam, kps):
    c = make_camera(cam)
    n = len(kps)
    dirs = np.zeros((max(n, 1), 3), dtype=np.float64)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    kps = np.ascontiguousarray(kps)
    lib().orc_backproject_keypoints(C.byref(c), _p(kps), n, _p(dirs), _p(valid))
    return dirs[:n], valid[:n]


def cam_backproject(cam, pt):
    c = make_camera(cam)
    p = (C.c_double * 2)(*pt)
    d = (C.c_double * 3)()
    ok = lib().orc_cam_backproject(C.byref(c), p, d)
    return bool(ok), np.array(d[:])


def cam_project(cam, p3, want_jac=False):
    c = make_camera(cam)
    p = (C.c_double * 3)(*p3)
    pt = (C.c_double * 2)()
    J = (C.c_double * 6)()
    st = lib().orc_cam_project(C.byref(c), p, pt, J if want_jac else None)
    return st, np.array(pt[:]), (np.array(J[:]).reshape(2, 3) if want_jac else None)


def triangulate_fast(p1, e1, p2, e2, sigma):
    a = [(C.c_double * 3)(*v) for v in (p1, e1, p2, e2)]
    hp = (C.c_double * 4)()
    v, par = C.c_int(), C.c_int()
    lib().orc_triangulate_fast(a[0], a[1], a[2], a[3], C.c_double(sigma), hp, C.byref(v),
                               C.byref(par))
    return np.array(hp[:]), bool(v.value), bool(par.value)


def match_stereo(desc0, kp0, bp0, bpv0, desc1, kp1, bp1, bpv1, T0, T1, f0, f1, thr):
    n0, n1 = len(kp0), len(kp1)
    out = np.zeros(max(n0, 1), dtype=STEREO_MATCH_DTYPE)
    P0, P1 = make_pose(*T0), make_pose(*T1)
    arrs = [np.ascontiguousarray(a) for a in (desc0, kp0, bp0, bpv0, desc1, kp1, bp1, bpv1)]
    lib().orc_match_stereo(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), n0, _p(arrs[4]),
                           _p(arrs[5]), _p(arrs[6]), _p(arrs[7]), n1, C.byref(P0), C.byref(P1),
                           C.c_double(f0), C.c_double(f1), C.c_double(thr), _p(out))
    return out[:n0]


def match_motion_stereo(desc0, kp0, bp0, bpv0, skip0, desc1, kp1, bp1, bpv1, matched1, T0, T1, cam,
                        thr):
    n0, n1 = len(kp0), len(kp1)
    out = np.zeros(max(n0, 1), dtype=MOTION_MATCH_DTYPE)
    P0, P1 = make_pose(*T0), make_pose(*T1)
    c = make_camera(cam)
    arrs = [None if a is None else np.ascontiguousarray(a)
            for a in (desc0, kp0, bp0, bpv0, skip0, desc1, kp1, bp1, bpv1, matched1)]
    lib().orc_match_motion_stereo(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
                                  n0, _p(arrs[5]), _p(arrs[6]), _p(arrs[7]), _p(arrs[8]),
                                  _p(arrs[9]), n1, C.byref(P0), C.byref(P1), C.byref(c),
                                  C.c_uint32(int(thr)), _p(out))
    return out[:n0]


def hamming_candidates(A, B, thr):
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    cap = max(len(A) * len(B), 1)
    out = np.empty(cap, dtype=CAND_DTYPE)
    n = lib().orc_hamming_candidates(_p(A), len(A), _p(B), len(B), int(thr), _p(out), cap)
    return out[:n].copy()


def hamming_argmin(A, B, thr):
    A = np.ascontiguousarray(A, dtype=np.uint8)
    B = np.ascontiguousarray(B, dtype=np.uint8)
    bj = np.empty(max(len(A), 1), dtype=np.int32)
    bd = np.empty(max(len(A), 1), dtype=np.uint32)
    lib().orc_hamming_argmin(_p(A), len(A), _p(B), len(B), C.c_uint32(int(thr)), _p(bj), _p(bd))
    return bj[:len(A)], bd[:len(A)]


def match_to_map(desc, kps, use, proj, desc_begin, pool, repr_thr, thr):
    n, nl = len(kps), len(desc_begin) - 1
    bl = np.zeros(max(n, 1), dtype=np.int32)
    bd = np.zeros(max(n, 1), dtype=np.int32)
    arrs = [np.ascontiguousarray(a) for a in (desc, kps, np.asarray(use, dtype=np.uint8),
                                             np.asarray(proj, dtype=np.float64),
                                             np.asarray(desc_begin, dtype=np.int32), pool)]
    lib().orc_match_to_map(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), n, _p(arrs[3]), _p(arrs[4]), nl,
                           _p(arrs[5]), C.c_double(repr_thr), C.c_double(thr), _p(bl), _p(bd))
    return bl[:n], bd[:n]


def match_to_map_uninit(desc, bp, use, previous, desc_begin, pool, e0_W, r0_W, T1, focal, thr):
    n, nl = len(desc), len(desc_begin) - 1
    bl = np.zeros(max(n, 1), dtype=np.int32)
    bd = np.zeros(max(n, 1), dtype=np.int32)
    hp = np.zeros((max(n, 1), 4), dtype=np.float64)
    hs = np.zeros(max(n, 1), dtype=np.uint8)
    ctr = C.c_int32()
    arrs = [np.ascontiguousarray(desc, dtype=np.uint8), np.ascontiguousarray(bp, dtype=np.float64),
            np.ascontiguousarray(use, dtype=np.uint8), np.ascontiguousarray(previous, dtype=np.int32),
            np.ascontiguousarray(desc_begin, dtype=np.int32), np.ascontiguousarray(pool, dtype=np.uint8),
            np.ascontiguousarray(e0_W, dtype=np.float64), np.ascontiguousarray(r0_W, dtype=np.float64)]
    P1 = make_pose(*T1)
    lib().orc_match_to_map_uninit(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), n, _p(arrs[4]),
                                  nl, _p(arrs[5]), _p(arrs[6]), _p(arrs[7]), C.byref(P1),
                                  C.c_double(focal), C.c_double(thr), _p(bl), _p(bd), _p(hp), _p(hs),
                                  C.byref(ctr))
    return bl[:n], bd[:n], hp[:n], hs[:n], ctr.value


def verify_place(pool, desc_begin, frame_desc, thr):
    pool = np.ascontiguousarray(pool, dtype=np.uint8).reshape(-1, 48)
    db = np.ascontiguousarray(desc_begin, dtype=np.int32)
    fd = np.ascontiguousarray(frame_desc, dtype=np.uint8).reshape(-1, 48)
    n = len(db) - 1
    k_min = np.zeros(max(n, 1), dtype=np.int32)
    d_min = np.zeros(max(n, 1), dtype=np.uint32)
    lib().orc_verify_place(_p(pool), _p(db), n, _p(fd), len(fd), C.c_uint32(thr), _p(k_min), _p(d_min))
    return k_min[:n], d_min[:n]


def voc_tree_arrays(parent):
    """children lists (child_begin, child_index) of a tree given parent[] (node 0 = root),
    children in ascending node id = the order DBoW2 appends them while loading a file."""
    parent = np.asarray(parent)
    n = len(parent)
    kids = [[] for _ in range(n)]
    for i in range(1, n):
        kids[int(parent[i])].append(i)
    begin = np.zeros(n + 1, dtype=np.int32)
    for i in range(n):
        begin[i + 1] = begin[i] + len(kids[i])
    index = np.array([c for k in kids for c in k], dtype=np.int32)
    return begin, index


def voc_transform(desc, node_desc, child_begin, child_index, word):
    d = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 48)
    nd = np.ascontiguousarray(node_desc, dtype=np.uint8).reshape(-1, 48)
    cb = np.ascontiguousarray(child_begin, dtype=np.int32)
    ci = np.ascontiguousarray(child_index, dtype=np.int32)
    w = np.ascontiguousarray(word, dtype=np.int32)
    wo = np.zeros(max(len(d), 1), dtype=np.int32)
    no = np.zeros(max(len(d), 1), dtype=np.int32)
    lib().orc_voc_transform(_p(d), len(d), _p(nd), _p(cb), _p(ci), _p(w), _p(wo), _p(no))
    return wo[:len(d)], no[:len(d)]


def acos_fixed(x) -> np.ndarray:
    f = lib().orc_acos_fixed
    f.restype, f.argtypes = C.c_double, [C.c_double]
    return np.array([f(float(v)) for v in np.atleast_1d(np.asarray(x, dtype=np.float64))])


def prepare_landmarks(hp_W, quality, obs_begin, obs_pose, obs_bp, poses, T_WC1, cam, repr_thr, exclusive):
    """orc_prepare_landmarks: Frontend.cpp:1219-1359 (projection + descriptor-view pooling)."""
    hp = np.ascontiguousarray(hp_W, dtype=np.float64).reshape(-1, 4)
    q = np.ascontiguousarray(quality, dtype=np.float64)
    ob = np.ascontiguousarray(obs_begin, dtype=np.int32)
    op = np.ascontiguousarray(obs_pose, dtype=np.int32)
    obp = np.ascontiguousarray(obs_bp, dtype=np.float64).reshape(-1, 3)
    P = (Pose * max(len(poses), 1))(*[make_pose(*p) for p in poses])
    T1 = make_pose(*T_WC1)
    c = make_camera(cam)
    nl = len(hp)
    out = {"status": np.zeros(max(nl, 1), np.int32), "n_desc": np.zeros(max(nl, 1), np.int32),
           "obs_rows": np.zeros((max(nl, 1), 3), np.int32), "projection": np.zeros((max(nl, 1), 2)),
           "e_W": np.zeros((max(nl, 1), 2, 3)), "r_W": np.zeros((max(nl, 1), 2, 3))}
    lib().orc_prepare_landmarks(_p(hp), _p(q), _p(ob), nl, _p(op), _p(obp), P, C.byref(T1), C.byref(c),
                                C.c_double(repr_thr), int(bool(exclusive)), _p(out["status"]),
                                _p(out["n_desc"]), _p(out["obs_rows"]), _p(out["projection"]),
                                _p(out["e_W"]), _p(out["r_W"]))
    return {k: v[:nl] for k, v in out.items()}


def bow_vector(word_ids, word_weight, weighting=0, normalise_l1=True):
    w = np.ascontiguousarray(word_ids, dtype=np.int32)
    ww = np.ascontiguousarray(word_weight, dtype=np.float64)
    ids = np.zeros(len(ww), dtype=np.int32)
    vals = np.zeros(len(ww), dtype=np.float64)
    n = lib().orc_bow_vector(_p(w), len(w), _p(ww), len(ww), int(weighting), int(bool(normalise_l1)),
                             _p(ids), _p(vals))
    return ids[:n].copy(), vals[:n].copy()


def bow_query_l1(db_begin, db_ids, db_values, q_ids, q_values, n_words):
    b = np.ascontiguousarray(db_begin, dtype=np.int32)
    ids = np.ascontiguousarray(db_ids, dtype=np.int32)
    vals = np.ascontiguousarray(db_values, dtype=np.float64)
    qi = np.ascontiguousarray(q_ids, dtype=np.int32)
    qv = np.ascontiguousarray(q_values, dtype=np.float64)
    n = len(b) - 1
    scores = np.zeros(max(n, 1), dtype=np.float64)
    lib().orc_bow_query_l1(_p(b), _p(ids), _p(vals), n, _p(qi), _p(qv), len(qi), int(n_words), _p(scores))
    return scores[:n]
