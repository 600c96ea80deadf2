"""GPU parity of the gated matchers beyond matchStereo, and of the remaining BASELINE configs.

 * matchMotionStereo (okvis_frontend/src/Frontend.cpp:1812-1905) -- FP64 gates + 4 px check
 * matchToMapByThread for 3-D landmarks (Frontend.cpp:1552-1589) -- reprojection gate + <=3
   descriptors per landmark
 * TUM-VI 1024x1024 (equidistant) and Hilti 720x540 x 5 cameras: detect + describe (+ FoV-overlap
   driven cross-camera matching) against the oracle.
"""
import numpy as np
import pytest

from okvis2_amd import capi, synth

import gpu_common as G

# every test of this module runs under both orders of the 3-term FP64 sums (conftest.fp64_order)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp64_order")]


def _scene(oracle, cam, n, seed, T):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.8, 0.8, n), rng.uniform(2.5, 9, n)], 1)
    Xc = X - np.asarray(T[1])
    kp = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
    kp["size"] = 12.0
    keep = np.zeros(n, bool)
    for i in range(n):
        st, pt, _ = oracle.cam_project(cam, Xc[i])
        if st == 0:
            kp["x"][i], kp["y"][i] = pt
            keep[i] = True
    return X, kp, keep


def test_match_motion_stereo(oracle):
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    rng = np.random.default_rng(3)
    n = 400
    T0 = (np.eye(3).reshape(-1), np.zeros(3))
    th = 0.05
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    T1 = (Rz.reshape(-1), np.array([0.35, 0.04, 0.02]))
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.8, 0.8, n), rng.uniform(2.5, 9, n)], 1)

    def observe(T, noise):
        Cm = np.asarray(T[0]).reshape(3, 3)
        Xc = (X - np.asarray(T[1])) @ Cm  # C^T (X - r)
        kp = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
        kp["size"] = 12.0
        for i in range(n):
            st, pt, _ = oracle.cam_project(cam, Xc[i])
            kp["x"][i], kp["y"][i] = pt if st == 0 else (5.0, 5.0)
        kp["x"] += rng.normal(0, noise, n).astype(np.float32)
        kp["y"] += rng.normal(0, noise, n).astype(np.float32)
        bp, bv = oracle.backproject_keypoints(cam, kp)
        return kp, bp, bv

    kp0, bp0, bv0 = observe(T0, 0.3)
    kp1, bp1, bv1 = observe(T1, 0.3)
    d0 = rng.integers(0, 256, (n, 48), dtype=np.uint8)
    d1 = d0 ^ (rng.random((n, 48)) < 0.03).astype(np.uint8) * rng.integers(0, 256, (n, 48), dtype=np.uint8)
    perm = rng.permutation(n)
    d1, kp1, bp1, bv1 = d1[perm], kp1[perm], bp1[perm], bv1[perm]
    skip0 = (rng.random(n) < 0.1).astype(np.uint8)
    matched1 = (rng.random(n) < 0.1).astype(np.uint8)
    bv0 = bv0.copy()
    bv0[::11] = 0
    for s0, m1 in ((skip0, matched1), (None, None)):
        ref = oracle.match_motion_stereo(d0, kp0, bp0, bv0, s0, d1, kp1, bp1, bv1, m1, T0, T1, cam,
                                         cfg.match_threshold)
        got = fe.match_motion_stereo(cam, d0, kp0, bp0, bv0, s0, d1, kp1, bp1, bv1, m1, T0, T1)
        for f in ("k1", "dist", "initialisable", "accepted"):
            assert np.array_equal(got[f], ref[f]), f
        assert np.array_equal(got["hp_W"].view(np.uint64), ref["hp_W"].view(np.uint64))
        hit = ref["k1"] >= 0
        # the reference stores acos(cos_quality); the host adaptor takes the acos
        import math
        q = np.array([math.acos(c) for c in got["cos_quality"][hit]])  # libm acos, as the C++ host does
        assert np.array_equal(q, ref["quality"][hit])
        assert hit.sum() > 100 and ref["accepted"].sum() > 50
    # device-resident variant: the same frames as gather blocks in HBM, flags in device arrays
    import torch
    from okvis2_amd import multigpu
    fe.set_camera(0, cam)
    K = cfg.max_kpts
    blk0 = torch.from_numpy(multigpu.pack_block_host(K, kp0, d0, bp0, bv0)).cuda()
    blk1 = torch.from_numpy(multigpu.pack_block_host(K, kp1, d1, bp1, bv1)).cuda()
    pad = lambda a: torch.from_numpy(np.concatenate([a, np.zeros(K - len(a), np.uint8)])).cuda()
    d_s0, d_m1 = pad(skip0), pad(matched1)
    d_out = torch.zeros((K, capi.MOTION_MATCH_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    for s0, m1, ps0, pm1 in ((skip0, matched1, d_s0.data_ptr(), d_m1.data_ptr()), (None, None, None, None)):
        ref = oracle.match_motion_stereo(d0, kp0, bp0, bv0, s0, d1, kp1, bp1, bv1, m1, T0, T1, cam,
                                         cfg.match_threshold)
        fe.match_motion_stereo_blocks_device(0, blk0.data_ptr(), blk1.data_ptr(), ps0, pm1, T0, T1,
                                             d_out.data_ptr())
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(capi.MOTION_MATCH_DTYPE).reshape(-1)[:n]
        for f in ("k1", "dist", "initialisable", "accepted"):
            assert np.array_equal(got[f], ref[f]), f
        assert np.array_equal(got["hp_W"].view(np.uint64), ref["hp_W"].view(np.uint64))
    # empty inputs
    assert len(fe.match_motion_stereo(cam, d0[:0], kp0[:0], bp0[:0], bv0[:0], None, d1, kp1, bp1, bv1,
                                      None, T0, T1)) == 0
    e = fe.match_motion_stereo(cam, d0, kp0, bp0, bv0, None, d1[:0], kp1[:0], bp1[:0], bv1[:0], None,
                               T0, T1)
    assert np.all(e["k1"] == -1)


@pytest.mark.parametrize("max_desc", [4, 8])
def test_match_to_map_3d(oracle, max_desc):
    """max_desc = 8: landmarks with up to 7 descriptors (more than the reference keeps), so some
    64-landmark chunks exceed the LDS staging area and take the direct-read path."""
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    rng = np.random.default_rng(8)
    n_k, n_lm = 650, 1800
    kps = np.zeros(n_k, dtype=oracle.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(30, 720, n_k)
    kps["y"] = rng.uniform(30, 450, n_k)
    desc = rng.integers(0, 256, (n_k, 48), dtype=np.uint8)
    use = (rng.random(n_k) > 0.15).astype(np.uint8)
    counts = rng.integers(1, max_desc, n_lm)
    counts[::17] = 0  # landmarks without descriptors
    desc_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    pool = rng.integers(0, 256, (desc_begin[-1], 48), dtype=np.uint8)
    proj = np.stack([rng.uniform(0, 752, n_lm), rng.uniform(0, 480, n_lm)], 1)
    # plant true matches: landmark l observes keypoint l (near its projection, few flipped bits)
    for l in range(0, min(n_k, n_lm), 2):
        if counts[l] == 0:
            continue
        proj[l] = (kps["x"][l] + rng.normal(0, 3), kps["y"][l] + rng.normal(0, 3))
        d = desc_begin[l] + rng.integers(0, counts[l])
        flips = (rng.random(48) < 0.05).astype(np.uint8) * rng.integers(0, 256, 48, dtype=np.uint8)
        pool[d] = desc[l] ^ flips
    # duplicates: two landmarks with the same best distance -> the first (lower index) wins
    pool[desc_begin[3]] = pool[desc_begin[1]] if counts[1] and counts[3] else pool[desc_begin[3]]
    for thr in (20.0, 150.0):
        rl, rd = oracle.match_to_map(desc, kps, use, proj, desc_begin, pool, thr, cfg.match_threshold)
        gl, gd = fe.match_to_map(desc, kps, use, proj, desc_begin, pool, thr)
        assert np.array_equal(gl, rl) and np.array_equal(gd, rd)
        assert (rl >= 0).sum() > 100
        assert np.all(rl[use == 0] == -1)
    gl, gd = fe.match_to_map(desc, kps, use, proj[:0], np.zeros(1, np.int32), pool[:0], 20.0)
    assert np.all(gl == -1) and np.all(gd == cfg.match_threshold)


def test_tumvi_1024_equidistant(oracle):
    """config/tumvi_slam_1024.yaml: 1024x1024, radius 50, threshold 5, <= 1000 keypoints.
    Detect + describe and the equidistant back-projections (fixed-sequence atan) are bit-exact."""
    cfg = synth.tumvi1024_config()
    fe = G.make_frontend(cfg)
    cam = cfg.cams[0]
    fe.set_camera(0, cam)
    rays, jac = oracle.awareness_maps(cam)
    img = synth.corners_image(cfg.w, cfg.h, 77)
    rk, rd = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                    oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu),
                                    (0.0, 1.0, 0.0))
    kps, desc, bp, bpv = fe.detect_describe(img, cam=0, gravity=(0.0, 1.0, 0.0))
    G.assert_keypoints_equal(kps, rk)
    assert np.array_equal(desc, rd) and len(kps) > 200
    rbp, rv = oracle.backproject_keypoints(cam, rk)
    assert np.array_equal(bpv, rv) and np.array_equal(bp.view(np.uint64), rbp.view(np.uint64))
    # the noise frame has ~50k NMS maxima: exercises the global-memory sort path
    noise = synth.noise_image(cfg.w, cfg.h, 78)
    G.assert_keypoints_equal(fe.detect(noise),
                             oracle.detect(noise, cfg.uniformity_radius, 0, cfg.abs_threshold,
                                           cfg.max_kpts))


def test_hilti_five_cameras_overlap_driven_matching(oracle):
    """config/hilti_challenge_2022.yaml shape: 5 equidistant 720x540 cameras.  Camera pairs are
    matched only where the fields of view overlap (Frontend.cpp:1998); here cameras 0/1 look
    forward, 2 looks backward: pairs (0,1) visited, (0,2), (1,2) skipped."""
    cfg = synth.hilti_config()
    # the forward pair shares one set of intrinsics so that the synthetic disparity is epipolar-consistent
    cams = [cfg.cams[0], cfg.cams[0], cfg.cams[2]]
    fe = G.make_frontend(cfg, num_cameras=3)
    eye, flip = np.eye(3), np.diag([-1.0, 1.0, -1.0])
    C_SC = [eye, eye, flip]
    small = [synth.Camera(c.w // 4, c.h // 4, c.fu / 4, c.fv / 4, c.cu / 4, c.cv / 4, c.dist_type, c.d)
             for c in cams]
    visit = [(i, j) for i in range(3) for j in range(i + 1, 3)
             if capi.camera_overlap(small[j], small[i], C_SC[i].T @ C_SC[j])]
    assert visit == [(0, 1)]
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 55)
    imgs = [L, R, synth.corners_image(cfg.w, cfg.h, 56)]
    res = []
    for ci, cam in enumerate(cams):
        fe.set_camera(ci, cam)
        rays, jac = oracle.awareness_maps(cam)
        rk, rd = oracle.detect_describe(imgs[ci], cfg.uniformity_radius, 0, cfg.abs_threshold,
                                        cfg.max_kpts, oracle.MODE_CAMERA_AWARE, rays, jac,
                                        np.float32(cam.fu), (0.0, 1.0, 0.0))
        k, d, bp, bv = fe.detect_describe(imgs[ci], cam=ci, gravity=(0.0, 1.0, 0.0))
        G.assert_keypoints_equal(k, rk)
        assert np.array_equal(d, rd) and len(k) > 50
        res.append((k, d, bp, bv))
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f = [0.5 * (c.fu + c.fv) for c in cams]
    for (i, j) in visit:
        (k0, d0, b0, v0), (k1, d1, b1, v1) = res[i], res[j]
        # the oracle consumes ITS OWN back-projections (bit-equal to the GPU's since round 2)
        ob0, ov0 = oracle.backproject_keypoints(cams[i], k0)
        ob1, ov1 = oracle.backproject_keypoints(cams[j], k1)
        assert np.array_equal(ob0.view(np.uint64), b0.view(np.uint64)) and np.array_equal(ov0, v0)
        ref = oracle.match_stereo(d0, k0, ob0, ov0, d1, k1, ob1, ov1, T0, T1, f[i], f[j],
                                  cfg.match_threshold)
        got = fe.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, T0, T1, f[i], f[j])
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert (ref["k1"] >= 0).sum() > 10


def test_match_to_map_uninitialised(oracle):
    """Frontend.cpp:1616-1719: landmarks that are not 3-D yet, matched through the epipolar /
    triangulation gates; includes the already-matched counting rule."""
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    rng = np.random.default_rng(12)
    n_k, n_lm = 500, 900
    focal = 0.5 * (cam.fu + cam.fv)
    X = np.stack([rng.uniform(-2, 2, n_lm), rng.uniform(-1, 1, n_lm), rng.uniform(2.5, 10, n_lm)], 1)
    th = 0.03
    Ry = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T1 = (Ry.reshape(-1), np.array([0.25, -0.03, 0.05]))
    # current frame: keypoint k observes landmark k (k < n_k) with pixel noise
    Xc = (X[:n_k] - T1[1]) @ Ry
    kps = np.zeros(n_k, dtype=oracle.KEYPOINT_DTYPE)
    for i in range(n_k):
        st, pt, _ = oracle.cam_project(cam, Xc[i])
        kps["x"][i], kps["y"][i] = pt if st == 0 else (9.0, 9.0)
    kps["x"] += rng.normal(0, 0.4, n_k).astype(np.float32)
    kps["y"] += rng.normal(0, 0.4, n_k).astype(np.float32)
    bp, bv = oracle.backproject_keypoints(cam, kps)
    desc = rng.integers(0, 256, (n_k, 48), dtype=np.uint8)
    # pool: 1..3 earlier observations per landmark from other camera centres
    counts = rng.integers(1, 4, n_lm)
    desc_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    m = desc_begin[-1]
    pool = rng.integers(0, 256, (m, 48), dtype=np.uint8)
    r0 = np.zeros((m, 3))
    e0 = np.zeros((m, 3))
    for l in range(n_lm):
        for d in range(desc_begin[l], desc_begin[l + 1]):
            r0[d] = rng.normal(0, 0.3, 3) + np.array([-0.2, 0, 0])
            ray = X[l] - r0[d] + rng.normal(0, 0.002, 3)
            e0[d] = ray / np.linalg.norm(ray)
            if l < n_k and rng.random() < 0.8:
                flips = (rng.random(48) < 0.04).astype(np.uint8) * rng.integers(0, 256, 48, dtype=np.uint8)
                pool[d] = desc[l] ^ flips
    use = (bv != 0) & (rng.random(n_k) > 0.1)
    previous = np.full(n_k, -1, dtype=np.int32)
    previous[::9] = np.arange(n_k)[::9]          # already carries the right landmark
    previous[4::9] = (np.arange(n_k)[4::9] + 1) % n_lm  # carries another one
    ref = oracle.match_to_map_uninit(desc, bp, use, previous, desc_begin, pool, e0, r0, T1, focal,
                                     cfg.match_threshold)
    got = fe.match_to_map_uninitialised(desc, bp, use, previous, desc_begin, pool, e0, r0, T1, focal)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[3], ref[3])
    assert np.array_equal(got[2].view(np.uint64), ref[2].view(np.uint64))
    assert got[4] == ref[4]
    assert (ref[0] >= 0).sum() > 100 and ref[4] > 10 and ref[3].sum() > 50
    e = fe.match_to_map_uninitialised(desc, bp, use, previous, np.zeros(1, np.int32), pool[:0], e0[:0],
                                      r0[:0], T1, focal)
    assert np.all(e[0] == -1) and e[4] == 0


def test_match_to_map_uninitialised_staircase_5000_landmarks(oracle):
    """5 000 landmarks in 8 ranges of the device kernel: every keypoint has ~14 noisy copies of its
    descriptor spread over the whole list with different Hamming distances, so the running minimum
    of the reference loop descends through several ranges; a third of the copies fail the
    epipolar / triangulation gate, far points triangulate as parallel (no hp stored, the hp of an
    EARLIER accepted pair survives), and the landmark a keypoint already carries holds several
    copies of which only a later one lies under the running best (the counting rule)."""
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    fe = G.make_frontend(cfg)
    rng = np.random.default_rng(2024)
    n_k, n_lm = 300, 5000
    focal = 0.5 * (cam.fu + cam.fv)
    th = -0.02
    Ry = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T1 = (Ry.reshape(-1), np.array([0.3, 0.02, -0.04]))
    far = rng.random(n_k) < 0.2
    X = np.stack([rng.uniform(-2, 2, n_k), rng.uniform(-1, 1, n_k), rng.uniform(2.5, 10, n_k)], 1)
    X[far] *= 300.0
    Xc = (X - T1[1]) @ Ry
    kps = np.zeros(n_k, dtype=oracle.KEYPOINT_DTYPE)
    for i in range(n_k):
        st, pt, _ = oracle.cam_project(cam, Xc[i])
        kps["x"][i], kps["y"][i] = pt if st == 0 else (9.0, 9.0)
    bp, bv = oracle.backproject_keypoints(cam, kps)
    desc = rng.integers(0, 256, (n_k, 48), dtype=np.uint8)
    counts = rng.integers(1, 6, n_lm)
    desc_begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    m = int(desc_begin[-1])
    pool = rng.integers(0, 256, (m, 48), dtype=np.uint8)
    r0 = rng.normal(0, 0.4, (m, 3)) + np.array([-0.3, 0, 0])
    e0 = rng.normal(0, 1, (m, 3)) + np.array([0, 0, 3.0])
    e0 /= np.linalg.norm(e0, axis=1, keepdims=True)
    previous = np.full(n_k, -1, dtype=np.int32)
    taken = np.zeros(m, bool)

    def plant(k, d, nbits, good):
        bits = rng.choice(384, nbits, replace=False)
        row = desc[k].copy()
        for b in bits:
            row[b >> 3] ^= np.uint8(1 << (b & 7))
        pool[d] = row
        ray = X[k] - r0[d]
        ray /= np.linalg.norm(ray)
        if not good:
            ray = ray + rng.normal(0, 0.25, 3)
            ray /= np.linalg.norm(ray)
        e0[d] = ray
        taken[d] = True

    for k in range(n_k):
        lms = rng.choice(n_lm, 14, replace=False)
        for l in lms:
            d = int(rng.integers(desc_begin[l], desc_begin[l + 1]))
            if not taken[d]:
                plant(k, d, int(rng.integers(3, 56)), rng.random() > 0.33)
        if k % 3 == 0:  # the carried landmark: all of its descriptors are copies, distances shuffled
            l = int(lms[rng.integers(0, 14)])
            previous[k] = l
            for d in range(desc_begin[l], desc_begin[l + 1]):
                plant(k, d, int(rng.integers(3, 56)), rng.random() > 0.25)
    use = (bv != 0) & (rng.random(n_k) > 0.05)
    ref = oracle.match_to_map_uninit(desc, bp, use, previous, desc_begin, pool, e0, r0, T1, focal,
                                     cfg.match_threshold)
    got = fe.match_to_map_uninitialised(desc, bp, use, previous, desc_begin, pool, e0, r0, T1, focal)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[3], ref[3])
    assert np.array_equal(got[2].view(np.uint64), ref[2].view(np.uint64))
    assert got[4] == ref[4]
    matched = ref[0] >= 0
    # the cases the fold has to get right are all present
    assert matched.sum() > 150 and ref[4] > 20
    assert (matched & (ref[3] == 0)).sum() > 10   # best pair parallel, no earlier non-parallel one
    assert (matched & (ref[3] != 0)).sum() > 100
    per = (n_lm + 7) // 8
    assert len(np.unique(ref[0][matched] // per)) == 8  # winners in every range


def test_gated_matchers_large_and_ambiguous(oracle):
    """1500 keypoints per side (segments longer than the LDS chunk: the scan reloads chunks every
    round) with clusters of near-identical descriptors, so that most rows have several candidates
    under the threshold whose geometric gate fails -- many rounds, both keys of a scan used."""
    cfg = synth.euroc_config()
    cam = cfg.cams[0]
    n = 1500
    fe = G.make_frontend(cfg, max_keypoints=2048)
    rng = np.random.default_rng(21)
    T0 = (np.eye(3).reshape(-1), np.zeros(3))
    T1 = (np.eye(3).reshape(-1), np.array([0.11, 0.0, 0.0]))
    X = np.stack([rng.uniform(-2.0, 2.0, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.0, 12.0, n)], 1)

    def observe(T):
        Xc = X - np.asarray(T[1])
        kp = np.zeros(n, dtype=oracle.KEYPOINT_DTYPE)
        kp["size"] = 12.0
        for i in range(n):
            st, pt, _ = oracle.cam_project(cam, Xc[i])
            kp["x"][i], kp["y"][i] = pt if st == 0 else (5.0, 5.0)
        kp["x"] += rng.normal(0, 0.2, n).astype(np.float32)
        kp["y"] += rng.normal(0, 0.2, n).astype(np.float32)
        bp, bv = oracle.backproject_keypoints(cam, kp)
        return kp, bp, bv

    kp0, bp0, bv0 = observe(T0)
    kp1, bp1, bv1 = observe(T1)
    # 60 descriptor clusters: every keypoint is a lightly perturbed copy of its cluster centre
    centres = rng.integers(0, 256, (60, 48), dtype=np.uint8)
    cl = rng.integers(0, 60, n)

    def perturb(p):
        return centres[cl] ^ ((rng.random((n, 48)) < p) * (1 << rng.integers(0, 8, (n, 48)))).astype(np.uint8)

    d0, d1 = perturb(0.08), perturb(0.08)
    perm = rng.permutation(n)
    d1, kp1, bp1, bv1 = d1[perm], kp1[perm], bp1[perm], bv1[perm]
    f = 0.5 * (cam.fu + cam.fv)
    ref = oracle.match_stereo(d0, kp0, bp0, bv0, d1, kp1, bp1, bv1, T0, T1, f, f, cfg.match_threshold)
    got = fe.match_stereo(d0, kp0, bp0, bv0, d1, kp1, bp1, bv1, T0, T1, f, f)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    # how ambiguous the data is: candidates under the threshold per row
    cnt = np.array([(np.unpackbits(d0[i] ^ d1, axis=1).sum(1) < cfg.match_threshold).sum()
                    for i in range(0, n, 50)])
    assert cnt.mean() > 5 and (ref["k1"] >= 0).sum() > 200
    skip0 = (rng.random(n) < 0.1).astype(np.uint8)
    matched1 = (rng.random(n) < 0.2).astype(np.uint8)
    refm = oracle.match_motion_stereo(d0, kp0, bp0, bv0, skip0, d1, kp1, bp1, bv1, matched1, T0, T1,
                                      cam, cfg.match_threshold)
    gotm = fe.match_motion_stereo(cam, d0, kp0, bp0, bv0, skip0, d1, kp1, bp1, bv1, matched1, T0, T1)
    for fld in ("k1", "dist", "initialisable", "accepted"):
        assert np.array_equal(gotm[fld], refm[fld]), fld
    assert np.array_equal(gotm["hp_W"].view(np.uint64), refm["hp_W"].view(np.uint64))
    assert (refm["k1"] >= 0).sum() > 100


def test_verify_place_batched_and_vocabulary_descent(oracle):
    """okvfe_verify_place_match (Frontend.cpp:330-355, all landmarks in one launch) and
    okvfe_fbrisk_transform (DBoW2 descent with the FBrisk trait) on the reference's real vocabulary
    data and on GPU-made descriptors, against the oracle."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    d = np.fromfile(os.path.join(gold, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    t = np.load(os.path.join(gold, "small_voc_tree.npz"))
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    fe.set_camera(0, cfg.cams[0])
    rng = np.random.default_rng(11)
    # landmarks with 1..4 descriptors each, frame of 700 descriptors incl. near-duplicates
    frame = np.concatenate([d[:500], d[500:700] ^ (rng.random((200, 48)) < 0.02).astype(np.uint8)])
    sizes = rng.integers(1, 5, 1500)
    begin = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pool = d[rng.integers(0, 819, begin[-1])] ^ ((rng.random((begin[-1], 48)) < 0.03) *
                                                  rng.integers(1, 256, (begin[-1], 48))).astype(np.uint8)
    for K in (700, 0, 1, 1500):  # 1500 > the LDS staging limit of the kernel
        fr = np.concatenate([frame, frame, frame])[:K]
        rk, rd = oracle.verify_place(pool, begin, fr, cfg.match_threshold)
        gk, gd = fe.verify_place_match(pool, begin, fr)
        assert np.array_equal(gk, rk) and np.array_equal(gd, rd), K
        if K == 700:
            assert (rd < cfg.match_threshold).sum() > 300
    gk, gd = fe.verify_place_match(pool[:0], np.zeros(1, np.int32), frame)
    assert len(gk) == 0
    # vocabulary descent
    cb, ci = oracle.voc_tree_arrays(t["parent"])
    img = synth.corners_image(cfg.w, cfg.h, 5)
    _, feats, _, _ = fe.detect_describe(img, cam=0, gravity=(0.0, 1.0, 0.0))
    feats = np.concatenate([feats, d, pool[:500]])
    rw, rn = oracle.voc_transform(feats, t["desc"], cb, ci, t["word"])
    gw, gn = fe.fbrisk_transform(feats, t["desc"], cb, ci, t["word"])
    assert np.array_equal(gw, rw) and np.array_equal(gn, rn)
    assert len(np.unique(rw)) > 300 and rw.min() >= 0 and rw.max() < 729
    with pytest.raises(capi.OkvfeError):
        fe.fbrisk_transform(feats[:4], t["desc"], cb, ci[::-1].copy(), t["word"])  # not a tree in id order


def test_place_recognition_query_descent_bow_vector_l1_scores(oracle):
    """dBow_->database.query(features, ...) of Frontend.cpp:756-766 end to end on the device path:
    vocabulary descent (okvfe_fbrisk_transform) -> BowVector (okvfe_bow_vector) -> L1 scores against
    every stored keyframe in one launch (okvfe_bow_query_l1), on the reference's real 9^3 vocabulary
    (node weights included) with keyframes made of GPU descriptors; scores as bit patterns against
    the oracle's inverted-file walk."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    t = np.load(os.path.join(gold, "small_voc_tree.npz"))
    word, weight = t["word"], t["weight"]
    n_words = int(word.max()) + 1
    ww = np.zeros(n_words)
    ww[word[word >= 0]] = weight[word >= 0]
    cb, ci = oracle.voc_tree_arrays(t["parent"])
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    fe.set_camera(0, cfg.cams[0])
    rng = np.random.default_rng(21)
    # 40 "places" (images), 6 keyframes each: the place's descriptors with a few bits flipped and a
    # random subset dropped -> 240 database entries; queries = fresh views of some places
    places = []
    for pl in range(40):
        _, feats, _, _ = fe.detect_describe(synth.corners_image(cfg.w, cfg.h, 900 + pl), cam=0,
                                            gravity=(0.0, 1.0, 0.0))
        places.append(feats)

    def view(pl):
        f = places[pl]
        keep = rng.random(len(f)) > 0.25
        flips = ((rng.random(f.shape) < 0.01) * rng.integers(1, 256, f.shape)).astype(np.uint8)
        return (f ^ flips)[keep]

    def bow(feats):
        w, _ = fe.fbrisk_transform(feats, t["desc"], cb, ci, t["word"])
        rw, _ = oracle.voc_transform(feats, t["desc"], cb, ci, t["word"])
        assert np.array_equal(w, rw)
        v = capi.bow_vector(w, ww, int(t["weighting"]), True)
        r = oracle.bow_vector(rw, ww, int(t["weighting"]), True)
        assert np.array_equal(v[0], r[0]) and np.array_equal(v[1].view(np.uint64), r[1].view(np.uint64))
        return v

    entries, owner = [], []
    for pl in range(40):
        for _ in range(6):
            entries.append(bow(view(pl)))
            owner.append(pl)
    entries.append((np.zeros(0, np.int32), np.zeros(0)))  # a keyframe without features
    owner.append(-1)
    begin = np.concatenate([[0], np.cumsum([len(e[0]) for e in entries])]).astype(np.int32)
    ids = np.concatenate([e[0] for e in entries]).astype(np.int32)
    vals = np.concatenate([e[1] for e in entries])
    owner = np.array(owner)
    hits = 0
    for pl in (0, 7, 19, 39):
        q = bow(view(pl))
        got = fe.bow_query_l1(begin, ids, vals, q[0], q[1])
        ref = oracle.bow_query_l1(begin, ids, vals, q[0], q[1], n_words)
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
        assert got[-1] == -1.0 and np.all(got[:-1] >= 0.0) and np.all(got[:-1] <= 1.0 + 1e-12)
        best = np.argsort(-got)[:6]
        hits += int((owner[best] == pl).sum())
    assert hits >= 20  # the six views of the queried place rank on top
    # a stored vector queried with itself: identical vectors score 1 up to the rounding of the norm
    e0 = entries[0]
    self_score = fe.bow_query_l1(begin, ids, vals, e0[0], e0[1])[0]
    assert abs(self_score - 1.0) < 1e-12
    # empty query, empty database
    assert np.all(fe.bow_query_l1(begin, ids, vals, np.zeros(0, np.int32), np.zeros(0)) == -1.0)
    assert len(fe.bow_query_l1(np.zeros(1, np.int32), ids[:0], vals[:0], q[0], q[1])) == 0
    with pytest.raises(capi.OkvfeError):
        fe.bow_query_l1(begin, ids[::-1].copy(), vals, q[0], q[1])  # not in ascending word order


def test_match_to_map_from_raw_landmark_table(oracle):
    """okvfe_match_to_map_landmarks (Frontend.cpp:1219-1411): projection, descriptor-view pooling and
    the 3-D matcher on the device from the raw landmark / observation tables, 6000 landmarks;
    every pooled field and every match against the oracle, then the un-initialised matcher on the
    status-2 landmarks the pooling returned."""
    import os
    import map_synth
    gold = os.path.join(os.path.dirname(__file__), "golden")
    voc = np.fromfile(os.path.join(gold, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    m = map_synth.make_map(6000, voc=voc)
    cfg = synth.euroc_config()
    fe = G.make_frontend(cfg)
    fe.set_camera(0, m["cam"])
    kps, desc, use = map_synth.make_frame(m, oracle)
    for exclusive, thr in ((False, 20.0), (True, 150.0)):
        ref = oracle.prepare_landmarks(m["hp"], m["quality"], m["obs_begin"], m["obs_pose"], m["obs_bp"],
                                       m["poses"], m["T1"], m["cam"], thr, exclusive)
        lm, bd, pool = fe.match_to_map_landmarks(0, m["hp"], m["quality"], m["obs_begin"], m["obs_pose"],
                                                 m["obs_desc"], m["obs_bp"], m["poses"], m["T1"], thr,
                                                 exclusive, desc, kps, use)
        for k in ("status", "n_desc", "obs_rows"):
            assert np.array_equal(pool[k], ref[k]), k
        for k in ("projection", "e_W", "r_W"):
            assert np.array_equal(pool[k].view(np.uint64), ref[k].view(np.uint64)), k
        assert (ref["status"] == 1).sum() > 500 and (ref["status"] == 2).sum() > 100
        assert (ref["status"] == 0).sum() > 500 and ref["n_desc"].max() == 2
        idx, proj, begin, rows = map_synth.packed_set(ref, m["obs_desc"], 1)
        rl, rd = oracle.match_to_map(desc, kps, use, proj, begin, rows, thr, cfg.match_threshold)
        rl = np.where(rl >= 0, idx[np.maximum(rl, 0)], -1)
        assert np.array_equal(lm, rl) and np.array_equal(bd, rd)
        assert (rl >= 0).sum() > 100
    # the status-2 landmarks feed matchToMapByThreadUnitialised unchanged
    idx, _, begin, rows = map_synth.packed_set(pool, m["obs_desc"], 2)
    e0 = np.concatenate([pool["e_W"][l, :pool["n_desc"][l]] for l in idx])
    r0 = np.concatenate([pool["r_W"][l, :pool["n_desc"][l]] for l in idx])
    bp, bv = oracle.backproject_keypoints(m["cam"], kps)
    prev = np.full(len(kps), -1, dtype=np.int32)
    f = 0.5 * (m["cam"].fu + m["cam"].fv)
    got = fe.match_to_map_uninitialised(desc, bp, use & bv, prev, begin, rows, e0, r0, m["T1"], f)
    want = oracle.match_to_map_uninit(desc, bp, use & bv, prev, begin, rows, e0, r0, m["T1"], f,
                                      cfg.match_threshold)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # empty table
    lm, bd, pool = fe.match_to_map_landmarks(0, m["hp"][:0], m["quality"][:0], np.zeros(1, np.int32),
                                             m["obs_pose"][:0], m["obs_desc"][:0], m["obs_bp"][:0], m["poses"],
                                             m["T1"], 20.0, False, desc, kps, use)
    assert np.all(lm == -1) and np.all(bd == cfg.match_threshold)
