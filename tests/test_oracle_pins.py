"""Pins the CPU oracle against every known answer the reference tree offers for the path
 -- these run without a GPU.

 * Hamming: popcnt(x^x)=0, popcnt(x^~x)=384 and the statistics of the 819 REAL BRISK2 descriptors
   of the reference's vocabulary resources/small_voc.yml.gz (fixture tests/golden/small_voc_desc.bin,
   extracted by tools/make_voc_fixture.py).
 * Camera model: the tolerances of okvis_cv/test/TestPinholeCamera.cpp:52-140 on the reference's
   createTestObject() cameras (PinholeCamera.hpp:389-393, distortion testObject()s).
 * FoV overlap truth table of okvis_cv/test/TestNCameraSystem.cpp:55-112.
 * triangulateFast (okvis_frontend/src/stereo_triangulation.cpp:50-132): intersecting, parallel,
   diverging rays.
"""
import os

import numpy as np
import pytest

from okvis2_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def voc():
    return np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)


def test_popcnt_known_answers(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = rng.integers(0, 256, 48, dtype=np.uint8)
        assert oracle.popcnt_xor(x, x) == 0
        assert oracle.popcnt_xor(x, ~x) == 384
        y = rng.integers(0, 256, 48, dtype=np.uint8)
        assert oracle.popcnt_xor(x, y) == int(np.unpackbits(x ^ y).sum())
        assert oracle.popcnt_xor(x, y, 1) == int(np.unpackbits((x ^ y)[:16]).sum())


def test_vocabulary_descriptor_statistics(oracle):
    """Values measured on the reference's own data file."""
    d = voc()
    assert d.shape == (819, 48)
    bits = np.unpackbits(d, axis=1).astype(np.int32)
    assert abs(bits.mean() - 0.498) < 0.002
    dist = bits @ (1 - bits).T
    dist = dist + dist.T  # pairwise Hamming
    iu = np.triu_indices(819, 1)
    pd = dist[iu]
    assert pd.min() == 8 and pd.max() == 372
    assert abs(pd.mean() - 191.9) < 0.1
    assert abs((pd < 60).mean() - 0.0027) < 0.0003
    # the oracle's popcount and its matchers agree with the numpy bit count on the real data
    rng = np.random.default_rng(1)
    for i, j in rng.integers(0, 819, (50, 2)):
        assert oracle.popcnt_xor(d[i], d[j]) == dist[i, j]
    cand = oracle.hamming_candidates(d[:200], d[200:500], 60)
    ii, jj = np.nonzero(dist[:200, 200:500] < 60)
    assert np.array_equal(cand["i"], ii) and np.array_equal(cand["j"], jj)
    assert np.array_equal(cand["dist"], dist[:200, 200:500][ii, jj])
    bj, bd = oracle.hamming_argmin(d[:200], d[200:500], 60)
    sub = dist[:200, 200:500]
    want_d = np.minimum(sub.min(1), 60)
    want_j = np.where(sub.min(1) < 60, sub.argmin(1), -1)
    assert np.array_equal(bd, want_d) and np.array_equal(bj, want_j)


def test_host_popcnt_of_product_library_matches(oracle):
    """okvfe_popcnt_xor is a host function of libokvfe.so (no GPU needed)."""
    from okvis2_amd import capi
    d = voc()
    for i in range(0, 800, 37):
        assert capi.popcnt_xor(d[i], d[i + 1]) == oracle.popcnt_xor(d[i], d[i + 1])
        assert capi.popcnt_xor(d[i], d[i + 1], 1) == oracle.popcnt_xor(d[i], d[i + 1], 1)


def _test_cameras():
    # createTestObject(): 752x480, f=(350,360), c=(378,238); distortion testObject() values
    return [synth.Camera(752, 480, 350.0, 360.0, 378.0, 238.0, 0, (0.0, 0.0, 0.0, 0.0)),
            synth.Camera(752, 480, 350.0, 360.0, 378.0, 238.0, 1, (-0.16, 0.15, 0.0003, 0.0002)),
            synth.Camera(752, 480, 350.0, 360.0, 378.0, 238.0, 2, (-0.21, 0.14, 0.0006, 0.0003))]


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_pinhole_roundtrip_and_jacobian(oracle, ci):
    cam = _test_cameras()[ci]
    rng = np.random.default_rng(42 + ci)
    for _ in range(100):
        # createRandomImagePoint(): uniform inside the image with a margin
        pt = np.array([rng.uniform(0.1 * cam.w, 0.9 * cam.w), rng.uniform(0.1 * cam.h, 0.9 * cam.h)])
        ok, ray = oracle.cam_backproject(cam, pt)
        assert ok
        ray = ray / np.linalg.norm(ray) * (0.2 + 8 * (rng.uniform(-1, 1) + 1.0))
        st, pt2, J = oracle.cam_project(cam, ray, want_jac=True)
        assert st == 0
        assert np.linalg.norm(pt2 - pt) < 0.01
        dp = 1.0e-7
        Jn = np.zeros((2, 3))
        for d in range(3):
            e = np.zeros(3)
            e[d] = dp
            _, pp, _ = oracle.cam_project(cam, ray + e)
            _, pm, _ = oracle.cam_project(cam, ray - e)
            Jn[:, d] = (pp - pm) / (2 * dp)
        assert np.linalg.norm(Jn - J) < 1e-4


def test_projection_status_codes(oracle):
    cam = _test_cameras()[1]
    assert oracle.cam_project(cam, (0.0, 0.0, 1.0))[0] == 0       # Successful
    assert oracle.cam_project(cam, (0.0, 0.0, 1e-13))[0] == 4     # Invalid
    assert oracle.cam_project(cam, (5.0, 0.0, 1.0))[0] == 1       # OutsideImage
    assert oracle.cam_project(cam, (0.0, 0.0, -1.0))[0] == 3      # Behind


def test_fov_overlap_truth_table(oracle):
    """TestNCameraSystem.cpp:73-112: cameras 0/1 look the same way, camera 2 the opposite way
    (quaternion (w,x,y,z) = (0,0,1,0) = 180 deg about y)."""
    cams = _test_cameras()
    # the test cameras are subsampled 4x for speed; intrinsics scaled accordingly
    small = [synth.Camera(c.w // 4, c.h // 4, c.fu / 4, c.fv / 4, c.cu / 4, c.cv / 4, c.dist_type, c.d)
             for c in cams]
    eye = np.eye(3)
    flip = np.diag([-1.0, 1.0, -1.0])
    C_SC = [eye, eye, flip]
    expect = {(0, 1): True, (1, 0): True, (1, 2): False, (2, 1): False, (0, 2): False, (2, 0): False}
    for (seen_by, idx), want in expect.items():
        R = C_SC[seen_by].T @ C_SC[idx]  # rotation of T_Cother_C = T_SC[seen_by]^-1 * T_SC[idx]
        assert oracle.cam_overlap(small[idx], small[seen_by], R) == want, (seen_by, idx)
    # the product's host helper computes the same masks (host code, no GPU needed)
    from okvis2_amd import capi
    has, mask = capi.camera_overlap(small[1], small[0], eye, want_mask=True)
    has_o, mask_o = oracle.cam_overlap(small[1], small[0], eye, want_mask=True)
    assert has == has_o and np.array_equal(mask, mask_o) and mask.mean() > 0.5


def test_awareness_maps_product_host_vs_oracle(oracle):
    """Input preparation is host code on both sides and must agree bit for bit."""
    from okvis2_amd import capi
    for cam in (_test_cameras()[1], synth.Camera(160, 120, 80.0, 82.0, 81.0, 59.0, 2,
                                                 (-0.0369, -0.0089, 0.0089, -0.0037))):
        if cam.w > 200:
            cam = synth.Camera(188, 120, cam.fu / 4, cam.fv / 4, cam.cu / 4, cam.cv / 4,
                               cam.dist_type, cam.d)
        r0, j0 = oracle.awareness_maps(cam)
        r1, j1 = capi.build_awareness_maps(cam)
        assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32))
        assert np.array_equal(j0.view(np.uint32), j1.view(np.uint32))
        c = r0[cam.h // 2, cam.w // 2]
        assert abs(np.linalg.norm(c) - 1.0) < 1e-6 and c[2] > 0.99


def test_triangulate_fast_cases(oracle):
    sigma = 12.0 / 458.0 * 0.125
    p1, p2 = np.zeros(3), np.array([0.11, 0.0, 0.0])
    X = np.array([0.3, -0.2, 4.0])
    e1, e2 = (X - p1) / np.linalg.norm(X - p1), (X - p2) / np.linalg.norm(X - p2)
    hp, valid, par = oracle.triangulate_fast(p1, e1, p2, e2, sigma)
    assert valid and not par and np.allclose(hp[:3] / hp[3], X, atol=1e-9)
    # identical directions: A not invertible -> parallel, valid, point far along the ray
    hp, valid, par = oracle.triangulate_fast(p1, e1, p2, e1, sigma)
    assert par and valid and hp[3] == 1.0 and np.linalg.norm(hp[:3]) > 0.3
    # far point (1 km): rays intersect but nearly parallel -> flagged parallel, still valid
    Xf = np.array([10.0, 5.0, 1000.0])
    f1, f2 = (Xf - p1) / np.linalg.norm(Xf - p1), (Xf - p2) / np.linalg.norm(Xf - p2)
    hp, valid, par = oracle.triangulate_fast(p1, f1, p2, f2, sigma)
    assert valid and par
    # diverging rays (intersection behind the cameras): lambda < 0.01 -> parallel branch, and the
    # mid-ray check fails -> invalid
    d1 = np.array([-0.3, 0.0, 1.0]) / np.linalg.norm([-0.3, 0.0, 1.0])
    d2 = np.array([0.3, 0.0, 1.0]) / np.linalg.norm([0.3, 0.0, 1.0])
    hp, valid, par = oracle.triangulate_fast(p1, d1, p2, d2, sigma)
    assert par and not valid
    # skew rays that miss each other by much more than sigma -> invalid
    s2 = np.array([0.3, 0.4, 4.0]) - p2
    s2 /= np.linalg.norm(s2)
    hp, valid, par = oracle.triangulate_fast(p1, e1, p2, s2, sigma)
    assert not valid


def test_fixed_atan_within_one_ulp_of_libm(oracle):
    """The reference's equidistant model calls libm atan (EquidistantDistortion.hpp:98,138); the
    oracle (and, with the same sequence, the product: okvis2_amd/csrc/atan_fixed.h) evaluates a
    fixed IEEE operation sequence instead.  Pin: never more than 1 ulp from this host's libm."""
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-10), np.log(1e6), 20000)),
                        np.array([0.0, 0.4375, 0.6875, 1.1875, 2.4375, 1.0, 1e300]),
                        np.nextafter(np.array([0.4375, 0.6875, 1.1875, 2.4375]), 0.0)])
    x = np.concatenate([x, -x])
    got = oracle.atan_fixed(x)
    want = np.arctan(x)
    ulps = np.abs(got.view(np.int64) - want.view(np.int64))
    assert ulps.max() <= 1
    assert (ulps != 0).mean() < 0.03
    assert np.isnan(oracle.atan_fixed(np.nan)[0])


def _voc_tree():
    t = np.load(os.path.join(GOLDEN, "small_voc_tree.npz"))
    return t


def test_vocabulary_descent_on_the_real_tree(oracle):
    """The reference's DBoW2 vocabulary (resources/small_voc.yml.gz: k = 9, L = 3, 819 nodes, 729
    words; fixture tests/golden/small_voc_tree.npz from tools/make_voc_tree_fixture.py).  Known
    answers of the descent (DBoW2 transform with FBrisk::distance): a node descriptor fed as a
    feature descends to that very node (distance 0 wins at its level) and, below it, keeps
    following minimum-Hamming children; every leaf maps to its own word; a numpy restatement of the
    descent agrees on perturbed descriptors."""
    t = _voc_tree()
    parent, word, desc = t["parent"], t["word"], t["desc"]
    assert int(t["k"]) == 9 and int(t["L"]) == 3 and len(parent) == 820 and (word >= 0).sum() == 729
    cb, ci = oracle.voc_tree_arrays(parent)
    assert np.all(np.diff(cb)[word < 0] == 9) and np.all(np.diff(cb)[word >= 0] == 0)
    leaves = np.flatnonzero(word >= 0)
    w, n = oracle.voc_transform(desc[leaves], desc, cb, ci, word)
    # a leaf descriptor reaches a leaf at distance 0 from it: its own word unless an earlier sibling
    # chain is equally close (never the case for these cluster centres)
    assert np.array_equal(n, leaves) and np.array_equal(w, word[leaves])
    # numpy restatement on noisy features
    rng = np.random.default_rng(3)
    feats = desc[rng.integers(1, 820, 300)] ^ (rng.random((300, 48)) < 0.08).astype(np.uint8) * \
        rng.integers(1, 256, (300, 48), dtype=np.uint8)
    bits = np.unpackbits(desc, axis=1).astype(np.int32)
    fb = np.unpackbits(feats, axis=1).astype(np.int32)
    want = []
    for i in range(len(feats)):
        node = 0
        while cb[node + 1] > cb[node]:
            kids = ci[cb[node]:cb[node + 1]]
            d = (bits[kids] != fb[i]).sum(axis=1)
            node = int(kids[np.argmin(d)])  # argmin = first minimum
        want.append(node)
    w, n = oracle.voc_transform(feats, desc, cb, ci, word)
    assert np.array_equal(n, want) and np.array_equal(w, word[n])


def test_verify_place_running_minimum(oracle):
    """Frontend.cpp:330-355 on the real vocabulary descriptors: per landmark the first-lowest
    (descriptor, k) below the threshold."""
    d = voc()
    frame = d[:300]
    pool = np.concatenate([d[300:420], d[10:14]])          # the last landmark contains frame rows
    begin = np.concatenate([np.arange(0, 121, 3), [124]]).astype(np.int32)
    k_min, d_min = oracle.verify_place(pool, begin, frame, 60)
    bits = np.unpackbits(d, axis=1).astype(np.int32)
    fb, pb = bits[:300], np.unpackbits(pool, axis=1).astype(np.int32)
    for l in range(len(begin) - 1):
        dist = (pb[begin[l]:begin[l + 1], None, :] != fb[None, :, :]).sum(axis=2)  # [desc, k]
        m = dist.min()
        if m < 60:
            dd, kk = np.unravel_index(np.argmin(dist), dist.shape)  # first minimum in (desc, k) order
            assert d_min[l] == m and k_min[l] == kk
        else:
            assert d_min[l] == 60 and k_min[l] == 0
    assert d_min[-1] == 0 and k_min[-1] == 10


def test_fixed_acos_within_one_ulp_of_libm(oracle):
    x = np.concatenate([np.linspace(-1, 1, 20001), [0.5, -0.5, 1e-20, 0.9999999999, -0.9999999999]])
    got, want = oracle.acos_fixed(x), np.arccos(x)
    assert np.abs(got.view(np.int64) - want.view(np.int64)).max() <= 1
    assert np.isnan(oracle.acos_fixed(1.5)[0])


def test_descriptor_view_pooling_follows_the_reference_buffer_rules(oracle):
    """Frontend.cpp:1305-1354 written out for three small cases: the three-slot buffer writes at row
    `o`, crops to `o` rows, and drops a landmark whose single view was accepted."""
    cam = synth.euroc_config().cams[0]
    eye = np.eye(3).reshape(-1)
    T1 = (eye, np.zeros(3))
    poses = [(eye, np.array([0.05 * i, 0.0, 0.0])) for i in range(5)]
    hp = np.array([[0.0, 0.0, 5.0, 1.0]] * 3)
    quality = np.array([1.0, 1.0, 1.0])
    # landmark 0: one view -> o stays 0 -> skipped; landmark 1: two views -> one row kept (the SECOND
    # view overwrote row 0); landmark 2: four views
    obs_begin = np.array([0, 1, 3, 7], dtype=np.int32)
    obs_pose = np.array([1, 1, 2, 1, 2, 3, 4], dtype=np.int32)
    obs_bp = np.tile(np.array([0.0, 0.0, 1.0]), (7, 1))
    out = oracle.prepare_landmarks(hp, quality, obs_begin, obs_pose, obs_bp, poses, T1, cam, 20.0, False)
    assert out["status"][0] == 0 and out["n_desc"][0] == 0
    assert out["status"][1] in (1, 2) and out["n_desc"][1] == 1 and out["obs_rows"][1, 0] == 2
    # four views: slots 0,1,2 filled in turn (rows 0,0->1,1->2 by the write-at-o rule), the fourth
    # (largest view-point change = worst score) is not better than the worst slot
    assert out["n_desc"][2] == 2 and list(out["obs_rows"][2]) == [4, 5, -1]
    assert np.allclose(out["projection"][0], [cam.cu, cam.cv], atol=1.0)
    assert np.allclose(out["r_W"][2, 0], poses[2][1]) and np.allclose(out["r_W"][2, 1], poses[3][1])


def test_fp64_reduction_order_is_a_live_switch(oracle):
    """The 3-term FP64 sums of the gate chain run in Eigen's unrolled-redux order x0 + (x1 + x2) by default
    (stereo_triangulation.cpp:62-76 evaluates them through fixed-size Eigen vectors) and left to right on request:
    same decisions on this content, different last bits of the triangulated points."""
    import real_image_cases as RC
    fx = RC.load()
    (k0, d0, b0, v0), (k1, d1, b1, v1) = RC.stereo_sides(oracle, fx["image"])
    out = {}
    for tree in (True, False):
        oracle.set_reduction(tree)
        try:
            out[tree] = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, *RC.stereo_geometry())
        finally:
            oracle.set_reduction(True)
    a, b = out[True], out[False]
    assert np.array_equal(a["k1"], b["k1"]) and np.array_equal(a["initialisable"], b["initialisable"])
    hit = a["k1"] >= 0
    differ = np.any(a["hp_W"][hit].view(np.uint64) != b["hp_W"][hit].view(np.uint64), axis=1)
    assert 0 < differ.sum() < hit.sum()
    assert np.allclose(a["hp_W"][hit], b["hp_W"][hit], rtol=1e-12, atol=0)
    assert np.array_equal(a["hp_W"].view(np.uint64), fx["stereo/match"]["hp_W"].view(np.uint64))  # committed = default
