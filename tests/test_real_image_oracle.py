"""CPU tests on the reference's only real camera image (tests/golden/real_image.npz =
okvis_multisensor_processing/test/testImage.jpg decoded in the build container, see
tools/make_real_image_fixture.py).

 * the oracle reproduces the committed vectors on every crop (guards the oracle against drift; the
   -m gpu twin of this file, test_gpu_real_image.py, holds the HIP path to the same vectors);
 * the oracle's descriptors of REAL content are compared with the only real BRISK2 outputs the
   reference tree holds, the 819 node descriptors of resources/small_voc.yml.gz: nearest-word
   Hamming distances through the real 9^3 tree, the bit-to-bit correlation structure, and the exact
   transitivity check of the pair table (the table was recovered from these descriptors in round 5,
   tools/pattern/README.md).
"""
import hashlib
import os

import numpy as np

import real_image_cases as RC

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _sha(k, d):
    return np.frombuffer(hashlib.sha256(k.tobytes() + d.tobytes()).digest(), dtype=np.uint8)


def test_oracle_reproduces_the_committed_vectors(oracle):
    fx = RC.load()
    full = fx["image"]
    assert full.shape == (960, 1280) and full.dtype == np.uint8
    assert abs(float(np.mean(full == 255)) - 0.278) < 0.001  # saturated board: plateaus
    for case in RC.CASES:
        img = RC.crop(full, case)
        kd = oracle.detect(img, case.radius, 0, case.thr, case.max_kpts)
        assert np.array_equal(kd.view(np.uint8), fx[f"{case.name}/kp_detect"].view(np.uint8)), case.name
        assert len(oracle.nms(oracle.harris_score(img), case.thr)) == int(fx[f"{case.name}/n_nms"])
        for mode, name in ((oracle.MODE_UPRIGHT, "upright"), (oracle.MODE_GRADIENT, "gradient")):
            k, d = oracle.describe(img, kd, mode)
            assert np.array_equal(_sha(k, d), fx[f"{case.name}/sha_{name}"]), (case.name, name)
        rays, jac = oracle.awareness_maps(case.cam)
        k, d = oracle.describe(img, kd, oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(case.cam.fu),
                               RC.GRAVITY)
        assert np.array_equal(k.view(np.uint8), fx[f"{case.name}/kp_aware"].view(np.uint8))
        assert np.array_equal(d, fx[f"{case.name}/desc_aware"])
        bp, bv = oracle.backproject_keypoints(case.cam, k)
        assert np.array_equal(bp.view(np.uint64), fx[f"{case.name}/bp"].view(np.uint64))
        assert np.array_equal(bv, fx[f"{case.name}/bpv"])


def test_oracle_stereo_on_the_shifted_real_pair(oracle):
    fx = RC.load()
    (k0, d0, b0, v0), (k1, d1, b1, v1) = RC.stereo_sides(oracle, fx["image"])
    assert np.array_equal(k1.view(np.uint8), fx["stereo/kp1"].view(np.uint8))
    assert np.array_equal(d1, fx["stereo/desc1"])
    m = oracle.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, *RC.stereo_geometry())
    want = fx["stereo/match"]
    for f in ("k1", "dist", "initialisable"):
        assert np.array_equal(m[f], want[f]), f
    assert np.array_equal(m["hp_W"].view(np.uint64), want["hp_W"].view(np.uint64))
    ok = m["k1"] >= 0
    assert ok.sum() >= 100
    # true correspondences: the matched partner sits STEREO_DISPARITY px to the left
    dx = k0["x"][ok] - k1["x"][m["k1"][ok]]
    assert np.mean(np.abs(dx - RC.STEREO_DISPARITY) < 1.5) > 0.9


def _bits(d):
    return np.unpackbits(np.ascontiguousarray(d), axis=1, bitorder="little").astype(np.float64)


def _hamming(A, B):
    a, b = _bits(A), _bits(B)
    return (a @ (1 - b).T + (1 - a) @ b.T).astype(np.int32)


def _voc_statistics(oracle, d):
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    bo, bv = _bits(d), _bits(voc)
    co, cv = np.corrcoef(bo.T), np.corrcoef(bv.T)
    iu = np.triu_indices(384, 1)
    return _hamming(d, voc).min(axis=1).mean(), np.corrcoef(co[iu], cv[iu])[0, 1], bo, bv


def test_oracle_descriptors_against_the_real_brisk2_vocabulary(oracle):
    """What the 819 real BRISK2 descriptors (resources/small_voc.yml.gz, FBrisk.hpp:35) say about the oracle's
    descriptor: the pair table and bit order of the built-in pattern were recovered from them
    (tools/pattern/README.md), and the oracle's descriptors of the only real image in the tree must now look
    like BRISK2 -- where rounds 1-4's restated table did not (recorded then: nearest node 149.6, matrix
    correlation 0.09, bit 383 dead).

     (1) mean distance to the nearest vocabulary node: random strings 163.5, vocabulary nodes among themselves
         47.7, oracle descriptors of another scene 114 (a blurred copy: 90);
     (2) correlation of the two 384 x 384 bit-correlation matrices: 0.63 on this image (the statistic also
         depends on image content and on the smoothing width, which stays an assumption: the forward simulator
         of tools/pattern reaches 0.87 when its descriptors are clustered like a vocabulary and smoothed with one published sigma);
     (3) all 384 bits live with densities near one half, as in the vocabulary."""
    fx = RC.load()
    full = fx["image"]
    # upright = the gravity-aligned extraction of an upright camera, which is how the vocabulary's bits are
    # distributed (densities 0.44..0.56); the gradient orientation skews the densities (0.28..0.72)
    k, d = oracle.detect_describe(full, 10.0, 0, 5, 4000, oracle.MODE_UPRIGHT)
    assert len(k) > 1500
    near, r, bo, bv = _voc_statistics(oracle, d)
    rng = np.random.default_rng(0)
    rnd = rng.integers(0, 256, (2000, 48), dtype=np.uint8)
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    near_random = _hamming(rnd, voc).min(axis=1).mean()
    hv = _hamming(voc, voc)
    np.fill_diagonal(hv, 999)
    assert abs(hv.min(axis=1).mean() - 47.7) < 0.5
    assert abs(near_random - 163.5) < 1.5
    assert near < 120.0, near            # measured 114.7 (rounds 1-4: 149.6)
    assert r > 0.55, r                   # measured 0.630 (rounds 1-4: 0.087)
    assert 0.40 < bo[:, 383].mean() < 0.52 and 0.40 < bv[:, 383].mean() < 0.52  # the last bit is live in both
    dens = bo.mean(axis=0)
    assert 0.40 < dens.min() and dens.max() < 0.60, (dens.min(), dens.max())
    # the committed camera-aware descriptors of the crops: an independent sample, other orientation rule
    d2 = np.concatenate([fx[f"{c.name}/desc_aware"] for c in RC.CASES])
    near2, r2, _, _ = _voc_statistics(oracle, d2)
    assert near2 < 120.0 and r2 > 0.4, (near2, r2)   # measured 111.9, 0.477 (checkerboard-heavy crops, tilted gravity)

    # the tree descent of the reference's vocabulary lands near the brute-force nearest node
    tree = np.load(os.path.join(GOLDEN, "small_voc_tree.npz"))
    begin, index = oracle.voc_tree_arrays(tree["parent"])
    words, nodes = oracle.voc_transform(d, tree["desc"], begin, index, tree["word"])
    assert words.min() >= 0 and words.max() < 729
    dist_word = np.array([oracle.popcnt_xor(d[i], tree["desc"][nodes[i]]) for i in range(0, len(d), 7)])
    assert dist_word.mean() < 150.0, dist_word.mean()   # rounds 1-4: 150..185 (random-like)


def test_pair_table_is_consistent_with_every_vocabulary_descriptor(oracle):
    """The exact part of the pin: when the comparisons among three sample points are all bits of the
    descriptor, the cyclic outcome is impossible.  Over the 740 point triangles the built-in table implies,
    the 819 node descriptors (bit-wise cluster majorities, so not strictly transitive) show 7 cyclic outcomes
    in total; a table with one pair or one position wrong shows hundreds (three unrelated bits: ~205 each)."""
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    B = np.unpackbits(voc, axis=1, bitorder="little").astype(bool)
    p = oracle.pattern()
    assert p.n_points == 66 and p.n_short == 384
    pairs = [(p.short_i[b], p.short_j[b]) for b in range(384)]
    assert all(i > j for i, j in pairs) and len(set(pairs)) == 384
    assert pairs == sorted(pairs)  # the generator's loop order: for i: for j < i
    bit = {q: b for b, q in enumerate(pairs)}
    below = {}
    for i, j in pairs:
        below.setdefault(i, []).append(j)
    total, n_tri, worst = 0, 0, 0
    for (z, y), c in bit.items():
        for x in below.get(y, ()):
            b = bit.get((z, x))
            if b is None:
                continue
            a = bit[(y, x)]   # a = [y > x], b = [z > x], c = [z > y]: (1,0,1) and (0,1,0) are cycles
            v = int(np.sum((B[:, a] & ~B[:, b] & B[:, c]) | (~B[:, a] & B[:, b] & ~B[:, c])))
            total += v
            n_tri += 1
            worst = max(worst, v)
    assert n_tri == 740 and total <= 7 and worst <= 2, (n_tri, total, worst)
    # and a shifted table (every pair one position late) is rejected by the same count
    shifted = pairs[-1:] + pairs[:-1]
    sb = {q: b for b, q in enumerate(shifted)}
    bad = 0
    for (z, y), c in sb.items():
        for x in below.get(y, ()):
            if (z, x) in sb:
                a, b = sb[(y, x)], sb[(z, x)]
                bad += int(np.sum((B[:, a] & ~B[:, b] & B[:, c]) | (~B[:, a] & B[:, b] & ~B[:, c])))
    assert bad > 20000, bad


def test_vocabulary_bit_densities_follow_the_vertical_pair_component(oracle):
    """Polarity and up/down orientation of the recovered pattern, which no transitivity argument can see: with
    bit b = [v_i > v_j] (published convention) and the pattern's +y axis along the extraction direction = gravity
    (orc_describe.c, camera-aware mode), the vocabulary's bit densities fall with the vertical component of
    p_i - p_j (r = -0.62; the fit explains 42 % of the density variance, whose sampling noise alone is 0.0175 of
    0.0202): the LOWER sample of a pair is darker on average -- scenes are lit from above.  The opposite polarity
    (or a mirrored pattern) would need scenes lit from below; both flipped at once is the one alternative this
    statistic cannot exclude."""
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    dens = _bits(voc).mean(axis=0) - 0.5
    p = oracle.pattern()
    i = np.array(p.short_i[:384]); j = np.array(p.short_j[:384])
    px, py = np.array(p.px[:p.n_points]), np.array(p.py[:p.n_points])
    dx, dy = px[i] - px[j], py[i] - py[j]
    r_y = np.corrcoef(dy, dens)[0, 1]
    r_x = np.corrcoef(dx, dens)[0, 1]
    assert r_y < -0.5, r_y          # measured -0.623
    assert abs(r_x) < 0.3, r_x      # measured -0.171: the gradient is (mostly) vertical


def _one_over_f_image(h, w, seed):
    rng = np.random.default_rng(seed)
    fy, fx = np.fft.fftfreq(h)[:, None], np.fft.fftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1
    img = np.real(np.fft.ifft2((rng.standard_normal((h, w)) + 1j * rng.standard_normal((h, w))) / f))
    img = (img - img.mean()) / img.std()
    return np.clip(128 + 50 * img, 0, 255).astype(np.uint8)


def _k_majority_tree(bits, k=9, levels=3, seed=1, iters=8):
    """node descriptors of a DBoW2-style vocabulary: hierarchical k-majority clustering (FBrisk::meanValue rule)"""
    rng = np.random.default_rng(seed)
    nodes = []

    def split(idx, lev):
        if lev == levels or len(idx) < k:
            return
        X = bits[idx].astype(np.float32)
        cent = X[rng.choice(len(idx), k, replace=False)]
        for _ in range(iters):
            lab = (X @ (1 - cent).T + (1 - X) @ cent.T).argmin(1)
            for c in range(k):
                if (lab == c).any():
                    cent[c] = X[lab == c].mean(0) >= 0.5
        for c in range(k):
            nodes.append(cent[c].astype(np.uint8))
            split(idx[lab == c], lev + 1)

    split(np.arange(len(bits)), 0)
    return np.array(nodes)


def test_a_vocabulary_of_oracle_descriptors_has_the_real_vocabularys_bit_structure(oracle):
    """Like with like: the reference's 819 descriptors are NODES of a 9^3 k-majority tree, i.e. denoised descriptors.
    The oracle's descriptors of natural-statistics (1/f) images, clustered the same way, give a bit-correlation
    matrix that correlates at 0.81 with the real vocabulary's (raw descriptors: 0.75; rounds 1-4's pair table: 0.09
    raw) -- the acceptance bar VERDICT r4 set for the recovered pattern was 0.8."""
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    ds = []
    for s in range(4):
        _, d = oracle.detect_describe(_one_over_f_image(960, 1280, s), 8.0, 0, 5, 6000, oracle.MODE_UPRIGHT)
        ds.append(d)
    bits = np.unpackbits(np.concatenate(ds), axis=1, bitorder="little")
    assert len(bits) > 15000
    nodes = _k_majority_tree(bits)
    assert len(nodes) == 819
    iu = np.triu_indices(384, 1)
    cv = np.corrcoef(_bits(voc).T)[iu]
    r_nodes = np.corrcoef(np.corrcoef(nodes.T.astype(np.float64))[iu], cv)[0, 1]
    r_raw = np.corrcoef(np.corrcoef(bits.T.astype(np.float64))[iu], cv)[0, 1]
    assert r_nodes > 0.78, r_nodes   # measured 0.814
    assert r_raw > 0.70, r_raw       # measured 0.747


def test_box_width_statistics_on_the_real_image_are_recorded(oracle, monkeypatch):
    """The one open parameter of the descriptor with evidence on both sides is the smoothing WIDTH (okvfe_config.box_scale,
    ABI 7).  Recorded here, on the reference's own camera image, so that the numbers cannot drift unnoticed
    (tools/pattern/README.md holds the table and the decision rule): for boxes 1.0 / 1.3 / 1.73 / 2.0 / 2.4 x the
    published half-side,
      * mean Hamming distance to the nearest vocabulary node:           114.7 / 106.8 / 96.7 / 91.6 / 86.1
      * correlation of the bit-correlation matrices, raw descriptors:   0.630 / 0.689 / 0.767 / 0.794 / 0.806
      * the same with the image's descriptors (uniformity radius 3: 3.1 k of them) clustered into a k-majority tree
        like a vocabulary:                                              0.670 / 0.736 / 0.781 / 0.825 / 0.810
    All monotone towards wide up to 2.0 -- and all BIASED towards wide: the vocabulary's nodes are cluster centres,
    i.e. denoised descriptors, and wider boxes denoise.  They cannot choose the width alone; nothing here favours 1.0."""
    import ctypes as C
    import math
    fx = RC.load()
    full = fx["image"]
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    iu = np.triu_indices(384, 1)
    cv = np.corrcoef(_bits(voc).T)[iu]
    base = oracle.pattern()
    want = {1.0: (114.7, 0.630, 0.670), 1.3: (106.8, 0.689, 0.736), 1.73: (96.7, 0.767, 0.781), 2.0: (91.6, 0.794, 0.825),
            2.4: (86.1, 0.806, 0.810)}
    got = {}
    for m in want:
        p = type(base)()
        C.memmove(C.byref(p), C.byref(base), C.sizeof(p))
        f, reach = float(np.float32(m)), 0.0
        for i in range(p.n_points):  # = scale_pattern_boxes (host_tables.cpp): half-side in double x the float factor
            p.sigma_half[i] = np.float32(float(base.sigma_half[i]) * f)
            reach = max(reach, math.hypot(p.px[i], p.py[i]) + p.sigma_half[i])
        p.border = int(math.ceil(reach)) + 1
        monkeypatch.setattr(oracle, "pattern", lambda p=p: p)
        k, d = oracle.detect_describe(full, 10.0, 0, 5, 4000, oracle.MODE_UPRIGHT)
        near, r, _, _ = _voc_statistics(oracle, d)
        _, d2 = oracle.detect_describe(full, 3.0, 0, 5, 30000, oracle.MODE_UPRIGHT)
        bits = np.unpackbits(d2, axis=1, bitorder="little")
        nodes = _k_majority_tree(bits)
        r_nodes = np.corrcoef(np.corrcoef(nodes.T.astype(np.float64))[iu], cv)[0, 1]
        got[m] = (near, r, r_nodes)
        assert abs(near - want[m][0]) < 0.6 and abs(r - want[m][1]) < 0.01 and abs(r_nodes - want[m][2]) < 0.03, (m, got[m])
    ms = sorted(want)
    assert all(got[a][0] > got[b][0] and got[a][1] < got[b][1] for a, b in zip(ms, ms[1:]))
