"""CPU tests on the reference's only real camera image (tests/golden/real_image.npz =
okvis_multisensor_processing/test/testImage.jpg decoded in the build container, see
tools/make_real_image_fixture.py).

 * the oracle reproduces the committed vectors on every crop (guards the oracle against drift; the
   -m gpu twin of this file, test_gpu_real_image.py, holds the HIP path to the same vectors);
 * the oracle's descriptors of REAL content are compared with the only real BRISK2 outputs the
   reference tree holds, the 819 node descriptors of resources/small_voc.yml.gz: nearest-word
   Hamming distances through the real 9^3 tree and the bit-to-bit correlation structure.  The
   outcome is recorded as it is (VERDICT r3 item 1b asked for the numbers "either way"): the
   built-in sampling pattern does NOT reproduce BRISK2's pair set / bit order.
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


def test_oracle_descriptors_against_the_real_brisk2_vocabulary(oracle):
    """What the 819 real BRISK2 descriptors say about the oracle's descriptor arithmetic.

    Two statistics, each with its "same extractor" and "unrelated bits" reference points:
     (1) distance from a descriptor to the nearest vocabulary node (brute force and through the
         real tree's descent): descriptors of one extractor on unrelated scenes lie much closer
         to each other than random bit strings do, because bits that share a sample point are
         correlated;
     (2) the correlation of the two 384x384 bit-correlation matrices: the same pair set in the
         same bit order gives a value near 1, an unrelated order a value near 0.
    Measured: (1) 149.6 (random 163.5; the oracle's descriptors of OTHER images among
    themselves 105); (2) 0.09.  And the vocabulary's bit 383 is live (density 0.46) where the
    oracle's 383 short pairs leave it zero.  So the restated pattern is distinguishable from
    real BRISK2: the pattern is data (okvfe_set_pattern), and bit-compatibility with stored BRISK2
    descriptors (maps, vocabularies) needs the real pattern installed."""
    voc = np.fromfile(os.path.join(GOLDEN, "small_voc_desc.bin"), dtype=np.uint8).reshape(-1, 48)
    fx = RC.load()
    full = fx["image"]
    k, d = oracle.detect_describe(full, 10.0, 0, 5, 4000, oracle.MODE_GRADIENT)
    assert len(k) > 1500
    # committed camera-aware descriptors of the crops: a second, independent sample
    d2 = np.concatenate([fx[f"{c.name}/desc_aware"] for c in RC.CASES])
    rng = np.random.default_rng(0)
    rnd = rng.integers(0, 256, (2000, 48), dtype=np.uint8)

    near_oracle = _hamming(d, voc).min(axis=1).mean()
    near_aware = _hamming(d2, voc).min(axis=1).mean()
    near_random = _hamming(rnd, voc).min(axis=1).mean()
    hv = _hamming(voc, voc)
    np.fill_diagonal(hv, 999)
    near_voc = hv.min(axis=1).mean()
    half = len(d) // 2
    near_self = _hamming(d[:half], d[half:]).min(axis=1).mean()
    # recorded values (tolerances = a few sigma of the sample means)
    assert abs(near_voc - 47.7) < 0.5
    assert abs(near_random - 163.5) < 1.5
    assert 140.0 < near_oracle < 158.0, near_oracle
    assert 140.0 < near_aware < 158.0, near_aware
    assert near_self < 115.0, near_self

    # the tree descent of the reference's vocabulary reaches words at the same (random-like) distance
    tree = np.load(os.path.join(GOLDEN, "small_voc_tree.npz"))
    begin, index = oracle.voc_tree_arrays(tree["parent"])
    words, nodes = oracle.voc_transform(d, tree["desc"], begin, index, tree["word"])
    assert words.min() >= 0 and words.max() < 729
    dist_word = np.array([oracle.popcnt_xor(d[i], tree["desc"][nodes[i]]) for i in range(0, len(d), 7)])
    assert 150.0 < dist_word.mean() < 185.0, dist_word.mean()

    # (2) bit-correlation structure
    bo, bv = _bits(d), _bits(voc)
    assert bo[:, 383].max() == 0.0          # the oracle's 383 pairs never set the last bit ...
    assert 0.40 < bv[:, 383].mean() < 0.52  # ... real BRISK2 does
    dens = bv.mean(axis=0)
    assert 0.44 < dens.min() and dens.max() < 0.56
    co = np.corrcoef(bo[:, :383].T)
    cv = np.corrcoef(bv[:, :383].T)
    iu = np.triu_indices(383, 1)
    r = np.corrcoef(co[iu], cv[iu])[0, 1]
    assert abs(r) < 0.2, r                  # measured 0.087: unrelated pair order
    # both have the banded structure of "consecutive bits share a sample point"
    assert np.mean(np.abs(np.diagonal(co, 1))) > 0.3 and np.mean(np.abs(np.diagonal(cv, 1))) > 0.3
