"""CPU: the scale-invariant extractor of the oracle (published BRISK scale ladder) and the product's
host-side scale index (okvfe_scale_index, the formula the device thresholds are bisected on)."""
import math

import numpy as np

from okvis2_amd import capi, synth


def test_scale_index_known_values(oracle):
    # max(int(64 / lb(30) * lb(size / 7.2) + 0.5), 0), at most 63
    assert oracle.scale_index(7.2) == 0
    assert oracle.scale_index(1.45 * 12.0) == 17  # the fixed-scale extractor's index
    assert oracle.scale_index(12.0) == 10
    assert oracle.scale_index(24.0) == 23
    assert oracle.scale_index(1.0e9) == 63
    for bad in (0.0, -3.0, float("nan")):
        assert oracle.scale_index(bad) == 0
    lb = math.log(30.0, 2.0)
    for size in np.linspace(5.0, 300.0, 400):
        want = min(max(int(64.0 / lb * math.log(float(np.float32(size)) / 7.2, 2.0) + 0.5), 0), 63)
        assert oracle.scale_index(size) == want


def test_product_scale_index_equals_oracle(oracle):
    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.uniform(0.0, 400.0, 4000), [0.0, -1.0, 7.2, 12.0, 17.4, 24.0, 1e9]]).astype(np.float32)
    # both sides of every index boundary
    lb = math.log(30.0, 2.0)
    edges = np.array([7.2 * 2.0 ** ((i - 0.5) * lb / 64.0) for i in range(1, 64)], dtype=np.float32)
    for e in edges:
        sizes = np.append(sizes, [np.nextafter(e, np.float32(0)), e, np.nextafter(e, np.float32(1e9))])
    idx = [capi.scale_index(s) for s in sizes]
    assert idx == [oracle.scale_index(s) for s in sizes]
    assert min(idx) == 0 and max(idx) == 63


def test_scaled_pattern(oracle):
    base = oracle.pattern()
    same = oracle.pattern_scaled(17)
    assert bytes(same) == bytes(base)
    lb = math.log(30.0, 2.0)
    borders = []
    for i in (0, 10, 23, 40, 63):
        p = oracle.pattern_scaled(i)
        rel = 2.0 ** ((i - 17) * lb / 64.0)
        assert np.allclose(np.array(p.px[:60]), np.array(base.px[:60]) * rel, rtol=1e-6, atol=1e-6)
        assert np.allclose(np.array(p.sigma_half[:60]), np.array(base.sigma_half[:60]) * rel, rtol=1e-6)
        reach = max(math.hypot(p.px[k], p.py[k]) + p.sigma_half[k] for k in range(60))
        assert p.border >= reach
        assert bytes(p.short_i) == bytes(base.short_i) and bytes(p.long_wdx) == bytes(base.long_wdx)
        borders.append(p.border)
    assert borders == sorted(borders)


def test_basic_size_reproduces_fixed_scale_descriptors(oracle):
    cfg = synth.euroc_config()
    img = synth.corners_image(cfg.w, cfg.h, 7)
    kps = oracle.detect(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts)
    fixed_k, fixed_d = oracle.describe(img, kps, oracle.MODE_GRADIENT)
    kps17 = kps.copy()
    kps17["size"] = np.float32(1.45 * 12.0)
    k, d = oracle.describe(img, kps17, oracle.MODE_GRADIENT, scale_invariant=True)
    assert len(k) == len(fixed_k) and np.array_equal(d, fixed_d)
    # at size 12 the pattern is smaller: more keypoints near the rim survive, descriptors differ
    k12, d12 = oracle.describe(img, kps, oracle.MODE_GRADIENT, scale_invariant=True)
    assert len(k12) >= len(fixed_k)
    assert not np.array_equal(d12[: len(fixed_d)], fixed_d)
