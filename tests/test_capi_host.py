"""No-GPU checks of the drop-in boundary: libokvfe.so loads, exports every symbol that
include/okvfe.h declares, its struct layouts match the ctypes mirror, and -- without a GPU --
the compute entry points fail loudly instead of falling back to any CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from okvis2_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "okvfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(okvfe_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    lib = C.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python mirror knows all of them
    assert sorted(capi.EXPORTS) == names


def test_abi_version_and_struct_layouts():
    lib = capi.lib()
    assert lib.okvfe_abi_version() == capi.ABI_VERSION
    assert C.sizeof(capi.Config) == 16 * 4
    assert capi.KEYPOINT_DTYPE.itemsize == 28            # cv::KeyPoint: 5 floats + 2 ints
    assert capi.STEREO_MATCH_DTYPE.itemsize == 48
    assert C.sizeof(capi.Pose) == 96 and C.sizeof(capi.Camera) == 80
    assert C.sizeof(capi.StereoPair) == 8 + 2 * 96 + 16


def test_no_silent_cpu_fallback():
    """On a machine without a usable gfx950 device okvfe_create must fail with
    OKVFE_ERR_NO_DEVICE (the GPU box runs the -m gpu tests instead)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(capi.OkvfeError) as e:
        capi.Frontend(752, 480, 38.0, 0, 150, 700)
    assert e.value.status == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value)


def test_argument_validation_precedes_device_probe():
    for kw, status in ((dict(octaves=5), capi.ERR_UNSUPPORTED),
                       (dict(absolute_threshold=0), capi.ERR_INVALID_ARGUMENT),
                       (dict(max_keypoints=5000), capi.ERR_INVALID_ARGUMENT),
                       (dict(width=32), capi.ERR_INVALID_ARGUMENT)):
        args = dict(width=752, height=480, uniformity_radius=38.0, octaves=0, absolute_threshold=150,
                    max_keypoints=700)
        args.update(kw)
        with pytest.raises(capi.OkvfeError) as e:
            capi.Frontend(**args)
        assert e.value.status == status, kw


def test_product_never_imports_the_oracle():
    """The package and the native sources must not reference oracle/ (checked textually)."""
    pkg = os.path.join(ROOT, "okvis2_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        if "build" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|okvfe_oracle|liboracle|orc_[a-z]+\(", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_synthetic_inputs_are_deterministic():
    a = synth.corners_image(752, 480, 5)
    b = synth.corners_image(752, 480, 5)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.shape == (480, 752)
    assert not np.array_equal(a, synth.corners_image(752, 480, 6))
    L, R, d = synth.stereo_pair(752, 480, 9)
    assert 4 <= d <= 40
    # the right image is the left content shifted by the disparity (up to +-2 sensor noise)
    diff = np.abs(L[:, d:].astype(int) - R[:, :752 - d].astype(int))
    assert diff.max() <= 2
    for mk in (synth.euroc_config, synth.mono640_config, synth.tumvi1024_config, synth.hilti_config):
        cfg = mk()
        assert all(c.w == cfg.w and c.h == cfg.h for c in cfg.cams)


def test_bow_vector_host_helper_against_the_oracle_and_a_plain_restatement(oracle):
    """okvfe_bow_vector (DBoW2 transform's weighting / normalisation half) on the reference's real
    vocabulary weights: every weighting mode, with and without L1 normalisation, against the C
    oracle and against a dictionary restatement written here."""
    import os
    voc = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_voc_tree.npz"))
    word, weight = voc["word"], voc["weight"]
    n_words = int(word.max()) + 1
    ww = np.zeros(n_words)
    ww[word[word >= 0]] = weight[word >= 0]
    assert int(voc["weighting"]) == 0 and int(voc["scoring"]) == 0   # TF_IDF, L1_NORM
    ww[5] = 0.0  # a stopped word: skipped
    rng = np.random.default_rng(4)
    for n_feat in (0, 1, 37, 700, 3000):
        ids_in = rng.integers(0, n_words, n_feat)
        for weighting in (0, 1, 2, 3):
            for norm in (True, False):
                got = capi.bow_vector(ids_in, ww, weighting, norm)
                ref = oracle.bow_vector(ids_in, ww, weighting, norm)
                assert np.array_equal(got[0], ref[0])
                assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64))
                acc = {}
                for x in ids_in:
                    x = int(x)
                    if not ww[x] > 0:
                        continue
                    if x not in acc:
                        acc[x] = ww[x]
                    elif weighting in (0, 1):
                        acc[x] = acc[x] + ww[x]
                keys = sorted(acc)
                vals = np.array([acc[k] for k in keys], dtype=np.float64)
                if norm:
                    s = 0.0
                    for t in vals:
                        s = s + abs(t)
                    if s > 0:
                        vals = vals / s
                elif weighting in (0, 1) and len(vals):
                    vals = vals / float(len(vals))
                assert np.array_equal(np.array(keys, dtype=np.int32), got[0])
                assert np.array_equal(vals.view(np.uint64), got[1].view(np.uint64))
                assert 5 not in got[0]
    with pytest.raises(capi.OkvfeError):
        capi.bow_vector([n_words], ww)


def test_product_library_reads_no_environment():
    """the shipping library decides no code path from the environment -- it does
    not even import getenv; the lab build (A/B knobs of LAB_NOTES.md) does."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    und = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(root, "okvis2_amd", "libokvfe.so")],
                         capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    lab = os.path.join(root, "okvis2_amd", "libokvfe_lab.so")
    if os.path.exists(lab):
        und = subprocess.run(["nm", "-D", "--undefined-only", lab], capture_output=True, text=True, check=True).stdout
        assert "getenv" in und
