"""Host-side formats either side of the path (no GPU needed): the map-file keypoint records of
okvis::Component::save/load (okvis_ceres/src/Component.cpp:235-266, 405-460) and
DBoW2::FBrisk::meanValue (okvis_frontend/src/FBrisk.cpp:25-58), against pure-Python
restatements of the reference code (the oracle for these tiny cases)."""
import os

import numpy as np
import pytest

from okvis2_amd import capi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz")


def ref_format(state_id, cam, kps, desc):
    """Component.cpp:450-459 -- operator<<(float) with std::setprecision(17) == %.17g of the
    widened value; descriptor as 96 lower-case hex digits."""
    out = []
    for k, d in zip(kps, desc):
        out.append("FRAME:KEYPOINT %d %d %s %s %s BRISK2 %s\n" % (
            state_id, cam, "%.17g" % float(k["x"]), "%.17g" % float(k["y"]), "%.17g" % float(k["size"]),
            "".join("%02x" % int(b) for b in d)))
    return "".join(out).encode()


def ref_mean(descs):
    """FBrisk.cpp:25-58."""
    s = len(descs) // 2
    bits = np.unpackbits(np.asarray(descs, dtype=np.uint8), axis=1, bitorder="little").sum(0)
    return np.packbits((bits > s).astype(np.uint8), bitorder="little")


def test_keypoint_records_roundtrip_and_reference_text():
    g = np.load(GOLDEN)
    kps, desc = g["kp_aware_0"], g["desc_aware_0"]
    text = capi.format_keypoint_lines(123456789012, 1, kps, desc)
    assert text == ref_format(123456789012, 1, kps, desc)
    first = text.split(b"\n")[0].split()
    assert first[0] == b"FRAME:KEYPOINT" and first[6] == b"BRISK2" and len(first[7]) == 96
    sid, cam, k2, d2, used = capi.parse_keypoint_lines(text)
    assert (sid, cam, used) == (123456789012, 1, len(text))
    # x, y, size survive exactly (17 significant digits); the other cv::KeyPoint fields are not
    # part of the record and come back as cv::KeyPoint defaults, as in Component::load
    for f in ("x", "y", "size"):
        assert np.array_equal(k2[f].view(np.uint32), kps[f].view(np.uint32))
    assert np.all(k2["angle"] == -1) and np.all(k2["response"] == 0) and np.all(k2["octave"] == 0)
    assert np.array_equal(d2, desc)


def test_keypoint_record_blocks_and_errors():
    g = np.load(GOLDEN)
    a = capi.format_keypoint_lines(7, 0, g["kp_aware_0"][:5], g["desc_aware_0"][:5])
    b = capi.format_keypoint_lines(7, 1, g["kp_aware_1"][:3], g["desc_aware_1"][:3])
    tail = b"FRAME 8 0 1 0 0 0 0 0 0 1 123\n"
    blob = a + b + tail
    sid, cam, k, d, used = capi.parse_keypoint_lines(blob)
    assert (sid, cam, len(k), used) == (7, 0, 5, len(a))          # stops at the camera change
    sid, cam, k, d, used2 = capi.parse_keypoint_lines(blob[used:])
    assert (sid, cam, len(k), used2) == (7, 1, 3, len(b))          # stops at the foreign line
    assert capi.parse_keypoint_lines(tail)[2].size == 0
    assert capi.format_keypoint_lines(1, 0, g["kp_aware_0"][:0], g["desc_aware_0"][:0]) == b""
    with pytest.raises(capi.OkvfeError) as e:                      # "only BRISK 2"
        capi.parse_keypoint_lines(a.replace(b"BRISK2", b"ORB"))
    assert e.value.status == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.OkvfeError):                           # truncated descriptor
        capi.parse_keypoint_lines(a[:-10] + b"\n")
    with pytest.raises(capi.OkvfeError) as e:                      # caller capacity
        capi.parse_keypoint_lines(a, cap=2)
    assert e.value.status == capi.ERR_CAPACITY


def test_fbrisk_mean():
    voc = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "small_voc_desc.bin"),
                      dtype=np.uint8).reshape(-1, 48)
    for n in (1, 2, 3, 8, 9, 100, 819):
        assert np.array_equal(capi.fbrisk_mean(voc[:n]), ref_mean(voc[:n])), n
    # tie (exactly half) is NOT a majority: strict '>'
    two = np.stack([np.zeros(48, np.uint8), np.full(48, 255, np.uint8)])
    assert not capi.fbrisk_mean(two).any()
    assert not capi.fbrisk_mean(voc[:0]).any()
