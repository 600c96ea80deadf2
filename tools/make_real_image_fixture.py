#!/usr/bin/env python3
"""Generates tests/golden/real_image.npz from the ONLY real camera image in the reference tree,
okvis_multisensor_processing/test/testImage.jpg (1280x960 gray: a calibration checkerboard on a
carpet, 28 % of the pixels saturated at 255, JPEG 8x8 block structure = plateaus and ties the
synthetic cells never produce).

Runs in the BUILD CONTAINER only (it needs /root/reference and Pillow); what is committed is
DATA: the decoded u8 pixels, the crop table and the CPU oracle's outputs on every crop
(keypoints, descriptors in the camera-aware mode, SHA-256 digests of the other two modes,
back-projections, the gated stereo matches of a shifted pair).  The JPEG is decoded by Pillow's
libjpeg; OpenCV's decoder may differ by +-1 gray level, which is irrelevant here: the committed
pixels are the test input.

    python tools/make_real_image_fixture.py

The fixture certifies "HIP path == oracle == committed vectors" on real content; like every golden
vector of this repository it does NOT certify "oracle == reference binary" (SURVEY.md 8 C3).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import real_image_cases as RC  # noqa: E402

SRC = sys.argv[1] if len(sys.argv) > 1 else \
    "/root/reference/okvis_multisensor_processing/test/testImage.jpg"


def digest(kps, desc):
    return np.frombuffer(hashlib.sha256(kps.tobytes() + desc.tobytes()).digest(), dtype=np.uint8)


def main():
    from PIL import Image
    full = np.asarray(Image.open(SRC).convert("L"), dtype=np.uint8).copy()
    assert full.shape == (960, 1280), full.shape
    out = {"image": full}
    for case in RC.CASES:
        img = RC.crop(full, case)
        cam = case.cam
        rays, jac = O.awareness_maps(cam)
        kd = O.detect(img, case.radius, 0, case.thr, case.max_kpts)
        out[f"{case.name}/kp_detect"] = kd
        out[f"{case.name}/n_nms"] = np.int64(len(O.nms(O.harris_score(img), case.thr)))
        for mode, name in ((O.MODE_UPRIGHT, "upright"), (O.MODE_GRADIENT, "gradient")):
            k, d = O.describe(img, kd, mode)
            out[f"{case.name}/sha_{name}"] = digest(k, d)
            out[f"{case.name}/n_{name}"] = np.int64(len(k))
        k, d = O.describe(img, kd, O.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), RC.GRAVITY)
        bp, bv = O.backproject_keypoints(cam, k)
        out[f"{case.name}/kp_aware"], out[f"{case.name}/desc_aware"] = k, d
        out[f"{case.name}/bp"], out[f"{case.name}/bpv"] = bp, bv
        print(f"{case.name}: {img.shape[1]}x{img.shape[0]} saturated {np.mean(img == 255):.2f} "
              f"nms {int(out[case.name + '/n_nms'])} detect {len(kd)} aware {len(k)}")
    # stereo: the same scene seen RC.STEREO_DISPARITY px further left by the second camera
    sides = RC.stereo_sides(O, full)
    (k0, d0, b0, v0), (k1, d1, b1, v1) = sides
    m = O.match_stereo(d0, k0, b0, v0, d1, k1, b1, v1, *RC.stereo_geometry())
    out["stereo/match"] = m
    out["stereo/kp1"], out["stereo/desc1"] = k1, d1
    print("stereo:", len(k0), "x", len(k1), "->", int((m["k1"] >= 0).sum()), "matches")
    dst = os.path.join(ROOT, "tests", "golden", "real_image.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
