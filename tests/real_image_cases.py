"""Crops of the reference's real test image (tests/golden/real_image.npz, made by
tools/make_real_image_fixture.py) and the detector / extractor parameters each crop runs with.
Shared by the fixture generator, the CPU test (oracle == committed vectors) and the -m gpu test
(HIP path == oracle == committed vectors)."""
import dataclasses
import os

import numpy as np

from okvis2_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_image.npz")
GRAVITY = (0.1, 0.98, -0.05)
STEREO_DISPARITY = 23


@dataclasses.dataclass
class Case:
    name: str
    w: int
    h: int
    x0: int
    y0: int
    radius: float
    thr: int
    max_kpts: int
    cam: synth.Camera


def _cases():
    e, m, t, hi = synth.euroc_config(), synth.mono640_config(), synth.tumvi1024_config(), synth.hilti_config()
    d_rt = e.cams[0].d
    d_eq = t.cams[0].d
    return [
        # BASELINE shapes with the front-end parameters of their shipped configuration
        Case("euroc", 752, 480, 264, 240, e.uniformity_radius, e.abs_threshold, e.max_kpts, e.cams[0]),
        Case("hilti", 720, 540, 100, 50, hi.uniformity_radius, hi.abs_threshold, hi.max_kpts, hi.cams[0]),
        Case("mono640", 640, 480, 600, 400, m.uniformity_radius, m.abs_threshold, m.max_kpts, m.cams[0]),
        # the whole image, and an odd shape between the BASELINE ones (TUM-VI parameters, packed last strip)
        Case("full", 1280, 960, 0, 0, e.uniformity_radius, e.abs_threshold, e.max_kpts,
             synth.Camera(1280, 960, 610.0, 612.0, 644.3, 478.9, 1, d_rt)),
        Case("odd1024", 1024, 960, 128, 0, t.uniformity_radius, t.abs_threshold, t.max_kpts,
             synth.Camera(1024, 960, 382.3307, 382.3203, 510.3634, 482.2949, 2, d_eq)),
        # carpet texture at a low threshold: thousands of weak maxima, the densest candidate lists
        Case("carpet", 752, 480, 500, 0, 10.0, 5, 1000, e.cams[1]),
    ]


CASES = _cases()


def load():
    return np.load(GOLDEN)


def crop(full, case):
    return np.ascontiguousarray(full[case.y0:case.y0 + case.h, case.x0:case.x0 + case.w])


def stereo_images(full):
    c = CASES[0]
    left = crop(full, c)
    right = np.ascontiguousarray(full[c.y0:c.y0 + c.h, c.x0 + STEREO_DISPARITY:c.x0 + STEREO_DISPARITY + c.w])
    return left, right


def stereo_geometry():
    """(T_WC0, T_WC1, f0, f1, threshold): the EuRoC rig of okvis2_amd.synth."""
    cfg = synth.euroc_config()
    T0, T1 = synth.stereo_poses(cfg.baseline)
    f0 = 0.5 * (cfg.cams[0].fu + cfg.cams[0].fv)
    f1 = 0.5 * (cfg.cams[1].fu + cfg.cams[1].fv)
    return T0, T1, f0, f1, cfg.match_threshold


def stereo_sides(oracle, full):
    """Oracle keypoints / descriptors / rays of the shifted real pair (camera-aware, EuRoC cams).
    Detector parameters: radius 20, threshold 20 -- enough points on the board's inner corners,
    which look alike (the matcher's gate decides between them)."""
    cfg = synth.euroc_config()
    out = []
    for ci, img in enumerate(stereo_images(full)):
        cam = cfg.cams[ci]
        rays, jac = oracle.awareness_maps(cam)
        k, d = oracle.detect_describe(img, 20.0, 0, 20, cfg.max_kpts, oracle.MODE_CAMERA_AWARE,
                                      rays, jac, np.float32(cam.fu), (0.0, 1.0, 0.0))
        bp, bv = oracle.backproject_keypoints(cam, k)
        out.append((k, d, bp, bv))
    return out
