"""Helpers shared by the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np

from okvis2_amd import capi, synth


def make_frontend(cfg, max_batch=1, num_cameras=None, **kw):
    args = dict(width=cfg.w, height=cfg.h, uniformity_radius=cfg.uniformity_radius,
                octaves=cfg.octaves, absolute_threshold=cfg.abs_threshold,
                max_keypoints=cfg.max_kpts, match_threshold=cfg.match_threshold,
                max_batch=max_batch, num_cameras=num_cameras or len(cfg.cams))
    args.update(kw)
    return capi.Frontend(**args)


def assert_keypoints_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("x", "y", "size", "angle", "response"):
        av, bv = a[f].view(np.uint32), b[f].view(np.uint32)
        assert np.array_equal(av, bv), (f, np.flatnonzero(av != bv)[:5], a[f][:3], b[f][:3])
    assert np.array_equal(a["octave"], b["octave"])
    assert np.array_equal(a["class_id"], b["class_id"])


def image_for(cfg, seed, kind="corners", cam=0):
    if kind == "noise":
        return synth.noise_image(cfg.w, cfg.h, seed + cam)
    return synth.corners_image(cfg.w, cfg.h, seed + cam)
