#!/usr/bin/env python3
"""Writes the seeded inputs of the reference bit-compare (tools/ref_compare/CMakeLists.txt):
8-bit PGM images, awareness maps as raw float32 and manifest.txt, one case per line.

Cases: the reference's own smoke-test call (752x480 iid-uniform image, detector (34, 2, 800, 450),
extractor (true, false): okvis_cv/test/TestFrame.cpp:75-85), the dbow2_test call ((36, 0, 100,
700), (false, false): okvis_apps/src/dbow2_test.cpp:100-101), and the shipped configurations
(euroc.yaml, tumvi_slam_1024.yaml, hilti_challenge_2022.yaml front-end blocks) in the production
camera-aware mode, on this repo's seeded corner / noise images.

usage: make_inputs.py <out_dir>      (needs only numpy + this repo; host code, no GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from okvis2_amd import capi, synth  # noqa: E402


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


def cases():
    """(name, image, radius, octaves, thr, max_kpts, rot_inv, scale_inv, mode, camera, gravity)"""
    out = []
    noise = synth.noise_image(752, 480, 0x0C0FFEE0)
    out.append(("testframe_noise_oct2", noise, 34.0, 2, 800, 450, 1, 0, 0, None, None))
    out.append(("testframe_noise_oct0", noise, 34.0, 0, 800, 450, 1, 0, 0, None, None))
    out.append(("dbow2_corners", synth.corners_image(752, 480, 11), 36.0, 0, 100, 700, 0, 0, 0, None,
                None))
    for cfg, seed in ((synth.euroc_config(), 21), (synth.tumvi1024_config(), 22),
                      (synth.hilti_config(), 23), (synth.mono640_config(), 24)):
        L, R, _ = synth.stereo_pair(cfg.w, cfg.h, seed)
        for ci, img in enumerate((L, R)[: min(2, len(cfg.cams))]):
            out.append((f"{cfg.name}_cam{ci}_aware", img, cfg.uniformity_radius, cfg.octaves,
                        cfg.abs_threshold, cfg.max_kpts, 1, 0, 2, cfg.cams[ci], (0.1, 0.98, -0.05)))
        out.append((f"{cfg.name}_cam0_gradient", L, cfg.uniformity_radius, cfg.octaves,
                    cfg.abs_threshold, cfg.max_kpts, 1, 0, 0, None, None))
        out.append((f"{cfg.name}_cam0_upright", L, cfg.uniformity_radius, cfg.octaves,
                    cfg.abs_threshold, cfg.max_kpts, 0, 0, 0, None, None))
    return out


PROBE_SIZE, PROBE_KP = 128, (64.0, 64.0)


def probes():
    """(name, image): images that make the extractor reveal its pattern around ONE keypoint at
    PROBE_KP.  A 5x5 bright blob at (r, phi) raises exactly the samples whose smoothing box covers
    it, so the set of bits it flips names the samples near (r, phi), i.e. positions, half-widths,
    pairs and bit order; edges at 16 orientations and a few noise images exercise the comparisons
    on graded input."""
    out = []
    n = PROBE_SIZE
    yy, xx = np.mgrid[0:n, 0:n]
    k = 0
    for r in (0.0, 3.0, 6.0, 9.0, 12.0, 15.5, 19.0, 22.5, 26.0):
        for a in range(1 if r == 0.0 else 24):
            phi = 2.0 * np.pi * a / 24.0
            cx, cy = PROBE_KP[0] + r * np.cos(phi), PROBE_KP[1] + r * np.sin(phi)
            img = np.full((n, n), 20, np.uint8)
            img[(np.abs(xx - cx) <= 2.0) & (np.abs(yy - cy) <= 2.0)] = 235
            out.append((f"probe_{k:03d}", img))
            k += 1
    for a in range(16):
        phi = 2.0 * np.pi * a / 16.0
        side = (xx - PROBE_KP[0]) * np.cos(phi) + (yy - PROBE_KP[1]) * np.sin(phi)
        out.append((f"probe_{k:03d}", np.where(side > 0.5, 200, 40).astype(np.uint8)))
        k += 1
    for seed in range(8):
        out.append((f"probe_{k:03d}", synth.noise_image(n, n, 900 + seed)))
        k += 1
    return out


N_TRI, N_BP = 400, 600


def fp64_inputs(oracle=None):
    """Inputs of tools/ref_compare/ref_dump_fp64.cpp as one byte string (layout: see that file) plus the
    parsed pieces for the comparison side.  Ray pairs: a 3-D point seen from two centres with angular
    noise (intersecting), near-parallel pairs (far points), diverging and crossing-behind pairs, exactly
    parallel rays.  Pixels: a grid + seeded random points incl. the image rim, for the EuRoC
    (radial-tangential) and TUM-VI (equidistant) intrinsics.  Stereo: the oracle's keypoints / descriptors
    of a seeded EuRoC pair (inputs only -- the dump recomputes every double with the reference's code)."""
    import struct
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if oracle is None:
        import oracle_lib as oracle
    rng = np.random.default_rng(20260928)
    tri = np.zeros((N_TRI, 13))
    for i in range(N_TRI):
        kind = i % 5
        p1 = rng.normal(0, 0.3, 3)
        p2 = p1 + np.array([rng.uniform(0.05, 0.3), rng.normal(0, 0.02), rng.normal(0, 0.02)])
        X = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.3, 60.0) if kind != 1 else rng.uniform(200, 5000)])
        e1, e2 = X - p1, X - p2
        if kind == 2:      # diverging: the second ray looks away
            e2 = e2 * np.array([1, 1, -1.0])
        if kind == 3:      # noisy: rays miss each other
            e1 = e1 + rng.normal(0, 0.02, 3)
            e2 = e2 + rng.normal(0, 0.02, 3)
        if kind == 4 and i % 10 == 4:  # exactly parallel
            e2 = e1.copy()
        e1, e2 = e1 / np.linalg.norm(e1), e2 / np.linalg.norm(e2)
        tri[i] = np.concatenate([p1, e1, p2, e2, [rng.choice([12.0, 18.0, 24.0]) / 458.0 * 0.125]])
    blob = struct.pack("<4i", N_TRI, N_BP, 0, 0)  # n0 / n1 patched below
    blob += tri.tobytes()
    cams = {"rt": synth.euroc_config().cams[0], "eq": synth.tumvi1024_config().cams[0]}
    pts = {}

    def intr(c):
        return struct.pack("<4i", c.w, c.h, c.dist_type, 0) + struct.pack("<8d", c.fu, c.fv, c.cu, c.cv, *c.d)

    for key in ("rt", "eq"):
        c = cams[key]
        g = np.stack(np.meshgrid(np.linspace(0, c.w - 1, 20), np.linspace(0, c.h - 1, 15)), -1).reshape(-1, 2)
        r = np.stack([rng.uniform(-8, c.w + 8, N_BP - len(g)), rng.uniform(-8, c.h + 8, N_BP - len(g))], 1)
        pts[key] = np.concatenate([g, r])[:N_BP].astype(np.float32).astype(np.float64)  # keypoints are floats
        blob += intr(c) + pts[key].tobytes()
    cfg = synth.euroc_config()
    L, R, _ = synth.stereo_pair(cfg.w, cfg.h, 77)
    sides = []
    for ci, img in enumerate((L, R)):
        cam = cfg.cams[ci]
        rays, jac = oracle.awareness_maps(cam)
        k, d = oracle.detect_describe(img, cfg.uniformity_radius, 0, cfg.abs_threshold, cfg.max_kpts,
                                      oracle.MODE_CAMERA_AWARE, rays, jac, np.float32(cam.fu), (0.0, 1.0, 0.0))
        sides.append((k, d))
    T0, T1 = synth.stereo_poses(cfg.baseline)
    # a rotated, translated rig: the products with C and the subtraction of r are not trivial
    a = 0.3
    Rw = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.1), -np.sin(0.1)], [0, np.sin(0.1), np.cos(0.1)]])
    tw = np.array([1.5, -0.25, 0.75])
    poses = [((Rw @ np.asarray(C).reshape(3, 3)).reshape(-1), Rw @ np.asarray(r) + tw) for C, r in (T0, T1)]
    blob += intr(cfg.cams[0]) + intr(cfg.cams[1])
    for C, r in poses:
        blob += np.asarray(C, np.float64).tobytes() + np.asarray(r, np.float64).tobytes()
    blob += struct.pack("<d", float(cfg.match_threshold))
    for k, d in sides:
        blob += np.stack([k["x"], k["y"], k["size"]], 1).astype(np.float32).tobytes() + d.tobytes()
    n0, n1 = len(sides[0][0]), len(sides[1][0])
    blob = blob[:8] + struct.pack("<2i", n0, n1) + blob[16:]
    return blob, dict(tri=tri, cams=cams, pts=pts, cfg=cfg, sides=sides, poses=poses)


def main():
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "fp64_inputs.bin"), "wb") as f:
        f.write(fp64_inputs()[0])
    with open(os.path.join(out_dir, "probes.txt"), "w") as f:
        f.write("# file W H kx ky\n")
        for name, img in probes():
            write_pgm(os.path.join(out_dir, name + ".pgm"), img)
            f.write(f"{name}.pgm {PROBE_SIZE} {PROBE_SIZE} {PROBE_KP[0]:.1f} {PROBE_KP[1]:.1f}\n")
    lines = ["# name image W H radius octaves abs_threshold max_kpts rot_inv scale_inv mode fu gx gy gz "
             "rays jac"]
    maps_done = {}
    for (name, img, radius, octaves, thr, maxk, rot, sc, mode, cam, grav) in cases():
        write_pgm(os.path.join(out_dir, name + ".pgm"), img)
        rays_f = jac_f = "-"
        fu, g = 1.0, (0.0, 1.0, 0.0)
        if mode == 2:
            key = (cam.w, cam.h, cam.fu, cam.cu, cam.dist_type, cam.d)
            if key not in maps_done:
                rays, jac = capi.build_awareness_maps(cam)
                stem = f"maps{len(maps_done)}"
                rays.tofile(os.path.join(out_dir, stem + ".rays.f32"))
                jac.tofile(os.path.join(out_dir, stem + ".jac.f32"))
                maps_done[key] = stem
            rays_f, jac_f = maps_done[key] + ".rays.f32", maps_done[key] + ".jac.f32"
            fu, g = float(np.float32(cam.fu)), grav
        h, w = img.shape
        lines.append(f"{name} {name}.pgm {w} {h} {radius:.9g} {octaves} {thr} {maxk} {rot} {sc} {mode} "
                     f"{fu:.9g} {g[0]:.9g} {g[1]:.9g} {g[2]:.9g} {rays_f} {jac_f}")
    with open(os.path.join(out_dir, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{len(lines) - 1} cases written to {out_dir}")


if __name__ == "__main__":
    main()
