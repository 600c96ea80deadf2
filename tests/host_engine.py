"""CPU stand-in for capi.Frontend in the world-size-2 gloo tests (no GPU in that tier).

okvis2_amd.multigpu.CrossCameraMatcher drives an "engine" through five methods; on the GPU box
the engine is capi.Frontend (libokvfe.so), here it is this class: the same calls, host memory
behind the pointers, the CPU oracle doing the arithmetic.  Test infrastructure only."""
import ctypes

import numpy as np

import oracle_lib as O
from okvis2_amd import capi, multigpu


def _bytes_at(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(int(ptr)))


class OracleEngine:
    def __init__(self, cfg, cam):
        self.cfg, self.cam = cfg, cam
        self.max_keypoints = cfg.max_kpts
        self.rays, self.jac = O.awareness_maps(cam)
        self.results = []
        self.calls = []

    def gather_block_bytes(self):
        return multigpu.block_layout(self.max_keypoints)["total"]

    def detect_describe_batch_device(self, images_ptr, n, cam_ids, gravity, stream=None):
        cfg, cam = self.cfg, self.cam
        imgs = _bytes_at(images_ptr, n * cfg.w * cfg.h).reshape(n, cfg.h, cfg.w)
        self.results = []
        for f in range(n):
            k, d = O.detect_describe(imgs[f], cfg.uniformity_radius, 0, cfg.abs_threshold,
                                     cfg.max_kpts, O.MODE_CAMERA_AWARE, self.rays, self.jac,
                                     np.float32(cam.fu), tuple(float(v) for v in gravity[f]))
            bp, bv = O.backproject_keypoints(cam, k)
            self.results.append((k, d, bp, bv))
        self.calls.append(("detect_describe", n))

    def pack_gather_blocks_device(self, first, n, blocks_ptr, stream=None):
        nb = self.gather_block_bytes()
        dst = _bytes_at(blocks_ptr, n * nb).reshape(n, nb)
        for f in range(n):
            dst[f] = multigpu.pack_block_host(self.max_keypoints, *self.results[first + f])
        self.calls.append(("pack", n))

    def match_stereo_blocks_batch_device(self, blocks0_ptr, blocks1_ptr, n, T0, T1, f0, f1,
                                         matches_ptr, stream=None):
        nb, cap = self.gather_block_bytes(), self.max_keypoints
        b0 = _bytes_at(blocks0_ptr, n * nb).reshape(n, nb)
        b1 = _bytes_at(blocks1_ptr, n * nb).reshape(n, nb)
        row = capi.STEREO_MATCH_DTYPE.itemsize
        out = _bytes_at(matches_ptr, n * cap * row).reshape(n, cap * row)
        for f in range(n):
            k0, d0, p0, v0 = multigpu.unpack_block_host(b0[f], cap)
            k1, d1, p1, v1 = multigpu.unpack_block_host(b1[f], cap)
            m = O.match_stereo(d0, k0, p0, v0, d1, k1, p1, v1, T0, T1, f0, f1,
                               self.cfg.match_threshold)
            out[f, :len(m) * row] = m.view(np.uint8).reshape(-1)
        self.calls.append(("match", n))
