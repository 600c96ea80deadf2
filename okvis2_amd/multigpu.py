"""Multi-GPU plumbing of the front-end: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

E1  detect+describe of an image depends on nothing but that image, so multiframes shard across
    ranks with no communication (`shard_range`); bench.py scales this way.
E2  the only exchange step of the path: when the cameras of ONE multiframe live on different GPUs
    (north_star: one camera per stream, Hilti 5-camera rig), Frontend::matchStereo
    (okvis_frontend/src/Frontend.cpp:1990-2026) needs both cameras' keypoints/descriptors/
    back-projections co-resident.  Each rank packs fixed-size per-image gather blocks
    (okvfe_pack_gather_block_device), ONE all-gather moves [frames x block] bytes per rank
    (latency-bound per frame, so many multiframes ride in one call), and camera pair (i, j) is
    matched on rank (i + j) % world (`pair_schedule`) -- only pairs with field-of-view overlap
    are visited, as in the reference (Frontend.cpp:1998, NCameraSystem.cpp:48-119).
"""
from __future__ import annotations

import numpy as np

from . import capi


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous [lo, hi) slice of n_items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pair_schedule(n_cams: int, overlap, world: int):
    """Static owner of every overlapping camera pair i < j: rank (i + j) % world.
    overlap(i, j) -> bool mirrors MultiFrame::hasOverlap."""
    return [(i, j, (i + j) % world) for i in range(n_cams) for j in range(i + 1, n_cams)
            if overlap(i, j)]


def block_layout(kp_cap: int):
    """Byte offsets of one gather block (mirror of block_layout() in csrc/okvfe_capi.cpp)."""
    def up(v, a):
        return (v + a - 1) // a * a
    o_kps = 16
    o_desc = up(o_kps + kp_cap * capi.KEYPOINT_DTYPE.itemsize, 16)
    o_bp = up(o_desc + kp_cap * capi.DESC_BYTES, 16)
    o_bpv = up(o_bp + kp_cap * 24, 16)
    total = up(o_bpv + kp_cap, 256)
    return {"count": 0, "kps": o_kps, "desc": o_desc, "bp": o_bp, "bpv": o_bpv, "total": total}


def pack_block_host(kp_cap: int, kps, desc, bp, bpv) -> np.ndarray:
    """Host-side packer with the same layout (tests / host-buffer callers)."""
    L = block_layout(kp_cap)
    n = len(kps)
    if n > kp_cap:
        raise ValueError(f"{n} keypoints exceed the block capacity {kp_cap}")
    b = np.zeros(L["total"], dtype=np.uint8)
    b[0:4] = np.array([n], dtype=np.int32).view(np.uint8)
    b[L["kps"]:L["kps"] + n * 28] = np.ascontiguousarray(kps).view(np.uint8).reshape(-1)
    b[L["desc"]:L["desc"] + n * 48] = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1)
    b[L["bp"]:L["bp"] + n * 24] = np.ascontiguousarray(bp, dtype=np.float64).view(np.uint8).reshape(-1)
    b[L["bpv"]:L["bpv"] + n] = np.ascontiguousarray(bpv, dtype=np.uint8)
    return b


def unpack_block_host(block: np.ndarray, kp_cap: int):
    L = block_layout(kp_cap)
    block = np.ascontiguousarray(block, dtype=np.uint8)
    n = int(block[0:4].view(np.int32)[0])
    kps = block[L["kps"]:L["kps"] + n * 28].view(capi.KEYPOINT_DTYPE).copy()
    desc = block[L["desc"]:L["desc"] + n * 48].reshape(n, 48).copy()
    bp = block[L["bp"]:L["bp"] + n * 24].view(np.float64).reshape(n, 3).copy()
    bpv = block[L["bpv"]:L["bpv"] + n].copy()
    return kps, desc, bp, bpv


def all_gather_blocks(local_blocks, group=None):
    """ONE collective for the whole batch: local_blocks is a uint8 tensor [frames, block_bytes]
    holding this rank's camera; returns [world, frames, block_bytes] (rank-major).
    With backend nccl this is an RCCL all-gather over xGMI on device memory."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    frames, nbytes = local_blocks.shape
    out = torch.empty((world * frames, nbytes), dtype=local_blocks.dtype,
                      device=local_blocks.device)
    dist.all_gather_into_tensor(out, local_blocks.contiguous(), group=group)
    return out.view(world, frames, nbytes)


class CrossCameraMatcher:
    """Rank r owns camera r of an n-camera rig (n == world).  `step` runs detect+describe for this
    rank's camera on a batch of frames, gathers all cameras' blocks and matches the camera pairs
    this rank owns.  Results: {(i, j): uint8 tensor [frames, kp_cap, 48]} of okvfe_stereo_match."""

    def __init__(self, fe: "capi.Frontend", cam_index: int, n_frames: int, poses_T_WC, focal,
                 overlap, world: int, rank: int, device):
        import torch
        self.fe, self.cam, self.n_frames = fe, cam_index, n_frames
        self.poses, self.focal = poses_T_WC, focal
        self.world, self.rank = world, rank
        self.block_bytes = fe.gather_block_bytes()
        self.local = torch.zeros((n_frames, self.block_bytes), dtype=torch.uint8, device=device)
        self.mine = [(i, j) for (i, j, o) in pair_schedule(world, overlap, world) if o == rank]
        self.out = {p: torch.zeros((n_frames, fe.max_keypoints, capi.STEREO_MATCH_DTYPE.itemsize),
                                   dtype=torch.uint8, device=device) for p in self.mine}

    def step(self, images_ptr, gravity, stream=None, group=None):
        fe, n = self.fe, self.n_frames
        cam_ids = np.full(n, 0, dtype=np.int32)  # this context holds its own camera in slot 0
        fe.detect_describe_batch_device(images_ptr, n, cam_ids, gravity, stream)
        fe.pack_gather_blocks_device(0, n, self.local.data_ptr(), stream)   # one kernel
        allb = all_gather_blocks(self.local, group)                          # one collective
        for (i, j) in self.mine:                                             # one launch per pair
            fe.match_stereo_blocks_batch_device(allb[i].data_ptr(), allb[j].data_ptr(), n,
                                                self.poses[i], self.poses[j], self.focal[i],
                                                self.focal[j], self.out[(i, j)].data_ptr(), stream)
        return allb, self.out
