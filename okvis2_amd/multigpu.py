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


def camera_owner(cam: int, world: int) -> int:
    """Rank that detects + describes camera `cam` of the rig: round robin."""
    return cam % world


class CrossCameraMatcher:
    """Cross-camera matchStereo of an n-camera rig whose cameras live on `world` ranks
    (Frontend.cpp:1990-2026).

    Camera c is owned by rank c % world, slot c // world of that rank; every rank holds
    slots = ceil(n_cams / world) gather-block rows (unused ones stay empty: count 0).  `step`
      1. runs detect + describe of every LOCAL camera on that camera's own engine (one context
         per camera, like one detector/extractor per camera in the reference,
         Frontend.cpp:2405-2413) and stream, and packs the results into gather blocks (one kernel
         per camera),
      2. moves all blocks with ONE all-gather ([slots, frames, block] bytes per rank; RCCL over
         xGMI with backend nccl),
      3. matches, with one launch per pair over all frames, the camera pairs this rank owns:
         pair (i, j), i < j, FoV-overlapping only (MultiFrame::hasOverlap, Frontend.cpp:1998),
         belongs to rank (i + j) % world.
    Stream discipline: every library call carries an explicit stream; the pack kernels' streams
    are joined into `main` before the collective, which is issued with `main` current, and the
    matchers run on `main` after it -- the exchange is ordered by streams, never by timing.

    engines: {cam: engine} for the local cameras.  An engine is a capi.Frontend or anything with
    its five methods used here (the CPU gloo tests pass an oracle-backed host engine):
    detect_describe_batch_device, pack_gather_blocks_device, match_stereo_blocks_batch_device,
    gather_block_bytes, max_keypoints.
    Results: {(i, j): uint8 tensor [frames, max_keypoints, sizeof(okvfe_stereo_match)]}."""

    def __init__(self, engines: dict, n_cams: int, n_frames: int, poses_T_WC, focal, overlap,
                 world: int, rank: int, device, group=None, comm="auto"):
        import torch
        self.torch = torch
        self.n_cams, self.n_frames, self.world, self.rank = n_cams, n_frames, world, rank
        self.local_cams = [c for c in range(n_cams) if camera_owner(c, world) == rank]
        if sorted(engines) != self.local_cams:
            raise ValueError(f"rank {rank} of {world} owns cameras {self.local_cams}, "
                             f"engines were given for {sorted(engines)}")
        self.engines = engines
        self.poses, self.focal, self.group = poses_T_WC, focal, group
        self.slots = (n_cams + world - 1) // world
        any_engine = engines[self.local_cams[0]] if self.local_cams else None
        if any_engine is None:
            raise ValueError("a rank without a camera cannot take part (world > n_cams)")
        self.matcher = any_engine
        self.block_bytes = any_engine.gather_block_bytes()
        self.kp_cap = any_engine.max_keypoints
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.local = torch.zeros((self.slots, n_frames, self.block_bytes), dtype=torch.uint8,
                                 device=self.device)
        self.gathered = torch.zeros((world * self.slots, n_frames, self.block_bytes),
                                    dtype=torch.uint8, device=self.device)
        self.schedule = pair_schedule(n_cams, overlap, world)
        self.mine = [(i, j) for (i, j, o) in self.schedule if o == rank]
        self.out = {p: torch.zeros((n_frames, self.kp_cap, capi.STEREO_MATCH_DTYPE.itemsize),
                                   dtype=torch.uint8, device=self.device) for p in self.mine}
        # The collective: on the GPU it is issued from C -- okvfe_gather_blocks = ncclAllGather on the
        # main stream, through a communicator made with okvfe_comm_create (the 128-byte id travels by
        # torch.distributed.broadcast); comm=None keeps torch.distributed's all_gather_into_tensor
        # (the CPU / gloo tests), a capi.Comm is used as given.
        self.comm = None
        if comm == "auto":
            if self.cuda:
                self.comm = self._make_comm()
        elif comm is not None:
            self.comm = comm
        if self.cuda:
            self.main = torch.cuda.Stream(device=self.device)
            self.cam_streams = {c: torch.cuda.Stream(device=self.device) for c in self.local_cams}
            torch.cuda.current_stream(self.device).synchronize()  # the zero fills above
        else:
            self.main, self.cam_streams = None, {c: None for c in self.local_cams}

    def block_of(self, gathered, cam: int):
        """[frames, block_bytes] view of camera `cam` in the all-gathered tensor."""
        return gathered[camera_owner(cam, self.world) * self.slots + cam // self.world]

    def step(self, images_ptrs: dict, gravities: dict):
        """images_ptrs[c]: device pointer of camera c's [frames, H, W] u8 images; gravities[c]:
        float32 [frames, 3] extraction directions (gravity in camera c's frame).  Returns
        (gathered blocks [world*slots, frames, block], {pair: matches}); the caller synchronises
        (`finish`) before reading."""
        torch, n = self.torch, self.n_frames
        cam_ids = np.zeros(n, dtype=np.int32)  # every engine holds its camera in slot 0
        for c in self.local_cams:
            st = self.cam_streams[c]
            if self.cuda:
                st.wait_stream(self.main)  # the previous step's collective has read self.local
            eng = self.engines[c]
            eng.detect_describe_batch_device(images_ptrs[c], n, cam_ids, gravities[c], st)
            eng.pack_gather_blocks_device(0, n, self.local[c // self.world].data_ptr(), st)
        if self.cuda:
            for c in self.local_cams:
                self.main.wait_stream(self.cam_streams[c])
            with torch.cuda.stream(self.main):
                self._gather()
        else:
            self._gather()
        for (i, j) in self.mine:  # one launch per pair
            self.matcher.match_stereo_blocks_batch_device(
                self.block_of(self.gathered, i).data_ptr(), self.block_of(self.gathered, j).data_ptr(),
                n, self.poses[i], self.poses[j], self.focal[i], self.focal[j],
                self.out[(i, j)].data_ptr(), self.main)
        return self.gathered, self.out

    def _make_comm(self):
        """One RCCL communicator over the ranks of `group`, created through the C ABI."""
        import torch.distributed as dist
        torch = self.torch
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if not (dist.is_available() and dist.is_initialized()) or self.world == 1:
            return capi.Comm.create(capi.Comm.unique_id(), 1, 0, dev_index)  # RCCL with one rank
        id_t = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            id_t.copy_(torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8))
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(id_t, src=src, group=self.group)
        return capi.Comm.create(bytes(id_t.cpu().numpy().tobytes()), self.world, self.rank, dev_index)

    def _gather(self):
        if self.comm is not None:  # C path: ncclAllGather on the current (main) stream
            st = self.main.cuda_stream if self.cuda else None
            self.comm.gather(self.local.data_ptr(), self.gathered.data_ptr(),
                             self.local.numel() * self.local.element_size(), st)
            return
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            self.gathered.copy_(self.local)
            return
        dist.all_gather_into_tensor(self.gathered.view(self.world, -1),
                                    self.local.view(1, -1), group=self.group)

    def finish(self):
        if self.cuda:
            self.main.synchronize()
