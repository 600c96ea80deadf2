"""ctypes binding of libokvfe.so (the C ABI declared in include/okvfe.h).

This is the product path used by bench.py and the GPU tests: every call goes through the C ABI
into the hand-written HIP kernels.  There is no Python or CPU implementation behind it -- if the
library is missing or no gfx950 device is present, loading / okvfe_create fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OKVFE_LIB") or os.path.join(_HERE, "libokvfe.so")  # OKVFE_LIB: A/B builds

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_OUT_OF_MEMORY, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_DEVICE, \
    ERR_NOT_READY = 1, 2, 3, 4, 5, 6, 7
ABI_VERSION = 7
SCORE_HARRIS, SCORE_AGAST_9_16, SCORE_BRISK_SCALESPACE = 0, 1, 2
DESC_BYTES = 48

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
STEREO_MATCH_DTYPE = np.dtype([("k1", "<i4"), ("dist", "<i4"), ("initialisable", "<i4"),
                               ("pad", "<i4"), ("hp_W", "<f8", (4,))])
CAND_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("dist", "<i4")])
MOTION_MATCH_DTYPE = np.dtype([("k1", "<i4"), ("dist", "<i4"), ("initialisable", "<i4"),
                               ("accepted", "<i4"), ("cos_quality", "<f8"), ("hp_W", "<f8", (4,))])


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("max_batch", C.c_int32), ("num_cameras", C.c_int32),
                ("uniformity_radius", C.c_float), ("octaves", C.c_int32),
                ("absolute_threshold", C.c_int32), ("max_keypoints", C.c_int32),
                ("rotation_invariant", C.c_int32), ("scale_invariant", C.c_int32),
                ("match_threshold", C.c_int32), ("max_candidates", C.c_int32),
                ("score_type", C.c_int32), ("box_scale", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fu", C.c_double), ("fv", C.c_double),
                ("cu", C.c_double), ("cv", C.c_double), ("distortion", C.c_int32),
                ("d", C.c_double * 4)]


class Pose(C.Structure):
    _fields_ = [("C", C.c_double * 9), ("r", C.c_double * 3)]


class StereoPair(C.Structure):
    _fields_ = [("image0", C.c_int32), ("image1", C.c_int32), ("T_WC0", Pose), ("T_WC1", Pose),
                ("f0", C.c_double), ("f1", C.c_double)]


class PatternData(C.Structure):
    """okvfe_pattern: the extractor's sampling pattern as data (okvfe_get_pattern / okvfe_set_pattern)."""
    _fields_ = [("n_points", C.c_int32), ("px", C.c_float * 72), ("py", C.c_float * 72),
                ("sigma_half", C.c_float * 72), ("n_short", C.c_int32),
                ("short_i", C.c_uint8 * 384), ("short_j", C.c_uint8 * 384),
                ("n_long", C.c_int32), ("long_i", C.c_uint8 * 1100), ("long_j", C.c_uint8 * 1100),
                ("long_wdx", C.c_int32 * 1100), ("long_wdy", C.c_int32 * 1100), ("border", C.c_int32)]


class MapDevice(C.Structure):
    """okvfe_map_device: the pooled landmark set in device memory (all members device pointers)."""
    _fields_ = [("n_landmarks", C.c_int32), ("desc_begin", C.c_void_p), ("pool", C.c_void_p),
                ("projections", C.c_void_p), ("e0_W", C.c_void_p), ("r0_W", C.c_void_p)]


class DeviceOutputs(C.Structure):
    _fields_ = [("max_keypoints", C.c_int32), ("counts", C.c_void_p), ("keypoints", C.c_void_p),
                ("descriptors", C.c_void_p), ("backproj", C.c_void_p),
                ("backproj_valid", C.c_void_p), ("scores", C.c_void_p),
                ("detect_counts", C.c_void_p), ("candidate_counts", C.c_void_p),
                ("score_pitch", C.c_int32), ("score_strips", C.c_int32)]


EXPORTS = [
    "okvfe_create", "okvfe_destroy", "okvfe_last_error", "okvfe_abi_version",
    "okvfe_set_camera_maps", "okvfe_set_camera", "okvfe_build_awareness_maps",
    "okvfe_detect_describe", "okvfe_detect", "okvfe_detect_ahead", "okvfe_detect_describe_batch_device",
    "okvfe_get_device_outputs", "okvfe_score_column", "okvfe_set_heavy_kernel_chaining", "okvfe_set_keep_score_map", "okvfe_set_internal_lanes", "okvfe_lanes_join", "okvfe_set_fp64_reduction", "okvfe_scale_index", "okvfe_download_image_result", "okvfe_harris_score_device", "okvfe_harris_byte_mover_device",
    "okvfe_match_stereo_batch_device", "okvfe_match_stereo", "okvfe_hamming_candidates",
    "okvfe_hamming_argmin", "okvfe_popcnt_xor", "okvfe_gather_block_bytes",
    "okvfe_pack_gather_block_device", "okvfe_match_stereo_blocks_device",
    "okvfe_profile_enable", "okvfe_profile_read", "okvfe_match_motion_stereo_blocks_device",
    "okvfe_detect_batch_device", "okvfe_describe_batch_device", "okvfe_camera_overlap", "okvfe_compute",
    "okvfe_match_motion_stereo", "okvfe_match_to_map",
    "okvfe_format_keypoint_lines", "okvfe_parse_keypoint_lines", "okvfe_fbrisk_mean",
    "okvfe_match_to_map_uninitialised", "okvfe_pack_gather_blocks_device",
    "okvfe_match_stereo_blocks_batch_device", "okvfe_check_capacity",
    "okvfe_detect_describe_batch_host", "okvfe_verify_place_match", "okvfe_fbrisk_transform",
    "okvfe_match_to_map_landmarks", "okvfe_bow_vector", "okvfe_bow_query_l1",
    "okvfe_get_pattern", "okvfe_set_pattern", "okvfe_pattern_kernel_class",
    "okvfe_match_to_map_blocks_device", "okvfe_match_to_map_uninitialised_blocks_device",
    "okvfe_verify_place_blocks_device",
    "okvfe_comm_unique_id", "okvfe_comm_create", "okvfe_comm_wrap", "okvfe_comm_destroy",
    "okvfe_comm_world", "okvfe_comm_rank", "okvfe_comm_last_error", "okvfe_gather_blocks",
    "okvfe_device_alloc", "okvfe_device_free", "okvfe_stream_create", "okvfe_stream_destroy",
    "okvfe_stream_synchronize", "okvfe_copy_to_device", "okvfe_copy_to_host", "okvfe_device_fill",
]

STAGES = ["harris", "nms", "sort", "select", "map", "describe", "compact", "match"]

_LIB = None


class LandmarkTable(C.Structure):
    _fields_ = [("n_landmarks", C.c_int32), ("n_observations", C.c_int32), ("n_poses", C.c_int32),
                ("hp_W", C.c_void_p), ("quality", C.c_void_p), ("obs_begin", C.c_void_p),
                ("obs_pose", C.c_void_p), ("obs_desc", C.c_void_p), ("obs_backproj", C.c_void_p),
                ("poses", C.c_void_p)]


class LandmarkPool(C.Structure):
    _fields_ = [("status", C.c_void_p), ("n_desc", C.c_void_p), ("obs_rows", C.c_void_p),
                ("projection", C.c_void_p), ("e_W", C.c_void_p), ("r_W", C.c_void_p)]


class OkvfeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"okvfe status {status}: {message}")
        self.status = status


def lib():
    """Loads libokvfe.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch-ROCm ships its own libamdhip64; whichever HIP runtime is loaded second finds no
        # GPU.  Importing torch first makes libokvfe.so bind to the runtime torch already loaded
        # (same soname), so device pointers and streams are shared.  Without torch installed the
        # system runtime under /opt/rocm is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.okvfe_last_error.restype = C.c_char_p
        L.okvfe_last_error.argtypes = [C.c_void_p]
        L.okvfe_popcnt_xor.restype = C.c_uint32
        L.okvfe_gather_block_bytes.restype = C.c_size_t
        L.okvfe_gather_block_bytes.argtypes = [C.c_void_p]
        L.okvfe_destroy.argtypes = [C.c_void_p]
        L.okvfe_destroy.restype = None
        L.okvfe_comm_last_error.restype = C.c_char_p
        L.okvfe_comm_destroy.argtypes = [C.c_void_p]
        L.okvfe_comm_destroy.restype = None
        L.okvfe_gather_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.okvfe_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        _LIB = L
    return _LIB


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


STREAM_LEGACY_DEFAULT = 1  # OKVFE_STREAM_LEGACY_DEFAULT (= hipStreamLegacy)
COMM_ID_BYTES = 128


class Comm:
    """okvfe_comm: the RCCL communicator of the cross-camera gather, driven through the C ABI
    (okvfe_comm_create = ncclCommInitRank, okvfe_gather_blocks = ncclAllGather on the caller's
    stream).  Comm.local() is the world-1 communicator that never touches RCCL."""

    def __init__(self, handle, world, rank):
        self._h, self.world, self.rank = handle, world, rank

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        st = lib().okvfe_comm_unique_id(buf)
        if st != 0:
            raise OkvfeError(st, lib().okvfe_comm_last_error().decode())
        return bytes(buf)

    @classmethod
    def create(cls, comm_id, world: int, rank: int, device: int):
        h = C.c_void_p()
        idbuf = None
        if comm_id is not None:
            idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        st = lib().okvfe_comm_create(C.cast(idbuf, C.c_void_p) if idbuf is not None else None,
                                     int(world), int(rank), int(device), C.byref(h))
        if st != 0:
            raise OkvfeError(st, lib().okvfe_comm_last_error().decode())
        return cls(h, world, rank)

    @classmethod
    def local(cls):
        return cls.create(None, 1, 0, 0)

    def gather(self, send_ptr, recv_ptr, bytes_per_rank: int, stream=None):
        """raw stream handle: torch.cuda.Stream.cuda_stream (0 / None = the HIP null stream)"""
        raw = getattr(stream, "cuda_stream", stream)
        st = lib().okvfe_gather_blocks(self._h, _p(send_ptr), _p(recv_ptr), int(bytes_per_rank),
                                       C.c_void_p(int(raw)) if raw else None)
        if st != 0:
            raise OkvfeError(st, lib().okvfe_comm_last_error().decode())

    def close(self):
        if self._h:
            lib().okvfe_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



def _s(stream):
    """Stream argument of the "_device" entry points: None = the context's own stream; a
    torch.cuda.Stream (anything with .cuda_stream) = that stream, torch's default stream (raw
    handle 0) becoming OKVFE_STREAM_LEGACY_DEFAULT; an int = a raw hipStream_t (0 = None)."""
    if stream is None:
        return None
    if hasattr(stream, "cuda_stream"):
        h = int(stream.cuda_stream)
        return C.c_void_p(h if h else STREAM_LEGACY_DEFAULT)
    return C.c_void_p(int(stream)) if int(stream) else None


def make_camera(cam) -> Camera:
    c = Camera()
    c.width, c.height, c.fu, c.fv, c.cu, c.cv = cam.w, cam.h, cam.fu, cam.fv, cam.cu, cam.cv
    c.distortion = cam.dist_type
    for i in range(4):
        c.d[i] = cam.d[i]
    return c


def make_pose(Cm, r) -> Pose:
    p = Pose()
    flat = np.asarray(Cm, dtype=np.float64).reshape(-1)
    for i in range(9):
        p.C[i] = float(flat[i])
    for i in range(3):
        p.r[i] = float(r[i])
    return p


def bow_vector(word_ids, word_weight, weighting=0, normalise_l1=True):
    """DBoW2 BowVector of a feature set from its word ids (okvfe_fbrisk_transform) and the
    vocabulary's word weights: (ascending word ids, values).  Host helper."""
    w = np.ascontiguousarray(word_ids, dtype=np.int32)
    ww = np.ascontiguousarray(word_weight, dtype=np.float64)
    ids = np.zeros(len(ww), dtype=np.int32)
    vals = np.zeros(len(ww), dtype=np.float64)
    n = C.c_int32()
    st = lib().okvfe_bow_vector(_p(w), len(w), _p(ww), len(ww), int(weighting), int(bool(normalise_l1)),
                                _p(ids), _p(vals), len(ww), C.byref(n))
    if st != OK:
        raise OkvfeError(st, "okvfe_bow_vector")
    return ids[:n.value].copy(), vals[:n.value].copy()


def set_heavy_kernel_chaining(mode: int):
    """okvfe_set_heavy_kernel_chaining: 0 off, 1 score kernels of all contexts in turn, 2 + describe."""
    st = lib().okvfe_set_heavy_kernel_chaining(int(mode))
    if st != OK:
        raise OkvfeError(st, "okvfe_set_heavy_kernel_chaining")


def popcnt_xor(a, b, n128=3) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().okvfe_popcnt_xor(_p(a), _p(b), int(n128)))


def scale_index(size) -> int:
    """Scale index (0..63) of the scale-invariant extractor for a keypoint diameter (host only)."""
    f = lib().okvfe_scale_index
    f.restype, f.argtypes = C.c_int32, [C.c_float]
    return int(f(float(size)))


def build_awareness_maps(cam):
    c = make_camera(cam)
    rays = np.zeros((cam.h, cam.w, 3), dtype=np.float32)
    jac = np.zeros((cam.h, cam.w, 6), dtype=np.float32)
    st = lib().okvfe_build_awareness_maps(C.byref(c), _p(rays), _p(jac))
    if st != OK:
        raise OkvfeError(st, lib().okvfe_last_error(None).decode())
    return rays, jac


def format_keypoint_lines(state_id, camera_idx, keypoints, descriptors) -> bytes:
    """Map-file text records (okvis::Component::save format)."""
    kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(descriptors, dtype=np.uint8)
    need = C.c_size_t()
    st = lib().okvfe_format_keypoint_lines(C.c_uint64(state_id), C.c_uint64(camera_idx), _p(kps),
                                           _p(desc), len(kps), None, C.c_size_t(0), C.byref(need))
    if st != OK:
        raise OkvfeError(st, "okvfe_format_keypoint_lines")
    buf = C.create_string_buffer(max(need.value, 1))
    st = lib().okvfe_format_keypoint_lines(C.c_uint64(state_id), C.c_uint64(camera_idx), _p(kps),
                                           _p(desc), len(kps), buf, C.c_size_t(need.value),
                                           C.byref(need))
    if st != OK:
        raise OkvfeError(st, "okvfe_format_keypoint_lines")
    return buf.raw[:need.value]


def parse_keypoint_lines(text: bytes, cap=4096):
    """Returns (state_id, camera_idx, keypoints, descriptors, bytes_consumed)."""
    kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
    desc = np.zeros((cap, DESC_BYTES), dtype=np.uint8)
    sid, cam, n, used = C.c_uint64(), C.c_uint64(), C.c_int32(), C.c_size_t()
    st = lib().okvfe_parse_keypoint_lines(text, C.c_size_t(len(text)), C.byref(sid), C.byref(cam),
                                          _p(kps), _p(desc), cap, C.byref(n), C.byref(used))
    if st != OK:
        raise OkvfeError(st, "okvfe_parse_keypoint_lines")
    return sid.value, cam.value, kps[:n.value].copy(), desc[:n.value].copy(), used.value


def fbrisk_mean(descriptors) -> np.ndarray:
    d = np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, DESC_BYTES)
    out = np.zeros(DESC_BYTES, dtype=np.uint8)
    st = lib().okvfe_fbrisk_mean(_p(d), len(d), _p(out))
    if st != OK:
        raise OkvfeError(st, "okvfe_fbrisk_mean")
    return out


def camera_overlap(cam, other, R_other_cam, want_mask=False):
    c, o = make_camera(cam), make_camera(other)
    R = (C.c_double * 9)(*[float(v) for v in np.asarray(R_other_cam).reshape(-1)])
    mask = np.zeros((cam.h, cam.w), dtype=np.uint8) if want_mask else None
    has = C.c_int32()
    st = lib().okvfe_camera_overlap(C.byref(c), C.byref(o), R, _p(mask), C.byref(has))
    if st != OK:
        raise OkvfeError(st, lib().okvfe_last_error(None).decode())
    return (bool(has.value), mask) if want_mask else bool(has.value)


class Frontend:
    """One okvfe context = the detector + extractor + matcher of one camera stream on one GPU.

    Mirrors the reference's per-camera objects (okvis_frontend/src/Frontend.cpp:2405-2413):
    constructor arguments are the brisk::ScaleSpaceFeatureDetector / BriskDescriptorExtractor
    ones plus image size and batch capacity.
    """

    def __init__(self, width, height, uniformity_radius, octaves, absolute_threshold,
                 max_keypoints, rotation_invariant=True, scale_invariant=False,
                 match_threshold=60, max_batch=1, num_cameras=1, device=0, max_candidates=0,
                 score_type=SCORE_HARRIS, box_scale=1.0):
        cfg = Config(ABI_VERSION, device, width, height, max_batch, num_cameras,
                     float(uniformity_radius), int(octaves), int(absolute_threshold),
                     int(max_keypoints), int(bool(rotation_invariant)), int(bool(scale_invariant)),
                     int(match_threshold), int(max_candidates), int(score_type), float(box_scale))
        self._h = C.c_void_p()
        # row capacity per image: a scale space (octaves > 0) has 2 * octaves layers and every layer
        # may deliver max_keypoints (okvfe_device_outputs.max_keypoints reports the same number)
        self.w, self.h, self.max_batch = width, height, max_batch
        self.max_keypoints = int(max_keypoints) * max(1, 2 * int(octaves))
        st = lib().okvfe_create(C.byref(cfg), C.byref(self._h))
        if st != OK:
            raise OkvfeError(st, lib().okvfe_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().okvfe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise OkvfeError(st, lib().okvfe_last_error(self._h).decode())

    # -- setup ----------------------------------------------------------------------------
    def set_camera(self, slot, cam):
        c = make_camera(cam)
        self._check(lib().okvfe_set_camera(self._h, int(slot), C.byref(c)))

    def set_camera_maps(self, slot, rays, jac, fu):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        jac = np.ascontiguousarray(jac, dtype=np.float32)
        self._check(lib().okvfe_set_camera_maps(self._h, int(slot), _p(rays), _p(jac),
                                                C.c_float(fu)))

    # -- host-buffer API ------------------------------------------------------------------
    def detect(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        cap = self.max_keypoints
        kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_int32()
        self._check(lib().okvfe_detect(self._h, _p(image), C.c_size_t(image.strides[0]), _p(kps),
                                       cap, C.byref(n)))
        return kps[:n.value].copy()

    def detect_ahead(self, image, cam=-1, gravity=None):
        """okvfe_detect_ahead: detect() whose compute() on the same image is answered from the kept result."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self._ahead_image = image  # the pairing is by pointer: keep the buffer alive for compute()
        cap = self.max_keypoints
        kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_int32()
        g = None if gravity is None else (C.c_float * 3)(*[float(v) for v in gravity])
        self._check(lib().okvfe_detect_ahead(self._h, _p(image), C.c_size_t(image.strides[0]), int(cam), g,
                                             _p(kps), cap, C.byref(n)))
        return kps[:n.value].copy()

    def harris_byte_mover_device(self, images_ptr, n_images, stream=None):
        """Diagnostic: the fused score kernel's loads and stores without its arithmetic."""
        self._check(lib().okvfe_harris_byte_mover_device(self._h, _p(images_ptr), int(n_images), _s(stream)))

    def detect_describe(self, image, cam=-1, gravity=None):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        cap = self.max_keypoints
        kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((cap, DESC_BYTES), dtype=np.uint8)
        bp = np.zeros((cap, 3), dtype=np.float64)
        bpv = np.zeros(cap, dtype=np.uint8)
        n = C.c_int32()
        g = None if gravity is None else (C.c_float * 3)(*[float(v) for v in gravity])
        self._check(lib().okvfe_detect_describe(self._h, _p(image), C.c_size_t(image.strides[0]),
                                                int(cam), g, _p(kps), _p(desc), _p(bp), _p(bpv),
                                                cap, C.byref(n)))
        k = n.value
        return kps[:k].copy(), desc[:k].copy(), bp[:k].copy(), bpv[:k].copy()

    def compute(self, image, keypoints, cam=-1, gravity=None):
        """cv::DescriptorExtractor::compute: describe the given keypoints (some may be removed)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE).copy()
        n_in = len(kps)
        if n_in == 0:
            kps = np.zeros(1, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((max(n_in, 1), DESC_BYTES), dtype=np.uint8)
        bp = np.zeros((max(n_in, 1), 3), dtype=np.float64)
        bpv = np.zeros(max(n_in, 1), dtype=np.uint8)
        n = C.c_int32()
        g = None if gravity is None else (C.c_float * 3)(*[float(v) for v in gravity])
        self._check(lib().okvfe_compute(self._h, _p(image), C.c_size_t(image.strides[0]), int(cam),
                                        g, _p(kps), n_in, _p(desc), _p(bp), _p(bpv), C.byref(n)))
        k = n.value
        return kps[:k].copy(), desc[:k].copy(), bp[:k].copy(), bpv[:k].copy()

    # -- device-resident API --------------------------------------------------------------
    def harris_score_device(self, images_ptr, n_images, scores_ptr, stream=None):
        self._check(lib().okvfe_harris_score_device(self._h, _p(images_ptr), int(n_images),
                                                    _p(scores_ptr), _s(stream)))

    def detect_batch_device(self, images_ptr, n_images, stream=None):
        self._check(lib().okvfe_detect_batch_device(self._h, _p(images_ptr), int(n_images),
                                                    _s(stream)))

    def describe_batch_device(self, images_ptr, n_images, cam_ids=None, gravity=None, stream=None):
        ids = None if cam_ids is None else np.ascontiguousarray(cam_ids, dtype=np.int32)
        g = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float32)
        self._check(lib().okvfe_describe_batch_device(self._h, _p(images_ptr), int(n_images),
                                                      _p(ids), _p(g), _s(stream)))

    def detect_describe_batch_device(self, images_ptr, n_images, cam_ids=None, gravity=None,
                                     stream=None):
        ids = None if cam_ids is None else np.ascontiguousarray(cam_ids, dtype=np.int32)
        g = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float32)
        self._check(lib().okvfe_detect_describe_batch_device(self._h, _p(images_ptr),
                                                             int(n_images), _p(ids), _p(g),
                                                             _s(stream)))

    def detect_describe_batch_host(self, images_host_ptr, n_images, cam_ids=None, gravity=None,
                                   stream=None):
        """images_host_ptr: address of n_images contiguous images in (preferably pinned) HOST
        memory; the copy overlaps the previous batch's kernels (okvfe_detect_describe_batch_host)."""
        ids = None if cam_ids is None else np.ascontiguousarray(cam_ids, dtype=np.int32)
        g = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float32)
        self._check(lib().okvfe_detect_describe_batch_host(self._h, _p(images_host_ptr),
                                                           int(n_images), _p(ids), _p(g),
                                                           _s(stream)))

    def set_keep_score_map(self, keep: bool = True):
        """okvfe_set_keep_score_map: single-scale Harris detection writes no score map by default
        (device_outputs().scores is then null); keep=True restores it for the following calls."""
        self._check(lib().okvfe_set_keep_score_map(self._h, int(bool(keep))))

    def set_internal_lanes(self, lanes: int = 0):
        """okvfe_set_internal_lanes: slices a batch call is cut into, each on a stream of the context's own
        (0 = automatic, 1 = off)."""
        self._check(lib().okvfe_set_internal_lanes(self._h, int(lanes)))

    def lanes_join(self, stream=None):
        """okvfe_lanes_join: `stream` (None: the context's own) waits for every pipelined lane (set_internal_lanes(-k))"""
        self._check(lib().okvfe_lanes_join(self._h, _s(stream)))

    def set_fp64_reduction(self, eigen_tree: bool = True):
        """okvfe_set_fp64_reduction: order of the 3-term FP64 sums of the gate chain -- Eigen's x0 + (x1 + x2)
        (default) or left to right; device-wide."""
        self._check(lib().okvfe_set_fp64_reduction(self._h, 1 if eigen_tree else 0))

    def device_outputs(self) -> DeviceOutputs:
        out = DeviceOutputs()
        self._check(lib().okvfe_get_device_outputs(self._h, C.byref(out)))
        return out

    def download(self, index):
        cap = self.max_keypoints
        kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((cap, DESC_BYTES), dtype=np.uint8)
        bp = np.zeros((cap, 3), dtype=np.float64)
        bpv = np.zeros(cap, dtype=np.uint8)
        n = C.c_int32()
        self._check(lib().okvfe_download_image_result(self._h, int(index), _p(kps), _p(desc),
                                                      _p(bp), _p(bpv), cap, C.byref(n)))
        k = n.value
        return kps[:k].copy(), desc[:k].copy(), bp[:k].copy(), bpv[:k].copy()

    def check_capacity(self, n_images):
        """Raises OkvfeError(ERR_CAPACITY) if an NMS candidate list of the last batch overflowed."""
        first = C.c_int32(-1)
        self._check(lib().okvfe_check_capacity(self._h, int(n_images), C.byref(first)))

    def match_stereo_batch_device(self, pairs, matches_ptr, stream=None):
        arr = pairs if isinstance(pairs, C.Array) else (StereoPair * len(pairs))(*pairs)
        self._check(lib().okvfe_match_stereo_batch_device(self._h, arr, len(pairs),
                                                          _p(matches_ptr), _s(stream)))

    # -- matching, host buffers -----------------------------------------------------------
    def match_stereo(self, desc0, kp0, bp0, bpv0, desc1, kp1, bp1, bpv1, T0, T1, f0, f1):
        n0, n1 = len(kp0), len(kp1)
        out = np.zeros(max(n0, 1), dtype=STEREO_MATCH_DTYPE)
        arrs = [np.ascontiguousarray(a) for a in (desc0, kp0, bp0, bpv0, desc1, kp1, bp1, bpv1)]
        P0, P1 = make_pose(*T0), make_pose(*T1)
        self._check(lib().okvfe_match_stereo(self._h, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]),
                                             _p(arrs[3]), n0, _p(arrs[4]), _p(arrs[5]),
                                             _p(arrs[6]), _p(arrs[7]), n1, C.byref(P0),
                                             C.byref(P1), C.c_double(f0), C.c_double(f1),
                                             _p(out)))
        return out[:n0]

    def match_motion_stereo(self, cam, desc0, kp0, bp0, bpv0, skip0, desc1, kp1, bp1, bpv1, matched1,
                            T0, T1):
        n0, n1 = len(kp0), len(kp1)
        out = np.zeros(max(n0, 1), dtype=MOTION_MATCH_DTYPE)
        arrs = [None if a is None else np.ascontiguousarray(a)
                for a in (desc0, kp0, bp0, bpv0, skip0, desc1, kp1, bp1, bpv1, matched1)]
        c = make_camera(cam)
        P0, P1 = make_pose(*T0), make_pose(*T1)
        self._check(lib().okvfe_match_motion_stereo(
            self._h, C.byref(c), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), n0,
            _p(arrs[5]), _p(arrs[6]), _p(arrs[7]), _p(arrs[8]), _p(arrs[9]), n1, C.byref(P0),
            C.byref(P1), _p(out)))
        return out[:n0]

    def match_to_map(self, desc, kps, use, proj, desc_begin, pool, repr_thr):
        n, nl = len(kps), len(desc_begin) - 1
        bl = np.zeros(max(n, 1), dtype=np.int32)
        bd = np.zeros(max(n, 1), dtype=np.int32)
        arrs = [np.ascontiguousarray(a) for a in (desc, kps, np.asarray(use, dtype=np.uint8),
                                                 np.asarray(proj, dtype=np.float64),
                                                 np.asarray(desc_begin, dtype=np.int32),
                                                 np.asarray(pool, dtype=np.uint8))]
        self._check(lib().okvfe_match_to_map(self._h, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), n,
                                             _p(arrs[3]), _p(arrs[4]), nl, _p(arrs[5]),
                                             C.c_double(repr_thr), _p(bl), _p(bd)))
        return bl[:n], bd[:n]

    def match_to_map_uninitialised(self, desc, bp, use, previous, desc_begin, pool, e0_W, r0_W, T1,
                                   focal):
        n, nl = len(desc), len(desc_begin) - 1
        bl = np.zeros(max(n, 1), dtype=np.int32)
        bd = np.zeros(max(n, 1), dtype=np.int32)
        hp = np.zeros((max(n, 1), 4), dtype=np.float64)
        hs = np.zeros(max(n, 1), dtype=np.uint8)
        ctr = C.c_int32()
        arrs = [np.ascontiguousarray(desc, dtype=np.uint8), np.ascontiguousarray(bp, dtype=np.float64),
                np.ascontiguousarray(use, dtype=np.uint8), np.ascontiguousarray(previous, dtype=np.int32),
                np.ascontiguousarray(desc_begin, dtype=np.int32), np.ascontiguousarray(pool, dtype=np.uint8),
                np.ascontiguousarray(e0_W, dtype=np.float64), np.ascontiguousarray(r0_W, dtype=np.float64)]
        P1 = make_pose(*T1)
        self._check(lib().okvfe_match_to_map_uninitialised(
            self._h, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), n, _p(arrs[4]), nl,
            _p(arrs[5]), _p(arrs[6]), _p(arrs[7]), C.byref(P1), C.c_double(focal), _p(bl), _p(bd),
            _p(hp), _p(hs), C.byref(ctr)))
        return bl[:n], bd[:n], hp[:n], hs[:n], ctr.value

    def hamming_candidates(self, A, B, thr, cap=None):
        A = np.ascontiguousarray(A, dtype=np.uint8)
        B = np.ascontiguousarray(B, dtype=np.uint8)
        cap = len(A) * len(B) if cap is None else cap
        out = np.zeros(max(cap, 1), dtype=CAND_DTYPE)
        n = C.c_int32()
        st = lib().okvfe_hamming_candidates(self._h, _p(A), len(A), _p(B), len(B), int(thr),
                                            _p(out), int(cap), C.byref(n))
        if st == ERR_CAPACITY:
            return out[:cap].copy(), n.value
        self._check(st)
        return out[:n.value].copy(), n.value

    def hamming_argmin(self, A, B, thr):
        A = np.ascontiguousarray(A, dtype=np.uint8)
        B = np.ascontiguousarray(B, dtype=np.uint8)
        bj = np.zeros(max(len(A), 1), dtype=np.int32)
        bd = np.zeros(max(len(A), 1), dtype=np.uint32)
        self._check(lib().okvfe_hamming_argmin(self._h, _p(A), len(A), _p(B), len(B),
                                               C.c_uint32(int(thr)), _p(bj), _p(bd)))
        return bj[:len(A)], bd[:len(A)]

    def match_to_map_landmarks(self, cam, hp_W, quality, obs_begin, obs_pose, obs_desc, obs_bp, poses,
                               T_WC1, repr_threshold, exclusive, desc, kps, use):
        """Frontend::matchToMap from the raw landmark table (projection + view pooling + 3-D match on
        the device).  poses: list of (C, r).  Returns (best_landmark, best_dist, pool dict)."""
        hp = np.ascontiguousarray(hp_W, dtype=np.float64).reshape(-1, 4)
        q = np.ascontiguousarray(quality, dtype=np.float64)
        ob = np.ascontiguousarray(obs_begin, dtype=np.int32)
        op = np.ascontiguousarray(obs_pose, dtype=np.int32)
        od = np.ascontiguousarray(obs_desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        obp = np.ascontiguousarray(obs_bp, dtype=np.float64).reshape(-1, 3)
        P = (Pose * max(len(poses), 1))(*[make_pose(*p) for p in poses])
        nl = len(hp)
        t = LandmarkTable(nl, len(op), len(poses), _p(hp).value, _p(q).value, _p(ob).value,
                          _p(op).value if len(op) else None, _p(od).value if len(od) else None,
                          _p(obp).value if len(obp) else None, C.addressof(P))
        pool = {"status": np.zeros(max(nl, 1), np.int32), "n_desc": np.zeros(max(nl, 1), np.int32),
                "obs_rows": np.zeros((max(nl, 1), 3), np.int32), "projection": np.zeros((max(nl, 1), 2)),
                "e_W": np.zeros((max(nl, 1), 2, 3)), "r_W": np.zeros((max(nl, 1), 2, 3))}
        lp = LandmarkPool(*[_p(pool[k]).value for k in ("status", "n_desc", "obs_rows", "projection",
                                                         "e_W", "r_W")])
        d = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        kk = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        u = np.ascontiguousarray(use, dtype=np.uint8)
        n = len(kk)
        lm = np.zeros(max(n, 1), dtype=np.int32)
        bd = np.zeros(max(n, 1), dtype=np.int32)
        T1 = make_pose(*T_WC1)
        self._check(lib().okvfe_match_to_map_landmarks(
            self._h, int(cam), C.byref(t), C.byref(T1), C.c_double(repr_threshold), int(bool(exclusive)),
            _p(d), _p(kk), _p(u), n, C.byref(lp), _p(lm), _p(bd)))
        return lm[:n], bd[:n], {k: v[:nl] for k, v in pool.items()}

    def verify_place_match(self, landmark_desc, desc_begin, frame_desc):
        """Frontend::verifyRecognisedPlace descriptor matching, all landmarks in one launch."""
        pool = np.ascontiguousarray(landmark_desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        db = np.ascontiguousarray(desc_begin, dtype=np.int32)
        fd = np.ascontiguousarray(frame_desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        n = len(db) - 1
        k_min = np.zeros(max(n, 1), dtype=np.int32)
        d_min = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(lib().okvfe_verify_place_match(self._h, _p(pool), _p(db), n, _p(fd), len(fd),
                                                   _p(k_min), _p(d_min)))
        return k_min[:n], d_min[:n]

    def fbrisk_transform(self, desc, node_desc, child_begin, child_index, node_word):
        """DBoW2 vocabulary descent (FBrisk trait): word id and leaf node per descriptor."""
        d = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        nd = np.ascontiguousarray(node_desc, dtype=np.uint8).reshape(-1, DESC_BYTES)
        cb = np.ascontiguousarray(child_begin, dtype=np.int32)
        ci = np.ascontiguousarray(child_index, dtype=np.int32)
        nw = np.ascontiguousarray(node_word, dtype=np.int32)
        words = np.zeros(max(len(d), 1), dtype=np.int32)
        leaves = np.zeros(max(len(d), 1), dtype=np.int32)
        self._check(lib().okvfe_fbrisk_transform(self._h, _p(d), len(d), _p(nd), len(nd), _p(cb), _p(ci),
                                                 _p(nw), _p(words), _p(leaves)))
        return words[:len(d)], leaves[:len(d)]

    def bow_query_l1(self, db_begin, db_ids, db_values, q_ids, q_values):
        """DBoW2 database query with L1 scoring against all entries: score per entry (-1 = no
        common word)."""
        b = np.ascontiguousarray(db_begin, dtype=np.int32)
        ids = np.ascontiguousarray(db_ids, dtype=np.int32)
        vals = np.ascontiguousarray(db_values, dtype=np.float64)
        qi = np.ascontiguousarray(q_ids, dtype=np.int32)
        qv = np.ascontiguousarray(q_values, dtype=np.float64)
        n = len(b) - 1
        scores = np.zeros(max(n, 1), dtype=np.float64)
        self._check(lib().okvfe_bow_query_l1(self._h, _p(b), _p(ids), _p(vals), n, _p(qi), _p(qv), len(qi),
                                             _p(scores)))
        return scores[:n]

    # -- stage profiling (HIP events on the launch stream) --------------------------------
    def profile_enable(self, on=True, stages=None):
        """on=True times every stage; stages=("harris", ...) only those (fewer event records)."""
        flag = int(bool(on))
        if on and stages:
            flag = 0
            for name in stages:
                flag |= 1 << (8 + list(STAGES).index(name))
        self._check(lib().okvfe_profile_enable(self._h, flag))

    def profile_read(self):
        ms = (C.c_double * len(STAGES))()
        n = (C.c_int32 * len(STAGES))()
        self._check(lib().okvfe_profile_read(self._h, ms, n))
        return {STAGES[i]: (ms[i], n[i]) for i in range(len(STAGES))}

    # -- sampling pattern as data ----------------------------------------------------------
    def get_pattern(self) -> PatternData:
        p = PatternData()
        self._check(lib().okvfe_get_pattern(self._h, C.byref(p)))
        return p

    def set_pattern(self, pattern: PatternData):
        self._check(lib().okvfe_set_pattern(self._h, C.byref(pattern)))

    def pattern_kernel_class(self) -> int:
        """0 / 1: the fast descriptor kernels (11 x 11 / 21 x 21 row slots); 2: the all-modes kernel (include/okvfe.h)"""
        return int(lib().okvfe_pattern_kernel_class(self._h))

    # -- device-resident, batched map matchers (frame f = gather block f) -----------------
    @staticmethod
    def make_map_device(n_landmarks, desc_begin_ptr, pool_ptr, projections_ptr=None, e0_ptr=None,
                        r0_ptr=None) -> MapDevice:
        m = MapDevice()
        m.n_landmarks = int(n_landmarks)
        m.desc_begin, m.pool = int(desc_begin_ptr), int(pool_ptr) if pool_ptr else None
        m.projections = int(projections_ptr) if projections_ptr else None
        m.e0_W = int(e0_ptr) if e0_ptr else None
        m.r0_W = int(r0_ptr) if r0_ptr else None
        return m

    def match_to_map_blocks_device(self, blocks_ptr, n_frames, use_ptr, map_dev, repr_thr, best_lm_ptr,
                                   best_d_ptr, stream=None):
        self._check(lib().okvfe_match_to_map_blocks_device(
            self._h, _p(blocks_ptr), int(n_frames), _p(use_ptr), C.byref(map_dev), C.c_double(repr_thr),
            _p(best_lm_ptr), _p(best_d_ptr), _s(stream)))

    def match_to_map_uninitialised_blocks_device(self, blocks_ptr, n_frames, use_ptr, previous_ptr, map_dev,
                                                 poses_T_WC1, focal, best_lm_ptr, best_d_ptr, hps_ptr,
                                                 hp_set_ptr, ctr_ptr, stream=None):
        P = (Pose * int(n_frames))(*[make_pose(*T) for T in poses_T_WC1])
        self._check(lib().okvfe_match_to_map_uninitialised_blocks_device(
            self._h, _p(blocks_ptr), int(n_frames), _p(use_ptr), _p(previous_ptr), C.byref(map_dev), P,
            C.c_double(focal), _p(best_lm_ptr), _p(best_d_ptr), _p(hps_ptr), _p(hp_set_ptr), _p(ctr_ptr),
            _s(stream)))

    def verify_place_blocks_device(self, blocks_ptr, n_frames, map_dev, k_min_ptr, dist_min_ptr, stream=None):
        self._check(lib().okvfe_verify_place_blocks_device(
            self._h, _p(blocks_ptr), int(n_frames), C.byref(map_dev), _p(k_min_ptr), _p(dist_min_ptr),
            _s(stream)))

    # -- gather blocks --------------------------------------------------------------------
    def gather_block_bytes(self) -> int:
        return int(lib().okvfe_gather_block_bytes(self._h))

    def pack_gather_block_device(self, index, block_ptr, stream=None):
        self._check(lib().okvfe_pack_gather_block_device(self._h, int(index), _p(block_ptr),
                                                         _s(stream)))

    def pack_gather_blocks_device(self, first, n, blocks_ptr, stream=None):
        self._check(lib().okvfe_pack_gather_blocks_device(self._h, int(first), int(n),
                                                          _p(blocks_ptr), _s(stream)))

    def match_stereo_blocks_batch_device(self, blocks0_ptr, blocks1_ptr, n_frames, T0, T1, f0, f1,
                                         matches_ptr, stream=None):
        P0, P1 = make_pose(*T0), make_pose(*T1)
        self._check(lib().okvfe_match_stereo_blocks_batch_device(
            self._h, _p(blocks0_ptr), _p(blocks1_ptr), int(n_frames), C.byref(P0), C.byref(P1),
            C.c_double(f0), C.c_double(f1), _p(matches_ptr), _s(stream)))

    def match_motion_stereo_blocks_device(self, cam, block0_ptr, block1_ptr, skip0_ptr,
                                          matched1_ptr, T0, T1, matches_ptr, stream=None):
        P0, P1 = make_pose(*T0), make_pose(*T1)
        self._check(lib().okvfe_match_motion_stereo_blocks_device(
            self._h, int(cam), _p(block0_ptr), _p(block1_ptr), _p(skip0_ptr), _p(matched1_ptr),
            C.byref(P0), C.byref(P1), _p(matches_ptr), _s(stream)))

    def match_stereo_blocks_device(self, block0_ptr, block1_ptr, T0, T1, f0, f1, matches_ptr,
                                   stream=None):
        P0, P1 = make_pose(*T0), make_pose(*T1)
        self._check(lib().okvfe_match_stereo_blocks_device(self._h, _p(block0_ptr), _p(block1_ptr),
                                                           C.byref(P0), C.byref(P1),
                                                           C.c_double(f0), C.c_double(f1),
                                                           _p(matches_ptr), _s(stream)))
