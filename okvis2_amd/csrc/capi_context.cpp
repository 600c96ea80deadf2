// capi_context.cpp -- C-ABI runtime of libokvfe.so: contexts, HBM workspaces, stream plumbing,
// cameras, the sampling pattern as data, stage profiling.
// Entry points and the reference interfaces they replace are documented in include/okvfe.h.
// There is no CPU fallback anywhere in this library: without a gfx950 device okvfe_create fails.
#include <atomic>

#include "okvfe_ctx.h"

using namespace okvfe;

namespace okvfe {
thread_local std::string g_create_error;


// OKVFE_SCORE_TOKEN=1: the score (+NMS) kernels of ALL contexts of the process on a device run one
// after the other, in the order they were enqueued (each waits for the previous one's completion
// event), while everything downstream of them is free to overlap.  With several contexts fed in
// turn from several streams this staggers the pipelines: the VALU-bound score kernel of one batch
// runs next to the latency-bound sort / greedy selection / matching of another one instead of next
// to another score kernel.  OKVFE_SCORE_TOKEN=2 also chains the describe kernels (for callers that
// enqueue detect for all contexts, then describe for all contexts).
std::mutex g_token_mutex;
hipEvent_t g_score_token[kMaxTokenDevices] = {};
std::atomic<int> g_heavy_chain_mode{0};  // okvfe_set_heavy_kernel_chaining
int score_token_mode() { return g_heavy_chain_mode.load(std::memory_order_relaxed); }

okvfe_status fail(okvfe_ctx* ctx, okvfe_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return st;
}



okvfe_status ensure_scratch(okvfe_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return OKVFE_OK;
  if (ctx->scratch) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return OKVFE_OK;
}

okvfe_status ensure_pinned(okvfe_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->h_pinned_bytes) return OKVFE_OK;
  if (ctx->h_pinned) {
    HIP_TRY(ctx, hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->h_pinned_bytes = 0;
  }
  void* p = nullptr;
  HIP_TRY(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault));
  ctx->h_pinned = static_cast<uint8_t*>(p);
  ctx->h_pinned_bytes = bytes;
  ctx->h_pinned_dev = nullptr;
  if (hipHostGetDevicePointer(&ctx->h_pinned_dev, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    ctx->h_pinned_dev = nullptr;  // copies then go through hipMemcpyAsync
  }
  return OKVFE_OK;
}

// (re)sizes a ring; only called while nothing of the ring is in flight (creation, or after a drain)
okvfe_status ring_reserve(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, size_t slot_bytes) {
  slot_bytes = align_up(std::max<size_t>(slot_bytes, 256), 256);
  if (slot_bytes <= r->slot_bytes) return OKVFE_OK;
  for (int i = 0; i < okvfe_ctx::ParamRing::kRingSlots; ++i)
    if (r->pending[i]) {
      if (r->recorded[i]) {
        HIP_TRY(ctx, hipEventSynchronize(r->done[i]));
      } else if (hipStreamSynchronize(r->used_on[i]) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(ctx, hipDeviceSynchronize());
      }
      r->pending[i] = false;
    }
  if (r->h) HIP_TRY(ctx, hipHostFree(r->h));
  if (r->d) HIP_TRY(ctx, hipFree(r->d));
  r->h = nullptr;
  r->d = nullptr;
  r->slot_bytes = 0;
  void* q = nullptr;
  HIP_TRY(ctx, hipHostMalloc(&q, slot_bytes * okvfe_ctx::ParamRing::kRingSlots, hipHostMallocDefault));
  r->h = static_cast<uint8_t*>(q);
  HIP_TRY(ctx, hipMalloc(&q, slot_bytes * okvfe_ctx::ParamRing::kRingSlots));
  r->d = static_cast<uint8_t*>(q);
  r->slot_bytes = slot_bytes;
  for (auto& e : r->done)
    if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return OKVFE_OK;
}

// takes the next slot, copies `bytes` from src through the pinned half to the device half on
// stream s (asynchronous: returns at once) and hands back the device address
okvfe_status ring_upload(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, const void* src, size_t bytes,
                         hipStream_t s, void** d_out, int* slot_out, int32_t* zero_dev, int n_zero,
                         bool* zeroed) {
  okvfe_status st = ring_reserve(ctx, r, bytes);
  if (st != OKVFE_OK) return st;
  const int slot = (int)(r->next++ % okvfe_ctx::ParamRing::kRingSlots);
  if (r->pending[slot]) {
    // released slots carry an event behind their last reader; a slot whose call returned early
    // (no ring_release) is waited for through its stream
    if (r->recorded[slot]) {
      HIP_TRY(ctx, hipEventSynchronize(r->done[slot]));
    } else if (hipStreamSynchronize(r->used_on[slot]) != hipSuccess) {  // (the caller's stream may be gone)
      (void)hipGetLastError();
      HIP_TRY(ctx, hipDeviceSynchronize());
    }
    r->pending[slot] = false;
  }
  uint8_t* h = r->h + (size_t)slot * r->slot_bytes;
  uint8_t* d = r->d + (size_t)slot * r->slot_bytes;
  std::memcpy(h, src, bytes);
  // a copy KERNEL reading the pinned slot keeps the hand-over inside the compute queue (k_util.hip);
  // OKVFE_PARAM_MEMCPY=1 restores the DMA-engine copy for A/B
  static const bool dma = lab_env("OKVFE_PARAM_MEMCPY") != nullptr;
  void* h_dev = nullptr;
  if (!dma && hipHostGetDevicePointer(&h_dev, h, 0) == hipSuccess && h_dev) {
    launch_param_copy(d, h_dev, bytes, zero_dev, n_zero, s);
    HIP_TRY(ctx, hipGetLastError());
    if (zeroed) *zeroed = n_zero > 0;
  } else {
    (void)hipGetLastError();
    HIP_TRY(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
  }
  // guarded from here on: the slot is pending on stream s; ring_release records the event behind its
  // last reader (an event here as well cost ~6 us of idle GPU per upload: an event record is a
  // barrier packet with a system-scope release)
  r->pending[slot] = true;
  r->recorded[slot] = false;
  r->used_on[slot] = s;
  *d_out = d;
  *slot_out = slot;
  return OKVFE_OK;
}

// marks the end of the slot's consumers on stream s
okvfe_status ring_release(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, int slot, hipStream_t s) {
  if (slot < 0) return OKVFE_OK;
  // test knob: behave like a call that returned before its release (the slot is then waited for
  // through its stream when it comes round again)
  static const bool skip = lab_env("OKVFE_TEST_SKIP_RING_RELEASE") != nullptr;
  if (skip) return OKVFE_OK;
  HIP_TRY(ctx, hipEventRecord(r->done[slot], s));
  r->pending[slot] = true;
  r->recorded[slot] = true;
  r->used_on[slot] = s;
  return OKVFE_OK;
}

void ring_destroy(okvfe_ctx::ParamRing* r) {
  for (auto& e : r->done)
    if (e) (void)hipEventDestroy(e);
  if (r->h) (void)hipHostFree(r->h);
  if (r->d) (void)hipFree(r->d);
}

DeviceCamera to_device_camera(const okvfe_camera& c) {
  DeviceCamera d{};
  d.fu = c.fu; d.fv = c.fv; d.cu = c.cu; d.cv = c.cv;
  d.one_over_fu = 1.0 / c.fu;
  d.one_over_fv = 1.0 / c.fv;
  for (int i = 0; i < 4; ++i) d.d[i] = c.d[i];
  d.distortion = c.distortion;
  return d;
}

PairParams to_pair_params(const okvfe_stereo_pair& p) {
  PairParams q{};
  q.image0 = p.image0;
  q.image1 = p.image1;
  std::memcpy(q.C0, p.T_WC0.C, sizeof(q.C0));
  std::memcpy(q.r0, p.T_WC0.r, sizeof(q.r0));
  std::memcpy(q.C1, p.T_WC1.C, sizeof(q.C1));
  std::memcpy(q.r1, p.T_WC1.r, sizeof(q.r1));
  q.f0 = p.f0;
  q.f1 = p.f1;
  // sigma = max(size0/f0, size1/f1) * 0.125 with size = 12 (single scale): Frontend.cpp:2035
  const double s0 = 12.0 / p.f0, s1 = 12.0 / p.f1;
  const double sigma = std::max(s0, s1) * 0.125;
  q.cos26 = std::cos(2.6 * sigma);  // stereo_triangulation.cpp:86,121
  q.cos6 = std::cos(6.0 * sigma);   // stereo_triangulation.cpp:127
  q.cls = nullptr;
  return q;
}

// keypoint size of scale-space layer l: 12 * scale(l) (exact in float)
double layer_keypoint_size(int l) {
  const int num = (l & 1) ? 3 << ((l - 1) / 2) : 1 << (l / 2);
  return 12.0 * (double)num / ((l & 1) ? 2.0 : 1.0);
}
// size-class table [2][kSizeClasses][kSizeClasses]: cos(2.6 sigma) then cos(6 sigma).
// stereo: sigma = max(size0/f0, size1/f1) * 0.125 (Frontend.cpp:2035);
// motion: sigma = size0/f0 * 0.125 (Frontend.cpp:1834)
void fill_class_table(double* t, double f0, double f1, bool motion) {
  for (int c0 = 0; c0 < kSizeClasses; ++c0)
    for (int c1 = 0; c1 < kSizeClasses; ++c1) {
      const double s0 = layer_keypoint_size(c0) / f0, s1 = layer_keypoint_size(c1) / f1;
      const double sigma = (motion ? s0 : std::max(s0, s1)) * 0.125;
      t[c0 * kSizeClasses + c1] = std::cos(2.6 * sigma);
      t[kSizeClasses * kSizeClasses + c0 * kSizeClasses + c1] = std::cos(6.0 * sigma);
    }
}

// host keypoints -> size classes present?  Fails unless every size is 12 * scale(octave).
okvfe_status check_size_classes(okvfe_ctx* ctx, const okvfe_keypoint* kp, int n, bool* multi) {
  for (int i = 0; kp && i < n; ++i) {
    const int l = kp[i].octave;
    if (l < 0 || l >= kSizeClasses || (double)kp[i].size != layer_keypoint_size(l))
      return fail(ctx, OKVFE_ERR_UNSUPPORTED, "keypoint %d: size %f is not 12 * scale(octave %d)", i, kp[i].size, l);
    if (l != 0) *multi = true;
  }
  return OKVFE_OK;
}


// NULL = the context's own non-blocking stream; OKVFE_STREAM_LEGACY_DEFAULT = the HIP legacy
// default (null) stream, which is what torch.cuda.default_stream() is: its handle is 0 and could
// not be told apart from "no stream given" otherwise.
hipStream_t pick_stream_raw(okvfe_ctx* ctx, void* stream) {
  if (!stream) return ctx->stream;
  if (stream == OKVFE_STREAM_LEGACY_DEFAULT) return static_cast<hipStream_t>(nullptr);
  return static_cast<hipStream_t>(stream);
}
hipStream_t pick_stream(okvfe_ctx* ctx, void* stream) {
  hipStream_t s = pick_stream_raw(ctx, stream);
  if (ctx->lanes_pending && ctx->join_done) {  // pipelined lanes: whatever this call queues on s sees their results
    if (hipStreamWaitEvent(s, ctx->join_done, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
    }
    ctx->lanes_pending = false;
  }
  return s;
}
okvfe_status lanes_join_host(okvfe_ctx* ctx) {
  if (!ctx->lanes_pending) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  if (ctx->join_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->join_stream));
  ctx->lanes_pending = false;
  return OKVFE_OK;
}

void layer_size(int w, int h, int l, int* lw, int* lh) {  // oracle: orc_layer_size
  if (l == 0) {
    *lw = w; *lh = h;
  } else if (l == 1) {
    *lw = (w / 3) * 2; *lh = (h / 3) * 2;
  } else {
    int pw, ph;
    layer_size(w, h, l - 2, &pw, &ph);
    *lw = pw / 2; *lh = ph / 2;
  }
}
void layer_scale(int l, int* num, int* den) {  // oracle: orc_layer_scale
  if ((l & 1) == 0) {
    *num = 1 << (l / 2); *den = 1;
  } else {
    *num = 3 << ((l - 1) / 2); *den = 2;
  }
}

BlockLayout block_layout(int kp_cap) {
  BlockLayout L;
  L.o_count = 0;
  L.o_kps = 16;
  L.o_desc = align_up(L.o_kps + (size_t)kp_cap * sizeof(okvfe_keypoint), 16);
  L.o_bp = align_up(L.o_desc + (size_t)kp_cap * OKVFE_DESC_BYTES, 16);
  L.o_bpv = align_up(L.o_bp + (size_t)kp_cap * 3 * sizeof(double), 16);
  L.total = align_up(L.o_bpv + (size_t)kp_cap, 256);
  return L;
}
}  // namespace okvfe

extern "C" {

int32_t okvfe_abi_version(void) { return OKVFE_ABI_VERSION; }

okvfe_status okvfe_set_heavy_kernel_chaining(int32_t mode) {
  if (mode < 0 || mode > 2) return OKVFE_ERR_INVALID_ARGUMENT;
  g_heavy_chain_mode.store(mode, std::memory_order_relaxed);
  return OKVFE_OK;
}

okvfe_status okvfe_set_keep_score_map(okvfe_ctx* ctx, int32_t keep) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  ctx->keep_score_map = keep != 0;
  return OKVFE_OK;
}

okvfe_status okvfe_set_internal_lanes(okvfe_ctx* ctx, int32_t lanes) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (lanes < -8 || lanes > 8) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_internal_lanes: %d (-8 .. 8)", lanes);
  okvfe_status st = lanes_join_host(ctx);
  if (st != OKVFE_OK) return st;
  ctx->internal_lanes = lanes < 0 ? -lanes : lanes;
  ctx->lanes_pipelined = lanes < -1;
  return OKVFE_OK;
}

okvfe_status okvfe_lanes_join(okvfe_ctx* ctx, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  (void)pick_stream(ctx, stream);
  return OKVFE_OK;
}

okvfe_status okvfe_set_fp64_reduction(okvfe_ctx* ctx, int32_t order) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (order != OKVFE_SUM3_LEFT_TO_RIGHT && order != OKVFE_SUM3_EIGEN_TREE)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_fp64_reduction: order %d", order);
  if (hipSetDevice(ctx->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return fail(ctx, OKVFE_ERR_DEVICE, "okvfe_set_fp64_reduction: device %d", ctx->cfg.device);
  if (!okvfe::set_fp64_tree_match(order) || !okvfe::set_fp64_tree_map(order))
    return fail(ctx, OKVFE_ERR_DEVICE, "okvfe_set_fp64_reduction: writing the device flag failed");
  return OKVFE_OK;
}

const char* okvfe_last_error(const okvfe_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

uint32_t okvfe_popcnt_xor(const uint8_t* a, const uint8_t* b, int32_t n128) {
  uint32_t c = 0;
  for (int i = 0; i < 2 * n128; ++i) {
    uint64_t x, y;
    std::memcpy(&x, a + 8 * i, 8);
    std::memcpy(&y, b + 8 * i, 8);
    c += static_cast<uint32_t>(__builtin_popcountll(x ^ y));
  }
  return c;
}

okvfe_status okvfe_build_awareness_maps(const okvfe_camera* camera, float* rays_hw3,
                                        float* jacobians_hw6) {
  if (!camera || !rays_hw3 || !jacobians_hw6 || camera->width <= 0 || camera->height <= 0)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_build_awareness_maps: bad argument");
  build_awareness_maps(*camera, rays_hw3, jacobians_hw6);
  return OKVFE_OK;
}

okvfe_status okvfe_camera_overlap(const okvfe_camera* camera, const okvfe_camera* other,
                                  const double R_other_cam[9], uint8_t* mask_hw, int32_t* has_overlap) {
  if (!camera || !other || !R_other_cam || !has_overlap || camera->width <= 0 || camera->height <= 0)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_camera_overlap: bad argument");
  *has_overlap = camera_overlap(*camera, *other, R_other_cam, mask_hw) ? 1 : 0;
  return OKVFE_OK;
}

}  // extern "C"

namespace {

// child = a detect-only layer context of a scale space (K1..K4 buffers only, any size >= 16)
okvfe_status create_impl(const okvfe_config* cfg, bool child, okvfe_ctx** out) {
  if (!cfg || !out) return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != OKVFE_ABI_VERSION)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: abi_version %d != %d",
                cfg->abi_version, OKVFE_ABI_VERSION);
  if (cfg->width < (child ? 16 : 64) || cfg->height < (child ? 16 : 64) || cfg->width > 4096 ||
      cfg->height > 4096 || (int64_t)cfg->width * cfg->height * 255 >= (int64_t)INT32_MAX)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT,
                "okvfe_create: image size %dx%d out of range (64..4096, w*h*255 < 2^31)", cfg->width,
                cfg->height);
  if (cfg->max_batch < 1 || cfg->num_cameras < 1 || cfg->num_cameras > 64)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: max_batch/num_cameras out of range");
  if (cfg->max_keypoints < 1 || cfg->max_keypoints > 4096)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: max_keypoints must be in 1..4096");
  if (cfg->absolute_threshold < 1)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: absolute_threshold must be >= 1");
  if (cfg->score_type != OKVFE_SCORE_HARRIS && cfg->score_type != OKVFE_SCORE_AGAST_9_16 &&
      cfg->score_type != OKVFE_SCORE_BRISK_SCALESPACE)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT,
                "okvfe_create: score_type %d (0 = Harris, 1 = AGAST 9-16, 2 = BRISK scale space)", cfg->score_type);
  if (cfg->match_threshold < 0 || cfg->match_threshold > 385)
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: match_threshold out of range");
  if (cfg->box_scale != 0.0f && !(cfg->box_scale > 0.25f && cfg->box_scale <= 2.5f))
    return fail(nullptr, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_create: box_scale %g (0 or 1 = the published boxes; range (0.25, 2.5])",
                (double)cfg->box_scale);
  if (cfg->octaves < 0 || cfg->octaves > 4)
    return fail(nullptr, OKVFE_ERR_UNSUPPORTED, "okvfe_create: octaves=%d out of range (0..4)", cfg->octaves);
  const int n_layers = cfg->octaves > 0 ? 2 * cfg->octaves : 1;
  if (cfg->octaves > 0) {
    int lw, lh;
    layer_size(cfg->width, cfg->height, n_layers - 1, &lw, &lh);
    if (lw < 16 || lh < 16)
      return fail(nullptr, OKVFE_ERR_UNSUPPORTED, "okvfe_create: %dx%d is too small for %d octaves (top layer %dx%d)",
                  cfg->width, cfg->height, cfg->octaves, lw, lh);
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, OKVFE_ERR_NO_DEVICE, "okvfe_create: no HIP device visible (no CPU fallback exists)");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, OKVFE_ERR_NO_DEVICE, "okvfe_create: device %d of %d not available", cfg->device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
    return fail(nullptr, OKVFE_ERR_NO_DEVICE, "okvfe_create: cannot query device %d", cfg->device);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, OKVFE_ERR_NO_DEVICE, "okvfe_create: device %d is %s; this library carries gfx950 code only",
                cfg->device, prop.gcnArchName);

  std::unique_ptr<okvfe_ctx> ctx(new okvfe_ctx());
  ctx->cfg = *cfg;
  ctx->w = cfg->width;
  ctx->h = cfg->height;
  ctx->B = cfg->max_batch;
  ctx->child = child;
  ctx->n_layers = n_layers;
  // row capacity per image: every layer of a scale space may deliver max_keypoints
  ctx->kp_cap = cfg->max_keypoints * n_layers;
  const int worst = (cfg->width / 2 + 1) * (cfg->height - 4);
  ctx->cand_cap = cfg->max_candidates > 0 ? std::min(cfg->max_candidates, worst) : worst;
  ctx->cand_cap = (std::max(ctx->cand_cap, 64) + 1) & ~1;  // even: the array doubles as 8-byte records
  ctx->ws_stride = 1;
  while (ctx->ws_stride < ctx->cand_cap) ctx->ws_stride <<= 1;
  ctx->mode_default = cfg->rotation_invariant ? kGradient : kUpright;
  if (cfg->uniformity_radius > 0.0f) {
    const float scaling = (float)(15.0 / (double)cfg->uniformity_radius);
    ctx->occ_rows = (int)((float)(ctx->h - 1) * scaling + 16.0f) + 17;
    ctx->occ_cols = (int)((float)(ctx->w - 1) * scaling + 16.0f) + 17;
  } else {
    ctx->occ_rows = ctx->occ_cols = 1;
  }
  ctx->occ_image_bytes = align_up((size_t)ctx->occ_rows * ctx->occ_cols, 256);

  okvfe_ctx* c = ctx.get();
  okvfe_status st = OKVFE_OK;
  auto run = [&]() -> okvfe_status {
    HIP_TRY(c, hipSetDevice(cfg->device));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t P = (size_t)c->w * c->h, B = (size_t)c->B, K = (size_t)c->kp_cap;
    okvfe_status s;
#define A(ptr, n) if ((s = dev_alloc(c, &c->ptr, (n))) != OKVFE_OK) return s
    const bool detects = n_layers == 1;  // a scale-space parent detects in its children
    const bool describes = !child;
    if (detects) {
      // AGAST score maps and maps of the unfused fall-back are dense; the fused Harris kernel
      // writes its slotted layout (okvfe_internal.h)
      c->score_layout = cfg->score_type == OKVFE_SCORE_HARRIS ? harris_nms_layout(c->w, c->h) : ScoreLayout{c->w, 0};
      c->live_layout = c->score_layout;
      A(d_scores, (size_t)c->score_layout.pitch * c->h * B);
      A(d_cand, (size_t)c->cand_cap * B);
      A(d_cand_count, 2 * B + (size_t)kFixListCap * B);  // candidate counts, fix-up counts, fix-up lists
      c->d_fix_count = c->d_cand_count + B;
      c->d_fix_list = c->d_cand_count + 2 * B;
      A(d_sort_ws, (size_t)c->ws_stride * B);
      A(d_occ, c->occ_image_bytes * B);
    }
    A(d_lut, kLutFloats);
    A(d_pattern, 1);
    if (cfg->scale_invariant && describes) A(d_scales, 1);
    A(d_kps_det, K * B + 1);  // + one record: launch_param_copy writes whole 16-byte chunks
    A(d_det_count, B);
    const size_t Kd = describes ? K : 1, Bd = describes ? B : 1;
    A(d_kps_tmp, Kd * Bd);
    A(d_desc_tmp, Kd * Bd * OKVFE_DESC_BYTES);
    A(d_valid_tmp, Kd * Bd);
    A(d_kps, Kd * Bd);
    A(d_desc, Kd * Bd * OKVFE_DESC_BYTES);
    A(d_bp, Kd * Bd * 3);
    A(d_bpv, Kd * Bd);
    A(d_count, Bd);
    A(d_cams, (size_t)cfg->num_cameras);
    A(d_rays_ptrs, (size_t)cfg->num_cameras);
    A(d_jac_ptrs, (size_t)cfg->num_cameras);
    A(d_img_stage, describes ? P : 1);
    A(d_match_stage, Kd);
#undef A
    if ((s = ring_reserve(c, &c->prm_ring, B * sizeof(ImageParams))) != OKVFE_OK) return s;
    if ((s = ring_reserve(c, &c->pair_ring, std::max<size_t>(1, B / 2) * sizeof(PairParams))) != OKVFE_OK) return s;
    float lut[kLutFloats];
    build_uniformity_lut(lut);
    build_pattern(&c->host_pattern);
    if (cfg->box_scale != 0.0f && cfg->box_scale != 1.0f) scale_pattern_boxes(&c->host_pattern, cfg->box_scale);
    HIP_TRY(c, hipMemcpy(c->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_pattern, &c->host_pattern, sizeof(Pattern), hipMemcpyHostToDevice));
    if (c->d_scales) {
      std::unique_ptr<PatternScales> ps(new PatternScales);
      build_pattern_scales(c->host_pattern, ps.get());
      HIP_TRY(c, hipMemcpy(c->d_scales, ps.get(), sizeof(PatternScales), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMemset(c->d_count, 0, Bd * sizeof(int32_t)));
    HIP_TRY(c, hipMemset(c->d_det_count, 0, B * sizeof(int32_t)));
    if (detects) HIP_TRY(c, hipMemset(c->d_cand_count, 0, 2 * B * sizeof(int32_t)));
    HIP_TRY(c, hipMemset(c->d_cams, 0, cfg->num_cameras * sizeof(DeviceCamera)));
    HIP_TRY(c, hipMemset(c->d_rays_ptrs, 0, cfg->num_cameras * sizeof(float*)));
    HIP_TRY(c, hipMemset(c->d_jac_ptrs, 0, cfg->num_cameras * sizeof(float*)));
    c->cam_rays.assign(cfg->num_cameras, nullptr);
    c->cam_jac.assign(cfg->num_cameras, nullptr);
    c->cam_fu.assign(cfg->num_cameras, 0.0f);
    c->cam_wide.assign(cfg->num_cameras, 0);
    c->cam_norms.assign(cfg->num_cameras, {});
    c->cam_aware_slow.assign(cfg->num_cameras, 0);
    c->h_cams.assign(cfg->num_cameras, DeviceCamera{});
    c->cam_has_intrinsics.assign(cfg->num_cameras, false);
    if (n_layers > 1) {
      // children: layer l at its own size, same detector parameters, single scale
      for (int l = 0; l < n_layers; ++l) {
        okvfe_config lc = *cfg;
        layer_size(cfg->width, cfg->height, l, &lc.width, &lc.height);
        lc.octaves = 0;
        lc.num_cameras = 1;
        okvfe_ctx* ch = nullptr;
        const okvfe_status cs = create_impl(&lc, true, &ch);
        if (cs != OKVFE_OK) return fail(c, cs, "layer %d (%dx%d): %s", l, lc.width, lc.height, g_create_error.c_str());
        ch->layer_child = true;
        c->layers.push_back(ch);
        c->layer_w.push_back(lc.width);
        c->layer_h.push_back(lc.height);
        uint8_t* img = nullptr;
        if (l > 0) {
          void* q = nullptr;
          HIP_TRY(c, hipMalloc(&q, (size_t)lc.width * lc.height * B));
          img = static_cast<uint8_t*>(q);
        }
        c->d_layer_img.push_back(img);
      }
      if (cfg->score_type == OKVFE_SCORE_BRISK_SCALESPACE) {  // FAST 5-8 map of c0: the layer below the first octave
        void* q = nullptr;
        HIP_TRY(c, hipMalloc(&q, (size_t)c->w * c->h * B * sizeof(int32_t)));
        c->d_virtual = static_cast<int32_t*>(q);
      }
      // single-scale views of the parent (score map of the full-resolution layer etc.)
      c->d_scores = c->layers[0]->d_scores;
      c->score_layout = c->layers[0]->score_layout;
      c->d_cand_count = c->layers[0]->d_cand_count;
      c->cand_cap = c->layers[0]->cand_cap;
    }
    return OKVFE_OK;
  };
  st = run();
  if (st != OKVFE_OK) {
    g_create_error = c->err;
    okvfe_destroy(ctx.release());
    return st;
  }
  *out = ctx.release();
  return OKVFE_OK;
}
}  // namespace

extern "C" {

okvfe_status okvfe_create(const okvfe_config* cfg, okvfe_ctx** out) { return create_impl(cfg, false, out); }

void okvfe_destroy(okvfe_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->cfg.device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->join_stream) {
    (void)hipStreamSynchronize(ctx->join_stream);
    (void)hipStreamDestroy(ctx->join_stream);
  }
  if (ctx->join_done) (void)hipEventDestroy(ctx->join_done);
  if (ctx->last_stream) (void)hipStreamSynchronize(ctx->last_stream);
  for (okvfe_ctx* ch : ctx->layers) okvfe_destroy(ch);
  for (okvfe_ctx* v : ctx->lane_ctx) {  // views: a stream and two chaining events each, no memory
    if (v->stream) {
      (void)hipStreamSynchronize(v->stream);
      (void)hipStreamDestroy(v->stream);
    }
    for (hipEvent_t ev : v->heavy_done) {
      if (!ev) continue;
      std::lock_guard<std::mutex> lock(g_token_mutex);
      for (auto& t : g_score_token)
        if (t == ev) t = nullptr;
      (void)hipEventDestroy(ev);
    }
    if (v->k1_done) (void)hipEventDestroy(v->k1_done);
    delete v;
  }
  for (hipEvent_t ev : ctx->lane_done)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
  if (ctx->score_stream && !ctx->lane_view) {
    (void)hipStreamSynchronize(ctx->score_stream);
    (void)hipStreamDestroy(ctx->score_stream);
  }
  for (hipEvent_t ev : ctx->layer_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->layer_fork) (void)hipEventDestroy(ctx->layer_fork);
  for (uint8_t* p : ctx->d_layer_img)
    if (p) (void)hipFree(p);
  if (ctx->d_virtual) (void)hipFree(ctx->d_virtual);
  for (auto& m : ctx->map_perm)
    if (m.d) (void)hipFree(m.d);
  for (void* p : ctx->allocs) (void)hipFree(p);
  for (float* p : ctx->cam_rays)
    if (p) (void)hipFree(p);
  for (float* p : ctx->cam_jac)
    if (p) (void)hipFree(p);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_result) (void)hipHostFree(ctx->h_result);
  ring_destroy(&ctx->prm_ring);
  ring_destroy(&ctx->pair_ring);
  ring_destroy(&ctx->cls_ring);
  if (ctx->feed_stream) (void)hipStreamSynchronize(ctx->feed_stream);
  for (int i = 0; i < 2; ++i) {
    if (ctx->d_feed[i]) (void)hipFree(ctx->d_feed[i]);
    if (ctx->feed_copied[i]) (void)hipEventDestroy(ctx->feed_copied[i]);
    if (ctx->feed_consumed[i]) (void)hipEventDestroy(ctx->feed_consumed[i]);
  }
  if (ctx->feed_stream) (void)hipStreamDestroy(ctx->feed_stream);
  for (auto& e : ctx->prof_events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t ev : ctx->heavy_done) {
    if (!ev) continue;
    std::lock_guard<std::mutex> lock(g_token_mutex);
    for (auto& t : g_score_token)
      if (t == ev) t = nullptr;
    (void)hipEventDestroy(ev);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// which descriptor kernel suits camera `cam` under the installed pattern (re-run by okvfe_set_pattern)
static void refresh_camera_patch_stats(okvfe_ctx* ctx, int cam) {
  const std::vector<float>& norms = ctx->cam_norms[cam];
  size_t seen = 0, large = 0, slow = 0;
  for (size_t i = 0; i + 1 < norms.size(); i += 2) {
    ++seen;
    if (!describe_patch_fits(norms[i], norms[i + 1], ctx->host_pattern.border)) ++large;
    if (describe_aware_patch_class(norms[i], norms[i + 1], ctx->host_pattern.reach) > 1) ++slow;
  }
  // (the 96-register instantiation with its 7.5 KB buffers pays when MANY patches need bands: measured with the
  // camera-aware-only six-wave form as the alternative -- 57 % / 60 % of the pixels (640x480 at fu 350, RealSense
  // D455): 1.13 against 1.33 ms, 0.53 against 0.58; 30 % (TUM-VI 512 / 1024): 0.27 against 0.22 ms, 0.26 against 0.27)
  ctx->cam_wide[cam] = seen > 0 && large * 5 > seen * 2;  // more than 40 %
  ctx->cam_aware_slow[cam] = seen == 0 || slow * 10 > seen;
}

okvfe_status okvfe_set_camera_maps(okvfe_ctx* ctx, int32_t cam, const float* rays_hw3,
                                   const float* jacobians_hw6, float fu) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  ctx->ahead.valid = false;  // a result computed ahead under the old extraction state must not answer okvfe_compute
  if (cam < 0 || cam >= ctx->cfg.num_cameras || !rays_hw3 || !jacobians_hw6 || !(fu > 0.0f))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_camera_maps: bad argument (cam=%d)", cam);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  const size_t P = (size_t)ctx->w * ctx->h;
  if (!ctx->cam_rays[cam]) {
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, P * 3 * sizeof(float)));
    ctx->cam_rays[cam] = static_cast<float*>(p);
    HIP_TRY(ctx, hipMalloc(&p, P * 6 * sizeof(float)));
    ctx->cam_jac[cam] = static_cast<float*>(p);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(ctx->cam_rays[cam], rays_hw3, P * 3 * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->cam_jac[cam], jacobians_hw6, P * 6 * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_rays_ptrs + cam, &ctx->cam_rays[cam], sizeof(float*), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_jac_ptrs + cam, &ctx->cam_jac[cam], sizeof(float*), hipMemcpyHostToDevice));
  ctx->cam_fu[cam] = fu;
  // how often does a keypoint's warped pattern exceed the one-piece LDS patch?  (row norms of the
  // image Jacobian bound the row norms of M = J [e_x e_y] / fu; every 8th pixel)
  std::vector<float>& norms = ctx->cam_norms[cam];
  norms.clear();
  for (int y = 0; y < ctx->h; y += 8)
    for (int x = 0; x < ctx->w; x += 8) {
      const float* J = jacobians_hw6 + ((size_t)y * ctx->w + x) * 6;
      const float nx = std::sqrt(J[0] * J[0] + J[1] * J[1] + J[2] * J[2]) / fu;
      const float ny = std::sqrt(J[3] * J[3] + J[4] * J[4] + J[5] * J[5]) / fu;
      if (!(nx == nx) || !(ny == ny)) continue;  // pixels without a ray
      norms.push_back(nx);
      norms.push_back(ny);
    }
  refresh_camera_patch_stats(ctx, cam);
  return OKVFE_OK;
}

okvfe_status okvfe_set_camera(okvfe_ctx* ctx, int32_t cam, const okvfe_camera* camera) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  ctx->ahead.valid = false;
  if (!camera || cam < 0 || cam >= ctx->cfg.num_cameras || camera->width != ctx->w ||
      camera->height != ctx->h || !(camera->fu > 0.0) || !(camera->fv > 0.0) ||
      camera->distortion < 0 || camera->distortion > 2)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_camera: bad argument (cam=%d)", cam);
  const size_t P = (size_t)ctx->w * ctx->h;
  std::vector<float> rays(P * 3), jac(P * 6);
  build_awareness_maps(*camera, rays.data(), jac.data());
  okvfe_status st = okvfe_set_camera_maps(ctx, cam, rays.data(), jac.data(), (float)camera->fu);
  if (st != OKVFE_OK) return st;
  ctx->h_cams[cam] = to_device_camera(*camera);
  ctx->cam_has_intrinsics[cam] = true;
  HIP_TRY(ctx, hipMemcpy(ctx->d_cams + cam, &ctx->h_cams[cam], sizeof(DeviceCamera), hipMemcpyHostToDevice));
  return OKVFE_OK;
}
okvfe_status okvfe_profile_enable(okvfe_ctx* ctx, int32_t enable) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  { const okvfe_status js = lanes_join_host(ctx); if (js != OKVFE_OK) return js; }
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& e : ctx->prof_events) {
    ctx->event_pool.push_back(e.a);
    ctx->event_pool.push_back(e.b);
  }
  ctx->prof_events.clear();
  ctx->prof_mask = enable == 1 ? 0xFFu : (enable > 1 ? ((uint32_t)enable >> 8) & 0xFFu : 0u);
  return OKVFE_OK;
}

okvfe_status okvfe_profile_read(okvfe_ctx* ctx, double total_ms[OKVFE_STAGE_COUNT],
                                int32_t launches[OKVFE_STAGE_COUNT]) {
  if (!ctx || !total_ms || !launches) return OKVFE_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  { const okvfe_status js = lanes_join_host(ctx); if (js != OKVFE_OK) return js; }
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < OKVFE_STAGE_COUNT; ++i) {
    total_ms[i] = 0.0;
    launches[i] = 0;
  }
  for (auto& e : ctx->prof_events) {
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e.a, e.b));
    total_ms[e.stage] += (double)ms;
    launches[e.stage] += 1;
  }
  return OKVFE_OK;
}

// ---- sampling pattern as data ---------------------------------------------------------------------
okvfe_status okvfe_get_pattern(const okvfe_ctx* ctx, okvfe_pattern* out) {
  if (!ctx || !out) return OKVFE_ERR_INVALID_ARGUMENT;
  const Pattern& P = ctx->host_pattern;
  std::memset(out, 0, sizeof(*out));
  out->n_points = P.n_points;
  std::memcpy(out->px, P.px, sizeof(out->px));
  std::memcpy(out->py, P.py, sizeof(out->py));
  std::memcpy(out->sigma_half, P.sigma_half, sizeof(out->sigma_half));
  out->n_short = P.n_short;
  std::memcpy(out->short_i, P.short_i, sizeof(out->short_i));
  std::memcpy(out->short_j, P.short_j, sizeof(out->short_j));
  out->n_long = P.n_long;
  std::memcpy(out->long_i, P.long_i, sizeof(out->long_i));
  std::memcpy(out->long_j, P.long_j, sizeof(out->long_j));
  std::memcpy(out->long_wdx, P.long_wdx, sizeof(out->long_wdx));
  std::memcpy(out->long_wdy, P.long_wdy, sizeof(out->long_wdy));
  out->border = P.border;
  return OKVFE_OK;
}

okvfe_status okvfe_set_pattern(okvfe_ctx* ctx, const okvfe_pattern* p) {
  if (!ctx || !p) return OKVFE_ERR_INVALID_ARGUMENT;
  ctx->ahead.valid = false;
  static_assert(OKVFE_PATTERN_POINTS == kPatternPoints && OKVFE_PATTERN_LONG_PAIRS == kMaxLongPairs, "pattern limits");
  if (p->n_points < 1 || p->n_points > kPatternPoints || p->n_short < 0 || p->n_short > OKVFE_PATTERN_SHORT_PAIRS ||
      p->n_long < 0 || p->n_long > kMaxLongPairs)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_pattern: %d points, %d short, %d long pairs (limits %d / %d / %d)",
                p->n_points, p->n_short, p->n_long, kPatternPoints, OKVFE_PATTERN_SHORT_PAIRS, kMaxLongPairs);
  float reach = 0.0f;
  for (int i = 0; i < p->n_points; ++i) {
    if (!(p->sigma_half[i] > 0.0f) || !std::isfinite(p->px[i]) || !std::isfinite(p->py[i]))
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_pattern: sample %d: half-width %g", i, (double)p->sigma_half[i]);
    reach = std::max(reach, std::sqrt(p->px[i] * p->px[i] + p->py[i] * p->py[i]) + p->sigma_half[i]);
  }
  if ((float)p->border < reach + 1.0f || p->border > 120)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_pattern: border %d, samples reach %.2f px", p->border, (double)reach);
  for (int b = 0; b < p->n_short; ++b)
    if (p->short_i[b] >= p->n_points || p->short_j[b] >= p->n_points)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_pattern: short pair %d names sample %d / %d", b, p->short_i[b], p->short_j[b]);
  for (int l = 0; l < p->n_long; ++l)
    if (p->long_i[l] >= p->n_points || p->long_j[l] >= p->n_points)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_set_pattern: long pair %d names sample %d / %d", l, p->long_i[l], p->long_j[l]);
  Pattern& P = ctx->host_pattern;  // rotation tables stay: they do not depend on the pattern
  P.n_points = p->n_points;
  std::memcpy(P.px, p->px, sizeof(P.px));
  std::memcpy(P.py, p->py, sizeof(P.py));
  std::memcpy(P.sigma_half, p->sigma_half, sizeof(P.sigma_half));
  P.n_short = p->n_short;
  std::memset(P.short_i, 0, sizeof(P.short_i));
  std::memset(P.short_j, 0, sizeof(P.short_j));
  std::memcpy(P.short_i, p->short_i, (size_t)p->n_short);
  std::memcpy(P.short_j, p->short_j, (size_t)p->n_short);
  P.n_long = p->n_long;
  std::memcpy(P.long_i, p->long_i, sizeof(P.long_i));
  std::memcpy(P.long_j, p->long_j, sizeof(P.long_j));
  std::memcpy(P.long_wdx, p->long_wdx, sizeof(P.long_wdx));
  std::memcpy(P.long_wdy, p->long_wdy, sizeof(P.long_wdy));
  P.border = p->border;
  P.reach = pattern_reach(P);
  for (int c = 0; c < ctx->cfg.num_cameras; ++c) refresh_camera_patch_stats(ctx, c);
  for (int i = 0; i < kPatternPoints; ++i) {  // same float sequence as build_pattern (host_tables.cpp)
    const float sg = i < P.n_points ? P.sigma_half[i] : 1.0f;
    float area = 4.0f * sg;
    area = area * sg;
    const int scaling = static_cast<int>(4194304.0f / area);
    const float s2 = static_cast<float>(scaling) * area;
    P.box_scaling[i] = scaling;
    P.box_scaling2[i] = static_cast<int>(s2 / 1024.0f);
  }
  fill_aware_lanes(&P);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  { const okvfe_status js = lanes_join_host(ctx); if (js != OKVFE_OK) return js; }
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(ctx->d_pattern, &P, sizeof(Pattern), hipMemcpyHostToDevice));
  if (ctx->d_scales) {  // the installed pattern is the base (index 17) of the scale ladder
    std::unique_ptr<PatternScales> ps(new PatternScales);
    build_pattern_scales(P, ps.get());
    HIP_TRY(ctx, hipMemcpy(ctx->d_scales, ps.get(), sizeof(PatternScales), hipMemcpyHostToDevice));
  }
  return OKVFE_OK;
}
}  // extern "C"
