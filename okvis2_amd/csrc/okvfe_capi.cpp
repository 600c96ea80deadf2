// okvfe_capi.cpp -- C-ABI runtime of libokvfe.so: contexts, HBM workspaces, stream plumbing.
// Entry points and the reference interfaces they replace are documented in include/okvfe.h.
// There is no CPU fallback anywhere in this file: without a gfx950 device okvfe_create fails.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>

#include "okvfe_internal.h"

using namespace okvfe;

namespace {
thread_local std::string g_create_error;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

struct okvfe_ctx {
  okvfe_config cfg{};
  hipStream_t stream = nullptr;
  hipEvent_t heavy_done[2] = {nullptr, nullptr};  // OKVFE_SCORE_TOKEN: after the score / describe kernel
  int detected_images = 0;  // images covered by the last okvfe_detect_batch_device
  std::string err;
  int w = 0, h = 0, B = 0, kp_cap = 0, cand_cap = 0, ws_stride = 0;
  int occ_rows = 0, occ_cols = 0;
  size_t occ_image_bytes = 0;
  int mode_default = kUpright;
  Pattern host_pattern{};

  std::vector<void*> allocs;
  int32_t* d_scores = nullptr;
  int32_t* d_virtual = nullptr;
  int32_t* d_map_perm = nullptr;  // okvfe_match_to_map_blocks_device: keypoint order per frame [frames][kp_cap]
  size_t map_perm_frames = 0;  // scale-space parent, OKVFE_SCORE_BRISK_SCALESPACE: FAST 5-8 map of layer 0
  ScoreLayout score_layout{0, 0};  // of d_scores: slotted where the fused score+NMS kernel applies
  ScoreLayout live_layout{0, 0};   // the layout the LAST score launch actually wrote (dense when the fused kernel refused the call)
  Candidate* d_cand = nullptr;
  int32_t* d_cand_count = nullptr;
  uint64_t* d_sort_ws = nullptr;
  uint8_t* d_occ = nullptr;
  float* d_lut = nullptr;
  Pattern* d_pattern = nullptr;
  PatternScales* d_scales = nullptr;  // scale_invariant extraction: the pattern at 64 scales
  okvfe_keypoint* d_kps_det = nullptr;
  int32_t* d_det_count = nullptr;
  okvfe_keypoint* d_kps_tmp = nullptr;
  uint8_t* d_desc_tmp = nullptr;
  uint8_t* d_valid_tmp = nullptr;
  okvfe_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  double* d_bp = nullptr;
  uint8_t* d_bpv = nullptr;
  int32_t* d_count = nullptr;
  ImageParams* d_prm = nullptr;  // current slot of prm_ring
  DeviceCamera* d_cams = nullptr;
  const float** d_rays_ptrs = nullptr;
  const float** d_jac_ptrs = nullptr;
  uint8_t* d_img_stage = nullptr;
  okvfe_stereo_match* d_match_stage = nullptr;
  // Per-call host parameters (ImageParams per image, PairParams per stereo pair) travel through
  // rings of pinned host slots + device slots: the call fills a pinned slot, enqueues ONE async
  // copy on its stream and the kernels read the device slot -- no host synchronisation.  A slot is
  // reused only after the event recorded behind its last consumer has completed (normally long
  // ago; the wait only bites when more than kRingSlots calls are in flight).
  struct ParamRing {
    static constexpr int kRingSlots = 8;
    uint8_t* h = nullptr;  // pinned, kRingSlots * slot_bytes
    uint8_t* d = nullptr;
    size_t slot_bytes = 0;
    hipEvent_t done[kRingSlots] = {};
    bool pending[kRingSlots] = {};   // in use by work enqueued on `used_on`
    bool recorded[kRingSlots] = {};  // ... and done[slot] was recorded behind its last reader (ring_release)
    hipStream_t used_on[kRingSlots] = {};
    unsigned next = 0;
  };
  ParamRing prm_ring, pair_ring, cls_ring;
  int prm_slot = -1;  // slot d_prm points into

  // scale space (octaves > 0): one detect-only child context per layer (K1..K4 at the layer's
  // size), layer images for l >= 1 owned here; this (parent) context keeps the merged keypoints
  // and everything from the descriptor stage on
  bool child = false;
  int n_layers = 1;
  std::vector<okvfe_ctx*> layers;
  std::vector<uint8_t*> d_layer_img;
  std::vector<int> layer_w, layer_h;

  // host-fed batches (okvfe_detect_describe_batch_host): two device image buffers filled by an
  // internal copy stream, so the PCIe copy of batch k+1 runs under the kernels of batch k
  uint8_t* d_feed[2] = {nullptr, nullptr};
  hipStream_t feed_stream = nullptr;
  hipEvent_t feed_copied[2] = {nullptr, nullptr}, feed_consumed[2] = {nullptr, nullptr};
  bool feed_busy[2] = {false, false};
  unsigned feed_next = 0;
  std::vector<float*> cam_rays, cam_jac;  // device maps per camera slot (nullptr = not set)
  std::vector<float> cam_fu;
  std::vector<uint8_t> cam_wide;  // camera-aware patches of this camera often exceed the LDS buffer (describe_kernel<5>)
  bool wide_patches = false;      // of the images of the current batch
  bool counters_cleared = false;  // upload_image_params zeroed d_cand_count on the call's stream
  bool fuse_setup = false;        // the current call describes what it detects: setup rides in the selection kernel
  bool setup_done = false;        // ... and did
  std::vector<DeviceCamera> h_cams;
  std::vector<bool> cam_has_intrinsics;
  int last_n_images = 0;
  hipStream_t last_stream = nullptr;

  // scratch for the explicit-array matchers (grown on demand)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  uint8_t* h_pinned = nullptr;  // pinned staging for the host-buffer API
  size_t h_pinned_bytes = 0;

  // stage profiling (okvfe_profile_*): event pairs per recorded stage launch
  uint32_t prof_mask = 0;  // bit s = stage s is timed
  struct StageEvents {
    int stage;
    hipEvent_t a, b;
  };
  std::vector<StageEvents> prof_events;
  std::vector<hipEvent_t> event_pool;
};

namespace {

// OKVFE_SCORE_TOKEN=1: the score (+NMS) kernels of ALL contexts of the process on a device run one
// after the other, in the order they were enqueued (each waits for the previous one's completion
// event), while everything downstream of them is free to overlap.  With several contexts fed in
// turn from several streams this staggers the pipelines: the VALU-bound score kernel of one batch
// runs next to the latency-bound sort / greedy selection / matching of another one instead of next
// to another score kernel.  OKVFE_SCORE_TOKEN=2 also chains the describe kernels (for callers that
// enqueue detect for all contexts, then describe for all contexts).
constexpr int kMaxTokenDevices = 64;
std::mutex g_token_mutex;
hipEvent_t g_score_token[kMaxTokenDevices] = {};
int score_token_mode() {  // 0 off, 1 score kernel only, 2 score and describe kernels
  static const int mode = [] {
    const char* e = getenv("OKVFE_SCORE_TOKEN");
    return e ? atoi(e) : 0;
  }();
  return mode;
}

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

#define HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail((ctx), e__ == hipErrorOutOfMemory ? OKVFE_ERR_OUT_OF_MEMORY               \
                                                    : OKVFE_ERR_DEVICE,                     \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

template <typename T>
okvfe_status dev_alloc(okvfe_ctx* ctx, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(ctx, hipMalloc(&q, std::max<size_t>(count * sizeof(T), 256)));
  ctx->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return OKVFE_OK;
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
                         hipStream_t s, void** d_out, int* slot_out, int32_t* zero_dev = nullptr, int n_zero = 0,
                         bool* zeroed = nullptr) {
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
  static const bool dma = getenv("OKVFE_PARAM_MEMCPY") != nullptr;
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
  static const bool skip = getenv("OKVFE_TEST_SKIP_RING_RELEASE") != nullptr;
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
constexpr size_t kClassTableDoubles = 2 * kSizeClasses * kSizeClasses;

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

// RAII-free stage timer: records an event pair around a launch when profiling is on
struct StageTimer {
  okvfe_ctx* ctx;
  hipStream_t s;
  int idx = -1;
  StageTimer(okvfe_ctx* c, int stage, hipStream_t st) : ctx(c), s(st) {
    if (!((c->prof_mask >> stage) & 1u) || c->prof_events.size() >= 65536) return;
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
      if (!c->event_pool.empty()) {
        e[i] = c->event_pool.back();
        c->event_pool.pop_back();
      } else if (hipEventCreate(&e[i]) != hipSuccess) {
        return;
      }
    }
    c->prof_events.push_back({stage, e[0], e[1]});
    idx = (int)c->prof_events.size() - 1;
    (void)hipEventRecord(e[0], s);
  }
  ~StageTimer() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof_events[idx].b, s);
  }
};

// NULL = the context's own non-blocking stream; OKVFE_STREAM_LEGACY_DEFAULT = the HIP legacy
// default (null) stream, which is what torch.cuda.default_stream() is: its handle is 0 and could
// not be told apart from "no stream given" otherwise.
hipStream_t pick_stream(okvfe_ctx* ctx, void* stream) {
  if (!stream) return ctx->stream;
  if (stream == OKVFE_STREAM_LEGACY_DEFAULT) return static_cast<hipStream_t>(nullptr);
  return static_cast<hipStream_t>(stream);
}

}  // namespace

extern "C" {

int32_t okvfe_abi_version(void) { return OKVFE_ABI_VERSION; }

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
      A(d_sort_ws, (size_t)c->ws_stride * B);
      A(d_occ, c->occ_image_bytes * B);
    }
    A(d_lut, kLutFloats);
    A(d_pattern, 1);
    if (cfg->scale_invariant && describes) A(d_scales, 1);
    A(d_kps_det, K * B);
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
  if (ctx->last_stream) (void)hipStreamSynchronize(ctx->last_stream);
  for (okvfe_ctx* ch : ctx->layers) okvfe_destroy(ch);
  for (uint8_t* p : ctx->d_layer_img)
    if (p) (void)hipFree(p);
  if (ctx->d_virtual) (void)hipFree(ctx->d_virtual);
  if (ctx->d_map_perm) (void)hipFree(ctx->d_map_perm);
  for (void* p : ctx->allocs) (void)hipFree(p);
  for (float* p : ctx->cam_rays)
    if (p) (void)hipFree(p);
  for (float* p : ctx->cam_jac)
    if (p) (void)hipFree(p);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
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

okvfe_status okvfe_set_camera_maps(okvfe_ctx* ctx, int32_t cam, const float* rays_hw3,
                                   const float* jacobians_hw6, float fu) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
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
  size_t seen = 0, large = 0;
  for (int y = 0; y < ctx->h; y += 8)
    for (int x = 0; x < ctx->w; x += 8) {
      const float* J = jacobians_hw6 + ((size_t)y * ctx->w + x) * 6;
      const float nx = std::sqrt(J[0] * J[0] + J[1] * J[1] + J[2] * J[2]) / fu;
      const float ny = std::sqrt(J[3] * J[3] + J[4] * J[4] + J[5] * J[5]) / fu;
      if (!(nx == nx) || !(ny == ny)) continue;  // pixels without a ray
      ++seen;
      if (!describe_patch_fits(nx, ny, ctx->host_pattern.border)) ++large;
    }
  ctx->cam_wide[cam] = seen > 0 && large * 20 > seen;  // more than 5 %
  return OKVFE_OK;
}

okvfe_status okvfe_set_camera(okvfe_ctx* ctx, int32_t cam, const okvfe_camera* camera) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
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

okvfe_status okvfe_harris_score_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                       int32_t* scores_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || !scores_dev || n_images < 0)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_harris_score_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  {
    hipStream_t s = pick_stream(ctx, stream);
    StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
    if (ctx->cfg.score_type != OKVFE_SCORE_HARRIS)
      launch_agast_score(images_dev, ctx->w, ctx->h, n_images, scores_dev, s);
    else
      launch_harris(images_dev, ctx->w, ctx->h, n_images, scores_dev, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  return OKVFE_OK;
}

okvfe_status okvfe_harris_byte_mover_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                            void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 0 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_harris_byte_mover_device: bad argument");
  if (ctx->cfg.score_type != OKVFE_SCORE_HARRIS || !ctx->d_scores || ctx->score_layout.strips < 1)
    return fail(ctx, OKVFE_ERR_UNSUPPORTED, "okvfe_harris_byte_mover_device: the fused score kernel does not apply to this context");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  bool ok;
  {
    StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
    ok = launch_harris_byte_mover(images_dev, ctx->w, ctx->h, n_images, ctx->d_scores, ctx->score_layout, s);
  }
  if (!ok) return fail(ctx, OKVFE_ERR_UNSUPPORTED, "okvfe_harris_byte_mover_device: image base or width not dword aligned");
  HIP_TRY(ctx, hipGetLastError());
  return OKVFE_OK;
}

static okvfe_status upload_image_params(okvfe_ctx* ctx, int n_images, const int32_t* cam_ids,
                                        const float* gravity, hipStream_t s, bool before_detect = false) {
  std::vector<ImageParams> prm(n_images);
  ctx->wide_patches = false;
  for (int i = 0; i < n_images; ++i) {
    ImageParams& p = prm[i];
    p.cam = cam_ids ? cam_ids[i] : -1;
    if (p.cam >= ctx->cfg.num_cameras)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "camera id %d out of range", p.cam);
    const bool aware = gravity != nullptr && p.cam >= 0;
    if (aware) {
      if (!ctx->cam_rays[p.cam])
        return fail(ctx, OKVFE_ERR_NOT_READY,
                    "camera-aware extraction requested for camera %d before okvfe_set_camera[_maps]", p.cam);
      p.mode = kCameraAware;
      p.dir[0] = gravity[3 * i];
      p.dir[1] = gravity[3 * i + 1];
      p.dir[2] = gravity[3 * i + 2];
      p.fu = ctx->cam_fu[p.cam];
      if (ctx->cam_wide[p.cam]) ctx->wide_patches = true;
    } else {
      p.mode = ctx->mode_default;
      p.dir[0] = 0.0f; p.dir[1] = 1.0f; p.dir[2] = 0.0f;
      p.fu = 1.0f;
    }
    if (p.cam >= 0 && !ctx->cam_has_intrinsics[p.cam]) p.cam = aware ? p.cam : -1;
  }
  // intrinsics are needed for back-projection; a slot with maps only (set_camera_maps) keeps its
  // cam id for the maps and gets invalid back-projections (DeviceCamera zeroed -> fu = 0)
  void* d = nullptr;
  // (single-scale detector: its candidate / fix-up counters are cleared by the same launch)
  ctx->counters_cleared = false;
  okvfe_status st = ring_upload(ctx, &ctx->prm_ring, prm.data(), n_images * sizeof(ImageParams), s, &d,
                                &ctx->prm_slot, before_detect && ctx->n_layers == 1 ? ctx->d_cand_count : nullptr,
                                before_detect && ctx->n_layers == 1 && ctx->d_cand_count ? 2 * ctx->B : 0,
                                &ctx->counters_cleared);
  if (st != OKVFE_OK) return st;
  ctx->d_prm = static_cast<ImageParams*>(d);
  return OKVFE_OK;
}

// ---- batch pipeline ----------------------------------------------------------------------------
// OKVFE_SCORE_TOKEN: see g_score_token.
namespace {
// The token mutex is held from the wait on the previous holder's event to the record of this
// launch's event, so two host threads can never chain on the same predecessor.
struct TokenScope {
  std::unique_lock<std::mutex> lock;
  bool on = false;
};
okvfe_status heavy_begin(okvfe_ctx* ctx, hipStream_t s, int which, TokenScope* t) {
  t->on = score_token_mode() > which && ctx->cfg.device >= 0 && ctx->cfg.device < kMaxTokenDevices;
  if (!t->on) return OKVFE_OK;
  t->lock = std::unique_lock<std::mutex>(g_token_mutex);
  hipEvent_t prev = g_score_token[ctx->cfg.device];
  if (prev) HIP_TRY(ctx, hipStreamWaitEvent(s, prev, 0));
  return OKVFE_OK;
}
okvfe_status heavy_end(okvfe_ctx* ctx, hipStream_t s, int which, TokenScope* t) {
  if (!t->on) return OKVFE_OK;
  hipEvent_t& ev = ctx->heavy_done[which];
  if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ev, s));
  g_score_token[ctx->cfg.device] = ev;
  t->lock.unlock();
  return OKVFE_OK;
}

// K1 + K2 of one layer context `L` (score map + NMS candidates), launched for `owner`
void layer_score_nms(okvfe_ctx* L, const uint8_t* images_dev, int n_images, hipStream_t s, bool* fused) {
  int32_t* d_fix_count = L->d_cand_count + L->B;  // [0, B) candidate counts, [B, 2B) flagged counts
  if (L->cfg.score_type != OKVFE_SCORE_HARRIS) {  // AGAST score map only; the stand-alone NMS follows
    launch_agast_score(images_dev, L->w, L->h, n_images, L->d_scores, s);
    *fused = false;
    L->live_layout = L->score_layout;
    return;
  }
  // a slotted score layout exists only where the fused kernel applies (decided at creation)
  *fused = L->score_layout.strips >= 1 &&
           launch_harris_nms(images_dev, L->w, L->h, n_images, L->d_scores, L->score_layout,
                             L->cfg.absolute_threshold, L->d_cand, L->cand_cap, L->d_cand_count, d_fix_count,
                             L->d_cand_count + 2 * (size_t)L->B, s);
  if (!*fused) launch_harris(images_dev, L->w, L->h, n_images, L->d_scores, s);
  // the unfused pair writes and reads a dense map (pitch w) into the same buffer: every later
  // reader of this call (selection, scale filter, okvfe_get_device_outputs) must follow it
  L->live_layout = *fused ? L->score_layout : ScoreLayout{L->w, 0};
}
void layer_nms_finish(okvfe_ctx* L, int n_images, hipStream_t s, bool fused) {
  int32_t* d_fix_count = L->d_cand_count + L->B;
  if (fused)
    launch_nms_fixup(L->d_scores, L->live_layout, L->w, L->h, n_images, L->cfg.absolute_threshold, L->d_cand, L->cand_cap,
                     L->d_cand_count, d_fix_count, L->d_cand_count + 2 * (size_t)L->B, s);
  else
    launch_nms(L->d_scores, L->w, L->h, n_images, L->cfg.absolute_threshold, L->d_cand, L->cand_cap,
               L->d_cand_count, s);
}
void layer_sort(okvfe_ctx* L, int n_images, hipStream_t s) {
  launch_sort(L->d_cand, L->cand_cap, L->d_cand_count, n_images, L->cfg.uniformity_radius, L->d_sort_ws, s);
}
void layer_select(okvfe_ctx* L, int n_images, hipStream_t s) {
  // detection + description in one call (single scale): the selection kernel also prepares the
  // extractor's per-keypoint inputs (describe_setup_dev.h)
  const DescribeSetup setup{L->d_pattern, L->d_prm, L->d_rays_ptrs, L->d_jac_ptrs, L->d_kps_tmp, L->d_desc_tmp,
                            L->d_valid_tmp, L->d_scales};
  const bool fuse = L->fuse_setup && L->n_layers == 1 && L->d_pattern && L->d_kps_tmp && L->d_prm;
  L->setup_done = launch_select(L->d_scores, L->live_layout, L->w, L->h, n_images, L->d_cand, L->cand_cap,
                                L->d_cand_count, L->cfg.uniformity_radius, L->cfg.max_keypoints, L->d_lut, L->d_occ,
                                L->occ_image_bytes, L->occ_rows, L->occ_cols, L->d_kps_det, L->kp_cap, L->d_det_count,
                                L->d_sort_ws, s, fuse ? &setup : nullptr);
}

// K1..K4: score map + NMS, sort, uniformity selection, sub-pixel -> d_kps_det / d_det_count.
// octaves > 0: the same per layer of the scale space (k_pyramid.hip), with the cross-layer maximum
// test between NMS and selection and the merge into image coordinates at the end.
okvfe_status detect_stage(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s) {
  TokenScope token;
  okvfe_status st;
  ctx->setup_done = false;  // set by this call's selection launch only (a failed earlier call must not leak it)
  if (ctx->n_layers == 1) {
    if (!ctx->counters_cleared)  // (cleared together with the parameter upload of the same call otherwise)
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_cand_count, 0, 2 * (size_t)ctx->B * sizeof(int32_t), s));
    ctx->counters_cleared = false;
    if ((st = heavy_begin(ctx, s, 0, &token)) != OKVFE_OK) return st;
    bool fused;
    {
      StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
      layer_score_nms(ctx, images_dev, n_images, s, &fused);
    }
    if ((st = heavy_end(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_NMS, s);
      layer_nms_finish(ctx, n_images, s, fused);
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SORT, s);
      layer_sort(ctx, n_images, s);
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SELECT, s);
      layer_select(ctx, n_images, s);
    }
  } else {
    const int L = ctx->n_layers;
    std::vector<const uint8_t*> img(L);
    std::vector<bool> fused(L);
    img[0] = images_dev;
    if ((st = heavy_begin(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
      for (int l = 1; l < L; ++l) {
        if (l == 1)
          launch_twothird(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_layer_img[1], s);
        else
          launch_halfsample(img[l - 2], ctx->layer_w[l - 2], ctx->layer_h[l - 2], n_images, ctx->d_layer_img[l], s);
        img[l] = ctx->d_layer_img[l];
      }
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        HIP_TRY(ctx, hipMemsetAsync(ch->d_cand_count, 0, 2 * (size_t)ch->B * sizeof(int32_t), s));
        bool f;
        layer_score_nms(ch, img[l], n_images, s, &f);
        fused[l] = f;
      }
      if (ctx->d_virtual) launch_fast58_score(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_virtual, s);
    }
    if ((st = heavy_end(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_NMS, s);
      for (int l = 0; l < L; ++l) layer_nms_finish(ctx->layers[l], n_images, s, fused[l]);
      // scale-space maxima: every layer against the finished score maps below and above
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        int sn, sd;
        layer_scale(l, &sn, &sd);
        const int32_t *below = nullptr, *above = nullptr;
        ScoreLayout lb{0, 0}, la{0, 0};
        int rb[2] = {1, 1}, ra[2] = {1, 1};
        auto ratio = [&](int m, int out[2]) {  // scale_l / scale_m, reduced
          int mn, md;
          layer_scale(m, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          out[0] = rn; out[1] = rd;
        };
        if (l > 0) { below = ctx->layers[l - 1]->d_scores; lb = ctx->layers[l - 1]->live_layout; ratio(l - 1, rb); }
        if (l == 0 && ctx->d_virtual) { below = ctx->d_virtual; lb = ScoreLayout{ctx->layer_w[0], 0}; }  // same grid: ratio 1
        if (l + 1 < L) { above = ctx->layers[l + 1]->d_scores; la = ctx->layers[l + 1]->live_layout; ratio(l + 1, ra); }
        launch_scale_filter(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, below, lb,
                            l > 0 ? ctx->layer_w[l - 1] : (below ? ctx->layer_w[0] : 0),
                            l > 0 ? ctx->layer_h[l - 1] : (below ? ctx->layer_h[0] : 0), rb[0], rb[1], above, la,
                            l + 1 < L ? ctx->layer_w[l + 1] : 0, l + 1 < L ? ctx->layer_h[l + 1] : 0, ra[0], ra[1], s);
      }
    }
    const bool brisk_ss = ctx->cfg.score_type == OKVFE_SCORE_BRISK_SCALESPACE;
    {
      StageTimer t(ctx, OKVFE_STAGE_SORT, s);
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        if (brisk_ss)  // always ordered (score desc, y, x): there is no uniformity radius to switch the sort on
          launch_sort(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, 1.0f, ch->d_sort_ws, s);
        else
          layer_sort(ch, n_images, s);
      }
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SELECT, s);
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        if (!brisk_ss) {
          layer_select(ch, n_images, s);
          continue;
        }
        // strongest maxima + continuous scale from the scores of the layers below and above
        int sn, sd;
        layer_scale(l, &sn, &sd);
        auto ratio2 = [&](int m, int out[2]) {
          int mn, md;
          layer_scale(m, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          out[0] = rn; out[1] = rd;
        };
        int rb2[2] = {1, 1}, ra2[2] = {1, 1};
        const int32_t* below = l == 0 ? ctx->d_virtual : ctx->layers[l - 1]->d_scores;
        const int wb = l == 0 ? ctx->layer_w[0] : ctx->layer_w[l - 1], hb = l == 0 ? ctx->layer_h[0] : ctx->layer_h[l - 1];
        if (l > 0) ratio2(l - 1, rb2);
        const int32_t* above = l + 1 < L ? ctx->layers[l + 1]->d_scores : nullptr;
        if (above) ratio2(l + 1, ra2);
        const double rel_b = (l & 1) ? 2.0 / 3.0 : 0.75, rel_a = (l & 1) ? 4.0 / 3.0 : 1.5;
        launch_brisk_refine(ch->d_scores, ch->w, ch->h, n_images, ch->cand_cap, ch->d_cand_count, ch->d_sort_ws,
                            ch->cfg.max_keypoints, below, wb, hb, rb2[0], rb2[1], above,
                            above ? ctx->layer_w[l + 1] : 0, above ? ctx->layer_h[l + 1] : 0, ra2[0], ra2[1], rel_b,
                            rel_a, ch->d_kps_det, ch->kp_cap, ch->d_det_count, s);
      }
      const okvfe_keypoint* kps[8];
      const int32_t* counts[8];
      float scale[8];
      for (int l = 0; l < L; ++l) {
        kps[l] = ctx->layers[l]->d_kps_det;
        counts[l] = ctx->layers[l]->d_det_count;
        int sn, sd;
        layer_scale(l, &sn, &sd);
        scale[l] = (float)sn / (float)sd;
      }
      launch_merge_layers(kps, counts, scale, L, ctx->cfg.max_keypoints, n_images, ctx->d_kps_det, ctx->kp_cap,
                          ctx->d_det_count, s);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_n_images = n_images;
  ctx->last_stream = s;
  return OKVFE_OK;
}

// first image of the last batch whose NMS candidate list overflowed in any layer (-1 = none);
// synchronises the last stream
okvfe_status find_overflow(okvfe_ctx* ctx, int first, int n_images, int* bad, int* count, int* cap) {
  *bad = -1;
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  std::vector<int32_t> counts(n_images);
  const int L = ctx->n_layers;
  for (int l = 0; l < L && *bad < 0; ++l) {
    okvfe_ctx* lc = L == 1 ? ctx : ctx->layers[l];
    HIP_TRY(ctx, hipMemcpy(counts.data(), lc->d_cand_count + first, (size_t)n_images * sizeof(int32_t),
                           hipMemcpyDeviceToHost));
    for (int i = 0; i < n_images; ++i)
      if (counts[i] > lc->cand_cap) {
        *bad = first + i;
        *count = counts[i];
        *cap = lc->cand_cap;
        break;
      }
  }
  return OKVFE_OK;
}

// K6 + compaction + back-projection of the keypoints detect_stage left in d_kps_det
okvfe_status describe_stage(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s) {
  const int w = ctx->w, h = ctx->h;
  TokenScope token;
  const bool setup_done = ctx->setup_done;  // consumed here, whatever happens below
  ctx->setup_done = false;
  okvfe_status st = heavy_begin(ctx, s, 1, &token);
  if (st != OKVFE_OK) return st;
  {
    StageTimer t(ctx, OKVFE_STAGE_DESCRIBE, s);
    launch_describe(images_dev, w, h, n_images, ctx->d_pattern, ctx->d_prm,
                    ctx->d_rays_ptrs, ctx->d_jac_ptrs, ctx->d_kps_det, ctx->kp_cap, ctx->d_det_count,
                    ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_scales, ctx->wide_patches, s, setup_done);
  }
  if ((st = heavy_end(ctx, s, 1, &token)) != OKVFE_OK) return st;
  {
    StageTimer t(ctx, OKVFE_STAGE_COMPACT, s);
    launch_compact(n_images, ctx->d_cams, ctx->d_prm, ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp,
                   ctx->d_det_count, ctx->kp_cap, ctx->d_kps, ctx->d_desc, ctx->d_bp, ctx->d_bpv,
                   ctx->d_count, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  const int slot = ctx->prm_slot;
  ctx->prm_slot = -1;
  return ring_release(ctx, &ctx->prm_ring, slot, s);  // the ImageParams slot has no reader after this
}
}  // namespace

okvfe_status okvfe_detect_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                       void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_batch_device: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->detected_images = 0;
  okvfe_status st = detect_stage(ctx, images_dev, n_images, pick_stream(ctx, stream));
  if (st == OKVFE_OK) ctx->detected_images = n_images;
  return st;
}

okvfe_status okvfe_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                         const int32_t* cam_ids, const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images != ctx->detected_images)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT,
                "okvfe_describe_batch_device: n_images=%d, but the last okvfe_detect_batch_device "
                "covered %d", n_images, ctx->detected_images);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s);
  if (st != OKVFE_OK) return st;
  return describe_stage(ctx, images_dev, n_images, s);
}

okvfe_status okvfe_detect_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                                int32_t n_images, const int32_t* cam_ids,
                                                const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe_batch_device: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s, true);
  if (st != OKVFE_OK) return st;
  static const bool no_fuse = getenv("OKVFE_NO_FUSED_SETUP") != nullptr;  // A/B knob
  ctx->fuse_setup = !no_fuse;
  st = detect_stage(ctx, images_dev, n_images, s);
  ctx->fuse_setup = false;
  if (st != OKVFE_OK) {
    ctx->setup_done = false;
    ctx->detected_images = 0;
    return st;
  }
  ctx->detected_images = n_images;
  return describe_stage(ctx, images_dev, n_images, s);
}

okvfe_status okvfe_detect_describe_batch_host(okvfe_ctx* ctx, const uint8_t* images_host, int32_t n_images,
                                              const int32_t* cam_ids, const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_host || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe_batch_host: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const size_t P = (size_t)ctx->w * ctx->h;
  if (!ctx->feed_stream) {
    // the feed state becomes visible only when ALL of it exists: a failed allocation leaves the
    // context as it was (the next call tries again) instead of a stream without buffers
    hipStream_t fs = nullptr;
    uint8_t* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      void* q = nullptr;
      e = hipMalloc(&q, P * (size_t)ctx->B);
      buf[i] = static_cast<uint8_t*>(q);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[2 * i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[2 * i + 1], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
      for (hipEvent_t v : ev)
        if (v) (void)hipEventDestroy(v);
      for (uint8_t* b : buf)
        if (b) (void)hipFree(b);
      if (fs) (void)hipStreamDestroy(fs);
      HIP_TRY(ctx, e);
    }
    for (int i = 0; i < 2; ++i) {
      ctx->d_feed[i] = buf[i];
      ctx->feed_copied[i] = ev[2 * i];
      ctx->feed_consumed[i] = ev[2 * i + 1];
    }
    ctx->feed_stream = fs;
  }
  const int slot = (int)(ctx->feed_next++ & 1u);
  // the buffer is rewritten only after the kernels of the batch that used it two calls ago
  if (ctx->feed_busy[slot]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->feed_stream, ctx->feed_consumed[slot], 0));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_feed[slot], images_host, P * (size_t)n_images, hipMemcpyHostToDevice,
                              ctx->feed_stream));
  HIP_TRY(ctx, hipEventRecord(ctx->feed_copied[slot], ctx->feed_stream));
  HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->feed_copied[slot], 0));
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s, true);
  if (st != OKVFE_OK) return st;
  ctx->fuse_setup = getenv("OKVFE_NO_FUSED_SETUP") == nullptr;
  st = detect_stage(ctx, ctx->d_feed[slot], n_images, s);
  ctx->fuse_setup = false;
  if (st != OKVFE_OK) {
    ctx->setup_done = false;
    ctx->detected_images = 0;
    return st;
  }
  ctx->detected_images = n_images;
  if ((st = describe_stage(ctx, ctx->d_feed[slot], n_images, s)) != OKVFE_OK) return st;
  HIP_TRY(ctx, hipEventRecord(ctx->feed_consumed[slot], s));
  ctx->feed_busy[slot] = true;
  return OKVFE_OK;
}

okvfe_status okvfe_get_device_outputs(okvfe_ctx* ctx, okvfe_device_outputs* out) {
  if (!ctx || !out) return OKVFE_ERR_INVALID_ARGUMENT;
  out->max_keypoints = ctx->kp_cap;
  out->counts = ctx->d_count;
  out->keypoints = ctx->d_kps;
  out->descriptors = ctx->d_desc;
  out->backproj = ctx->d_bp;
  out->backproj_valid = ctx->d_bpv;
  out->scores = ctx->d_scores;
  out->detect_counts = ctx->d_det_count;
  out->candidate_counts = ctx->d_cand_count;
  // the layout of the LAST batch's map (dense if that call took the unfused score + NMS kernels)
  const ScoreLayout& sl = ctx->n_layers > 1 ? ctx->layers[0]->live_layout : ctx->live_layout;
  out->score_pitch = sl.pitch;
  out->score_strips = sl.strips;
  return OKVFE_OK;
}

int32_t okvfe_scale_index(float keypoint_size) { return pattern_scale_index(keypoint_size); }

int32_t okvfe_score_column(const okvfe_ctx* ctx, int32_t x) {
  return ctx ? score_col(ctx->n_layers > 1 ? ctx->layers[0]->live_layout : ctx->live_layout, x) : x;
}

okvfe_status okvfe_check_capacity(okvfe_ctx* ctx, int32_t n_images, int32_t* first_overflowed) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_images < 0 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_check_capacity: n_images=%d (max_batch %d)", n_images, ctx->B);
  if (first_overflowed) *first_overflowed = -1;
  if (n_images == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  int bad = -1, count = 0, cap = 0;
  okvfe_status st = find_overflow(ctx, 0, n_images, &bad, &count, &cap);
  if (st != OKVFE_OK) return st;
  if (bad >= 0) {
    if (first_overflowed) *first_overflowed = bad;
    return fail(ctx, OKVFE_ERR_CAPACITY,
                "image %d produced %d NMS maxima, candidate capacity is %d (its keypoint list was left empty)", bad,
                count, cap);
  }
  return OKVFE_OK;
}

okvfe_status okvfe_download_image_result(okvfe_ctx* ctx, int32_t index, okvfe_keypoint* keypoints,
                                         uint8_t* descriptors, double* backproj,
                                         uint8_t* backproj_valid, int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (index < 0 || index >= ctx->last_n_images || !n_out || cap < 0)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_download_image_result: index %d of %d", index,
                ctx->last_n_images);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  int32_t counts[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpy(&counts[0], ctx->d_count + index, sizeof(int32_t), hipMemcpyDeviceToHost));
  {
    int bad = -1, cnt = 0, cap_c = 0;
    okvfe_status st = find_overflow(ctx, index, 1, &bad, &cnt, &cap_c);
    if (st != OKVFE_OK) return st;
    if (bad >= 0)
      return fail(ctx, OKVFE_ERR_CAPACITY, "image %d produced %d NMS maxima, candidate capacity is %d", index, cnt,
                  cap_c);
  }
  const int n = counts[0];
  *n_out = n;
  if (n > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  const size_t off = (size_t)index * ctx->kp_cap;
  if (n > 0) {
    if (keypoints)
      HIP_TRY(ctx, hipMemcpy(keypoints, ctx->d_kps + off, n * sizeof(okvfe_keypoint), hipMemcpyDeviceToHost));
    if (descriptors)
      HIP_TRY(ctx, hipMemcpy(descriptors, ctx->d_desc + off * OKVFE_DESC_BYTES, (size_t)n * OKVFE_DESC_BYTES,
                             hipMemcpyDeviceToHost));
    if (backproj)
      HIP_TRY(ctx, hipMemcpy(backproj, ctx->d_bp + off * 3, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (backproj_valid)
      HIP_TRY(ctx, hipMemcpy(backproj_valid, ctx->d_bpv + off, n, hipMemcpyDeviceToHost));
  }
  return OKVFE_OK;
}

static okvfe_status stage_image(okvfe_ctx* ctx, const uint8_t* image, size_t stride) {
  const size_t P = (size_t)ctx->w * ctx->h;
  if (stride < (size_t)ctx->w) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "stride %zu < width %d", stride, ctx->w);
  okvfe_status st = ensure_pinned(ctx, P);
  if (st != OKVFE_OK) return st;
  for (int y = 0; y < ctx->h; ++y) std::memcpy(ctx->h_pinned + (size_t)y * ctx->w, image + (size_t)y * stride, ctx->w);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_img_stage, ctx->h_pinned, P, hipMemcpyHostToDevice, ctx->stream));
  return OKVFE_OK;
}

okvfe_status okvfe_detect_describe(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                                   const float gravity_C[3], okvfe_keypoint* keypoints,
                                   uint8_t* descriptors, double* backproj, uint8_t* backproj_valid,
                                   int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  const int32_t cam_id = cam;
  st = okvfe_detect_describe_batch_device(ctx, ctx->d_img_stage, 1, &cam_id, (cam >= 0) ? gravity_C : nullptr,
                                          ctx->stream);
  if (st != OKVFE_OK) return st;
  return okvfe_download_image_result(ctx, 0, keypoints, descriptors, backproj, backproj_valid, cap, n_out);
}

okvfe_status okvfe_detect(okvfe_ctx* ctx, const uint8_t* image, size_t stride, okvfe_keypoint* keypoints,
                          int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  hipStream_t s = ctx->stream;
  if ((st = detect_stage(ctx, ctx->d_img_stage, 1, s)) != OKVFE_OK) return st;
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpy(&n, ctx->d_det_count, sizeof(int32_t), hipMemcpyDeviceToHost));
  {
    int bad = -1, cnt = 0, cap_c = 0;
    if ((st = find_overflow(ctx, 0, 1, &bad, &cnt, &cap_c)) != OKVFE_OK) return st;
    if (bad >= 0) return fail(ctx, OKVFE_ERR_CAPACITY, "%d NMS maxima, candidate capacity is %d", cnt, cap_c);
  }
  *n_out = n;
  if (n > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  if (n > 0 && keypoints)
    HIP_TRY(ctx, hipMemcpy(keypoints, ctx->d_kps_det, n * sizeof(okvfe_keypoint), hipMemcpyDeviceToHost));
  return OKVFE_OK;
}

okvfe_status okvfe_compute(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                           const float gravity_C[3], okvfe_keypoint* keypoints, int32_t n_in,
                           uint8_t* descriptors, double* backproj, uint8_t* backproj_valid,
                           int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out || n_in < 0 || (n_in > 0 && !keypoints))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_compute: bad argument");
  if (n_in > ctx->kp_cap)
    return fail(ctx, OKVFE_ERR_CAPACITY, "okvfe_compute: %d keypoints exceed max_keypoints %d", n_in, ctx->kp_cap);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  hipStream_t s = ctx->stream;
  const int32_t cam_id = cam;
  st = upload_image_params(ctx, 1, &cam_id, (cam >= 0) ? gravity_C : nullptr, s);
  if (st != OKVFE_OK) return st;
  const int w = ctx->w, h = ctx->h;
  if (n_in > 0)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kps_det, keypoints, n_in * sizeof(okvfe_keypoint), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_det_count, &n_in, sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_cand_count, 0, sizeof(int32_t), s));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable sources
  launch_describe(ctx->d_img_stage, w, h, 1, ctx->d_pattern, ctx->d_prm, ctx->d_rays_ptrs,
                  ctx->d_jac_ptrs, ctx->d_kps_det, ctx->kp_cap, ctx->d_det_count, ctx->d_kps_tmp,
                  ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_scales, ctx->wide_patches, s);
  launch_compact(1, ctx->d_cams, ctx->d_prm, ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_det_count,
                 ctx->kp_cap, ctx->d_kps, ctx->d_desc, ctx->d_bp, ctx->d_bpv, ctx->d_count, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_n_images = 1;
  ctx->last_stream = s;
  {
    const int slot = ctx->prm_slot;
    ctx->prm_slot = -1;
    if ((st = ring_release(ctx, &ctx->prm_ring, slot, s)) != OKVFE_OK) return st;
  }
  return okvfe_download_image_result(ctx, 0, keypoints, descriptors, backproj, backproj_valid, n_in, n_out);
}

okvfe_status okvfe_match_stereo_batch_device(okvfe_ctx* ctx, const okvfe_stereo_pair* pairs,
                                             int32_t n_pairs, okvfe_stereo_match* matches_dev,
                                             void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!pairs || !matches_dev || n_pairs < 1)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo_batch_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  std::vector<PairParams> pp(n_pairs);
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs[i].image0 < 0 || pairs[i].image0 >= ctx->B || pairs[i].image1 < 0 || pairs[i].image1 >= ctx->B ||
        !(pairs[i].f0 > 0.0) || !(pairs[i].f1 > 0.0))
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "pair %d: image index or focal length out of range", i);
    pp[i] = to_pair_params(pairs[i]);
  }
  okvfe_status st;
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    // one table per distinct (f0, f1); usually one for the whole call
    std::vector<double> tables;
    std::vector<std::pair<double, double>> seen;
    std::vector<int> which(n_pairs);
    for (int i = 0; i < n_pairs; ++i) {
      int j = 0;
      for (; j < (int)seen.size(); ++j)
        if (seen[j].first == pairs[i].f0 && seen[j].second == pairs[i].f1) break;
      if (j == (int)seen.size()) {
        seen.emplace_back(pairs[i].f0, pairs[i].f1);
        tables.resize(tables.size() + kClassTableDoubles);
        fill_class_table(tables.data() + (size_t)j * kClassTableDoubles, pairs[i].f0, pairs[i].f1, false);
      }
      which[i] = j;
    }
    void* d_tab = nullptr;
    if ((st = ring_upload(ctx, &ctx->cls_ring, tables.data(), tables.size() * sizeof(double), s, &d_tab,
                          &cls_slot)) != OKVFE_OK)
      return st;
    for (int i = 0; i < n_pairs; ++i)
      pp[i].cls = static_cast<const double*>(d_tab) + (size_t)which[i] * kClassTableDoubles;
  }
  void* d_pairs = nullptr;
  int slot = -1;
  st = ring_upload(ctx, &ctx->pair_ring, pp.data(), (size_t)n_pairs * sizeof(PairParams), s, &d_pairs, &slot);
  if (st != OKVFE_OK) return st;
  {
    StageTimer t(ctx, OKVFE_STAGE_MATCH, s);
    launch_match_stereo(static_cast<const PairParams*>(d_pairs), n_pairs, ctx->d_kps, ctx->d_desc, ctx->d_bp,
                        ctx->d_bpv, ctx->d_count, ctx->kp_cap, ctx->cfg.match_threshold, matches_dev, s);
  }
  if ((st = ring_release(ctx, &ctx->pair_ring, slot, s)) != OKVFE_OK) return st;
  if ((st = ring_release(ctx, &ctx->cls_ring, cls_slot, s)) != OKVFE_OK) return st;
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_match_stereo(okvfe_ctx* ctx, const uint8_t* desc0, const okvfe_keypoint* kp0,
                                const double* backproj0, const uint8_t* valid0, int32_t n0,
                                const uint8_t* desc1, const okvfe_keypoint* kp1, const double* backproj1,
                                const uint8_t* valid1, int32_t n1, const okvfe_pose* T_WC0,
                                const okvfe_pose* T_WC1, double f0, double f1, okvfe_stereo_match* matches) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n0 < 0 || n1 < 0 || !T_WC0 || !T_WC1 || !(f0 > 0.0) || !(f1 > 0.0) ||
      (n0 > 0 && (!desc0 || !backproj0 || !valid0 || !matches)) || (n1 > 0 && (!desc1 || !backproj1 || !valid1)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo: bad argument");
  // keypoint sizes select the triangulation sigma (Frontend.cpp:2031-2035): sizes must be
  // 12 * scale(octave); only a scale-space detector produces anything but 12
  bool multi = false;
  okvfe_status st = check_size_classes(ctx, kp0, n0, &multi);
  if (st == OKVFE_OK) st = check_size_classes(ctx, kp1, n1, &multi);
  if (st != OKVFE_OK) return st;
  if (multi && (!kp0 || !kp1)) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo: keypoints needed");
  if (n0 == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_cls = take(kClassTableDoubles * sizeof(double));
  const size_t o_d0 = take((size_t)n0 * 48), o_b0 = take((size_t)n0 * 24), o_v0 = take(n0),
               o_k0 = take((size_t)n0 * sizeof(okvfe_keypoint));
  const size_t o_d1 = take((size_t)n1 * 48), o_b1 = take((size_t)n1 * 24), o_v1 = take(n1),
               o_k1 = take((size_t)n1 * sizeof(okvfe_keypoint));
  const size_t o_out = take((size_t)n0 * sizeof(okvfe_stereo_match));
  if ((st = ensure_scratch(ctx, off)) != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  okvfe_stereo_pair sp{};
  sp.image0 = 0; sp.image1 = 0; sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1; sp.f0 = f0; sp.f1 = f1;
  PairParams pp = to_pair_params(sp);
  double table[kClassTableDoubles];
  if (multi) {
    fill_class_table(table, f0, f1, false);
    pp.cls = reinterpret_cast<const double*>(base + o_cls);
  }
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_pair, &pp, sizeof(pp)));
  if (multi) {
    HIP_TRY(ctx, up(o_cls, table, sizeof(table)));
    HIP_TRY(ctx, up(o_k0, kp0, (size_t)n0 * sizeof(okvfe_keypoint)));
    HIP_TRY(ctx, up(o_k1, kp1, (size_t)n1 * sizeof(okvfe_keypoint)));
  }
  HIP_TRY(ctx, up(o_d0, desc0, (size_t)n0 * 48));
  HIP_TRY(ctx, up(o_b0, backproj0, (size_t)n0 * 24));
  HIP_TRY(ctx, up(o_v0, valid0, (size_t)n0));
  HIP_TRY(ctx, up(o_d1, desc1, (size_t)n1 * 48));
  HIP_TRY(ctx, up(o_b1, backproj1, (size_t)n1 * 24));
  HIP_TRY(ctx, up(o_v1, valid1, (size_t)n1));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable sources must stay valid until copied
  launch_match_stereo_arrays(reinterpret_cast<PairParams*>(base + o_pair), base + o_d0,
                             reinterpret_cast<double*>(base + o_b0), base + o_v0, nullptr, n0, base + o_d1,
                             reinterpret_cast<double*>(base + o_b1), base + o_v1, nullptr, n1, n0,
                             ctx->cfg.match_threshold, reinterpret_cast<okvfe_stereo_match*>(base + o_out), s,
                             multi ? reinterpret_cast<const okvfe_keypoint*>(base + o_k0) : nullptr,
                             multi ? reinterpret_cast<const okvfe_keypoint*>(base + o_k1) : nullptr);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(matches, base + o_out, (size_t)n0 * sizeof(okvfe_stereo_match), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_match_motion_stereo(okvfe_ctx* ctx, const okvfe_camera* camera, const uint8_t* desc0,
                                       const okvfe_keypoint* kp0, const double* backproj0, const uint8_t* valid0,
                                       const uint8_t* skip0, int32_t n0, const uint8_t* desc1,
                                       const okvfe_keypoint* kp1, const double* backproj1, const uint8_t* valid1,
                                       const uint8_t* matched1, int32_t n1, const okvfe_pose* T_WC0,
                                       const okvfe_pose* T_WC1, okvfe_motion_match* matches) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!camera || n0 < 0 || n1 < 0 || !T_WC0 || !T_WC1 ||
      (n0 > 0 && (!desc0 || !kp0 || !backproj0 || !valid0 || !matches)) ||
      (n1 > 0 && (!desc1 || !kp1 || !backproj1 || !valid1)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_motion_stereo: bad argument");
  bool multi = false;
  {
    okvfe_status cst = check_size_classes(ctx, kp0, n0, &multi);
    if (cst == OKVFE_OK) cst = check_size_classes(ctx, kp1, n1, &multi);
    if (cst != OKVFE_OK) return cst;
  }
  if (n0 == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), a); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_cam = take(sizeof(DeviceCamera)),
               o_cls = take(kClassTableDoubles * sizeof(double));
  const size_t o_d0 = take((size_t)n0 * 48), o_k0 = take((size_t)n0 * sizeof(okvfe_keypoint)), o_b0 = take((size_t)n0 * 24),
               o_v0 = take(n0), o_s0 = take(n0);
  const size_t o_d1 = take((size_t)n1 * 48), o_k1 = take((size_t)n1 * sizeof(okvfe_keypoint)), o_b1 = take((size_t)n1 * 24),
               o_v1 = take(n1), o_m1 = take(n1);
  const size_t o_out = take((size_t)n0 * sizeof(okvfe_motion_match));
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1;
  sp.f0 = sp.f1 = 0.5 * (camera->fu + camera->fv);  // sigma = size0 / f0 * 0.125 (Frontend.cpp:1834)
  PairParams pp = to_pair_params(sp);
  const DeviceCamera dc = to_device_camera(*camera);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  double table[kClassTableDoubles];
  if (multi) {
    fill_class_table(table, sp.f0, sp.f1, true);
    pp.cls = reinterpret_cast<const double*>(base + o_cls);
    HIP_TRY(ctx, up(o_cls, table, sizeof(table)));
  }
  HIP_TRY(ctx, up(o_pair, &pp, sizeof(pp)));
  HIP_TRY(ctx, up(o_cam, &dc, sizeof(dc)));
  HIP_TRY(ctx, up(o_d0, desc0, (size_t)n0 * 48));
  HIP_TRY(ctx, up(o_k0, kp0, (size_t)n0 * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_b0, backproj0, (size_t)n0 * 24));
  HIP_TRY(ctx, up(o_v0, valid0, n0));
  if (skip0) HIP_TRY(ctx, up(o_s0, skip0, n0));
  HIP_TRY(ctx, up(o_d1, desc1, (size_t)n1 * 48));
  HIP_TRY(ctx, up(o_k1, kp1, (size_t)n1 * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_b1, backproj1, (size_t)n1 * 24));
  HIP_TRY(ctx, up(o_v1, valid1, n1));
  if (matched1) HIP_TRY(ctx, up(o_m1, matched1, n1));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_match_motion(reinterpret_cast<PairParams*>(base + o_pair), reinterpret_cast<DeviceCamera*>(base + o_cam),
                      camera->width, camera->height, base + o_d0, reinterpret_cast<okvfe_keypoint*>(base + o_k0),
                      reinterpret_cast<double*>(base + o_b0), base + o_v0, skip0 ? base + o_s0 : nullptr, n0,
                      base + o_d1, reinterpret_cast<okvfe_keypoint*>(base + o_k1),
                      reinterpret_cast<double*>(base + o_b1), base + o_v1, matched1 ? base + o_m1 : nullptr, n1,
                      ctx->cfg.match_threshold, reinterpret_cast<okvfe_motion_match*>(base + o_out), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(matches, base + o_out, (size_t)n0 * sizeof(okvfe_motion_match), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map(okvfe_ctx* ctx, const uint8_t* desc, const okvfe_keypoint* kps, const uint8_t* use,
                                int32_t n_kps, const double* projections_l2, const int32_t* desc_begin,
                                int32_t n_landmarks, const uint8_t* pool, double reprojection_threshold,
                                int32_t* best_landmark, int32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_kps < 0 || n_landmarks < 0 || !desc_begin || !(reprojection_threshold >= 0.0) ||
      (n_kps > 0 && (!desc || !kps || !use || !best_landmark || !best_dist)) ||
      (n_landmarks > 0 && (!projections_l2 || !pool)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map: desc_begin not monotone at %d", l);
  if (n_kps == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const int n_pool = desc_begin[n_landmarks];
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_d = take((size_t)n_kps * 48), o_k = take((size_t)n_kps * sizeof(okvfe_keypoint)), o_u = take(n_kps);
  const size_t o_p = take((size_t)n_landmarks * 16), o_b = take((size_t)(n_landmarks + 1) * 4), o_pool = take((size_t)n_pool * 48);
  const size_t o_lm = take((size_t)n_kps * 4), o_bd = take((size_t)n_kps * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_k, kps, (size_t)n_kps * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, up(o_p, projections_l2, (size_t)n_landmarks * 16));
  HIP_TRY(ctx, up(o_b, desc_begin, (size_t)(n_landmarks + 1) * 4));
  HIP_TRY(ctx, up(o_pool, pool, (size_t)n_pool * 48));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_match_to_map(base + o_d, reinterpret_cast<okvfe_keypoint*>(base + o_k), base + o_u, n_kps,
                      reinterpret_cast<double*>(base + o_p), reinterpret_cast<int32_t*>(base + o_b), n_landmarks,
                      base + o_pool, reprojection_threshold * reprojection_threshold, ctx->cfg.match_threshold,
                      reinterpret_cast<int32_t*>(base + o_lm), reinterpret_cast<int32_t*>(base + o_bd), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_landmarks(okvfe_ctx* ctx, int32_t cam, const okvfe_landmark_table* T,
                                          const okvfe_pose* T_WC1, double reprojection_threshold, int32_t exclusive,
                                          const uint8_t* desc, const okvfe_keypoint* kps, const uint8_t* use,
                                          int32_t n_kps, okvfe_landmark_pool* pool_out, int32_t* best_landmark,
                                          int32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!T || !T_WC1 || n_kps < 0 || !(reprojection_threshold >= 0.0) || T->n_landmarks < 0 || T->n_observations < 0 ||
      T->n_poses < 0 || !T->obs_begin || (n_kps > 0 && (!desc || !kps || !use || !best_landmark || !best_dist)) ||
      (T->n_landmarks > 0 && (!T->hp_W || !T->quality)) ||
      (T->n_observations > 0 && (!T->obs_pose || !T->obs_desc || !T->obs_backproj || !T->poses)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: bad argument");
  if (cam < 0 || cam >= (int)ctx->h_cams.size() || !(ctx->h_cams[cam].fu > 0.0))
    return fail(ctx, OKVFE_ERR_NOT_READY, "okvfe_match_to_map_landmarks: camera slot %d has no intrinsics (okvfe_set_camera)", cam);
  const int nl = T->n_landmarks, no = T->n_observations;
  for (int l = 0; l < nl; ++l)
    if (T->obs_begin[l + 1] < T->obs_begin[l] || T->obs_begin[l] < 0 || T->obs_begin[l + 1] > no)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: obs_begin not monotone at %d", l);
  for (int o = 0; o < no; ++o)
    if (T->obs_pose[o] < 0 || T->obs_pose[o] >= T->n_poses)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: observation %d: pose index out of range", o);
  for (int k = 0; k < n_kps; ++k) {
    best_landmark[k] = -1;
    best_dist[k] = ctx->cfg.match_threshold;
  }
  if (nl == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  // inputs
  const size_t o_hp = take((size_t)nl * 32), o_q = take((size_t)nl * 8), o_ob = take((size_t)(nl + 1) * 4),
               o_op = take((size_t)no * 4), o_od = take((size_t)no * 48), o_obp = take((size_t)no * 24),
               o_poses = take((size_t)T->n_poses * sizeof(okvfe_pose));
  const size_t o_d = take((size_t)n_kps * 48), o_k = take((size_t)n_kps * sizeof(okvfe_keypoint)), o_u = take(n_kps);
  // pooling results
  const size_t o_st = take((size_t)nl * 4), o_nd = take((size_t)nl * 4), o_rows = take((size_t)nl * 12),
               o_proj = take((size_t)nl * 16), o_e = take((size_t)nl * 48), o_r = take((size_t)nl * 48);
  // packed 3-D set + matcher outputs
  const size_t o_idx = take((size_t)nl * 4), o_p3 = take((size_t)nl * 16), o_b3 = take((size_t)(nl + 1) * 4),
               o_pool3 = take((size_t)nl * 2 * 48), o_n3 = take(8), o_lm = take((size_t)n_kps * 4),
               o_bd = take((size_t)n_kps * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_hp, T->hp_W, (size_t)nl * 32));
  HIP_TRY(ctx, up(o_q, T->quality, (size_t)nl * 8));
  HIP_TRY(ctx, up(o_ob, T->obs_begin, (size_t)(nl + 1) * 4));
  HIP_TRY(ctx, up(o_op, T->obs_pose, (size_t)no * 4));
  HIP_TRY(ctx, up(o_od, T->obs_desc, (size_t)no * 48));
  HIP_TRY(ctx, up(o_obp, T->obs_backproj, (size_t)no * 24));
  HIP_TRY(ctx, up(o_poses, T->poses, (size_t)T->n_poses * sizeof(okvfe_pose)));
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_k, kps, (size_t)n_kps * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable sources
  const DeviceCamera& dc = ctx->h_cams[cam];
  const double focal = dc.fu + dc.fv;  // the SUM, as at Frontend.cpp:1213-1215
  auto I = [&](size_t o) { return reinterpret_cast<int32_t*>(base + o); };
  auto D = [&](size_t o) { return reinterpret_cast<double*>(base + o); };
  launch_prepare_landmarks(D(o_hp), D(o_q), I(o_ob), nl, I(o_op), D(o_obp),
                           reinterpret_cast<const okvfe_pose*>(base + o_poses), *T_WC1, ctx->d_cams + cam, ctx->w,
                           ctx->h, reprojection_threshold, exclusive ? 1 : 0, std::cos(10.0 / focal), std::cos(0.6),
                           I(o_st), I(o_nd), I(o_rows), D(o_proj), D(o_e), D(o_r), s);
  launch_compact_landmarks(I(o_st), I(o_nd), I(o_rows), D(o_proj), base + o_od, nl, 1, I(o_idx), D(o_p3), I(o_b3),
                           base + o_pool3, I(o_n3), s);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n3[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(n3, base + o_n3, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // the matcher's grid depends on the number of 3-D landmarks only
  if (n_kps > 0 && n3[0] > 0) {
    launch_match_to_map(base + o_d, reinterpret_cast<okvfe_keypoint*>(base + o_k), base + o_u, n_kps, D(o_p3),
                        I(o_b3), n3[0], base + o_pool3, reprojection_threshold * reprojection_threshold,
                        ctx->cfg.match_threshold, I(o_lm), I(o_bd), s);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int32_t> idx(n3[0]);
    HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(idx.data(), base + o_idx, (size_t)n3[0] * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    for (int k = 0; k < n_kps; ++k)
      if (best_landmark[k] >= 0) best_landmark[k] = idx[best_landmark[k]];  // packed -> table index
  }
  if (pool_out) {
    if (pool_out->status) HIP_TRY(ctx, hipMemcpyAsync(pool_out->status, base + o_st, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
    if (pool_out->n_desc) HIP_TRY(ctx, hipMemcpyAsync(pool_out->n_desc, base + o_nd, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
    if (pool_out->obs_rows) HIP_TRY(ctx, hipMemcpyAsync(pool_out->obs_rows, base + o_rows, (size_t)nl * 12, hipMemcpyDeviceToHost, s));
    if (pool_out->projection) HIP_TRY(ctx, hipMemcpyAsync(pool_out->projection, base + o_proj, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
    if (pool_out->e_W) HIP_TRY(ctx, hipMemcpyAsync(pool_out->e_W, base + o_e, (size_t)nl * 48, hipMemcpyDeviceToHost, s));
    if (pool_out->r_W) HIP_TRY(ctx, hipMemcpyAsync(pool_out->r_W, base + o_r, (size_t)nl * 48, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
  }
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_uninitialised(okvfe_ctx* ctx, const uint8_t* desc, const double* backproj,
                                              const uint8_t* use, const int32_t* previous_landmark,
                                              int32_t n_kps, const int32_t* desc_begin, int32_t n_landmarks,
                                              const uint8_t* pool, const double* e0_W, const double* r0_W,
                                              const okvfe_pose* T_WC1, double focal_length,
                                              int32_t* best_landmark, int32_t* best_dist, double* hps_W,
                                              uint8_t* hp_set, int32_t* already_matched) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_kps < 0 || n_landmarks < 0 || !desc_begin || !T_WC1 || !(focal_length > 0.0) || !already_matched ||
      (n_kps > 0 && (!desc || !backproj || !use || !previous_landmark || !best_landmark || !best_dist || !hps_W || !hp_set)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: desc_begin not monotone at %d", l);
  *already_matched = 0;
  if (n_kps == 0) return OKVFE_OK;
  const int n_pool = desc_begin[n_landmarks];
  if (n_pool > 0 && (!pool || !e0_W || !r0_W))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: null pool");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_d = take((size_t)n_kps * 48), o_bp = take((size_t)n_kps * 24),
               o_u = take(n_kps), o_prev = take((size_t)n_kps * 4), o_b = take((size_t)(n_landmarks + 1) * 4),
               o_pool = take((size_t)n_pool * 48), o_e = take((size_t)n_pool * 24), o_r = take((size_t)n_pool * 24),
               o_lm = take((size_t)n_kps * 4), o_bd = take((size_t)n_kps * 4), o_hp = take((size_t)n_kps * 32),
               o_hs = take(n_kps), o_ctr = take(4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  PairParams pp{};
  std::memcpy(pp.C1, T_WC1->C, sizeof(pp.C1));
  std::memcpy(pp.r1, T_WC1->r, sizeof(pp.r1));
  const double sigma = 1.0 / focal_length;  // Frontend.cpp:1636
  pp.cos26 = std::cos(2.6 * sigma);
  pp.cos6 = std::cos(6.0 * sigma);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_pair, &pp, sizeof(pp)));
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_bp, backproj, (size_t)n_kps * 24));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, up(o_prev, previous_landmark, (size_t)n_kps * 4));
  HIP_TRY(ctx, up(o_b, desc_begin, (size_t)(n_landmarks + 1) * 4));
  HIP_TRY(ctx, up(o_pool, pool, (size_t)n_pool * 48));
  HIP_TRY(ctx, up(o_e, e0_W, (size_t)n_pool * 24));
  HIP_TRY(ctx, up(o_r, r0_W, (size_t)n_pool * 24));
  HIP_TRY(ctx, hipMemsetAsync(base + o_ctr, 0, 4, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_match_to_map_uninit(reinterpret_cast<PairParams*>(base + o_pair), base + o_d,
                             reinterpret_cast<double*>(base + o_bp), base + o_u,
                             reinterpret_cast<int32_t*>(base + o_prev), n_kps, reinterpret_cast<int32_t*>(base + o_b),
                             n_landmarks, base + o_pool, reinterpret_cast<double*>(base + o_e),
                             reinterpret_cast<double*>(base + o_r), ctx->cfg.match_threshold,
                             reinterpret_cast<int32_t*>(base + o_lm), reinterpret_cast<int32_t*>(base + o_bd),
                             reinterpret_cast<double*>(base + o_hp), base + o_hs,
                             reinterpret_cast<int32_t*>(base + o_ctr), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(hps_W, base + o_hp, (size_t)n_kps * 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(hp_set, base + o_hs, n_kps, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(already_matched, base + o_ctr, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_hamming_candidates(okvfe_ctx* ctx, const uint8_t* A, int32_t nA, const uint8_t* B,
                                      int32_t nB, int32_t threshold, okvfe_candidate* out, int32_t cap,
                                      int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (nA < 0 || nB < 0 || !n_out || cap < 0 || (nA > 0 && !A) || (nB > 0 && !B) || (cap > 0 && !out))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_hamming_candidates: bad argument");
  *n_out = 0;
  if (nA == 0 || nB == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  const size_t o_A = 0;
  const size_t o_B = align_up(o_A + (size_t)nA * 48, a);
  const size_t o_rows = align_up(o_B + (size_t)nB * 48, a);
  const size_t o_out = align_up(o_rows + (size_t)nA * 4, a);
  const size_t total = o_out + (size_t)std::max(cap, 1) * sizeof(okvfe_candidate);
  okvfe_status st = ensure_scratch(ctx, total);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_A, A, (size_t)nA * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_B, B, (size_t)nB * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int32_t* d_rows = reinterpret_cast<int32_t*>(base + o_rows);
  launch_hamming_count(base + o_A, nA, base + o_B, nB, threshold, d_rows, s);
  std::vector<int32_t> rows(nA);
  HIP_TRY(ctx, hipMemcpyAsync(rows.data(), d_rows, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int64_t total_c = 0;
  for (int i = 0; i < nA; ++i) {
    const int32_t c = rows[i];
    rows[i] = (int32_t)std::min<int64_t>(total_c, INT32_MAX);
    total_c += c;
  }
  *n_out = (int32_t)std::min<int64_t>(total_c, INT32_MAX);
  HIP_TRY(ctx, hipMemcpyAsync(d_rows, rows.data(), (size_t)nA * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_hamming_emit(base + o_A, nA, base + o_B, nB, threshold, d_rows,
                      reinterpret_cast<okvfe_candidate*>(base + o_out), cap, s);
  HIP_TRY(ctx, hipGetLastError());
  const int32_t ncopy = (int32_t)std::min<int64_t>(total_c, cap);
  if (ncopy > 0)
    HIP_TRY(ctx, hipMemcpyAsync(out, base + o_out, (size_t)ncopy * sizeof(okvfe_candidate), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (total_c > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%lld candidates, caller capacity %d", (long long)total_c, cap);
  return OKVFE_OK;
}

okvfe_status okvfe_hamming_argmin(okvfe_ctx* ctx, const uint8_t* A, int32_t nA, const uint8_t* B, int32_t nB,
                                  uint32_t threshold, int32_t* best_j, uint32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (nA < 0 || nB < 0 || (nA > 0 && (!A || !best_j || !best_dist)) || (nB > 0 && !B))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_hamming_argmin: bad argument");
  if (nA == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  const size_t o_A = 0;
  const size_t o_B = align_up(o_A + (size_t)nA * 48, a);
  const size_t o_j = align_up(o_B + (size_t)std::max(nB, 1) * 48, a);
  const size_t o_d = align_up(o_j + (size_t)nA * 4, a);
  const size_t total = o_d + (size_t)nA * 4;
  okvfe_status st = ensure_scratch(ctx, total);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_A, A, (size_t)nA * 48, hipMemcpyHostToDevice, s));
  if (nB > 0) HIP_TRY(ctx, hipMemcpyAsync(base + o_B, B, (size_t)nB * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_hamming_argmin(base + o_A, nA, base + o_B, nB, threshold, reinterpret_cast<int32_t*>(base + o_j),
                        reinterpret_cast<uint32_t*>(base + o_d), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_j, base + o_j, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_d, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_verify_place_match(okvfe_ctx* ctx, const uint8_t* landmark_desc, const int32_t* desc_begin,
                                      int32_t n_landmarks, const uint8_t* frame_desc, int32_t n_kps,
                                      int32_t* k_min, uint32_t* dist_min) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_landmarks < 0 || n_kps < 0 || !desc_begin || (n_landmarks > 0 && (!k_min || !dist_min)) ||
      (n_kps > 0 && !frame_desc))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: desc_begin not monotone at %d", l);
  if (n_landmarks == 0) return OKVFE_OK;
  const int n_pool = desc_begin[n_landmarks];
  if (n_pool > 0 && !landmark_desc) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: null pool");
  const uint32_t thr = (uint32_t)ctx->cfg.match_threshold;
  if (n_kps == 0 || n_pool == 0) {  // Frontend.cpp:333-335: a camera without keypoints is skipped
    for (int l = 0; l < n_landmarks; ++l) { k_min[l] = 0; dist_min[l] = thr; }
    return OKVFE_OK;
  }
  if ((int64_t)3 * n_kps >= (int64_t)1 << 31) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "too many keypoints");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pool = take((size_t)n_pool * 48), o_b = take((size_t)(n_landmarks + 1) * 4),
               o_f = take((size_t)n_kps * 48), o_k = take((size_t)n_landmarks * 4), o_d = take((size_t)n_landmarks * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_pool, landmark_desc, (size_t)n_pool * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_b, desc_begin, (size_t)(n_landmarks + 1) * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_f, frame_desc, (size_t)n_kps * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_verify_place(base + o_pool, reinterpret_cast<int32_t*>(base + o_b), n_landmarks, base + o_f, n_kps, thr,
                      reinterpret_cast<int32_t*>(base + o_k), reinterpret_cast<uint32_t*>(base + o_d), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(k_min, base + o_k, (size_t)n_landmarks * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(dist_min, base + o_d, (size_t)n_landmarks * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_fbrisk_transform(okvfe_ctx* ctx, const uint8_t* descriptors, int32_t n,
                                    const uint8_t* node_descriptors, int32_t n_nodes, const int32_t* child_begin,
                                    const int32_t* child_index, const int32_t* node_word, int32_t* word_ids,
                                    int32_t* leaf_nodes) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n < 0 || n_nodes < 1 || !node_descriptors || !child_begin || !node_word || (n > 0 && (!descriptors || !word_ids)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: bad argument");
  // the tree must be a tree: children lists monotone, indices in range and pointing downwards
  if (child_begin[0] != 0) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: child_begin[0] != 0");
  for (int i = 0; i < n_nodes; ++i)
    if (child_begin[i + 1] < child_begin[i])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: child_begin not monotone at %d", i);
  const int n_child = child_begin[n_nodes];
  if (n_child > 0 && !child_index) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: null children");
  {
    std::vector<int> depth(n_nodes, -1);
    depth[0] = 0;
    for (int i = 0; i < n_nodes; ++i)  // nodes are numbered so that a parent precedes its children
      for (int c = child_begin[i]; c < child_begin[i + 1]; ++c) {
        const int id = child_index[c];
        if (id <= i || id >= n_nodes || depth[i] < 0)
          return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: node %d has child %d (not a tree in id order)", i, id);
        depth[id] = depth[i] + 1;
      }
  }
  if (n == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_d = take((size_t)n * 48), o_n = take((size_t)n_nodes * 48), o_cb = take((size_t)(n_nodes + 1) * 4),
               o_ci = take((size_t)n_child * 4), o_w = take((size_t)n_nodes * 4), o_wo = take((size_t)n * 4),
               o_no = take((size_t)n * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_d, descriptors, (size_t)n * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_n, node_descriptors, (size_t)n_nodes * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_cb, child_begin, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice, s));
  if (n_child) HIP_TRY(ctx, hipMemcpyAsync(base + o_ci, child_index, (size_t)n_child * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_w, node_word, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_voc_transform(base + o_d, n, base + o_n, n_nodes, reinterpret_cast<int32_t*>(base + o_cb),
                       reinterpret_cast<int32_t*>(base + o_ci), reinterpret_cast<int32_t*>(base + o_w),
                       reinterpret_cast<int32_t*>(base + o_wo), reinterpret_cast<int32_t*>(base + o_no), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(word_ids, base + o_wo, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  if (leaf_nodes) HIP_TRY(ctx, hipMemcpyAsync(leaf_nodes, base + o_no, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

// ---- stage profiling -------------------------------------------------------------------------
okvfe_status okvfe_bow_vector(const int32_t* word_ids, int32_t n_features, const double* word_weight,
                              int32_t n_words, int32_t weighting, int32_t normalise_l1, int32_t* ids_out,
                              double* values_out, int32_t cap, int32_t* n_out) {
  if (n_features < 0 || n_words < 1 || !word_weight || !n_out || weighting < 0 || weighting > 3 ||
      (n_features > 0 && !word_ids) || cap < 0 || (cap > 0 && (!ids_out || !values_out)))
    return OKVFE_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n_features; ++i)
    if (word_ids[i] < 0 || word_ids[i] >= n_words) return OKVFE_ERR_INVALID_ARGUMENT;
  // value per word in feature order (the sums are sequential additions of the same weight, as
  // BowVector::addWeight performs them), then the words in ascending order
  const bool sums = weighting == 0 || weighting == 1;  // TF_IDF, TF
  std::vector<double> acc(n_words, 0.0);
  std::vector<uint8_t> seen(n_words, 0);
  for (int i = 0; i < n_features; ++i) {
    const int id = word_ids[i];
    const double wgt = word_weight[id];
    if (!(wgt > 0)) continue;
    if (!seen[id]) {
      seen[id] = 1;
      acc[id] = wgt;
    } else if (sums) {
      acc[id] = acc[id] + wgt;
    }
  }
  int n = 0;
  for (int id = 0; id < n_words; ++id) n += seen[id];
  *n_out = n;
  if (n > cap) return OKVFE_ERR_CAPACITY;
  int k = 0;
  for (int id = 0; id < n_words; ++id)
    if (seen[id]) {
      ids_out[k] = id;
      values_out[k] = acc[id];
      ++k;
    }
  if (normalise_l1) {
    double norm = 0.0;
    for (int i = 0; i < n; ++i) norm = norm + std::fabs(values_out[i]);
    if (norm > 0.0)
      for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / norm;
  } else if (sums && n > 0) {
    const double nd = (double)n;
    for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / nd;
  }
  return OKVFE_OK;
}

okvfe_status okvfe_bow_query_l1(okvfe_ctx* ctx, const int32_t* db_begin, const int32_t* db_ids,
                                const double* db_values, int32_t n_entries, const int32_t* q_ids,
                                const double* q_values, int32_t n_q, double* scores) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_entries < 0 || n_q < 0 || !db_begin || (n_entries > 0 && !scores) || (n_q > 0 && (!q_ids || !q_values)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: bad argument");
  if (db_begin[0] != 0) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: db_begin[0] != 0");
  for (int e = 0; e < n_entries; ++e) {
    if (db_begin[e + 1] < db_begin[e])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: db_begin not monotone at %d", e);
    for (int i = db_begin[e] + 1; i < db_begin[e + 1]; ++i)
      if (db_ids[i] <= db_ids[i - 1])
        return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: entry %d is not in ascending word order", e);
  }
  for (int j = 1; j < n_q; ++j)
    if (q_ids[j] <= q_ids[j - 1])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: query is not in ascending word order");
  if (n_entries == 0) return OKVFE_OK;
  const int m = db_begin[n_entries];
  if (m > 0 && (!db_ids || !db_values)) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: null database");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_b = take((size_t)(n_entries + 1) * 4), o_i = take((size_t)m * 4), o_v = take((size_t)m * 8),
               o_qi = take((size_t)n_q * 4), o_qv = take((size_t)n_q * 8), o_s = take((size_t)n_entries * 8);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_b, db_begin, (size_t)(n_entries + 1) * 4, hipMemcpyHostToDevice, s));
  if (m) {
    HIP_TRY(ctx, hipMemcpyAsync(base + o_i, db_ids, (size_t)m * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_v, db_values, (size_t)m * 8, hipMemcpyHostToDevice, s));
  }
  if (n_q) {
    HIP_TRY(ctx, hipMemcpyAsync(base + o_qi, q_ids, (size_t)n_q * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_qv, q_values, (size_t)n_q * 8, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable host buffers: the copies above have completed
  launch_bow_query_l1(reinterpret_cast<int32_t*>(base + o_b), reinterpret_cast<int32_t*>(base + o_i),
                      reinterpret_cast<double*>(base + o_v), n_entries, reinterpret_cast<int32_t*>(base + o_qi),
                      reinterpret_cast<double*>(base + o_qv), n_q, reinterpret_cast<double*>(base + o_s), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(scores, base + o_s, (size_t)n_entries * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_profile_enable(okvfe_ctx* ctx, int32_t enable) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
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

// ---- gather blocks ---------------------------------------------------------------------------
namespace {
struct BlockLayout {
  size_t o_count, o_kps, o_desc, o_bp, o_bpv, total;
};
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
}  // namespace

size_t okvfe_gather_block_bytes(const okvfe_ctx* ctx) { return ctx ? block_layout(ctx->kp_cap).total : 0; }

okvfe_status okvfe_pack_gather_blocks_device(okvfe_ctx* ctx, int32_t first_index, int32_t n, void* blocks_dev,
                                             void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (first_index < 0 || n < 1 || first_index + n > ctx->B || !blocks_dev)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_pack_gather_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  launch_pack_blocks(offs, first_index, n, ctx->kp_cap, ctx->d_count, ctx->d_kps, ctx->d_desc, ctx->d_bp,
                     ctx->d_bpv, static_cast<uint8_t*>(blocks_dev), s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_pack_gather_block_device(okvfe_ctx* ctx, int32_t index, void* block_dev, void* stream) {
  return okvfe_pack_gather_blocks_device(ctx, index, 1, block_dev, stream);
}

okvfe_status okvfe_match_stereo_blocks_batch_device(okvfe_ctx* ctx, const void* blocks0_dev,
                                                    const void* blocks1_dev, int32_t n_frames,
                                                    const okvfe_pose* T_WC0, const okvfe_pose* T_WC1, double f0,
                                                    double f1, okvfe_stereo_match* matches_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!blocks0_dev || !blocks1_dev || n_frames < 1 || !T_WC0 || !T_WC1 || !matches_dev || !(f0 > 0.0) || !(f1 > 0.0))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo_blocks_batch_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1; sp.f0 = f0; sp.f1 = f1;
  PairParams pp = to_pair_params(sp);
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    double table[kClassTableDoubles];
    fill_class_table(table, f0, f1, false);
    void* d_tab = nullptr;
    okvfe_status st = ring_upload(ctx, &ctx->cls_ring, table, sizeof(table), s, &d_tab, &cls_slot);
    if (st != OKVFE_OK) return st;
    pp.cls = static_cast<const double*>(d_tab);
  }
  // the pair record travels by value as a kernel argument: nothing to keep alive
  launch_match_stereo_blocks(pp, offs, static_cast<const uint8_t*>(blocks0_dev),
                             static_cast<const uint8_t*>(blocks1_dev), n_frames, ctx->kp_cap,
                             ctx->cfg.match_threshold, matches_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return ring_release(ctx, &ctx->cls_ring, cls_slot, s);
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
  for (int i = 0; i < kPatternPoints; ++i) {  // same float sequence as build_pattern (host_tables.cpp)
    const float sg = i < P.n_points ? P.sigma_half[i] : 1.0f;
    float area = 4.0f * sg;
    area = area * sg;
    const int scaling = static_cast<int>(4194304.0f / area);
    const float s2 = static_cast<float>(scaling) * area;
    P.box_scaling[i] = scaling;
    P.box_scaling2[i] = static_cast<int>(s2 / 1024.0f);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
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

// ---- device-resident, batched map matchers (frame f = gather block f) --------------------------
namespace {
okvfe_status map_args_ok(okvfe_ctx* ctx, const char* who, const void* blocks, int n_frames, const okvfe_map_device* map) {
  if (!blocks || !map || n_frames < 1 || map->n_landmarks < 0 || !map->desc_begin ||
      (map->n_landmarks > 0 && !map->pool))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
  return OKVFE_OK;
}
}  // namespace

okvfe_status okvfe_match_to_map_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const uint8_t* use_dev, const okvfe_map_device* map,
                                              double reprojection_threshold, int32_t* best_landmark_dev,
                                              int32_t* best_dist_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_match_to_map_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!best_landmark_dev || !best_dist_dev || !(reprojection_threshold >= 0.0) ||
      (map->n_landmarks > 0 && !map->projections))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  if ((size_t)n_frames > ctx->map_perm_frames) {  // workspace of the region order: grown on demand (synchronises once)
    if (ctx->d_map_perm) HIP_TRY(ctx, hipFree(ctx->d_map_perm));
    ctx->d_map_perm = nullptr;
    ctx->map_perm_frames = 0;
    void* q = nullptr;
    HIP_TRY(ctx, hipMalloc(&q, (size_t)n_frames * ctx->kp_cap * sizeof(int32_t)));
    ctx->d_map_perm = static_cast<int32_t*>(q);
    ctx->map_perm_frames = (size_t)n_frames;
  }
  {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_match_to_map_blocks(offs, static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap, use_dev,
                               map->projections, (size_t)map->n_landmarks * 2, map->desc_begin, map->n_landmarks,
                               map->pool, reprojection_threshold * reprojection_threshold, ctx->cfg.match_threshold,
                               best_landmark_dev, best_dist_dev, ctx->d_map_perm, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_uninitialised_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                                            const uint8_t* use_dev, const int32_t* previous_landmark_dev,
                                                            const okvfe_map_device* map, const okvfe_pose* T_WC1,
                                                            double focal_length, int32_t* best_landmark_dev,
                                                            int32_t* best_dist_dev, double* hps_W_dev, uint8_t* hp_set_dev,
                                                            int32_t* already_matched_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_match_to_map_uninitialised_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!T_WC1 || !(focal_length > 0.0) || !best_landmark_dev || !best_dist_dev || !hps_W_dev || !hp_set_dev ||
      !already_matched_dev || (map->n_landmarks > 0 && (!map->e0_W || !map->r0_W)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  // one pose record per frame, through the pinned parameter ring (one asynchronous copy, no host sync)
  std::vector<PairParams> pp((size_t)n_frames);
  const double sigma = 1.0 / focal_length;  // Frontend.cpp:1636
  const double c26 = std::cos(2.6 * sigma), c6 = std::cos(6.0 * sigma);
  for (int f = 0; f < n_frames; ++f) {
    pp[(size_t)f] = PairParams{};
    std::memcpy(pp[(size_t)f].C1, T_WC1[f].C, sizeof(pp[(size_t)f].C1));
    std::memcpy(pp[(size_t)f].r1, T_WC1[f].r, sizeof(pp[(size_t)f].r1));
    pp[(size_t)f].cos26 = c26;
    pp[(size_t)f].cos6 = c6;
  }
  void* d_pairs = nullptr;
  int slot = -1;
  st = ring_upload(ctx, &ctx->pair_ring, pp.data(), pp.size() * sizeof(PairParams), s, &d_pairs, &slot);
  if (st != OKVFE_OK) return st;
  hipError_t e = hipMemsetAsync(already_matched_dev, 0, (size_t)n_frames * sizeof(int32_t), s);
  if (e == hipSuccess) {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_match_to_map_uninit_blocks(static_cast<const PairParams*>(d_pairs), offs,
                                      static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap, use_dev,
                                      previous_landmark_dev, map->desc_begin, map->n_landmarks, map->pool, map->e0_W,
                                      map->r0_W, ctx->cfg.match_threshold, best_landmark_dev, best_dist_dev, hps_W_dev,
                                      hp_set_dev, already_matched_dev, s);
    e = hipGetLastError();
  }
  const okvfe_status rel = ring_release(ctx, &ctx->pair_ring, slot, s);  // on every path: the slot has a reader or not
  HIP_TRY(ctx, e);
  ctx->last_stream = s;
  return rel;
}

okvfe_status okvfe_verify_place_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const okvfe_map_device* map, int32_t* k_min_dev,
                                              uint32_t* dist_min_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_verify_place_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!k_min_dev || !dist_min_dev)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_verify_place_blocks(map->pool, map->desc_begin, map->n_landmarks, offs,
                               static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap,
                               (uint32_t)ctx->cfg.match_threshold, k_min_dev, dist_min_dev, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_match_motion_stereo_blocks_device(okvfe_ctx* ctx, int32_t cam, const void* block0_dev,
                                                     const void* block1_dev, const uint8_t* skip0_dev,
                                                     const uint8_t* matched1_dev, const okvfe_pose* T_WC0,
                                                     const okvfe_pose* T_WC1, okvfe_motion_match* matches_dev,
                                                     void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!block0_dev || !block1_dev || !T_WC0 || !T_WC1 || !matches_dev || cam < 0 ||
      cam >= (int)ctx->h_cams.size())
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_motion_stereo_blocks_device: bad argument");
  const DeviceCamera& dc = ctx->h_cams[cam];
  if (!(dc.fu > 0.0))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "camera slot %d has no intrinsics (okvfe_set_camera)", cam);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1;
  sp.f0 = sp.f1 = 0.5 * (dc.fu + dc.fv);  // sigma = size0 / f0 * 0.125 (Frontend.cpp:1834)
  PairParams pp = to_pair_params(sp);
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    double table[kClassTableDoubles];
    fill_class_table(table, sp.f0, sp.f1, true);
    void* d_tab = nullptr;
    okvfe_status st = ring_upload(ctx, &ctx->cls_ring, table, sizeof(table), s, &d_tab, &cls_slot);
    if (st != OKVFE_OK) return st;
    pp.cls = static_cast<const double*>(d_tab);
  }
  launch_match_motion_blocks(pp, ctx->d_cams + cam, ctx->w, ctx->h, offs,
                             static_cast<const uint8_t*>(block0_dev), static_cast<const uint8_t*>(block1_dev),
                             skip0_dev, matched1_dev, ctx->kp_cap, ctx->cfg.match_threshold, matches_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return ring_release(ctx, &ctx->cls_ring, cls_slot, s);
}

okvfe_status okvfe_match_stereo_blocks_device(okvfe_ctx* ctx, const void* block0_dev, const void* block1_dev,
                                              const okvfe_pose* T_WC0, const okvfe_pose* T_WC1, double f0,
                                              double f1, okvfe_stereo_match* matches_dev, void* stream) {
  return okvfe_match_stereo_blocks_batch_device(ctx, block0_dev, block1_dev, 1, T_WC0, T_WC1, f0, f1, matches_dev,
                                                stream);
}

}  // extern "C"
