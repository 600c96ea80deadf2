// okvfe_ctx.h -- the context behind the C ABI and the runtime helpers its entry points share
// (capi_context.cpp: creation, cameras, pattern, profiling; capi_detect.cpp: detector / extractor
// pipeline; capi_match.cpp: stereo / motion / Hamming matchers and gather blocks; capi_map.cpp: map
// matchers and place recognition).  Internal to libokvfe.so.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "okvfe_internal.h"

using namespace okvfe;  // internal header of the runtime's own translation units

struct okvfe_ctx {
  okvfe_config cfg{};
  hipStream_t stream = nullptr;
  hipEvent_t heavy_done[2] = {nullptr, nullptr};  // okvfe_set_heavy_kernel_chaining: after the score / describe kernel
  int detected_images = 0;  // images covered by the last okvfe_detect_batch_device
  std::string err;
  int w = 0, h = 0, B = 0, kp_cap = 0, cand_cap = 0, ws_stride = 0;
  int occ_rows = 0, occ_cols = 0;
  size_t occ_image_bytes = 0;
  int mode_default = kUpright;
  Pattern host_pattern{};

  std::vector<void*> allocs;
  int32_t* d_scores = nullptr;
  int32_t* d_virtual = nullptr;   // scale-space parent, OKVFE_SCORE_BRISK_SCALESPACE: FAST 5-8 map of layer 0
  // okvfe_match_to_map_blocks_device: keypoint order per frame [frames][kp_cap], ONE workspace PER STREAM the entry
  // point was called on (calls on one stream are ordered; calls on different streams no longer share a buffer)
  struct MapPerm {
    hipStream_t stream = nullptr;
    int32_t* d = nullptr;
    size_t frames = 0;  // frames d holds
  };
  std::vector<MapPerm> map_perm;
  ScoreLayout score_layout{0, 0};  // of d_scores: slotted where the fused score+NMS kernel applies
  ScoreLayout live_layout{0, 0};   // the layout the LAST score launch actually wrote (dense when the fused kernel refused the call)
  // Map-free detection (round 4): single-scale Harris calls whose selection kernel can recompute the nine
  // sub-pixel scores from the image write NO score map (okvfe_set_keep_score_map(ctx, 1) restores it).
  bool keep_score_map = false;
  bool layer_child = false;        // detect-only context of one scale-space layer: its map is read by the other layers
  bool map_free_live = false;      // the LAST detect call wrote no map: okvfe_device_outputs.scores is null
  const uint8_t* live_images = nullptr;  // images of the running detect call (map-free: read again by the selection)
  Candidate* d_cand = nullptr;
  int32_t* d_cand_count = nullptr;
  int32_t* d_fix_count = nullptr;  // flagged-record counts [B] and their lists [B][kFixListCap]: inside d_cand_count's
  int32_t* d_fix_list = nullptr;   // allocation (creation) -- separate names so that a lane view can offset them
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

  // Lanes inside one call (round 6; okvfe_set_internal_lanes): a device-resident batch is cut into `internal_lanes`
  // slices, each run on a stream of the context's own -- the VALU-bound score kernel of one slice under the LDS- and
  // latency-bound selection / descriptor / matcher kernels of another (the camera-parallel shape of
  // okvis_multisensor_processing/src/ThreadedSlam.cpp:434-448, inside the call).  A lane is a VIEW: an okvfe_ctx
  // whose per-image pointers are this context's, offset to the slice (bind_lane, capi_detect.cpp); it owns a stream
  // and two events, nothing else.
  int internal_lanes = 0;  // 0 = automatic (4 from 512 images per call), 1 = off
  bool lane_view = false;
  okvfe_ctx* prof_owner = nullptr;  // lane view: stage timers record into the owning context
  hipEvent_t k1_wait = nullptr;     // lane view: its score kernel starts behind this event (the previous lane's k1_done)
  hipEvent_t k1_done = nullptr;     // ... and records this one behind itself
  std::vector<okvfe_ctx*> lane_ctx;
  std::vector<hipEvent_t> lane_done;
  hipEvent_t lane_fork = nullptr;
  // priority form of the lanes (lab knob OKVFE_LANES_PRIO / okvfe_set_internal_lanes(-k)): ONE low-priority stream runs the
  // score kernels of all slices back to back, the slices' tails (fix-up .. compaction) run on high-priority lane streams
  hipStream_t score_stream = nullptr;  // owner: the low-priority stream; lane view: where its score kernel goes (or null)
  bool lanes_prio = false;             // owner: lane streams were created with priorities
  // PIPELINED lanes (okvfe_set_internal_lanes(ctx, -k)): the call does not join its lanes onto the caller's stream; the
  // slices' chains -- and those of the okvfe_match_stereo_batch_device call that follows -- stay on the lane streams, so
  // lane l starts the next call's score kernel behind ITS OWN previous work and the lanes drift out of phase, as separate
  // contexts do.  The join happens when something needs the results: any other entry point of this context (the stream
  // it is given waits for `join_done`; host-side readers synchronise), or okvfe_lanes_join.
  bool lanes_pipelined = false;        // owner: internal_lanes was set negative
  bool lanes_pending = false;          // owner: lane work has been issued that the caller's streams have not waited for
  int lanes_used = 0;                  // owner: lanes of the pending call
  int lane_chunk = 0;                  // owner: images per lane of the pending call
  hipStream_t join_stream = nullptr;   // owner: waits for every lane, releases the parameter slots, records join_done
  hipEvent_t join_done = nullptr;

  // scale space (octaves > 0): one detect-only child context per layer (K1..K4 at the layer's
  // size), layer images for l >= 1 owned here; this (parent) context keeps the merged keypoints
  // and everything from the descriptor stage on
  bool child = false;
  int n_layers = 1;
  std::vector<okvfe_ctx*> layers;
  std::vector<uint8_t*> d_layer_img;
  // scale-space parent: the layers of a call run concurrently on the layer contexts' own streams (round 6) -- per layer
  // {image ready, score map + candidates ready, keypoints ready} and the fork of the call
  std::vector<hipEvent_t> layer_ev;
  hipEvent_t layer_fork = nullptr;
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
  // row norms (nx, ny) of the image Jacobian / fu at every 8th pixel, kept so that okvfe_set_pattern can re-derive
  // the two statistics; cam_aware_slow: more than a tenth of the camera's keypoints would have a patch of neither LDS
  // class of describe_aware_kernel (k_describe_aware.hip) -> such a camera keeps describe_kernel
  std::vector<std::vector<float>> cam_norms;
  std::vector<uint8_t> cam_aware_slow;
  bool aware_fast = false;        // of the images of the current batch: none from a cam_aware_slow camera
  bool none_aware = false;        // no image of the last parameter upload is camera-aware (describe_rot_kernel's call)
  bool rot_fast_call = false;     // lane view: pattern_rot_ok of the owner's pattern
  int box_class_call = 0;         // lane view: pattern_box_class of the owner's pattern
  int aware_extra_box = -1;       // of the running call (aware_box_for_call): >= 0 = describe_aware_kernel serves it
  bool wide_patches = false;      // of the images of the current batch
  bool all_aware = false;         // every image of the current batch is extracted camera-aware
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
  uint8_t* h_pinned = nullptr;  // pinned staging for the host-buffer API: [image][keypoints in]
  void* h_pinned_dev = nullptr; // its device-visible address (the copy kernels read it in place)
  size_t h_pinned_bytes = 0;
  // single-image host-buffer API: ONE image's results land in this pinned block (export_result_kernel),
  // so a caller waiting for a frame pays one stream synchronisation
  uint8_t* h_result = nullptr;
  void* h_result_dev = nullptr;
  // okvfe_detect_ahead: the whole detect + describe chain ran for this image with this extraction
  // set-up; an okvfe_compute that asks for exactly that is answered from h_result
  struct Ahead {
    bool valid = false;
    const uint8_t* image = nullptr;
    size_t stride = 0;
    int32_t cam = -1;
    bool aware = false;
    float g[3] = {0.0f, 0.0f, 0.0f};
  } ahead;
  std::vector<uint8_t> ahead_shadow;  // the image okvfe_detect_ahead ran on (ordinary memory)

  // stage profiling (okvfe_profile_*): event pairs per recorded stage launch
  uint32_t prof_mask = 0;  // bit s = stage s is timed
  struct StageEvents {
    int stage;
    hipEvent_t a, b;
  };
  std::vector<StageEvents> prof_events;
  std::vector<hipEvent_t> event_pool;
};

namespace okvfe {

extern thread_local std::string g_create_error;  // message of a failed okvfe_create (no context yet)
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

okvfe_status fail(okvfe_ctx* ctx, okvfe_status st, const char* fmt, ...);

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

okvfe_status ensure_scratch(okvfe_ctx* ctx, size_t bytes);
okvfe_status ensure_pinned(okvfe_ctx* ctx, size_t bytes);
okvfe_status ring_reserve(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, size_t slot_bytes);
okvfe_status ring_upload(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, const void* src, size_t bytes, hipStream_t s,
                         void** d_out, int* slot_out, int32_t* zero_dev = nullptr, int n_zero = 0,
                         bool* zeroed = nullptr);
okvfe_status ring_release(okvfe_ctx* ctx, okvfe_ctx::ParamRing* r, int slot, hipStream_t s);
void ring_destroy(okvfe_ctx::ParamRing* r);
DeviceCamera to_device_camera(const okvfe_camera& c);
PairParams to_pair_params(const okvfe_stereo_pair& p);
double layer_keypoint_size(int l);
void fill_class_table(double* t, double f0, double f1, bool motion);
constexpr size_t kClassTableDoubles = 2 * kSizeClasses * kSizeClasses;
okvfe_status check_size_classes(okvfe_ctx* ctx, const okvfe_keypoint* kp, int n, bool* multi);
hipStream_t pick_stream(okvfe_ctx* ctx, void* stream);      // joins pending pipelined lanes onto the stream it returns
hipStream_t pick_stream_raw(okvfe_ctx* ctx, void* stream);  // no join (the lane-aware entry points)
okvfe_status lanes_join_host(okvfe_ctx* ctx);               // host-side join of pending pipelined lanes
void layer_size(int w, int h, int l, int* lw, int* lh);
void layer_scale(int l, int* num, int* den);

// serialisation of the heavy kernels across the contexts of a process (okvfe_set_heavy_kernel_chaining)
constexpr int kMaxTokenDevices = 64;
extern std::mutex g_token_mutex;
extern hipEvent_t g_score_token[kMaxTokenDevices];
int score_token_mode();

// RAII-free stage timer: records an event pair around a launch when profiling is on
struct StageTimer {
  okvfe_ctx* ctx;
  hipStream_t s;
  int idx = -1;
  StageTimer(okvfe_ctx* c0, int stage, hipStream_t st) : ctx(c0->prof_owner ? c0->prof_owner : c0), s(st) {
    okvfe_ctx* c = ctx;
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

// gather block = what travels between GPUs and what the device-resident matchers read
struct BlockLayout {
  size_t o_count, o_kps, o_desc, o_bp, o_bpv, total;
};
BlockLayout block_layout(int kp_cap);

}  // namespace okvfe
