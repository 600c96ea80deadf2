// okvfe_internal.h -- shared declarations of the libokvfe.so runtime (host + HIP kernels).
// Product code; never includes or links anything under oracle/.
#pragma once
#include <atomic>
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/okvfe.h"

namespace okvfe {

// Lab switches = the A/B knobs of the experiments recorded in LAB_NOTES.md.  The product library
// reads NO environment variable: lab_env() is a constant nullptr there, so every knob folds away at
// compile time.  `make lab` builds libokvfe_lab.so with -DOKVFE_LAB, where the knobs are live
// (tests/test_gpu_detector_paths.py loads that build for the tests that flip them).
#ifdef OKVFE_LAB
inline const char* lab_env(const char* name) { return std::getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif

// A function attribute (opt-in to > 64 KiB of dynamic LDS) belongs to the function ON THE CURRENT DEVICE: one process may
// hold contexts on several GPUs, so the one-shot is kept per device ordinal, thread-safe.
constexpr int kMaxAttrDevices = 64;
struct PerDeviceOnce {
  std::atomic<int> state[kMaxAttrDevices];  // 0 = not yet, 1 = being set, 2 = set
  template <typename F>
  void run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxAttrDevices) {
      f();  // (unknown ordinal: set it every time)
      return;
    }
    int expect = 0;
    if (state[dev].load(std::memory_order_acquire) == 2) return;
    if (state[dev].compare_exchange_strong(expect, 1)) {
      f();
      state[dev].store(2, std::memory_order_release);
    } else {
      while (state[dev].load(std::memory_order_acquire) != 2) {}  // another host thread is setting it right now
    }
  }
};



constexpr int kPatternPoints = 72;  // capacity (lane i and, past 64, a second sample on lane i - 64); built-in: 66
constexpr int kMaxLongPairs = 1100;
constexpr int kRot = 1024;

// BRISK2-style sampling pattern, host-built (host_tables.cpp), uploaded once per context.
struct Pattern {
  int32_t n_points;
  float px[kPatternPoints];
  float py[kPatternPoints];
  float sigma_half[kPatternPoints];
  int32_t n_short;
  uint8_t short_i[384], short_j[384];
  int32_t n_long;
  uint8_t long_i[kMaxLongPairs], long_j[kMaxLongPairs];
  int32_t long_wdx[kMaxLongPairs], long_wdy[kMaxLongPairs];
  int32_t border;
  int32_t rot_cos[kRot], rot_sin[kRot];
  float rot_cosf[kRot], rot_sinf[kRot];
  // per sample, constants of the box mean that depend on sigma_half only: scaling =
  // (int)(4194304 / (4 sigma^2)), scaling2 = (int)(scaling * 4 sigma^2 / 1024) (same float
  // sequence as the device code used per keypoint before they were tabulated)
  int32_t box_scaling[kPatternPoints], box_scaling2[kPatternPoints];
  // max_i |p_i| + sigma_half_i, rounded up: the patch geometry of describe_setup_one (describe_setup_dev.h)
  float reach;
  // describe_aware_kernel's per-lane constants in ONE 32-byte record (two 16-byte loads, no dependent look-ups in a
  // wave's prologue): lane l samples point extra + l (a lane without a point: point 0) -- {px, py, sigma_half (float
  // bits), box_scaling, box_scaling2, short pairs of words 0|1, 2|3, 4|5 (i | j << 8 each, two per dword)}
  int32_t aware_lane[64][8];
};
void fill_aware_lanes(Pattern* p);
void build_pattern(Pattern* p);
void scale_pattern_boxes(Pattern* p, float s);
float pattern_reach(const Pattern& p);
// scale_invariant = true (Frontend.hpp:235-237): the published BRISK extractor keeps the pattern at 64
// scales spanning a factor of 30 and picks index max(int(64 / lb(30) * lb(size / 7.2) + 0.5), 0)
// (<= 63) from the keypoint's diameter; the fixed-scale extractor is index 17 of the same ladder.
// Scale i = the base pattern with offsets, box half-sides and reach times 2^((i - 17) lb(30) / 64).
constexpr int kPatternScales = 64;
constexpr int kBasicScale = 17;
struct PatternScales {
  float px[kPatternScales][kPatternPoints], py[kPatternScales][kPatternPoints];
  float sigma_half[kPatternScales][kPatternPoints];
  int32_t box_scaling[kPatternScales][kPatternPoints], box_scaling2[kPatternScales][kPatternPoints];
  int32_t border[kPatternScales];
  float size_from[kPatternScales];  // index(size) = #{i >= 1 : size >= size_from[i]} (bisected on the host formula)
};
int pattern_scale_index(float size);
void build_pattern_scales(const Pattern& base, PatternScales* out);
// d_lut layout: [0, 961) the 31x31 weights; from kStampTableOffset the compacted stamp, one
// {(ry << 8) | rx, weight} pair per non-zero cell (697 of them), padded to kStampSlots.
constexpr int kStampSlots = 704;
constexpr int kStampCells = 697;
constexpr int kStampTableOffset = 1024;
constexpr int kLutFloats = kStampTableOffset + 2 * kStampSlots;
void build_uniformity_lut(float lut[kLutFloats]);
void build_awareness_maps(const okvfe_camera& cam, float* rays_hw3, float* jac_hw6);
bool camera_overlap(const okvfe_camera& cam, const okvfe_camera& other, const double R_other_cam[9],
                    uint8_t* mask_hw);
bool host_backproject(const okvfe_camera& cam, double px, double py, double dir[3]);

struct Candidate {  // NMS maximum
  int32_t x, y, score;
};
// set in Candidate::y by the fused score+NMS kernel: acceptance still depends on the raster-scan
// rule over a run of equal maxima, settled by nms_fixup_kernel
constexpr int32_t kCandidateFixupFlag = 0x40000000;
// the fused kernel also lists the positions of the flagged records, up to this many per image (more:
// nms_fixup_kernel scans the whole list)
constexpr int kFixListCap = 1024;

struct DeviceCamera {  // intrinsics for on-device back-projection
  double fu, fv, cu, cv;
  double one_over_fu, one_over_fv;
  double d[4];
  int32_t distortion;
  int32_t pad;
};

enum ExtractMode : int32_t { kUpright = 0, kGradient = 1, kCameraAware = 2 };

// per-image launch parameters of one batch (host-filled, uploaded per call)
struct ImageParams {
  int32_t cam;     // camera slot or -1
  int32_t mode;    // ExtractMode
  float dir[3];    // extraction direction
  float fu;
};

struct PairParams {  // one stereo pair on device
  int32_t image0, image1;
  double C0[9], r0[3], C1[9], r1[3];
  double f0, f1;
  double cos26, cos6;  // cos(2.6 sigma), cos(6 sigma) of size class (0, 0): keypoint size 12
  // scale space (octaves > 0): keypoint sizes are 12 * scale(layer) and sigma depends on the sizes
  // of the two keypoints; cls = device table [2][kSizeClasses][kSizeClasses] of cos(2.6 sigma) /
  // cos(6 sigma) indexed by the layers (= okvfe_keypoint::octave) of keypoint 0 and keypoint 1,
  // or null when every keypoint has size 12
  const double* cls;
};
constexpr int kSizeClasses = 8;

// Layout of a context's int32 score map in HBM.
//   dense   (strips <= 1): pixel (x, y) at y * pitch + x, pitch == w;
//   slotted (strips >= 2): what the fused score+NMS kernel (k_harris.hip) writes.  A wave of that
//     kernel owns a strip of 62 dwords-of-pixels (quads) of a row plus one halo lane on either side;
//     in the slotted layout every strip has a 1024-byte slot of its own in the row and ALL 64 lanes
//     store, so that each store instruction writes eight whole 128-byte lines (partial-line stores
//     cost the memory system as much as whole lines: profiles/round2_k1_store_pattern.txt).  Quad
//     d = x >> 2 belongs to strip s(d) = d <= 62 ? 0 : min((d - 1) / 62, strips - 1) and sits at
//     column x + 8 * s(d); the 2 * s quads in between are pad (halo-lane values, never read).
//     pitch = 4 * (w / 4 + 2 * (strips - 1)) rounded up to 32 ints (128 bytes).
struct ScoreLayout {
  int32_t pitch;   // ints per row
  int32_t strips;  // 0 / 1 = dense
};
__host__ __device__ inline int score_col(const ScoreLayout& L, int x) {
  if (L.strips <= 1) return x;
  const int d = x >> 2;
  int s = d <= 62 ? 0 : (d - 1) / 62;
  s = s > L.strips - 1 ? L.strips - 1 : s;
  return x + 8 * s;
}
__host__ __device__ inline size_t score_index(const ScoreLayout& L, int x, int y) {
  return (size_t)y * (size_t)L.pitch + (size_t)score_col(L, x);
}

}  // namespace okvfe

// XCD-aware tile mapping (device): the dispatcher is observed to place linear workgroup id L on
// XCD L % 8 (used for speed only, never for correctness).  All tiles of one image are handed to
// ONE XCD, so row halos and the cache lines shared by neighbouring strips are served by that
// XCD's L2 instead of being fetched from HBM once per XCD.  1-D grid of n_images * tiles blocks.
#ifdef __HIPCC__
__device__ __forceinline__ void xcd_tile(int tiles, int n_images, int* image, int* tile) {
  const int L = blockIdx.x;
  const int n8 = n_images & ~7;
  const int full = n8 * tiles;
  if (L < full) {
    const int xcd = L & 7, slot = L >> 3;
    const int g = slot / tiles;
    *image = g * 8 + xcd;
    *tile = slot - g * tiles;
  } else {
    const int r = L - full;
    const int i = r / tiles;
    *image = n8 + i;
    *tile = r - i * tiles;
  }
}
#endif

// ---- kernel launchers (defined in the .hip files) ----------------------------------------------
namespace okvfe {

// 3-term FP64 sum order of the matcher / landmark kernels (one device flag per translation unit)
bool set_fp64_tree_match(int tree);
bool set_fp64_tree_map(int tree);
void launch_bow_query_l1(const int32_t* db_begin, const int32_t* db_ids, const double* db_values, int n_entries,
                         const int32_t* q_ids, const double* q_values, int n_q, double* scores,
                         hipStream_t stream);
void launch_agast_score(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                        hipStream_t stream);
void launch_fast58_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream);
void launch_brisk_refine(const int32_t* score, int w, int h, int n_images, int cand_cap, const int32_t* cand_count,
                         const uint64_t* sort_ws, int max_kpts, const int32_t* below, int wb, int hb, int rn_b,
                         int rd_b, const int32_t* above, int wa, int ha, int rn_a, int rd_a, double rb, double ra, double lo,
                         okvfe_keypoint* kps, int kp_cap, int32_t* kp_count, hipStream_t stream);
void launch_harris(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                   hipStream_t stream);
// Score map and NMS candidates in one pass (k_harris.hip); false = not applicable (unaligned
// width, >32 rows per wave): the caller then runs launch_harris + launch_nms.  Must be followed by
// launch_nms_fixup.
bool launch_harris_nms(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                       ScoreLayout layout, int abs_threshold, Candidate* cand, int cand_cap,
                       int32_t* cand_count, int32_t* fix_count, int32_t* fix_list, hipStream_t stream,
                       bool store_map = true);
void launch_param_copy(void* dst_dev, const void* src_host_mapped, size_t bytes, int32_t* zero_dev, int n_zero,
                       hipStream_t stream, int32_t* poke_dev = nullptr, int32_t poke_value = 0);
// one image's results -> one block of pinned host memory (k_util.hip); null sources are skipped
struct ResultSrc {
  const int32_t* count;          // [B] final keypoint counts
  const okvfe_keypoint* kps;     // [B][kp_cap]
  const uint8_t* desc;           // [B][kp_cap][48]
  const double* bp;              // [B][kp_cap][3]
  const uint8_t* bpv;            // [B][kp_cap]
  const int32_t* det_count;      // [B] detector's counts
  const okvfe_keypoint* det_kps; // [B][kp_cap]
  const int32_t* cand_count;     // [B] NMS maxima found (may exceed the capacity)
};
struct ResultLayout {  // byte offsets inside the host block: header {n, candidates, detected, 0}
  int32_t o_count, o_kps, o_desc, o_bp, o_bpv, o_det, total;
};
void launch_export_result(const ResultSrc& src, int index, int kp_cap, const ResultLayout& layout,
                          void* dst_host_mapped, hipStream_t stream);
bool launch_harris_byte_mover(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                              ScoreLayout layout, hipStream_t stream);
// the layout launch_harris_nms writes for w x h images (dense when the fused kernel does not apply)
ScoreLayout harris_nms_layout(int w, int h);
void launch_nms_fixup(const int32_t* score, ScoreLayout layout, int w, int h, int n_images,
                      int abs_threshold, Candidate* cand, int cand_cap, int32_t* cand_count,
                      const int32_t* fix_count, const int32_t* fix_list, hipStream_t stream, bool map_free = false,
                      int32_t* idle_score_buffer = nullptr);
void launch_nms(const int32_t* score, int w, int h, int n_images, int abs_threshold,
                Candidate* cand, int cand_cap, int32_t* cand_count, hipStream_t stream);
void launch_sort(const Candidate* cand, int cand_cap, const int32_t* cand_count, int n_images,
                 float radius, uint64_t* sort_ws, hipStream_t stream);
// per-keypoint preparation of the extractor (describe_setup_dev.h), optionally run by the selection kernel
struct DescribeSetup {  // pat == nullptr: not requested
  const Pattern* pat;
  const ImageParams* prm;
  const float* const* rays;
  const float* const* jac;
  okvfe_keypoint* kps_tmp;
  uint8_t* desc_tmp;
  uint8_t* valid_tmp;
  const PatternScales* scales;
  // describe_aware_kernel's inputs (k_describe_aware.hip): the set-up thread also evaluates the pattern's samples
  // beyond 64 ("extra", <= kAwareMaxExtra of them) from `images` -- boxes of at most extra_box + 1 pixels a side
  // (4 or 9); 0 = not wanted (the call will run describe_kernel)
  const uint8_t* images = nullptr;
  int extra_box = 0;
};
constexpr int kAwareMaxExtra = 6;  // ints that fit the descriptor slot behind M (16 B) and the patch geometry (8 B)
// true: launch_select orders the candidates itself for this configuration (array-bin lazy selection): the
// caller launches no sort before it
bool select_sorts_candidates(float radius, int max_kpts, int kp_cap, const uint8_t* occupancy, size_t occ_image_bytes,
                             int occ_rows, int occ_cols);
// grid fall-backs of launch_select (k_select_grid.hip)
void launch_select_grid(const int32_t* score, ScoreLayout layout, int w, int h, int n_images, Candidate* cand,
                        int cand_cap, const int32_t* cand_count, float radius, int max_kpts, const float* lut,
                        uint8_t* occupancy, size_t occ_image_bytes, int occ_rows, int occ_cols, okvfe_keypoint* kps,
                        int kp_cap, int32_t* kp_count, uint64_t* sort_ws, hipStream_t stream);
bool launch_select(const int32_t* score, ScoreLayout layout, int w, int h, int n_images, Candidate* cand,
                   int cand_cap, const int32_t* cand_count, float radius, int max_kpts,
                   const float* lut, uint8_t* occupancy, size_t occ_pitch_bytes, int occ_rows,
                   int occ_cols, okvfe_keypoint* kps, int kp_cap, int32_t* kp_count,
                   uint64_t* sort_ws, hipStream_t stream, const DescribeSetup* setup = nullptr,
                   const uint8_t* images = nullptr);  // images != null: map-free call, see select_recomputes_scores
bool select_recomputes_scores(float radius, int max_kpts, int kp_cap, const uint8_t* occupancy, size_t occ_image_bytes,
                              int occ_rows, int occ_cols);
void launch_describe(const uint8_t* img, int w, int h, int n_images, const Pattern* pat,
                     const ImageParams* prm, const float* const* rays, const float* const* jac,
                     const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in,
                     okvfe_keypoint* kps_tmp, uint8_t* desc_tmp, uint8_t* valid_tmp,
                     const PatternScales* scales, bool wide_patches, hipStream_t stream, bool setup_done = false,
                     bool all_camera_aware = false, int box_class = 0,  // (every image of the call has mode kCameraAware)
                     int aware_extra_box = -1,  // >= 0: describe_aware_kernel serves the call (capi_detect.cpp: aware_box_for_call)
                     bool rot_fast = false);    // no image of the call is camera-aware and the pattern suits describe_rot_kernel
bool describe_patch_fits(float nx, float ny, int border);
// k_describe_aware.hip: the camera-aware-only extractor with batched extra samples (round 6)
int describe_aware_patch_class(float nx, float ny, float reach);
void launch_describe_aware(const uint8_t* img, int w, int h, int n_images, const Pattern* pat,
                           const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in, uint8_t* desc_tmp,
                           uint8_t* valid_tmp, bool wide_boxes, hipStream_t stream, int extras_now, int extra_box);
// true: the set-up threads (selection kernel's tail / describe_setup_kernel) evaluate the extra samples; false: a
// kernel of their own does (describe_extras_kernel)
bool aware_extras_in_setup();
// Quarter-wave lookup of the pattern's 1024-step rotation tables: sin(k) = +-Q[r or 256 - r] with Q = the first 257
// entries of the sine table, cos(k) = sin(k + 256).  describe_rot_kernel keeps Q (ints and floats, 2 KB) in LDS instead
// of gathering from the 16 KB of global tables inside every keypoint's chain; the host verifies once per pattern that the
// rule reproduces all 4096 table entries bit for bit (pattern_rot_ok, capi_detect.cpp) and keeps the all-modes kernel
// otherwise.
template <typename V>
__host__ __device__ inline V quarter_sin(const V* Q, int k) {
  k &= 1023;
  const int r = k & 255, q = k >> 8;
  const V v = (q & 1) ? Q[256 - r] : Q[r];
  return (q & 2) ? -v : v;
}
template <typename V>
__host__ __device__ inline V quarter_cos(const V* Q, int k) { return quarter_sin(Q, k + 256); }
void launch_describe_rot(const uint8_t* img, int w, int h, int n_images, const Pattern* pat, const ImageParams* prm,
                         const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in, okvfe_keypoint* kps_tmp,
                         uint8_t* desc_tmp, uint8_t* valid_tmp, hipStream_t stream);
void launch_compact(int n_images, const DeviceCamera* cams, const ImageParams* prm,
                    const okvfe_keypoint* kps_tmp, const uint8_t* desc_tmp,
                    const uint8_t* valid_tmp, const int32_t* kp_count_in, int kp_cap,
                    okvfe_keypoint* kps, uint8_t* desc, double* bp, uint8_t* bpv,
                    int32_t* kp_count, hipStream_t stream);
void launch_match_stereo(const PairParams* pairs, int n_pairs, const okvfe_keypoint* kps,
                         const uint8_t* desc, const double* bp, const uint8_t* bpv,
                         const int32_t* counts, int kp_cap, int threshold,
                         okvfe_stereo_match* out, hipStream_t stream);
void launch_match_stereo_arrays(const PairParams* pair, const uint8_t* desc0, const double* bp0,
                                const uint8_t* bpv0, const int32_t* n0p, int n0,
                                const uint8_t* desc1, const double* bp1, const uint8_t* bpv1,
                                const int32_t* n1p, int n1, int max_rows, int threshold,
                                okvfe_stereo_match* out, hipStream_t stream,
                                const okvfe_keypoint* kp0 = nullptr, const okvfe_keypoint* kp1 = nullptr);
void launch_match_motion(const PairParams* pair, const DeviceCamera* camera, int w, int h,
                         const uint8_t* desc0, const okvfe_keypoint* kp0, const double* bp0,
                         const uint8_t* bpv0, const uint8_t* skip0, int n0, const uint8_t* desc1,
                         const okvfe_keypoint* kp1, const double* bp1, const uint8_t* bpv1,
                         const uint8_t* matched1, int n1, int threshold, okvfe_motion_match* out,
                         hipStream_t stream);
void launch_match_to_map(const uint8_t* desc_k, const okvfe_keypoint* kps, const uint8_t* use, int n_k,
                         const double* projections, const int32_t* desc_begin, int n_lm,
                         const uint8_t* pool, double thr_sq, int threshold, int32_t* best_lm,
                         int32_t* best_d, hipStream_t stream);
void launch_match_to_map_uninit(const PairParams* pair, const uint8_t* desc_k, const double* bp,
                                const uint8_t* use, const int32_t* previous, int n_k,
                                const int32_t* desc_begin, int n_lm, const uint8_t* pool,
                                const double* e0_W, const double* r0_W, int threshold,
                                int32_t* best_lm, int32_t* best_d, double* hps_W, uint8_t* hp_set,
                                int32_t* ctr_total, hipStream_t stream);
// device-resident batches of the map matchers (frame f reads gather block f; k_match.hip, MapBatch)
void launch_match_to_map_blocks(const int offs[6], const uint8_t* blocks, int n_frames, int kp_cap,
                                const uint8_t* use, const double* projections, size_t proj_stride,
                                const int32_t* desc_begin, int n_lm, const uint8_t* pool, double thr_sq,
                                int threshold, int32_t* best_lm, int32_t* best_d, int32_t* perm_ws,
                                hipStream_t stream);
void launch_match_to_map_uninit_blocks(const PairParams* pairs, const int offs[6], const uint8_t* blocks,
                                       int n_frames, int kp_cap, const uint8_t* use, const int32_t* previous,
                                       const int32_t* desc_begin, int n_lm, const uint8_t* pool,
                                       const double* e0_W, const double* r0_W, int threshold, int32_t* best_lm,
                                       int32_t* best_d, double* hps_W, uint8_t* hp_set, int32_t* ctr_total,
                                       hipStream_t stream);
void launch_verify_place_blocks(const uint8_t* pool, const int32_t* desc_begin, int n_landmarks, const int offs[6],
                                const uint8_t* blocks, int n_frames, int kp_cap, uint32_t threshold,
                                int32_t* k_min, uint32_t* dist_min, hipStream_t stream);
void launch_pack_blocks(const int offs[6], int first, int n, int kp_cap, const int32_t* counts,
                        const okvfe_keypoint* kps, const uint8_t* desc, const double* bp,
                        const uint8_t* bpv, uint8_t* blocks, hipStream_t stream);
void launch_match_motion_blocks(const PairParams& pair, const DeviceCamera* camera, int w, int h,
                                const int offs[6], const uint8_t* block0, const uint8_t* block1,
                                const uint8_t* skip0, const uint8_t* matched1, int kp_cap,
                                int threshold, okvfe_motion_match* out, hipStream_t stream);
void launch_match_stereo_blocks(const PairParams& pair, const int offs[6], const uint8_t* blocks0,
                                const uint8_t* blocks1, int n_frames, int kp_cap, int threshold,
                                okvfe_stereo_match* out, hipStream_t stream);
void launch_verify_place(const uint8_t* pool, const int32_t* desc_begin, int n_landmarks,
                         const uint8_t* frame_desc, int K, uint32_t threshold, int32_t* k_min,
                         uint32_t* dist_min, hipStream_t stream);
void launch_voc_transform(const uint8_t* desc, int n, const uint8_t* node_desc, int n_nodes,
                          const int32_t* child_begin, const int32_t* child_index, const int32_t* word,
                          int32_t* word_out, int32_t* node_out, hipStream_t stream);
// matchToMap preparation (k_map.hip)
void launch_prepare_landmarks(const double* hp_W, const double* quality, const int32_t* obs_begin,
                              int n_landmarks, const int32_t* obs_pose, const double* obs_bp,
                              const okvfe_pose* poses, const okvfe_pose& T_WC1, const DeviceCamera* camera,
                              int w, int h, double repr, int exclusive, double cos10, double cos06,
                              int32_t* status, int32_t* n_desc, int32_t* obs_rows, double* projection,
                              double* e_W, double* r_W, hipStream_t stream);
void launch_compact_landmarks(const int32_t* status, const int32_t* n_desc, const int32_t* obs_rows,
                              const double* projection, const uint8_t* obs_desc, int n_landmarks, int want,
                              int32_t* index_out, double* proj_out, int32_t* begin_out, uint8_t* pool_out,
                              int32_t* n_out, hipStream_t stream);
// scale space (k_pyramid.hip)
void launch_halfsample(const uint8_t* src, int w, int h, int n_images, uint8_t* dst, hipStream_t stream);
void launch_twothird(const uint8_t* src, int w, int h, int n_images, uint8_t* dst, hipStream_t stream);
void launch_scale_filter(Candidate* cand, int cand_cap, int32_t* cand_count, int n_images,
                         const int32_t* below, ScoreLayout lb, int wb, int hb, int rn_b, int rd_b,
                         const int32_t* above, ScoreLayout la, int wa, int ha, int rn_a, int rd_a,
                         hipStream_t stream);
void launch_merge_layers(const okvfe_keypoint* const* kps, const int32_t* const* counts, const float* scale,
                         int n_layers, int layer_cap, int n_images, okvfe_keypoint* out, int out_cap,
                         int32_t* out_count, hipStream_t stream);
void launch_hamming_argmin(const uint8_t* A, int nA, const uint8_t* B, int nB, uint32_t thr,
                           int32_t* best_j, uint32_t* best_d, hipStream_t stream);
void launch_hamming_count(const uint8_t* A, int nA, const uint8_t* B, int nB, int thr,
                          int32_t* row_counts, hipStream_t stream);
void launch_hamming_emit(const uint8_t* A, int nA, const uint8_t* B, int nB, int thr,
                         const int32_t* row_offsets, okvfe_candidate* out, int cap,
                         hipStream_t stream);

}  // namespace okvfe
