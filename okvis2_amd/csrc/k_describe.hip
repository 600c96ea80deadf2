// k_describe.hip -- K6 BRISK2 descriptor, compaction + back-projection.
//
// Replaces brisk::BriskDescriptorExtractor::compute (behind cv::DescriptorExtractor::compute,
// okvis_cv/include/okvis/implementation/Frame.hpp:167; extractor built at
// okvis_frontend/src/Frontend.cpp:2410-2412, configured by setCameraProperties /
// setExtractionDirection at :239-251) and Frame::computeBackProjections
// (okvis_cv/include/okvis/implementation/Frame.hpp:178-193 ->
// cameras/implementation/PinholeCamera.hpp:574-593).
//
//   describe_setup_kernel  one thread per keypoint: border test, camera-aware matrix M.
//   describe_kernel  one wave per keypoint at a time, lane l = pattern point extra + l: the built-in pattern has 66
//                    points, its first two (centre, first hexagon point) are a second, 5 x 5-slot pass of lanes
//                    0..1 over the same patch.  <6, AWARE> / <5, AWARE>: camera-aware extraction only (the
//                    production mode), 6 / 5 waves per SIMD; <4, generic>: every mode (gradient orientation with
//                    its long pairs in LDS, upright, scale ladder, unaligned images, boxes beyond 11 x 11).  16
//                    workgroups of 4 waves per image, each wave walks the image's keypoints with stride 64
//                    (per-lane pattern constants, the lane's six pair entries, image parameters and the buffer
//                    resource are set up once per wave; wave-uniform values -- keypoint, M -- live in SGPRs, the next
//                    keypoint's arrive through scalar loads).  The pixels under the keypoint's pattern go straight
//                    from the image into a dense LDS patch (buffer_load ... lds, whole rows per instruction) while
//                    the per-sample set-up runs; every sample is a box sum with sub-pixel rim weights read from LDS
//                    (fixed trip counts, v_msad_u8 over dwords masked from an LDS table); 384 pair comparisons
//                    become 6 wave ballots -> 6 x u64 = 48 bytes.  All blocks of an image run on one XCD.  No
//                    integral image: the 4 B/px integral pass of the classic CPU formulation is gone.
//                    Bound by vector-ALU issue, the LDS gather and the latency of each keypoint's dependent chain
//                    together (540 VALU + 82 LDS wave-instructions per keypoint; ONE global round trip per
//                    keypoint: its patch); ~4.2 KB in + 48 B out per keypoint.
//   compact_kernel   removes the keypoints the extractor dropped (order preserved) and
//                    back-projects the survivors in FP64 (Gauss-Newton undistortion).
#include <limits.h>

#include "camera_dev.h"
#include "describe_setup_dev.h"
#include "okvfe_internal.h"

namespace okvfe {
namespace {

// ---- K6 -----------------------------------------------------------------------------------
// Box of half-side sigma_half centred at (xf, yf); returns 1024 * mean intensity.  Same integer /
// float operation sequence as the published BRISK smoothedIntensity (sub-pixel rim weights); the
// interior / edge sums are taken directly over the pixels (identical to integral-image sums).
[[maybe_unused]] constexpr int kMaxBox = 10;  // fast path: boxes of at most 11 x 11 pixels (sigma_half <= 4.75)
constexpr int kSmallBox = 4;  // second pass of the camera-aware-only kernel: boxes of at most 5 x 5 (sigma_half <= 2.0)
// WIDE-box instantiations (round 5: a pattern with wider smoothing -- what the vocabulary's statistics favour,
// tools/pattern/README.md -- must not fall off the fast path): boxes up to 21 x 21 (sigma_half <= 9.75) in the
// first pass, up to 10 x 10 (sigma_half <= 4.25) in the second
constexpr int kWideBox = 20;
constexpr int kWideSmallBox = 9;
// mask table geometry per kernel: counts = interior byte counts 0 .. MAXB - 1, row = dwords per entry
template <bool WIDE> struct BoxTab {
  static constexpr int kCounts = WIDE ? kWideBox : 10;
  static constexpr int kRowDw = WIDE ? 8 : 4;   // (MAXB + 5) / 4 mask dwords, padded to whole 16-byte reads
  static constexpr int kMaskDw = WIDE ? 6 : 3;
};
// LDS patch of one wave: [kZeroRowBytes of zeros][pixel rows, dense: pitch = 4 * dwords per row]
constexpr int kZeroRowBytes = 160;  // >= the widest patch row (152 B) + the 3-dword reads past a box
// LDS per workgroup = 4 patch buffers + values (1152 B) + short pairs (768 B) + box masks (640 B) + second-pass
// constants (160 B): <= 26880 B = 21
// allocation granules of 1280 B, six workgroups per CU (the 6-waves-per-SIMD form)
constexpr int kPatchBufBytes = 6032;
#ifndef OKVFE_DESC_WIDE_BUF
#define OKVFE_DESC_WIDE_BUF 7312  // 4 x 7312 + 2720 B of tables <= 32000 B = 25 allocation granules: five workgroups per CU
#endif
constexpr int kPatchDataBytes = kPatchBufBytes - kZeroRowBytes - 16;  // 16 B slack: 3-dword row reads
// floor(num / den) for 0 <= num < 2^31, den >= 1 and a quotient below 2^22 (here: 1024 * mean
// intensity): float reciprocal estimate (off by at most 1) + exact integer correction, instead of
// the ~40-instruction generic 32-bit division
// (q < 2^22 and den = Pattern::box_scaling2 ~ 4096: the 24-bit multiply is exact)
__device__ __forceinline__ int mul24i(int a, int b) {  // (__mul24 sign-extends both operands first)
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int div_nonneg(int num, int den) {
  int q = (int)((float)num * __builtin_amdgcn_rcpf((float)den));
  int r = num - mul24i(q, den);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (r < 0) {
      --q;
      r += den;
    } else if (r >= den) {
      ++q;
      r -= den;
    }
  }
  return q;
}

// FASTONLY: every box of the pattern is at most 11 x 11 (sigma_half <= 4.75, checked on the host): the plain-loop
// form and the test for it are compiled out.
// MAXB: largest box side minus one the fixed-trip form serves (kMaxBox; the second-pass samples of a pattern
// with more than 64 points are its innermost, smallest boxes and take kSmallBox: 4 row slots, 2 dwords).
// LATE_WAIT: the patch loads (buffer_load ... lds) are still in flight when the function is entered; it waits for
// them just before its first pixel read, so the ~80 instructions of per-sample set-up run under the loads' latency
// (only with FASTONLY: the one code path then reaches the wait in every lane).
template <bool FASTONLY = false, int MAXB = 10, bool LATE_WAIT = false, bool WIDETAB = false, typename PX>
__device__ __forceinline__ int smoothed_intensity(const PX& px, float xf, float yf,
                                                  float sigma_half, int scaling, int scaling2) {
  if (sigma_half < 0.5f) {
    const int x = (int)xf, y = (int)yf;
    const int r_x = (int)((xf - (float)x) * 1024.0f);
    const int r_y = (int)((yf - (float)y) * 1024.0f);
    const int r_x_1 = 1024 - r_x, r_y_1 = 1024 - r_y;
    int ret = r_x_1 * r_y_1 * px(y, x);
    ret += r_x * r_y_1 * px(y, x + 1);
    ret += r_x * r_y * px(y + 1, x + 1);
    ret += r_x_1 * r_y * px(y + 1, x);
    return (ret + 512) / 1024;
  }
  // scaling = (int)(4194304 / (4 sigma^2)) and scaling2 = (int)(scaling * 4 sigma^2 / 1024) come
  // from the pattern table (Pattern::box_scaling*, host_tables.cpp)
  const float x_1 = xf - sigma_half, x1 = xf + sigma_half;
  const float y_1 = yf - sigma_half, y1 = yf + sigma_half;
  const int x_left = (int)(x_1 + 0.5f), y_top = (int)(y_1 + 0.5f);
  const int x_right = (int)(x1 + 0.5f), y_bottom = (int)(y1 + 0.5f);
  float r_x_1 = (float)x_left - x_1;  r_x_1 = r_x_1 + 0.5f;
  float r_y_1 = (float)y_top - y_1;   r_y_1 = r_y_1 + 0.5f;
  float r_x1 = x1 - (float)x_right;   r_x1 = r_x1 + 0.5f;
  float r_y1 = y1 - (float)y_bottom;  r_y1 = r_y1 + 0.5f;
  const float fs = (float)scaling;
  float t;
  t = r_x_1 * r_y_1; const int A = (int)(t * fs);
  t = r_x1 * r_y_1;  const int B = (int)(t * fs);
  t = r_x1 * r_y1;   const int C = (int)(t * fs);
  t = r_x_1 * r_y1;  const int D = (int)(t * fs);
  const int r_x_1_i = (int)(r_x_1 * fs), r_y_1_i = (int)(r_y_1 * fs);
  const int r_x1_i = (int)(r_x1 * fs), r_y1_i = (int)(r_y1 * fs);
  int ret;
  int upper = 0, middle = 0, left = 0, right = 0, bottom = 0;
  const int bw = x_right - x_left, bh = y_bottom - y_top;  // >= 1 for sigma_half >= 0.5
  if constexpr (PX::kFixedTrip) {
  if (FASTONLY || __all(bw <= MAXB && bh <= MAXB)) {
    // Fixed trip counts, no data-dependent selects: the top and the bottom row are read once each,
    // then MAXB - 1 interior slots, where a slot past the box (dy >= bh) reads the all-zero
    // row of the patch instead, so every accumulation is unconditional.  Interior columns
    // x_left+1 .. x_right-1 (<= 9 bytes) lie in at most 3 aligned dwords of the patch row: one
    // byte mask per dword, then one v_msad_u8 per dword sums 4 pixels.
    const int cl = x_left - px.x0, cr = x_right - px.x0;  // patch columns of the rim pixels
    const int xi0 = cl + 1;
    const int q0 = xi0 >> 2;
    const int lo = xi0 & 3, ni = bw - 1;
    constexpr int kDw = (MAXB + 5) / 4;  // interior <= MAXB - 1 bytes from byte 0..3 of the first dword
    // byte masks of the interior columns in those dwords: one 16-byte read of the (first byte, count) table in LDS
    // (kernel prologue: fill_box_masks) instead of ~8 VALU per dword
    static_assert(kDw <= BoxTab<WIDETAB>::kMaskDw && MAXB <= BoxTab<WIDETAB>::kCounts, "mask table of the kernel");
    const uint32_t* mp = px.masks + ((mul24i(lo, BoxTab<WIDETAB>::kCounts) + ni) << (BoxTab<WIDETAB>::kRowDw == 8 ? 3 : 2));
    const uint4 mrow = *reinterpret_cast<const uint4*>(mp);
    uint32_t m[kDw];
    m[0] = mrow.x;
    if constexpr (kDw > 1) m[1] = mrow.y;
    if constexpr (kDw > 2) m[2] = mrow.z;
    if constexpr (kDw > 3) m[3] = mrow.w;
    if constexpr (kDw > 4) {
      const uint2 mhi = *reinterpret_cast<const uint2*>(mp + 4);
      m[4] = mhi.x;
      if constexpr (kDw > 5) m[5] = mhi.y;
    }
    // Rows are addressed by 32-bit byte offsets from the first patch row (the zero row lies
    // kZeroRowBytes before it); every product below has both factors under 2^23 (weights <= 2^22,
    // pixel sums <= 81 * 255), so the 24-bit multiplies give the same low 32 bits as the
    // v_mul_lo_u32 / 64-bit v_mad_u64_u32 chains (pointer arithmetic included) the plain expressions
    // compile to.  (Measured: no change of the kernel's time -- it is latency-bound, and
    // v_mul_lo_u32 issues at the same rate as the 24-bit forms on this part.)
    const int pitch = px.pitch;
    int run = mul24i(y_top - px.y0, pitch);
    int pl, pr;
    auto read_row = [&](int off, uint32_t acc) -> uint32_t {
      pl = px.patch[off + cl];
      pr = px.patch[off + cr];
      const uint32_t* d = reinterpret_cast<const uint32_t*>(px.patch + off) + q0;
      // v_msad_u8 adds |pixel - 0xFF| = 255 - pixel for the bytes the mask selects and skips the others: one
      // instruction per dword instead of v_and + v_sad_u8; the row sums come back as 255 * count - accumulator.
      // A slot past the box reads the zero row and adds 255 per selected byte, which the same identity absorbs:
      // sum over the real rows = 255 * ni * (slots) - accumulator, whatever the number of real rows.
#pragma unroll
      for (int j = 0; j < kDw; ++j) acc = __builtin_amdgcn_msad_u8(d[j], m[j], acc);
      return acc;
    };
    const int full = mul24i(ni, 255);
    if constexpr (LATE_WAIT) {
      static_assert(!LATE_WAIT || FASTONLY, "late wait needs the single code path");
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the staged rows have landed in LDS
      __builtin_amdgcn_wave_barrier();
    }
    upper = full - (int)read_row(run, 0u);
    ret = mul24i(A, pl) + mul24i(B, pr);
    bottom = full - (int)read_row(run + mul24i(bh, pitch), 0u);
    ret += mul24i(D, pl) + mul24i(C, pr);
    uint32_t mid = 0u;
#pragma unroll
    for (int dy = 1; dy < MAXB; ++dy) {
      run += pitch;
      mid = read_row(dy < bh ? run : -kZeroRowBytes, mid);
      left += pl;
      right += pr;
    }
    middle = mul24i(full, MAXB - 1) - (int)mid;
    ret += mul24i(upper, r_y_1_i) + mul24i(middle, scaling) + mul24i(left, r_x_1_i) + mul24i(right, r_x1_i) +
           mul24i(bottom, r_y1_i);
    return div_nonneg(ret + scaling2 / 2, scaling2);
  }
  }
  {
    ret = A * px(y_top, x_left);
    ret += B * px(y_top, x_right);
    ret += C * px(y_bottom, x_right);
    ret += D * px(y_bottom, x_left);
    for (int x = x_left + 1; x < x_right; ++x) {
      upper += px(y_top, x);
      bottom += px(y_bottom, x);
    }
    for (int y = y_top + 1; y < y_bottom; ++y) {
      left += px(y, x_left);
      right += px(y, x_right);
      for (int x = x_left + 1; x < x_right; ++x) middle += px(y, x);
    }
  }
  ret += upper * r_y_1_i + middle * scaling + left * r_x_1_i + right * r_x1_i + bottom * r_y1_i;
  return div_nonneg(ret + scaling2 / 2, scaling2);
}

// sample position of this lane's pattern point under M; ok = box inside the image (NaN-safe)
__device__ __forceinline__ bool sample_pos(const float M[4], float kx, float ky, float px, float py,
                                           float sg, int w, int h, float* xf, float* yf) {
  float a = M[0] * px;
  float b = M[1] * py;
  a = a + b;
  *xf = kx + a;
  float c = M[2] * px;
  float d = M[3] * py;
  c = c + d;
  *yf = ky + c;
  const float x_1 = *xf - sg, x1 = *xf + sg, y_1 = *yf - sg, y1 = *yf + sg;
  return (x_1 >= 0.0f && y_1 >= 0.0f && x1 < (float)(w - 1) && y1 < (float)(h - 1));
}

// A/B switches of the round-5 latency experiments (tools/lab/descab5.sh; defaults = the kept forms)
#ifndef OKVFE_DESC_LATE_WAIT
#define OKVFE_DESC_LATE_WAIT 1   // per-sample set-up runs under the patch loads' latency
#endif
#ifndef OKVFE_DESC_PAIRS_REG
#define OKVFE_DESC_PAIRS_REG 1   // the lane's six pair entries live in registers (0: read from the LDS table per keypoint)
#endif
#ifndef OKVFE_DESC_PAIRS_OPAQUE
#define OKVFE_DESC_PAIRS_OPAQUE 0  // 1: the gather addresses derived from them are recomputed per keypoint (no hoisting)
#endif
#ifndef OKVFE_DESC_WG_WAVES
#define OKVFE_DESC_WG_WAVES 4
#endif
constexpr int kDescWaves = OKVFE_DESC_WG_WAVES;
#ifndef OKVFE_DESC_BLOCKS
#define OKVFE_DESC_BLOCKS 16
#endif
constexpr int kDescBlocksPerImage = OKVFE_DESC_BLOCKS;

// table[((first byte 0..3) * kCounts + count) * kRowDw + j]: 0xFF in every byte of dword j (j < kMaskDw) that belongs
// to the run of `count` interior bytes starting at byte `first byte` of the first dword (the rest of a row: 0)
template <bool WIDE>
__device__ __forceinline__ void fill_box_masks(uint32_t* table, int tid, int nthreads) {
  using T = BoxTab<WIDE>;
  for (int e = tid; e < 4 * T::kCounts; e += nthreads) {
    const int lo = e / T::kCounts, ni = e - lo * T::kCounts;
    for (int j = 0; j < T::kRowDw; ++j) {
      uint32_t mk = 0u;
      for (int b2 = 0; b2 < 4; ++b2) {
        const int pos = 4 * j + b2;
        if (j < T::kMaskDw && pos >= lo && pos < lo + ni) mk |= 0xFFu << (8 * b2);
      }
      table[T::kRowDw * e + j] = mk;
    }
  }
}

struct GlobalPx {  // direct reads from the image (fallback when the patch does not fit in LDS)
  static constexpr bool kFixedTrip = false;
  const uint8_t* img;
  int w;
  int x0 = 0, pitch = 0;  // unused: the fixed-trip path is compiled out for this reader
  __device__ __forceinline__ int operator()(int y, int x) const { return img[(size_t)y * w + x]; }
  __device__ __forceinline__ const uint8_t* row8(int) const { return nullptr; }
  __device__ __forceinline__ const uint8_t* zero_row8() const { return nullptr; }
};
struct PatchPx {   // reads from the keypoint's patch staged in LDS
  static constexpr bool kFixedTrip = true;
  const uint8_t* patch;  // first pixel row; the zero row lies kZeroRowBytes before it
  int x0, y0, pitch;
  const uint32_t* masks;  // LDS table of fill_box_masks
  __device__ __forceinline__ int operator()(int y, int x) const {
    return patch[(y - y0) * pitch + (x - x0)];
  }
  __device__ __forceinline__ const uint8_t* row8(int y) const {
    return patch + (y - y0) * pitch;
  }
  __device__ __forceinline__ const uint8_t* zero_row8() const {
    return patch - kZeroRowBytes;
  }
};

// One THREAD per keypoint: border test and, in camera-aware mode, M = J [e_x e_y] / fu.  Done here
// because in the wave-per-keypoint kernel all 64 lanes would repeat the same ~150 instructions.
// M goes to the first 16 bytes of the keypoint's (not yet written) descriptor slot.
__global__ __launch_bounds__(256) void describe_setup_kernel(
    int w, int h, const Pattern* __restrict__ pat, const ImageParams* __restrict__ prm,
    const float* const* __restrict__ rays, const float* const* __restrict__ jac,
    const okvfe_keypoint* __restrict__ kps_in, int kp_cap, const int32_t* __restrict__ kp_count_in,
    okvfe_keypoint* __restrict__ kps_tmp, uint8_t* __restrict__ desc_tmp, uint8_t* __restrict__ valid_tmp,
    const PatternScales* __restrict__ scales, const uint8_t* __restrict__ images, int extra_box) {
  const int img = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kp_count_in[img]) return;
  const size_t slot = (size_t)img * kp_cap + k;
  const DescribeSetup ds{pat, prm, rays, jac, kps_tmp, desc_tmp, valid_tmp, scales, images, extra_box};
  describe_setup_one(ds, w, h, img, slot, kps_in[slot]);
}

// One wave per keypoint, lane i = pattern point i.  The pixels under the keypoint's pattern
// (<= 80 x 96) are staged once in LDS with coalesced dword loads; all box sums then read LDS.
// kWavesPerSimd = 6 (80 VGPRs) is the fastest form when nearly every patch fits the LDS buffer; the
// banded path of the wide-angle cameras spills at 80 registers and runs 26 % faster with 96 (five
// waves per SIMD): launch_describe picks the instantiation from the cameras' patch statistics.
// AWARE: every image of the launch is extracted camera-aware (the production mode, Frontend.cpp:2410-2412 with
// setExtractionDirection): the gradient-orientation pass and the fixed-box staging are compiled out -- a
// third of the kernel's code and the registers that had to live across it.  It is launched for patterns whose boxes
// all fit the fixed-trip box sum only (sigma_half <= 4.75: capi_detect.cpp), so the plain-loop form is compiled out too.
// WIDE: the pattern's boxes go up to 21 x 21 (first pass) / 9 x 9 (second pass): same code with more row slots and a
// wider mask table, four workgroups per CU.
template <int kWavesPerSimd, bool AWARE = false, bool WIDE = false>
__global__ __launch_bounds__(64 * kDescWaves) __attribute__((amdgpu_waves_per_eu(kWavesPerSimd, 8))) void describe_kernel(
    const uint8_t* __restrict__ images, int w, int h, const Pattern* __restrict__ pat,
    const ImageParams* __restrict__ prm, const float* const* __restrict__ rays,
    const float* const* __restrict__ jac, const okvfe_keypoint* __restrict__ kps_in, int kp_cap,
    const int32_t* __restrict__ kp_count_in, okvfe_keypoint* __restrict__ kps_tmp,
    uint8_t* __restrict__ desc_tmp, uint8_t* __restrict__ valid_tmp, int n_images, int tiles,
    uint32_t inv_tiles, const PatternScales* __restrict__ scales) {
  // (the 96-register instantiation leaves LDS for five workgroups of 32 KB: its waves get 7.5 KB buffers)
  // (the generic form runs four workgroups per CU: 4 x 7312 + 2720 + 8800 B of long pairs = 40768 B, 128 registers)
  // (WIDE: 4 x 9072 + 4640 B of tables, resp. 4 x 6864 + 13444 with the long pairs: <= 40960 B, four workgroups per CU)
  constexpr int kBufBytes = WIDE ? (AWARE ? 9072 : 6864) : (kWavesPerSimd >= 6 ? kPatchBufBytes : OKVFE_DESC_WIDE_BUF);
  constexpr int kDataBytes = kBufBytes - kZeroRowBytes - (WIDE ? 32 : 16);  // slack: the row reads past a box (3 / 6 dwords)
  constexpr int kFirstBox = WIDE ? kWideBox : kMaxBox, kSecondBox = WIDE ? kWideSmallBox : kSmallBox;
  __shared__ __attribute__((aligned(16))) uint8_t patches[kDescWaves][kBufBytes];
  __shared__ int values[kDescWaves][kPatternPoints];
  // the short pairs (i | j << 8), once per workgroup: read 12 x per keypoint from global memory
  // they were a third dependent round trip in every keypoint's chain
  __shared__ __attribute__((aligned(16))) uint32_t box_masks[4 * BoxTab<WIDE>::kCounts * BoxTab<WIDE>::kRowDw];
  fill_box_masks<WIDE>(box_masks, threadIdx.x, 64 * kDescWaves);
  // constants of the second-pass samples (points 0 .. kPatternPoints - 65) at the base scale: read from LDS per
  // keypoint (global loads here were a second memory round trip in every keypoint's chain)
  __shared__ float second_f[3][kPatternPoints - 64];
  __shared__ int second_i[2][kPatternPoints - 64];
  if (threadIdx.x < kPatternPoints - 64) {
    second_f[0][threadIdx.x] = pat->px[threadIdx.x];
    second_f[1][threadIdx.x] = pat->py[threadIdx.x];
    second_f[2][threadIdx.x] = pat->sigma_half[threadIdx.x];
    second_i[0][threadIdx.x] = pat->box_scaling[threadIdx.x];
    second_i[1][threadIdx.x] = pat->box_scaling2[threadIdx.x];
  }
  // generic form: the long pairs of the gradient orientation as {i | j << 8, wdx | wdy << 16} in LDS (their four
  // global tables were ~64 loads per lane in every keypoint's chain); weights beyond 16 bits keep the global path
  __shared__ uint2 long_tab[AWARE ? 1 : kMaxLongPairs];
  __shared__ int long_wide;
  if constexpr (!AWARE) {
    if (threadIdx.x == 0) long_wide = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < pat->n_long; t += 64 * kDescWaves) {
      const int wx = pat->long_wdx[t], wy = pat->long_wdy[t];
      if (wx < -32768 || wx > 32767 || wy < -32768 || wy > 32767) long_wide = 1;
      long_tab[t] = make_uint2((uint32_t)pat->long_i[t] | ((uint32_t)pat->long_j[t] << 8),
                               ((uint32_t)wx & 0xFFFFu) | ((uint32_t)wy << 16));
    }
  }
  __shared__ uint16_t short_pairs[384];
  for (int t = threadIdx.x; t < 384; t += 64 * kDescWaves)
    short_pairs[t] = t < pat->n_short ? (uint16_t)(pat->short_i[t] | (pat->short_j[t] << 8)) : (uint16_t)0;
  __syncthreads();
  // all keypoint blocks of an image run on the same XCD (block L -> XCD L % 8), so its pixels
  // are fetched from HBM into ONE L2 instead of into all eight
  // (two thirds of the blocks find no keypoint in their slots and leave right here, so the
  // block -> (image, tile) split uses a host-computed reciprocal instead of integer divisions)
  int img, tile;
  {
    const uint32_t L = blockIdx.x, n8 = (uint32_t)n_images & ~7u, full = n8 * (uint32_t)tiles;
    const uint32_t slot = L < full ? L >> 3 : L - full;
    uint32_t g = (uint32_t)(((uint64_t)slot * inv_tiles) >> 32);  // slot / tiles, off by <= 1
    if (g * (uint32_t)tiles > slot) --g;
    if ((g + 1) * (uint32_t)tiles <= slot) ++g;
    img = L < full ? (int)(g * 8u + (L & 7u)) : (int)(n8 + g);
    tile = (int)(slot - g * (uint32_t)tiles);
  }
  // wave index as a scalar: everything indexed by the keypoint (slot, kp, M) is wave-uniform and
  // belongs in SGPRs
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  const int n = kp_count_in[img];
  if (tile * kDescWaves + wv >= n) return;  // whole wave exits; no block-wide barriers below
  const uint8_t* im = images + (size_t)img * w * h;
  const ImageParams ip = prm[img];
  int border = pat->border;  // (per keypoint with scale-invariant extraction)
  // lane l samples point extra + l; a pattern of more than 64 points (the built-in one has 66) gives its first
  // `extra` points -- the innermost, smallest boxes -- to lanes 0..extra-1 in a second pass over the same patch
  const int extra = __builtin_amdgcn_readfirstlane(pat->n_points > 64 ? pat->n_points - 64 : 0);
  const bool active = extra + lane < pat->n_points;  // <= kPatternPoints (okvfe_set_pattern may install fewer samples)
  const bool active2 = lane < extra;
  const int li = active ? extra + lane : 0;
  float px = pat->px[li], py = pat->py[li], sg = pat->sigma_half[li];
  int bsc = pat->box_scaling[li], bsc2 = pat->box_scaling2[li];
  okvfe_keypoint kp;
  float M[4];
  float xf, yf;
  int scale_idx = 0;  // this keypoint's rung of the scale ladder (wave-uniform)
  uint32_t my_pairs[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    my_pairs[j] = (uint32_t)short_pairs[(2 * j) * 64 + lane] | ((uint32_t)short_pairs[(2 * j + 1) * 64 + lane] << 16);
  int* vals = values[wv];
  uint8_t* patch = patches[wv] + kZeroRowBytes;
  if (lane < kZeroRowBytes / 4) reinterpret_cast<uint32_t*>(patches[wv])[lane] = 0u;
  // (the AWARE form is only launched on dword-aligned images of a width that is a multiple of 4)
  const bool dword_ok = AWARE || ((w % 4 == 0) && ((reinterpret_cast<uintptr_t>(images) & 3) == 0));
  const __amdgpu_buffer_rsrc_t img_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(im), 0, w * h, 0x00027000);

  // stages the pixels [bx0..bx1] x [by0..by1] (inside the image) into LDS; false if too large
  auto stage_patch = [&](int bx0, int bx1, int by0, int by1, PatchPx* ppx, bool* in_flight = nullptr) -> bool {
    // the bounds derive from the keypoint (same in every lane): tell the compiler they are scalar
    bx0 = __builtin_amdgcn_readfirstlane(bx0);
    bx1 = __builtin_amdgcn_readfirstlane(bx1);
    by0 = __builtin_amdgcn_readfirstlane(by0);
    by1 = __builtin_amdgcn_readfirstlane(by1);
    const int px0 = bx0 & ~3;
    const int pw = bx1 - px0 + 1, ph = by1 - by0 + 1;
    // 16 bytes per lane (buffer_load_dwordx4 ... lds, new on gfx950): a quarter of the load
    // instructions for the same pixels.  The staging is bound by the number of vector-memory
    // requests in flight at L2 latency, not by bytes (stage-only 0.43 ms of the kernel's 0.51 with
    // dword requests), so fewer, wider requests are what shortens it.  Rows are padded to 16 B.
    int nq = (pw + 15) >> 4;  // 16-byte chunks per patch row
#ifdef OKVFE_DESC_ODD_PITCH
    nq |= 1;  // A/B: odd chunk count = rows step through all bank phases (tools/lab: LDS bank conflicts of the box sums)
#endif
    if (dword_ok && nq >= 1 && nq * 16 <= kZeroRowBytes - 8) {
      const int pitch = nq * 16;
      const uint32_t inv = (65536u + (uint32_t)nq - 1u) / (uint32_t)nq;
      const int R = (int)((64u * inv) >> 16);
      const int trips = (ph + R - 1) / R;
      if (trips * R * pitch <= kDataBytes) {  // wave-uniform
        __builtin_amdgcn_wave_barrier();
        const uint32_t rr = ((uint32_t)lane * inv) >> 16;
        const uint32_t c = (uint32_t)lane - rr * (uint32_t)nq;
        const uint32_t src_lane = rr * (uint32_t)w + c * 16u;
        const int src0 = by0 * w + px0;
        if ((int)rr < R) {
          for (int it = 0; it < trips; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                img_rsrc, (__attribute__((address_space(3))) void*)(patch + it * R * pitch), 16,
                (int)src_lane, src0 + it * R * w, 0, 0);
        }
        if (AWARE && in_flight && OKVFE_DESC_LATE_WAIT) {
          *in_flight = true;  // the caller's first box sum waits (smoothed_intensity<.., LATE_WAIT>)
        } else {
          __builtin_amdgcn_s_waitcnt(0);
          __builtin_amdgcn_wave_barrier();
        }
        ppx->patch = patch;
        ppx->x0 = px0;
        ppx->y0 = by0;
        ppx->pitch = pitch;
        ppx->masks = box_masks;
        return true;
      }
    }
    const int ndw = (pw + 3) >> 2;  // dwords per patch row
    const int pitch = ndw * 4;
    if (pitch > kZeroRowBytes - 8 || ndw < 1) return false;  // wave-uniform
    __builtin_amdgcn_wave_barrier();
    if (dword_ok) {
      // Each trip moves R = 64 / ndw whole patch rows straight from the image into LDS
      // (buffer_load ... lds: lane l lands at the trip's LDS base + 4 l, which is exactly the
      // dense row-major patch when lane = row * ndw + dword).  No VGPRs, no LDS stores, no index
      // arithmetic per trip: scalar row offset, constant per-lane offset; lanes whose address falls
      // outside the image read 0.
      const uint32_t inv = (65536u + (uint32_t)ndw - 1u) / (uint32_t)ndw;  // i / ndw, i < 2730
      const int R = (int)((64u * inv) >> 16);
      const int trips = (ph + R - 1) / R;
      if (trips * R * pitch > kDataBytes) return false;  // wave-uniform
      const uint32_t rr = ((uint32_t)lane * inv) >> 16;
      const uint32_t c = (uint32_t)lane - rr * (uint32_t)ndw;
      const uint32_t src_lane = rr * (uint32_t)w + c * 4u;
      const int src0 = by0 * w + px0;
      if ((int)rr < R) {
        for (int it = 0; it < trips; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              img_rsrc, (__attribute__((address_space(3))) void*)(patch + it * R * pitch), 4,
              (int)src_lane, src0 + it * R * w, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0);
    } else {
      if (ph * pitch > kDataBytes) return false;
      for (int r = 0; r < ph; ++r)
        for (int c = lane; c < pw; c += 64) patch[r * pitch + c] = im[(size_t)(by0 + r) * w + px0 + c];
    }
    __builtin_amdgcn_wave_barrier();
    ppx->patch = patch;
    ppx->x0 = px0;
    ppx->y0 = by0;
    ppx->pitch = pitch;
    ppx->masks = box_masks;
    return true;
  };
  // values of all samples under the current M; false when a box leaves the image
  // gradient mode samples twice under the same fixed box (unrotated for the orientation, then rotated): the patch
  // staged for the first call serves the second one
  PatchPx kept_patch{};
  bool have_kept = false;
  auto sample_all = [&](bool fixed_box, bool reuse = false) -> bool {
    const bool ok = sample_pos(M, kp.x, kp.y, px, py, sg, w, h, &xf, &yf);
    // second-pass sample of lanes < extra: its constants are re-read per keypoint and its position is computed
    // twice (here for the inside-the-image test, again after the first pass) so that nothing of it stays in
    // registers across the first pass's box sum
    int l2 = active2 ? lane : 0;
    const bool ladder = !AWARE && scales != nullptr;
    const int sc2 = ladder ? scale_idx : 0;
    bool ok2 = true;
    auto second_pos = [&](float* x2, float* y2, float* s2) -> bool {
      const float px2 = ladder ? scales->px[sc2][l2] : second_f[0][l2];
      const float py2 = ladder ? scales->py[sc2][l2] : second_f[1][l2];
      *s2 = ladder ? scales->sigma_half[sc2][l2] : second_f[2][l2];
      return sample_pos(M, kp.x, kp.y, px2, py2, *s2, w, h, x2, y2);
    };
    if (extra > 0) {  // wave-uniform
      float x2, y2, s2;
      ok2 = second_pos(&x2, &y2, &s2) || !active2;
    }
    if (!__all((ok || !active) && ok2)) return false;
    int bx0, bx1, by0, by1;
    if (fixed_box) {  // circle of the pattern: covers every rotation (upright / gradient modes)
      const int cx = (int)kp.x, cy = (int)kp.y;
      bx0 = cx - border; bx1 = cx + border + 1; by0 = cy - border; by1 = cy + border + 1;
      bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
      bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
    } else {
      // superset of all boxes under M: |M p|_x <= |row_x(M)| * |p| and |p| + sigma_half stays
      // below border - 1 for this pattern; the boxes themselves were checked to lie in the image
      // (only a superset is needed: hardware sqrt estimate, rounded up by the 1.001 factor)
      float nx = M[0] * M[0], t = M[1] * M[1];
      nx = __builtin_amdgcn_sqrtf(nx + t) * 1.001f;
      float ny = M[2] * M[2];
      t = M[3] * M[3];
      ny = __builtin_amdgcn_sqrtf(ny + t) * 1.001f;
      const float ex = fmaxf(nx, 1.0f) * (float)(border - 1) + 1.5f;
      const float ey = fmaxf(ny, 1.0f) * (float)(border - 1) + 1.5f;
      bx0 = (int)floorf(kp.x - ex); bx1 = (int)ceilf(kp.x + ex) + 1;
      by0 = (int)floorf(kp.y - ey); by1 = (int)ceilf(kp.y + ey) + 1;
      bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
      bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
    }
    PatchPx ppx;
    int v = 0, v2 = 0;
    bool in_flight = false;
    const bool reused = !AWARE && reuse && have_kept;  // wave-uniform
    if (reused) ppx = kept_patch;
    if (reused || stage_patch(bx0, bx1, by0, by1, &ppx, &in_flight)) {
      if (!AWARE && fixed_box) {
        kept_patch = ppx;
        have_kept = true;
      }
      // (no exec-mask change around the box sums: lanes without a sample carry point 0's constants, whose box lies
      // in the patch, so both passes are straight-line code the scheduler may interleave)
      if (AWARE && in_flight)  // wave-uniform
        v = smoothed_intensity<AWARE, kFirstBox, AWARE, WIDE>(ppx, xf, yf, sg, bsc, bsc2);
      else
        v = smoothed_intensity<AWARE, kFirstBox, false, WIDE>(ppx, xf, yf, sg, bsc, bsc2);
      if (extra > 0) {  // wave-uniform; AWARE: the host checked sigma_half <= 2.0 for these points (5 x 5 boxes)
        const int b1 = ladder ? scales->box_scaling[sc2][l2] : second_i[0][l2];
        const int b2 = ladder ? scales->box_scaling2[sc2][l2] : second_i[1][l2];
        asm volatile("" : "+v"(l2));  // opaque: the position is recomputed, not carried over the first pass
        float xf2, yf2, sg2;
        second_pos(&xf2, &yf2, &sg2);
        v2 = smoothed_intensity<AWARE, kSecondBox, false, WIDE>(ppx, xf2, yf2, sg2, b1, b2);  // (generic form: run-time check, plain loops beyond 5 x 5)
      }
    } else {
      if (extra > 0 && active2) {  // rare path (patch larger than the wave's buffer): the few extra samples read the image
        const GlobalPx gpx2{im, w};
        float xf2, yf2, sg2;
        second_pos(&xf2, &yf2, &sg2);
        const int b1 = ladder ? scales->box_scaling[sc2][l2] : second_i[0][l2];
        const int b2 = ladder ? scales->box_scaling2[sc2][l2] : second_i[1][l2];
        v2 = smoothed_intensity(gpx2, xf2, yf2, sg2, b1, b2);
      }
      // The patch does not fit in the wave's LDS buffer (wide-angle cameras stretch the camera-aware
      // pattern towards the image rim: fu = 350 on 640 px gives |M| up to ~1.6).  It is staged in
      // horizontal BANDS instead: a band holds as many rows as fit, every lane computes its sample in
      // the first band that contains all rows of its box, and the next band starts at the topmost
      // row still needed (a box is at most 12 rows tall, a band at least 39).  Only patches wider
      // than 152 px fall back to direct image reads.
      const int px0 = __builtin_amdgcn_readfirstlane(bx0) & ~3;
      const int pitch16 = ((__builtin_amdgcn_readfirstlane(bx1) - px0 + 1 + 15) >> 4) << 4;
      int rows_fit = 0;  // whole trips of R rows (stage_patch: R = 64 / chunks per row)
      if (pitch16 <= kZeroRowBytes - 8) {
        const int R = 64 / (pitch16 >> 4);
        rows_fit = (kDataBytes / (R * pitch16)) * R;
      }
      // rows this lane's box touches (one spare row either side, inside the staged rectangle)
      const int fy0 = __builtin_amdgcn_readfirstlane(by0), fy1 = __builtin_amdgcn_readfirstlane(by1);
      int ly0 = (int)floorf(yf - sg) - 1, ly1 = (int)ceilf(yf + sg) + 1;
      ly0 = ly0 < fy0 ? fy0 : ly0;
      ly1 = ly1 > fy1 ? fy1 : ly1;
      bool done = !active;
      int band0 = __builtin_amdgcn_readfirstlane(by0);
      bool banded = rows_fit >= 24 && dword_ok;
      while (banded && !__all(done)) {
        int band1 = band0 + rows_fit - 1;
        band1 = band1 > by1 ? by1 : band1;
        if (!stage_patch(bx0, bx1, band0, band1, &ppx)) {  // cannot happen for rows_fit rows; stay exact anyway
          banded = false;
          break;
        }
        const bool mine = !done && ly0 >= band0 && ly1 <= band1;
        if (mine) {
          v = smoothed_intensity<AWARE, kFirstBox, false, WIDE>(ppx, xf, yf, sg, bsc, bsc2);
          done = true;
        }
        if (band1 >= by1) break;
        // next band: the topmost row a remaining lane needs (wave minimum), at least one row further
        int need = done ? INT_MAX : ly0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) need = min(need, __shfl_xor(need, d));
        need = __builtin_amdgcn_readfirstlane(need);
        if (need == INT_MAX) break;
        band0 = need > band0 ? need : band0 + 1;
      }
      if (!__all(done)) {  // too wide for the buffer (or a box that fits no band): direct reads
        const GlobalPx gpx{im, w};
        if (!done) v = smoothed_intensity(gpx, xf, yf, sg, bsc, bsc2);
      }
    }
    __builtin_amdgcn_wave_barrier();
    vals[extra + lane] = v;  // extra + 63 < kPatternPoints
    if (active2) vals[lane] = v2;
    __builtin_amdgcn_wave_barrier();
    return true;
  };

  // The wave walks the image's keypoints wv, wv + tiles * kDescWaves, ...: the per-lane pattern
  // constants, the image's parameters and the buffer resource above are set up once per wave
  // instead of once per keypoint.
  // (x, y), M and the valid flag of a keypoint are loaded ONE KEYPOINT AHEAD: with the pattern
  // tables in LDS the chain of a keypoint is then a single global round trip (the patch) instead of
  // three.  The loads are wave-uniform; their values move to SGPRs when the keypoint's turn comes.
  const int k_step = tiles * kDescWaves;
  // scalar loads (the constant address space forces s_load): the prefetched values of the NEXT keypoint then wait in
  // SGPRs instead of seven VGPRs across this keypoint's box sums (that was what spilled at 80 registers).  Safe on
  // the scalar cache: a slot's position / M / valid byte are written before this kernel starts and read here before
  // this wave -- the only writer of the slot -- overwrites them.
  typedef const float __attribute__((address_space(4))) * cfloat_p;
  typedef const uint8_t __attribute__((address_space(4))) * cbyte_p;
  float nxt_x = 0.f, nxt_y = 0.f, nxt_M0 = 0.f, nxt_M1 = 0.f, nxt_M2 = 0.f, nxt_M3 = 0.f;
  int nxt_valid = 0;
  auto fetch = [&](int kk) {
    const size_t sl = (size_t)img * kp_cap + kk;
    const cfloat_p pxy = (cfloat_p)(uintptr_t)(&kps_in[sl].x);
    const cfloat_p pm = (cfloat_p)(uintptr_t)(desc_tmp + sl * OKVFE_DESC_BYTES);
    nxt_x = pxy[0];
    nxt_y = pxy[1];
    nxt_M0 = pm[0];
    nxt_M1 = pm[1];
    nxt_M2 = pm[2];
    nxt_M3 = pm[3];
    nxt_valid = (int)((cbyte_p)(uintptr_t)valid_tmp)[sl];
  };
  fetch(tile * kDescWaves + wv);
  for (int k = tile * kDescWaves + wv; k < n; k += k_step) {
  // opaque to the optimiser: expressions of the lane constants are NOT hoisted out of the loop
  // (they would cost ~25 more live VGPRs and push the kernel below 6 waves/SIMD)
  asm volatile("" : "+v"(px), "+v"(py), "+v"(sg), "+v"(bsc), "+v"(bsc2), "+v"(lane));
#if OKVFE_DESC_PAIRS_OPAQUE
  asm volatile("" : "+v"(my_pairs[0]), "+v"(my_pairs[1]), "+v"(my_pairs[2]));  // (nor the 12 gather addresses derived from these)
#endif
  const size_t slot = (size_t)img * kp_cap + k;
  auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  kp.x = uni(nxt_x);
  kp.y = uni(nxt_y);
  // border test and (camera-aware mode) the matrix M come from describe_setup_kernel
  const int vbyte = __builtin_amdgcn_readfirstlane(nxt_valid);  // valid | scale index << 1
  bool valid = (vbyte & 1) != 0;
  if (!AWARE && scales) {  // wave-uniform: this keypoint's pattern scale (the AWARE form is launched without a ladder)
    const int sc = vbyte >> 1;
    scale_idx = sc;
    px = scales->px[sc][li];
    py = scales->py[sc][li];
    sg = scales->sigma_half[sc][li];
    bsc = scales->box_scaling[sc][li];
    bsc2 = scales->box_scaling2[sc][li];
    border = scales->border[sc];
  }
  M[0] = uni(nxt_M0); M[1] = uni(nxt_M1); M[2] = uni(nxt_M2); M[3] = uni(nxt_M3);
  if (k + k_step < n) fetch(k + k_step);  // scalar branch
  bool new_angle = false;
  const int mode = AWARE ? (int)kCameraAware : (int)ip.mode;
  if (valid && mode == kGradient) {
    valid = sample_all(true);
    if (valid) {
      int d0 = 0, d1 = 0;
      if constexpr (!AWARE) {
        if (long_wide == 0) {  // block-uniform
          const int nl = pat->n_long;
          for (int l = lane; l < nl; l += 64) {
            const uint2 e = long_tab[l];
            const int delta_t = vals[e.x & 255u] - vals[(e.x >> 8) & 255u];
            d0 += delta_t * (int)(short)(e.y & 0xFFFFu) / 1024;
            d1 += delta_t * ((int)e.y >> 16) / 1024;
          }
        } else {
          for (int l = lane; l < pat->n_long; l += 64) {
            const int delta_t = vals[pat->long_i[l]] - vals[pat->long_j[l]];
            d0 += delta_t * pat->long_wdx[l] / 1024;
            d1 += delta_t * pat->long_wdy[l] / 1024;
          }
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        d0 += __shfl_xor(d0, d);
        d1 += __shfl_xor(d1, d);
      }
      int best_k = 0;
      if (d0 != 0 || d1 != 0) {
        // exact arg-max over the 1024 directions (smallest index among equal maxima, as the serial scan of the
        // published code finds it), evaluated only where it can lie: the dot product is a sampled sinusoid, 32 steps
        // from its peak it has dropped by 632 |d| while the rounding of the tables moves any sample by < |d|, and the
        // float estimate of the peak is good to a millionth of a step -- so lane l tests step k_est - 32 + l
        const float ang = atan2f((float)d1, (float)d0);
        const int k_est = (int)lrintf(ang * (1024.0f / 6.2831853071795864769f));
        int bk = (k_est - 32 + lane) & 1023;
        long long best = (long long)d0 * pat->rot_cos[bk] + (long long)d1 * pat->rot_sin[bk];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const long long ob = __shfl_xor(best, d);
          const int ok2 = __shfl_xor(bk, d);
          if (ob > best || (ob == best && ok2 < bk)) {
            best = ob;
            bk = ok2;
          }
        }
        best_k = bk;
      }
      kp.angle = (float)best_k * 0.3515625f;
      new_angle = true;
      M[0] = pat->rot_cosf[best_k];
      M[1] = -pat->rot_sinf[best_k];
      M[2] = pat->rot_sinf[best_k];
      M[3] = pat->rot_cosf[best_k];
    }
  }
  if (valid) valid = sample_all(mode != kCameraAware, /*reuse=*/new_angle);
  have_kept = false;
  if (valid) {
    unsigned long long words[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      // this lane's pair of word j, held in registers for the whole wave (two pairs per dword): no table read in a
      // keypoint's chain, only the two value gathers
#if OKVFE_DESC_PAIRS_REG
      const uint32_t pr = (j & 1) ? my_pairs[j >> 1] >> 16 : my_pairs[j >> 1] & 0xFFFFu;  // slots past n_short: 0 | 0
#else
      const uint32_t pr = short_pairs[j * 64 + lane];
#endif
      const bool bit = vals[pr & 255u] > vals[pr >> 8];
      words[j] = __ballot(bit);
    }
    if (lane < 6) {
      unsigned long long wsel = words[0];
#pragma unroll
      for (int j = 1; j < 6; ++j)
        if (lane == j) wsel = words[j];
      reinterpret_cast<unsigned long long*>(desc_tmp + slot * OKVFE_DESC_BYTES)[lane] = wsel;
    }
  }
  if (lane == 0) {
    if (new_angle) kps_tmp[slot].angle = kp.angle;  // the rest of the record: describe_setup_kernel
    valid_tmp[slot] = valid ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();  // vals[] / the patch are rewritten for the next keypoint
  }
}

// ---- compaction + back-projection -----------------------------------------------------------
__global__ __launch_bounds__(256) void compact_kernel(
    const DeviceCamera* __restrict__ cams, const ImageParams* __restrict__ prm,
    const okvfe_keypoint* __restrict__ kps_tmp, const uint8_t* __restrict__ desc_tmp,
    const uint8_t* __restrict__ valid_tmp, const int32_t* __restrict__ kp_count_in, int kp_cap,
    okvfe_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, double* __restrict__ bp,
    uint8_t* __restrict__ bpv, int32_t* __restrict__ kp_count) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = kp_count_in[img];
  const size_t off = (size_t)img * kp_cap;
  const int cam = prm[img].cam;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += 256) {
    const int k = k0 + tid;
    const bool v = k < n && valid_tmp[off + k] != 0;
    const unsigned long long b = __ballot(v);
    if (lane == 0) wave_cnt[wv] = __popcll(b);
    __syncthreads();
    int pos = base_s;
    for (int i = 0; i < wv; ++i) pos += wave_cnt[i];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    pos += __popcll(b & lt);
    if (v) {
      const okvfe_keypoint kp = kps_tmp[off + k];
      kps[off + pos] = kp;
      const uint4* s = reinterpret_cast<const uint4*>(desc_tmp + (off + k) * OKVFE_DESC_BYTES);
      uint4* d = reinterpret_cast<uint4*>(desc + (off + pos) * OKVFE_DESC_BYTES);
      d[0] = s[0];
      d[1] = s[1];
      d[2] = s[2];
      double dir[3] = {0.0, 0.0, 0.0};
      bool ok = false;
      if (cam >= 0 && cams[cam].fu > 0.0) ok = cam::backproject(cams[cam], (double)kp.x, (double)kp.y, dir);
      bp[(off + pos) * 3 + 0] = dir[0];
      bp[(off + pos) * 3 + 1] = dir[1];
      bp[(off + pos) * 3 + 2] = dir[2];
      bpv[off + pos] = ok ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) kp_count[img] = base_s;
}

}  // namespace

void launch_describe(const uint8_t* img, int w, int h, int n_images, const Pattern* pat,
                     const ImageParams* prm, const float* const* rays, const float* const* jac,
                     const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in,
                     okvfe_keypoint* kps_tmp, uint8_t* desc_tmp, uint8_t* valid_tmp,
                     const PatternScales* scales, bool wide_patches, hipStream_t stream, bool setup_done,
                     bool all_camera_aware, int box_class, int aware_extra_box, bool rot_fast) {
  if (n_images <= 0) return;
  static const char* force = lab_env("OKVFE_DESC_WAVES");  // A/B knob: 5 / 6
  if (force) wide_patches = force[0] == '5';
  if (!setup_done)  // (done by select_lazy_kernel when detection and description were one call)
  hipLaunchKernelGGL(describe_setup_kernel, dim3((kp_cap + 255) / 256, n_images), dim3(256), 0,
                     stream, w, h, pat, prm, rays, jac, kps_in, kp_cap, kp_count_in, kps_tmp, desc_tmp,
                     valid_tmp, scales, img, aware_extra_box > 0 && aware_extras_in_setup() ? (aware_extra_box & 0xFF) : 0);
  // blocks per image: enough waves to fill the machine with one image's ~300 keypoints spread
  // over them (a wave then describes ~9 keypoints of its image in a row)
  int tiles = (kp_cap + kDescWaves - 1) / kDescWaves;
  if (tiles > kDescBlocksPerImage) tiles = kDescBlocksPerImage;
  const uint32_t inv_tiles = (uint32_t)((0x100000000ull + (uint64_t)tiles - 1) / (uint64_t)tiles);
#define OKVFE_DESC_LAUNCH(WAVES, AWARE, WIDE)                                                                    \
  hipLaunchKernelGGL((describe_kernel<WAVES, AWARE, WIDE>), dim3(tiles * n_images), dim3(64 * kDescWaves), 0, stream, \
                     img, w, h, pat, prm, rays, jac, kps_in, kp_cap, kp_count_in, kps_tmp, desc_tmp, valid_tmp,        \
                     n_images, tiles, inv_tiles, scales)
  static const bool no_aware = lab_env("OKVFE_DESC_GENERIC") != nullptr;  // A/B knob: the all-modes kernel
  if (no_aware || scales != nullptr || w % 4 != 0 || (reinterpret_cast<uintptr_t>(img) & 3) != 0)
    all_camera_aware = false;  // (scale-invariant extraction, unaligned images: generic form)
  // round 6: the production mode on cameras whose patches fit the two LDS classes -> k_describe_aware.hip (extra
  // samples in batches, patch geometry from the set-up thread, compile-time row pitches)
  // (aware_extra_box >= 0: capi_detect.cpp decided for it -- cameras, pattern, alignment -- and the set-up threads
  // have left the extra samples in the slots)
  if (all_camera_aware && aware_extra_box >= 0 && box_class <= 1) {
    launch_describe_aware(img, w, h, n_images, pat, kps_in, kp_cap, kp_count_in, desc_tmp, valid_tmp, box_class == 1,
                          stream, aware_extra_box > 0 && !aware_extras_in_setup() ? (aware_extra_box >> 8) : 0,
                          aware_extra_box & 0xFF);
    return;
  }
  // round 6: upright / gradient-orientation calls (the BRISK scale-space path, callers without a camera) on the same box
  // sums: describe_rot_kernel (k_describe_aware.hip)
  static const bool no_rot = lab_env("OKVFE_DESC_NO_ROT") != nullptr;  // A/B knob
  if (rot_fast && !no_rot && !no_aware && box_class == 0 && scales == nullptr && w % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(img) & 3) == 0) {
    launch_describe_rot(img, w, h, n_images, pat, prm, kps_in, kp_cap, kp_count_in, kps_tmp, desc_tmp, valid_tmp, stream);
    return;
  }
  // box_class (capi_detect.cpp: pattern_box_class): 0 = every box fits the 11 x 11 / 5 x 5 slots, 1 = the 21 x 21 /
  // 9 x 9 slots of the WIDE instantiations, 2 = wider still: the all-modes form's plain box loops
  if (box_class == 1 && scales == nullptr) {
    if (all_camera_aware) OKVFE_DESC_LAUNCH(4, true, true); else OKVFE_DESC_LAUNCH(4, false, true);
  } else if (!all_camera_aware || box_class != 0) {
    // the all-modes form: 128 registers, the long pairs in LDS: four workgroups per CU (measured on the BRISK
    // scale-space path, 512 images x 2780 keypoints in gradient mode: 5.5 ms with 80 registers and the long-pair /
    // rotation tables in global memory, 4.3 ms in this form; the 5- and 6-wave forms of it 4.7 ms)
    OKVFE_DESC_LAUNCH(4, false, false);
  } else if (wide_patches) {
    OKVFE_DESC_LAUNCH(5, true, false);
  } else {
    OKVFE_DESC_LAUNCH(6, true, false);
  }
#undef OKVFE_DESC_LAUNCH
}

// Does the camera-aware patch of a keypoint with the row norms (nx, ny) of M fit the wave's LDS
// buffer in one piece?  (the same arithmetic as sample_all / stage_patch, on the host: used by
// okvfe_set_camera_maps to find out which describe_kernel instantiation suits a camera)
bool describe_patch_fits(float nx, float ny, int border) {
  const float ex = fmaxf(nx * 1.001f, 1.0f) * (float)(border - 1) + 1.5f;
  const float ey = fmaxf(ny * 1.001f, 1.0f) * (float)(border - 1) + 1.5f;
  const int pw = 2 * (int)ceilf(ex) + 2 + 3, ph = 2 * (int)ceilf(ey) + 2;  // + 3: the row start is aligned down to a dword
  const int nq = (pw + 15) >> 4;
  if (nq * 16 > kZeroRowBytes - 8) return false;
  const int R = 64 / nq, trips = (ph + R - 1) / R;
  return trips * R * nq * 16 <= kPatchDataBytes;
}

void launch_compact(int n_images, const DeviceCamera* cams, const ImageParams* prm,
                    const okvfe_keypoint* kps_tmp, const uint8_t* desc_tmp,
                    const uint8_t* valid_tmp, const int32_t* kp_count_in, int kp_cap,
                    okvfe_keypoint* kps, uint8_t* desc, double* bp, uint8_t* bpv,
                    int32_t* kp_count, hipStream_t stream) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(compact_kernel, dim3(n_images), dim3(256), 0, stream, cams, prm, kps_tmp,
                     desc_tmp, valid_tmp, kp_count_in, kp_cap, kps, desc, bp, bpv, kp_count);
}

}  // namespace okvfe
