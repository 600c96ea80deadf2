// k_describe_aware.hip -- K6, the production form: BRISK2 descriptors under camera-aware extraction.
//
// Replaces brisk::BriskDescriptorExtractor::compute (behind cv::DescriptorExtractor::compute,
// okvis_cv/include/okvis/implementation/Frame.hpp:167; extractor built at okvis_frontend/src/Frontend.cpp:2410-2412,
// configured by setCameraProperties / setExtractionDirection at :239-251) for calls in which EVERY image is extracted
// camera-aware with the fixed-scale pattern -- what OKVIS2 runs.  Everything else (gradient orientation, upright,
// scale ladder, unaligned images, boxes beyond the fixed-trip slots, cameras whose warped patterns mostly exceed the
// LDS patch classes below) stays with describe_kernel (k_describe.hip).
//
// Round 6.  The round-5 kernel spent 518 wave64 VALU + 83 LDS instructions per keypoint; a third of both was a SECOND
// pass of the whole wave for the two samples (of 66) that do not fit 64 lanes, another fifth was wave-uniform patch
// geometry recomputed by 64 lanes, and two thirds of the first pass's row loop was address arithmetic.  This kernel:
//
//   * EXTRA SAMPLES FROM THE SET-UP THREAD.  describe_setup_one (describe_setup_dev.h: one THREAD per keypoint, in
//     the tail of the selection kernel, where the keypoint's pixels are still in the L2) evaluates the samples beyond
//     64 with the same box sum and leaves them in the keypoint's descriptor slot; the wave that owns the keypoint
//     loads them with the patch.  (First form of this round: lane j * extra + e of the wave computed them for a round
//     of its keypoints before walking them -- 8.8 k of a wave's 41 k cycles, latency of two dependent global round
//     trips: tools/lab/aware_prof.py.)
//   * PATCH GEOMETRY FROM THE SET-UP THREAD.  describe_setup_one (one THREAD per keypoint, in the tail of the
//     selection kernel) leaves {first byte offset, x0 / y0 / rows / class} next to M in the keypoint's descriptor
//     slot; they arrive here through the scalar prefetch of the next keypoint.
//   * TWO COMPILE-TIME ROW PITCHES (64 and 80 bytes: |M| up to ~1.05 / ~1.25) so that every row read of the box sum
//     is base + immediate; rows past a lane's box are masked off by exec instead of being redirected to a zero row.
//     Patches of neither class (wide-angle rims) are sampled straight from the image by plain loops -- correct, slow,
//     and rare on the cameras this kernel is launched for (capi_context.cpp: cam_slow).
//
// Bit-exact to describe_kernel and to the oracle: the float set-up and the integer sums are the same operation
// sequences (published BRISK smoothedIntensity with sub-pixel rim weights).
#include "describe_setup_dev.h"
#include "okvfe_internal.h"

namespace okvfe {
namespace {

#ifdef OKVFE_LAB
__device__ unsigned long long g_aware_prof[16];
#define AW_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define AW_T(x)
#endif
constexpr int kAwWaves = 4;
constexpr int kAwBufRows1 = 72, kAwPitch0 = 64, kAwPitch1 = 80;
constexpr int kAwBufBytes = kAwBufRows1 * kAwPitch1 + 32;  // 5792: class 0 needs 64 x 64 = 4096; 32 B slack: row reads past a box

template <bool WIDE> struct AwCfg {
  static constexpr int kMaxB = WIDE ? 20 : 10;   // first-pass samples: box side - 1 (21 x 21 / 11 x 11 pixels)
  static constexpr int kMaxB2 = WIDE ? 9 : 4;    // extra samples (10 x 10 / 5 x 5)
  static constexpr int kCounts = WIDE ? 20 : 10; // mask table: interior byte counts 0 .. kCounts - 1
  // A row's interior (<= 9 / 19 bytes) is read as ALIGNED 8-byte windows from the multiple of 8 at or below its first
  // byte: ds_read_b64 serves 32 lanes per cycle on 64 banks where ds_read2_b32 + ds_read_b32 took three dword slots
  // on 32, and the LDS gather is what bounds the kernel.  Window q >= 1 is skipped (exec) by the lanes whose interior
  // ends before it.
  static constexpr int kAlign = 8;
  static constexpr int kQ = WIDE ? 4 : 2;        // windows per row: 7 + 9 <= 16, 7 + 19 <= 32 bytes
  static constexpr int kRowDw = 2 * kQ;          // dwords per table entry
  static constexpr int kMaskDw = 2 * kQ;
};

__device__ __forceinline__ int mul24i(int a, int b) {
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mad24i(int a, int b, int c) {  // low 32 bits of a[23:0] * b[23:0] + c
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// floor(num / den) for 0 <= num < 2^31, 1 <= den < 2^24, quotient < 2^22: float estimate (rcp = v_rcp_f32 of
// (float)den, off by at most one either way -- relative error of the product < 2^-22) + one exact correction
__device__ __forceinline__ int div_nonneg_rcp(int num, int den, float rcp) {
  int q = (int)((float)num * rcp);
  const int r = num - mul24i(q, den);
  q += r >= den ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

// table[((first byte 0..kAlign-1) * kCounts + count) * kRowDw + j]: 0xFF in every byte of dword j that belongs to the
// run of `count` interior bytes starting at byte `first byte` of the window
template <bool WIDE>
__device__ __forceinline__ void fill_box_masks(uint32_t* table, int tid, int nthreads) {
  using T = AwCfg<WIDE>;
  for (int e = tid; e < T::kAlign * T::kCounts; e += nthreads) {
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

// ---- pixel readers of the box sum ---------------------------------------------------------------------------------
// byte(off) / qword(off) at byte offset `off` from the patch's first byte; compile-time row pitch
template <int PITCH>
struct LdsReader {  // the keypoint's patch in LDS, row pitch PITCH
  static constexpr int kPitch = PITCH;
  const uint8_t* base;  // LDS
  static constexpr int pitch = PITCH;
  __device__ __forceinline__ int byte(int off) const { return base[off]; }
  __device__ __forceinline__ uint2 qword(int off) const { return *reinterpret_cast<const uint2*>(base + off); }  // 8-aligned
};
// Box of half-side sigma_half centred at (xf, yf), 1024 * mean intensity: the published BRISK smoothedIntensity with
// sub-pixel rim weights, fixed trip counts.  `x0`, `y0`: image coordinates of the reader's origin; MAXB: largest box
// side minus one served.  Same float / integer sequence as smoothed_intensity (k_describe.hip).
template <int MAXB, bool WIDE, typename RD>
__device__ __forceinline__ int box_mean(const RD& rd, const uint32_t* __restrict__ masks, int x0, int y0, float xf,
                                        float yf, float sigma_half, int scaling, int scaling2, float rcp2) {
  using T = AwCfg<WIDE>;
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
  const int bw = x_right - x_left, bh = y_bottom - y_top;  // 1 .. MAXB
  constexpr int kAl = T::kAlign, kQ = T::kQ, kDw = 2 * kQ;
  static_assert(MAXB <= T::kCounts && MAXB + kAl - 2 <= 8 * kQ, "mask table of the kernel");
  const int cl = x_left - x0;
  const int xi0 = cl + 1, lo = xi0 & (kAl - 1), ni = bw - 1;
  uint32_t m[kDw];
  {
    const uint4* mp = reinterpret_cast<const uint4*>(masks + ((mul24i(lo, T::kCounts) + ni) * T::kRowDw));
#pragma unroll
    for (int q = 0; q < kQ; q += 2) {
      const uint4 mrow = mp[q >> 1];
      m[2 * q] = mrow.x;
      m[2 * q + 1] = mrow.y;
      m[2 * q + 2] = mrow.z;
      m[2 * q + 3] = mrow.w;
    }
  }
  const int pitch = RD::kPitch > 0 ? RD::kPitch : rd.pitch;
  const int row0 = mul24i(y_top - y0, pitch);
  const int aL = row0 + cl, aR = aL + bw, aQ = row0 + (xi0 & ~(kAl - 1));
  const int nq = (lo + ni + 7) >> 3;  // windows the interior reaches (0 for an empty interior: window 0 is read anyway)
  // v_msad_u8 adds 255 - pixel for the bytes the mask selects and skips the others: a row's interior sum comes back
  // as 255 * ni - accumulator
  struct Row {
    uint32_t d[kDw];
    int pl, pr;
  };
  // all reads of a row are issued before its first sum; a window the lane skips keeps zeros (its masks are zero anyway)
  auto issue = [&](int off, Row& r) {
    r.pl = rd.byte(aL + off);
    r.pr = rd.byte(aR + off);
    const uint2 a = rd.qword(aQ + off);
    r.d[0] = a.x;
    r.d[1] = a.y;
#pragma unroll
    for (int q = 1; q < kQ; ++q) {
      uint2 b = make_uint2(0u, 0u);
      if (nq > q) b = rd.qword(aQ + off + 8 * q);
      r.d[2 * q] = b.x;
      r.d[2 * q + 1] = b.y;
    }
  };
  auto row_sum = [&](const Row& r, uint32_t acc) -> uint32_t {
#pragma unroll
    for (int j = 0; j < kDw; ++j) acc = __builtin_amdgcn_msad_u8(r.d[j], m[j], acc);
    return acc;
  };
  const int full = mul24i(ni, 255);
  // top and bottom row
  const int ob = mul24i(bh, pitch);
  Row rt, rb;
  issue(0, rt);
  issue(ob, rb);
  const int pl_t = rt.pl, pr_t = rt.pr, pl_b = rb.pl, pr_b = rb.pr;
  const uint32_t up = row_sum(rt, 0u), bot = row_sum(rb, 0u);
  int ret = mad24i(A, pl_t, scaling2 / 2);
  ret = mad24i(B, pr_t, ret);
  ret = mad24i(C, pr_b, ret);
  ret = mad24i(D, pl_b, ret);
  // interior rows 1 .. bh - 1: fixed trip count, rows past the lane's box masked off by exec -- the LDS pipe is this
  // kernel's bound (bank conflicts of the gather), so a lane that has no row left must not read.  Rows go in groups of
  // three nested tests (dy < bh is monotone): the reads of a group are all issued on the way in and summed on the way
  // out, one LDS round trip per group instead of one per row.  (Measured against it: the reads of three rows issued
  // for every lane and the sums selected afterwards, 1.45 vs 1.36 ms.)
  uint32_t mid = 0u;
  int left = 0, right = 0;
  auto row_off = [&](int dy) -> int { return RD::kPitch > 0 ? dy * RD::kPitch : mul24i(dy, pitch); };
  auto consume = [&](const Row& r) {
    mid = row_sum(r, mid);
    left += r.pl;
    right += r.pr;
  };
#pragma unroll
  for (int d0 = 1; d0 < MAXB; d0 += 3) {
    if (d0 < bh) {
      Row r0;
      issue(row_off(d0), r0);
      if (d0 + 1 < MAXB && d0 + 1 < bh) {
        Row r1;
        issue(row_off(d0 + 1), r1);
        if (d0 + 2 < MAXB && d0 + 2 < bh) {
          Row r2;
          issue(row_off(d0 + 2), r2);
          consume(r2);
        }
        consume(r1);
      }
      consume(r0);
    }
  }
  const int upper = full - (int)up, bottom = full - (int)bot;
  const int middle = mul24i(full, bh - 1) - (int)mid;
  ret = mad24i(upper, r_y_1_i, ret);
  ret = mad24i(middle, scaling, ret);
  ret = mad24i(left, r_x_1_i, ret);
  ret = mad24i(right, r_x1_i, ret);
  ret = mad24i(bottom, r_y1_i, ret);
  return div_nonneg_rcp(ret, scaling2, rcp2);
}

// any box size, straight from the image (patches of neither LDS class): plain loops
__device__ __forceinline__ int box_mean_plain(const uint8_t* __restrict__ im, int w, float xf, float yf, float sigma_half,
                                              int scaling, int scaling2, float rcp2) {
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
  auto px = [&](int y, int x) -> int { return im[(size_t)y * w + x]; };
  int ret = A * px(y_top, x_left);
  ret += B * px(y_top, x_right);
  ret += C * px(y_bottom, x_right);
  ret += D * px(y_bottom, x_left);
  int upper = 0, middle = 0, left = 0, right = 0, bottom = 0;
  for (int x = x_left + 1; x < x_right; ++x) {
    upper += px(y_top, x);
    bottom += px(y_bottom, x);
  }
  for (int y = y_top + 1; y < y_bottom; ++y) {
    left += px(y, x_left);
    right += px(y, x_right);
    for (int x = x_left + 1; x < x_right; ++x) middle += px(y, x);
  }
  ret += upper * r_y_1_i + middle * scaling + left * r_x_1_i + right * r_x1_i + bottom * r_y1_i;
  return div_nonneg_rcp(ret + scaling2 / 2, scaling2, rcp2);
}

// sample position of a pattern point under M; ok = box inside the image (NaN-safe)
__device__ __forceinline__ bool sample_pos(float M0, float M1, float M2, float M3, float kx, float ky, float px,
                                           float py, float sg, int w, int h, float* xf, float* yf) {
  float a = M0 * px;
  float b = M1 * py;
  a = a + b;
  *xf = kx + a;
  float c = M2 * px;
  float d = M3 * py;
  c = c + d;
  *yf = ky + c;
  const float x_1 = *xf - sg, x1 = *xf + sg, y_1 = *yf - sg, y1 = *yf + sg;
  return (x_1 >= 0.0f && y_1 >= 0.0f && x1 < (float)(w - 1) && y1 < (float)(h - 1));
}

// The samples beyond the 64 lanes, one THREAD per (keypoint, extra sample): M and the valid byte come from the set-up
// (describe_setup_one), the value goes to bytes 24.. of the keypoint's slot, a box outside the image drops the keypoint.
// (Alternative home: the tail of the selection kernel, where the pixels are still in the L2 -- okvfe_internal.h
// DescribeSetup::extra_box; it costs that kernel its register allocation: scratch 68 -> 160 bytes, + 0.08-0.1 ms.)
__global__ __launch_bounds__(256) void describe_extras_kernel(const uint8_t* __restrict__ images, int w, int h,
                                                              const Pattern* __restrict__ pat, int extra, int extra_box,
                                                              const okvfe_keypoint* __restrict__ kps_in, int kp_cap,
                                                              const int32_t* __restrict__ kp_count_in,
                                                              uint8_t* __restrict__ desc_tmp, uint8_t* __restrict__ valid_tmp) {
  const int img = blockIdx.y;
  const int item = blockIdx.x * 256 + threadIdx.x;
  const int k = item / extra, e = item - k * extra;
  if (k >= kp_count_in[img]) return;
  const size_t slot = (size_t)img * kp_cap + k;
  if ((valid_tmp[slot] & 1) == 0) return;
  const float4 Mv = *reinterpret_cast<const float4*>(desc_tmp + slot * OKVFE_DESC_BYTES);
  const float M[4] = {Mv.x, Mv.y, Mv.z, Mv.w};
  const float2 xy = *reinterpret_cast<const float2*>(&kps_in[slot].x);
  if (!extra_sample_one(pat, images + (size_t)img * w * h, w, h, extra_box, e, M, xy.x, xy.y,
                        desc_tmp + slot * OKVFE_DESC_BYTES))
    valid_tmp[slot] = 0;  // (only ever cleared, by any of the keypoint's threads)
}

// One wave per keypoint at a time, lane l = pattern point extra + l; 4 waves per workgroup, `tiles` workgroups per
// image (all on one XCD), a wave walks the image's keypoints wave, wave + 4 * tiles, ...
template <bool WIDE>
__global__ __launch_bounds__(64 * kAwWaves) __attribute__((amdgpu_waves_per_eu(WIDE ? 5 : 6, 8))) void describe_aware_kernel(
    const uint8_t* __restrict__ images, int w, int h, const Pattern* __restrict__ pat,
    const okvfe_keypoint* __restrict__ kps_in, int kp_cap, const int32_t* __restrict__ kp_count_in,
    uint8_t* __restrict__ desc_tmp, uint8_t* __restrict__ valid_tmp, int n_images, int tiles, uint32_t inv_tiles) {
  using T = AwCfg<WIDE>;
  AW_T(t_entry);
  __shared__ __attribute__((aligned(16))) uint8_t patches[kAwWaves][kAwBufBytes];
  __shared__ int values[kAwWaves][kPatternPoints];
  __shared__ __attribute__((aligned(16))) uint32_t box_masks[T::kAlign * T::kCounts * T::kRowDw];
  // everything a wave needs before its first keypoint is requested up front -- the image's keypoint count, the lane's
  // constants (one 32-byte record, Pattern::aware_lane), the first keypoint's scalars -- so that the prologue is ONE
  // global round trip (it was five dependent ones: 7.6 k of a wave's 41 k cycles, tools/lab/aware_prof.py)
  int img, tile;
  {
    const uint32_t L = blockIdx.x, n8 = (uint32_t)n_images & ~7u, full = n8 * (uint32_t)tiles;
    const uint32_t slot = L < full ? L >> 3 : L - full;
    uint32_t g = (uint32_t)(((uint64_t)slot * inv_tiles) >> 32);  // slot / tiles, off by <= 1
    if (g * (uint32_t)tiles > slot) --g;
    if ((g + 1) * (uint32_t)tiles <= slot) ++g;
    // all keypoint blocks of an image run on the same XCD (block L -> XCD L % 8): its pixels are fetched into ONE L2
    img = L < full ? (int)(g * 8u + (L & 7u)) : (int)(n8 + g);
    tile = (int)(slot - g * (uint32_t)tiles);
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  const int k_first = tile * kAwWaves + wv, k_step = tiles * kAwWaves;  // (k_first < kp_cap: launch_describe_aware)
  const size_t slot0 = (size_t)img * kp_cap;
  const int n_raw = kp_count_in[img];
  const int n_points = pat->n_points;
  const int4 lc0 = *reinterpret_cast<const int4*>(&pat->aware_lane[lane][0]);
  const int4 lc1 = *reinterpret_cast<const int4*>(&pat->aware_lane[lane][4]);
  typedef const float __attribute__((address_space(4))) * cfloat_p;
  typedef const int __attribute__((address_space(4))) * cint_p;
  typedef const uint8_t __attribute__((address_space(4))) * cbyte_p;
  float nxt_x = 0.f, nxt_y = 0.f, nxt_M0 = 0.f, nxt_M1 = 0.f, nxt_M2 = 0.f, nxt_M3 = 0.f;
  int nxt_g0 = 0, nxt_g1 = 0, nxt_valid = 0;
  // scalar loads (the constant address space forces s_load): the NEXT keypoint's values wait in SGPRs.  Safe on the
  // scalar cache: a slot's position / M / geometry / valid byte are written before this kernel starts and read here
  // before this wave -- the only writer of the slot -- overwrites them.
  auto fetch = [&](int kk) {
    const size_t sl = slot0 + kk;
    const cfloat_p pxy = (cfloat_p)(uintptr_t)(&kps_in[sl].x);
    const cfloat_p pm = (cfloat_p)(uintptr_t)(desc_tmp + sl * OKVFE_DESC_BYTES);
    nxt_x = pxy[0];
    nxt_y = pxy[1];
    nxt_M0 = pm[0];
    nxt_M1 = pm[1];
    nxt_M2 = pm[2];
    nxt_M3 = pm[3];
    nxt_g0 = ((cint_p)pm)[4];
    nxt_g1 = ((cint_p)pm)[5];
    nxt_valid = (int)((cbyte_p)(uintptr_t)valid_tmp)[sl];
  };
  fetch(k_first < kp_cap ? k_first : kp_cap - 1);  // (a slot past the image's count holds stale bytes: read, never used)
  fill_box_masks<WIDE>(box_masks, threadIdx.x, 64 * kAwWaves);
  __syncthreads();
  const int n = __builtin_amdgcn_readfirstlane(n_raw);
  if (k_first >= n) return;  // whole wave exits; no block-wide barriers below
  const uint8_t* im = images + (size_t)img * w * h;
  const int extra = __builtin_amdgcn_readfirstlane(n_points > 64 ? n_points - 64 : 0);
  const bool active = extra + lane < n_points;
  float px = __int_as_float(lc0.x), py = __int_as_float(lc0.y), sg = __int_as_float(lc0.z);
  int bsc = lc0.w, bsc2 = lc1.x;
  float rcp2 = __builtin_amdgcn_rcpf((float)bsc2);
  const uint32_t my_pairs[3] = {(uint32_t)lc1.y, (uint32_t)lc1.z, (uint32_t)lc1.w};
  int* vals = values[wv];
  uint8_t* patch = patches[wv];
  const __amdgpu_buffer_rsrc_t img_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(im), 0, w * h, 0x00027000);
  // per-lane source offsets of the two patch classes: lane = row * chunks + 16-byte chunk
  const int src_lane0 = (lane >> 2) * w + (lane & 3) * 16;                  // pitch 64: 16 rows per trip
  const int rr1 = (int)(((uint32_t)lane * 13108u) >> 16);                   // lane / 5
  const int src_lane1 = rr1 * w + (lane - rr1 * 5) * 16;                    // pitch 80: 12 rows per trip (lanes 0..59)
  AW_T(t_pro);
#ifdef OKVFE_LAB
  unsigned long long a_dma = 0, a_box = 0, a_bits = 0, a_kp = 0;
#endif
  for (int k = k_first; k < n; k += k_step) {  // (k: wave-uniform)
    {
      AW_T(t_a);
      // opaque to the optimiser: expressions of the lane constants are NOT hoisted out of the loop
      asm volatile("" : "+v"(px), "+v"(py), "+v"(sg), "+v"(bsc), "+v"(bsc2), "+v"(rcp2), "+v"(lane));
      const size_t slot = slot0 + k;
      auto unif = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
      const float kx = unif(nxt_x), ky = unif(nxt_y);
      const float M0 = unif(nxt_M0), M1 = unif(nxt_M1), M2 = unif(nxt_M2), M3 = unif(nxt_M3);
      const int g0 = __builtin_amdgcn_readfirstlane(nxt_g0), g1 = __builtin_amdgcn_readfirstlane(nxt_g1);
      bool valid = (__builtin_amdgcn_readfirstlane(nxt_valid) & 1) != 0;
      if (k + k_step < n) fetch(k + k_step);  // scalar branch
      if (valid) {
        const int x0 = (g1 & 0x3FF) << 2, y0 = (g1 >> 10) & 0xFFF, ph = (g1 >> 22) & 0x7F, cls = (g1 >> 29) & 3;
        // the patch goes straight from the image into LDS (buffer_load ... lds, 16 bytes per lane, whole rows per
        // instruction) while the per-sample set-up runs
        __builtin_amdgcn_wave_barrier();
        if (cls == 0) {
          const int trips = (ph + 15) >> 4;
          for (int it = 0; it < trips; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rsrc, (__attribute__((address_space(3))) void*)(patch + it * 1024),
                                                     16, src_lane0, g0 + it * 16 * w, 0, 0);
        } else if (cls == 1) {
          const int trips = (ph + 11) / 12;
          if (lane < 60)
            for (int it = 0; it < trips; ++it)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rsrc, (__attribute__((address_space(3))) void*)(patch + it * 960),
                                                       16, src_lane1, g0 + it * 12 * w, 0, 0);
        }
        // the samples beyond 64, evaluated by the set-up thread (which also dropped the keypoint if one of their boxes
        // left the image): lanes 0 .. extra - 1 pick them up with the patch
        int xv = 0;
        if (lane < extra) xv = *reinterpret_cast<const int*>(desc_tmp + slot * OKVFE_DESC_BYTES + 24 + 4 * lane);
        float xf, yf;
        const bool ok = sample_pos(M0, M1, M2, M3, kx, ky, px, py, sg, w, h, &xf, &yf);
        valid = __all(ok || !active);
        // (the loads are waited for even when the keypoint is dropped: the next keypoint's may not overtake them)
        int v = 0;
#ifdef OKVFE_LAB
        unsigned long long t_b = 0, t_c = 0;
#endif
        if (cls == 0) {
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the staged rows have landed in LDS
          __builtin_amdgcn_wave_barrier();
#ifdef OKVFE_LAB
          t_b = __builtin_amdgcn_s_memtime();
#endif
          if (valid) {
            const LdsReader<kAwPitch0> rd{patch};
            v = box_mean<T::kMaxB, WIDE>(rd, box_masks, x0, y0, xf, yf, sg, bsc, bsc2, rcp2);
          }
        } else if (cls == 1) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __builtin_amdgcn_wave_barrier();
          if (valid) {
            const LdsReader<kAwPitch1> rd{patch};
            v = box_mean<T::kMaxB, WIDE>(rd, box_masks, x0, y0, xf, yf, sg, bsc, bsc2, rcp2);
          }
        } else if (valid) {
          v = box_mean_plain(im, w, xf, yf, sg, bsc, bsc2, rcp2);
        }
#ifdef OKVFE_LAB
        asm volatile("" :: "v"(v));
        t_c = __builtin_amdgcn_s_memtime();
        if (t_b) { a_dma += t_b - t_a; a_box += t_c - t_b; }
#endif
        if (valid) {
          __builtin_amdgcn_wave_barrier();
          vals[extra + lane] = v;  // extra + 63 < kPatternPoints
          if (lane < extra) vals[lane] = xv;
          __builtin_amdgcn_wave_barrier();
          unsigned long long words[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const uint32_t pr = (j & 1) ? my_pairs[j >> 1] >> 16 : my_pairs[j >> 1] & 0xFFFFu;  // slots past n_short: 0 | 0
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
      }
      if (lane == 0) valid_tmp[slot] = valid ? 1 : 0;
      __builtin_amdgcn_wave_barrier();  // vals[] / the patch are rewritten for the next keypoint
#ifdef OKVFE_LAB
      {
        AW_T(t_d);
        a_bits += t_d - t_a;
        a_kp += 1;
      }
#endif
    }
  }
#ifdef OKVFE_LAB
  if (lane == 0 && wv == 0 && (blockIdx.x & 63) == 5) {
    AW_T(t_end);
    atomicAdd(&g_aware_prof[0], 1ull);
    atomicAdd(&g_aware_prof[1], t_pro - t_entry);
    atomicAdd(&g_aware_prof[2], 0ull);
    atomicAdd(&g_aware_prof[3], a_dma);
    atomicAdd(&g_aware_prof[4], a_box);
    atomicAdd(&g_aware_prof[5], a_bits);
    atomicAdd(&g_aware_prof[6], a_kp);
    atomicAdd(&g_aware_prof[7], t_end - t_entry);
  }
#endif
}


// ---- upright / gradient-orientation extraction on the same box sums (round 6) ----------------------------------------
// The modes the BRISK scale-space path (brisk::BriskFeatureDetector + extractor, okvis_cv/test/TestFrame.cpp:71-77) and
// cv::Feature2D-style callers without a camera use: rotationInvariant = false (upright) or true with the published
// gradient orientation -- the pattern sampled UPRIGHT, the 968 long pairs summed into a gradient, the best of 1024
// tabulated directions, the pattern sampled again under that rotation.  Until round 6 these ran on describe_kernel's
// all-modes form (128 registers, row loops with computed addresses: 4.3 ms per 512 images x 2780 keypoints); here they
// get the camera-aware kernel's box sum: a fixed 64-byte row pitch (the pattern's circle, 2 * border + 2 <= 64 pixels,
// always fits), aligned 8-byte row windows, exec-masked rows.  The patch is staged ONCE per keypoint and serves both
// samplings.  The samples beyond the 64 lanes are a second, exec-masked call of the small-box form on lanes 0 .. extra-1
// (their M is not known to any set-up thread in gradient mode).  Bit-exact to describe_kernel and the oracle.
constexpr int kRotBufBytes = 64 * kAwPitch0 + 32;
__global__ __launch_bounds__(64 * kAwWaves) __attribute__((amdgpu_waves_per_eu(5, 8))) void describe_rot_kernel(
    const uint8_t* __restrict__ images, int w, int h, const Pattern* __restrict__ pat, const ImageParams* __restrict__ prm,
    const okvfe_keypoint* __restrict__ kps_in, int kp_cap, const int32_t* __restrict__ kp_count_in,
    okvfe_keypoint* __restrict__ kps_tmp, uint8_t* __restrict__ desc_tmp, uint8_t* __restrict__ valid_tmp, int n_images,
    int tiles, uint32_t inv_tiles) {
  using T = AwCfg<false>;
  __shared__ __attribute__((aligned(16))) uint8_t patches[kAwWaves][kRotBufBytes];
  __shared__ int values[kAwWaves][kPatternPoints];
  __shared__ __attribute__((aligned(16))) uint32_t box_masks[T::kAlign * T::kCounts * T::kRowDw];
  __shared__ uint2 long_tab[kMaxLongPairs];  // {i | j << 8, wdx (16 bit) | wdy << 16}: the host checked the ranges
  __shared__ float ex_f[3][kAwareMaxExtra];
  __shared__ int ex_i[2][kAwareMaxExtra];
  // quarter waves of the rotation tables (okvfe_internal.h: quarter_sin; the host verified the rule for this pattern):
  // no gather from global memory inside a keypoint's chain
  __shared__ int32_t q_sin_i[257];
  __shared__ float q_sin_f[257];
  int img, tile;
  {
    const uint32_t L = blockIdx.x, n8 = (uint32_t)n_images & ~7u, full = n8 * (uint32_t)tiles;
    const uint32_t slot = L < full ? L >> 3 : L - full;
    uint32_t g = (uint32_t)(((uint64_t)slot * inv_tiles) >> 32);  // slot / tiles, off by <= 1
    if (g * (uint32_t)tiles > slot) --g;
    if ((g + 1) * (uint32_t)tiles <= slot) ++g;
    img = L < full ? (int)(g * 8u + (L & 7u)) : (int)(n8 + g);  // an image's blocks on one XCD
    tile = (int)(slot - g * (uint32_t)tiles);
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  const int k_first = tile * kAwWaves + wv, k_step = tiles * kAwWaves;
  const size_t slot0 = (size_t)img * kp_cap;
  const int n_raw = kp_count_in[img];
  const int n_points = pat->n_points;
  const int4 lc0 = *reinterpret_cast<const int4*>(&pat->aware_lane[lane][0]);
  const int4 lc1 = *reinterpret_cast<const int4*>(&pat->aware_lane[lane][4]);
  const int mode = __builtin_amdgcn_readfirstlane(prm[img].mode);
  const int border = __builtin_amdgcn_readfirstlane(pat->border);
  fill_box_masks<false>(box_masks, threadIdx.x, 64 * kAwWaves);
  for (int t = threadIdx.x; t < pat->n_long; t += 64 * kAwWaves)
    long_tab[t] = make_uint2((uint32_t)pat->long_i[t] | ((uint32_t)pat->long_j[t] << 8),
                             ((uint32_t)pat->long_wdx[t] & 0xFFFFu) | ((uint32_t)pat->long_wdy[t] << 16));
  for (int t = threadIdx.x; t < 257; t += 64 * kAwWaves) {
    q_sin_i[t] = pat->rot_sin[t];
    q_sin_f[t] = pat->rot_sinf[t];
  }
  // the three float entries the rule does not reproduce (sin(pi), cos(pi / 2), cos(3 pi / 2) are 1e-16 in double, not 0)
  const float f_s512 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pat->rot_sinf[512])));
  const float f_c256 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pat->rot_cosf[256])));
  const float f_c768 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pat->rot_cosf[768])));
  if (threadIdx.x < kAwareMaxExtra) {
    ex_f[0][threadIdx.x] = pat->px[threadIdx.x];
    ex_f[1][threadIdx.x] = pat->py[threadIdx.x];
    ex_f[2][threadIdx.x] = pat->sigma_half[threadIdx.x];
    ex_i[0][threadIdx.x] = pat->box_scaling[threadIdx.x];
    ex_i[1][threadIdx.x] = pat->box_scaling2[threadIdx.x];
  }
  typedef const float __attribute__((address_space(4))) * cfloat_p;
  typedef const uint8_t __attribute__((address_space(4))) * cbyte_p;
  float nxt_x = 0.f, nxt_y = 0.f;
  int nxt_valid = 0;
  auto fetch = [&](int kk) {  // scalar loads: the next keypoint waits in SGPRs (see describe_aware_kernel)
    const size_t sl = slot0 + kk;
    const cfloat_p pxy = (cfloat_p)(uintptr_t)(&kps_in[sl].x);
    nxt_x = pxy[0];
    nxt_y = pxy[1];
    nxt_valid = (int)((cbyte_p)(uintptr_t)valid_tmp)[sl];
  };
  fetch(k_first < kp_cap ? k_first : kp_cap - 1);
  __syncthreads();
  const int n = __builtin_amdgcn_readfirstlane(n_raw);
  if (k_first >= n) return;  // whole wave exits; no block-wide barriers below
  const uint8_t* im = images + (size_t)img * w * h;
  const int extra = __builtin_amdgcn_readfirstlane(n_points > 64 ? n_points - 64 : 0);
  const bool active = extra + lane < n_points;
  const bool active2 = lane < extra;
  float px = __int_as_float(lc0.x), py = __int_as_float(lc0.y), sg = __int_as_float(lc0.z);
  int bsc = lc0.w, bsc2 = lc1.x;
  const uint32_t my_pairs[3] = {(uint32_t)lc1.y, (uint32_t)lc1.z, (uint32_t)lc1.w};
  int* vals = values[wv];
  uint8_t* patch = patches[wv];
  const __amdgpu_buffer_rsrc_t img_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(im), 0, w * h, 0x00027000);
  const int src_lane0 = (lane >> 2) * w + (lane & 3) * 16;  // pitch 64: 16 rows per trip
  const int nl = pat->n_long;
  for (int k = k_first; k < n; k += k_step) {  // (k: wave-uniform)
    asm volatile("" : "+v"(px), "+v"(py), "+v"(sg), "+v"(bsc), "+v"(bsc2), "+v"(lane));  // (no hoisting: registers)
    const size_t slot = slot0 + k;
    auto unif = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float kx = unif(nxt_x), ky = unif(nxt_y);
    bool valid = (__builtin_amdgcn_readfirstlane(nxt_valid) & 1) != 0;  // the set-up kernel's rim test
    if (k + k_step < n) fetch(k + k_step);
    bool new_angle = false;
    float angle = 0.0f;
    if (valid) {
      // the pattern's circle, clipped to the image: covers every rotation
      const int cx = (int)kx, cy = (int)ky;
      int bx0 = cx - border, bx1 = cx + border + 1, by0 = cy - border, by1 = cy + border + 1;
      bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
      bx1 = bx1 > w - 1 ? w - 1 : bx1; by1 = by1 > h - 1 ? h - 1 : by1;
      const int x0 = bx0 & ~3, y0 = by0, ph = by1 - by0 + 1;  // (bx1 - x0 + 1 <= 2 * border + 5 <= 64: launch_describe)
      __builtin_amdgcn_wave_barrier();
      {
        const int trips = (ph + 15) >> 4;
        const int g0 = y0 * w + x0;
        for (int it = 0; it < trips; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rsrc, (__attribute__((address_space(3))) void*)(patch + it * 1024),
                                                   16, src_lane0, g0 + it * 16 * w, 0, 0);
      }
      const LdsReader<kAwPitch0> rd{patch};
      bool staged = false;
      // all samples under M -> vals[]; false: a box leaves the image
      auto sample_all = [&](float M0, float M1, float M2, float M3) -> bool {
        float xf, yf;
        const bool ok = sample_pos(M0, M1, M2, M3, kx, ky, px, py, sg, w, h, &xf, &yf);
        const int l2 = active2 ? lane : 0;
        float xf2 = 0.f, yf2 = 0.f;
        const float sg2 = ex_f[2][l2];
        bool ok2 = true;
        if (extra > 0) ok2 = sample_pos(M0, M1, M2, M3, kx, ky, ex_f[0][l2], ex_f[1][l2], sg2, w, h, &xf2, &yf2) || !active2;
        const bool all_ok = __all((ok || !active) && ok2);
        if (!staged) {  // (the loads are waited for even when the keypoint is dropped: the next one's may not overtake them)
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the staged rows have landed in LDS
          __builtin_amdgcn_wave_barrier();
          staged = true;
        }
        if (!all_ok) return false;
        const int v = box_mean<T::kMaxB, false>(rd, box_masks, x0, y0, xf, yf, sg, bsc, bsc2,
                                                __builtin_amdgcn_rcpf((float)bsc2));
        int v2 = 0;
        if (active2) {  // (exec-masked second call: the samples beyond the 64 lanes, 5 x 5 row slots)
          const int b1 = ex_i[0][l2], b2 = ex_i[1][l2];
          v2 = box_mean<T::kMaxB2, false>(rd, box_masks, x0, y0, xf2, yf2, sg2, b1, b2, __builtin_amdgcn_rcpf((float)b2));
        }
        __builtin_amdgcn_wave_barrier();
        vals[extra + lane] = v;  // extra + 63 < kPatternPoints
        if (active2) vals[lane] = v2;
        __builtin_amdgcn_wave_barrier();
        return true;
      };
      // one or two samplings through ONE copy of the box-sum code (a loop the compiler must not unroll: two inlined
      // copies of the row groups do not fit the instruction cache next to each other)
      float M0 = 1.0f, M1 = 0.0f, M2 = 0.0f, M3 = 1.0f;
      int passes = mode == kGradient ? 2 : 1;
      asm volatile("" : "+s"(passes));
      for (int pass = 0; pass < passes && valid; ++pass) {
        valid = sample_all(M0, M1, M2, M3);
        if (!valid || pass + 1 >= passes) break;
          int d0 = 0, d1 = 0;
          // four long pairs per lane and trip: the table reads, then the eight value gathers, are issued together (one at
          // a time the loop was 15 x two dependent LDS round trips); slots past n_long carry weight 0
          for (int l0 = 0; l0 < nl; l0 += 256) {  // wave-uniform
            uint2 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int l = l0 + 64 * u + lane;
              e[u] = long_tab[l < kMaxLongPairs ? l : 0];
              if (l >= nl) e[u] = make_uint2(0u, 0u);
            }
            int dt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) dt[u] = vals[e[u].x & 255u] - vals[(e[u].x >> 8) & 255u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              d0 += dt[u] * (int)(short)(e[u].y & 0xFFFFu) / 1024;
              d1 += dt[u] * ((int)e[u].y >> 16) / 1024;
            }
          }
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) {
            d0 += __shfl_xor(d0, d);
            d1 += __shfl_xor(d1, d);
          }
          int best_k = 0;
          if (d0 != 0 || d1 != 0) {
            // exact arg-max over the 1024 directions in a 64-step window around the float estimate (k_describe.hip)
            const float ang = atan2f((float)d1, (float)d0);
            const int k_est = (int)lrintf(ang * (1024.0f / 6.2831853071795864769f));
            int bk = (k_est - 32 + lane) & 1023;
            long long best = (long long)d0 * quarter_cos(q_sin_i, bk) + (long long)d1 * quarter_sin(q_sin_i, bk);
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
          best_k = __builtin_amdgcn_readfirstlane(best_k);
          angle = (float)best_k * 0.3515625f;
          new_angle = true;
          float c = quarter_cos(q_sin_f, best_k), sn = quarter_sin(q_sin_f, best_k);
          sn = best_k == 512 ? f_s512 : sn;
          c = best_k == 256 ? f_c256 : (best_k == 768 ? f_c768 : c);
          M0 = c; M1 = -sn; M2 = sn; M3 = c;
      }
      if (valid) {
        unsigned long long words[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t pr = (j & 1) ? my_pairs[j >> 1] >> 16 : my_pairs[j >> 1] & 0xFFFFu;  // slots past n_short: 0 | 0
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
    }
    if (lane == 0) {
      if (new_angle) kps_tmp[slot].angle = angle;  // the rest of the record: describe_setup_kernel
      valid_tmp[slot] = valid ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();  // vals[] / the patch are rewritten for the next keypoint
  }
}

}  // namespace

// Patch class of a camera-aware keypoint (describe_setup_dev.h computes the same on the device): 0 / 1 = LDS patch of
// pitch 64 / 80, 3 = neither.  Host copy for the cameras' statistics (capi_context.cpp).
int describe_aware_patch_class(float nx, float ny, float reach) {
  const float ex = fmaxf(nx * 1.001f, 1.0f) * reach + 0.75f;
  const float ey = fmaxf(ny * 1.001f, 1.0f) * reach + 0.75f;
  const int pw = 2 * (int)ceilf(ex) + 1 + 3, ph = 2 * (int)ceilf(ey) + 1;  // + 3: the row start is aligned down to a dword
  if (pw <= kAwPitch0 && ph <= 64) return 0;
  if (pw <= kAwPitch1 && ph <= kAwBufRows1) return 1;
  return 3;
}

#ifdef OKVFE_LAB
extern "C" int okvfe_lab_aware_prof(unsigned long long out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_aware_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_aware_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

bool aware_extras_in_setup() {
  // A/B knob: a kernel of their own instead (describe_extras_kernel: 0.18 ms per 6144 EuRoC images where the selection
  // kernel's tail takes 0.085 for the same work -- 688 k vs 705 k stereo-frames/s)
  static const bool own_kernel = lab_env("OKVFE_EXTRAS_KERNEL") != nullptr;
  return !own_kernel;
}

// upright / gradient modes of a whole call on describe_rot_kernel (launch_describe decides: no camera-aware image, boxes
// of class 0, pattern circle within 64 pixels, long-pair weights within 16 bits)
void launch_describe_rot(const uint8_t* img, int w, int h, int n_images, const Pattern* pat, const ImageParams* prm,
                         const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in, okvfe_keypoint* kps_tmp,
                         uint8_t* desc_tmp, uint8_t* valid_tmp, hipStream_t stream) {
  if (n_images <= 0) return;
  int tiles = (kp_cap + kAwWaves - 1) / kAwWaves;
  if (tiles > 16) tiles = 16;
  const uint32_t inv_tiles = (uint32_t)((0x100000000ull + (uint64_t)tiles - 1) / (uint64_t)tiles);
  hipLaunchKernelGGL(describe_rot_kernel, dim3(tiles * n_images), dim3(64 * kAwWaves), 0, stream, img, w, h, pat, prm, kps_in,
                     kp_cap, kp_count_in, kps_tmp, desc_tmp, valid_tmp, n_images, tiles, inv_tiles);
}

void launch_describe_aware(const uint8_t* img, int w, int h, int n_images, const Pattern* pat,
                           const okvfe_keypoint* kps_in, int kp_cap, const int32_t* kp_count_in, uint8_t* desc_tmp,
                           uint8_t* valid_tmp, bool wide_boxes, hipStream_t stream, int extras_now, int extra_box) {
  if (n_images <= 0) return;
  if (extras_now > 0)  // (not done by the set-up threads)
    hipLaunchKernelGGL(describe_extras_kernel, dim3((kp_cap * extras_now + 255) / 256, n_images), dim3(256), 0, stream, img,
                       w, h, pat, extras_now, extra_box, kps_in, kp_cap, kp_count_in, desc_tmp, valid_tmp);
  static const char* tiles_env = lab_env("OKVFE_DESC_TILES");  // A/B knob: workgroups per image
  int tiles = (kp_cap + kAwWaves - 1) / kAwWaves;
  const int want = tiles_env ? atoi(tiles_env) : 16;
  if (tiles > want) tiles = want;
  const uint32_t inv_tiles = (uint32_t)((0x100000000ull + (uint64_t)tiles - 1) / (uint64_t)tiles);
  if (wide_boxes)
    hipLaunchKernelGGL((describe_aware_kernel<true>), dim3(tiles * n_images), dim3(64 * kAwWaves), 0, stream, img, w, h,
                       pat, kps_in, kp_cap, kp_count_in, desc_tmp, valid_tmp, n_images, tiles, inv_tiles);
  else
    hipLaunchKernelGGL((describe_aware_kernel<false>), dim3(tiles * n_images), dim3(64 * kAwWaves), 0, stream, img, w, h,
                       pat, kps_in, kp_cap, kp_count_in, desc_tmp, valid_tmp, n_images, tiles, inv_tiles);
}

}  // namespace okvfe
