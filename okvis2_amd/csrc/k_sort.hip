// k_sort.hip -- K3: score-ordered candidate lists, one workgroup per image.
//
//   sort_rb_kernel   register-blocked bitonic sort of up to 8192 64-bit keys (score descending, y, x
//                    ascending -- a total order, so the result does not depend on K2's append order)
//   sort_kernel      the classic LDS / HBM-workspace form for larger candidate sets
// Since round 4 the lazy selection orders its own candidates; these serve the grid fall-backs, the
// AGAST / scale-space paths and the lab knob OKVFE_SELECT_PRESORTED.  Replaces the sort in
// brisk::ScaleSpaceLayer::DetectScaleSpaceMaxima (behind Frame.hpp:152).
#include <mutex>

#include "select_common_dev.h"

namespace okvfe {
namespace {

__device__ __forceinline__ void sort_classic_body(const Candidate* __restrict__ cand,
                                                  int cand_cap,
                                                  const int32_t* __restrict__ cand_count,
                                                  uint64_t* __restrict__ sort_ws,
                                                  int ws_stride, int lds_lo_keys,
                                                  int lds_keys, int img) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* lds = reinterpret_cast<uint64_t*>(smem_raw);
  int n = cand_count[img];
  // overflowed candidate list: WHICH maxima were dropped depends on the order of the atomics, so
  // the image keeps no keypoints at all (deterministic) and okvfe_check_capacity reports it
  n = n > cand_cap ? 0 : n;
  const Candidate* c = cand + (size_t)img * cand_cap;
  uint64_t* ws = sort_ws + (size_t)img * ws_stride;
  int np = 1;
  while (np < n) np <<= 1;
  const int tid = threadIdx.x;
  if (np <= lds_lo_keys) return;  // handled by the launch with the smaller LDS allocation
  if (np > lds_keys && lds_keys < 2 * kLdsSortKeys) return;  // left to the second launch
  if (np <= lds_keys) {
    for (int i = tid; i < np; i += kThreads) lds[i] = i < n ? make_key(c[i]) : ~0ull;
    __syncthreads();
    // Two consecutive strides (2j, j) of a phase touch the same 4 elements {b, b+j, b+2j, b+3j}
    // and all 4 lie in one k-block (same direction), so they are done in ONE pass with the keys in
    // registers: half the LDS traffic and half the barriers of the plain network.
    auto cswap = [](uint64_t& a, uint64_t& b, bool up) {
      const bool sw = (a > b) == up;
      const uint64_t x = sw ? b : a, y = sw ? a : b;
      a = x;
      b = y;
    };
    for (int k = 2; k <= np; k <<= 1) {
      int lj = 31 - __builtin_clz(k >> 1);  // largest stride of the phase = 1 << lj
      for (; lj >= 1; lj -= 2) {            // strides 1 << lj and 1 << (lj - 1)
        const int j = 1 << (lj - 1);
        for (int t = tid; t < (np >> 2); t += kThreads) {
          const int b = ((t >> (lj - 1)) << (lj + 1)) | (t & (j - 1));
          const bool up = ((b & k) == 0);
          uint64_t e0 = lds[b], e1 = lds[b + j], e2 = lds[b + 2 * j], e3 = lds[b + 3 * j];
          cswap(e0, e2, up);
          cswap(e1, e3, up);
          cswap(e0, e1, up);
          cswap(e2, e3, up);
          lds[b] = e0;
          lds[b + j] = e1;
          lds[b + 2 * j] = e2;
          lds[b + 3 * j] = e3;
        }
        __syncthreads();
      }
      if (lj == 0) {  // odd number of strides in this phase: the last one (stride 1) alone
        for (int t = tid; t < (np >> 1); t += kThreads) {
          const int lo = t << 1;
          const bool up = ((lo & k) == 0);
          uint64_t a = lds[lo], b = lds[lo + 1];
          cswap(a, b, up);
          lds[lo] = a;
          lds[lo + 1] = b;
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += kThreads) ws[i] = lds[i];
  } else {
    for (int i = tid; i < np; i += kThreads) ws[i] = i < n ? make_key(c[i]) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1) {
      for (int lj = 31 - __builtin_clz(k >> 1); lj >= 0; --lj) {
        const int j = 1 << lj;
        for (int t = tid; t < (np >> 1); t += kThreads) {
          const int lo = ((t >> lj) << (lj + 1)) | (t & (j - 1));
          const int hi = lo + j;
          const bool up = ((lo & k) == 0);
          const uint64_t a = ws[lo], b = ws[hi];
          if ((a > b) == up) {
            ws[lo] = b;
            ws[hi] = a;
          }
        }
        __syncthreads();
      }
    }
  }
}

// ---- register-blocked sort (up to 8192 keys), the production path -------------------------------
// Same total order, different network: the ALL-ASCENDING form of the bitonic sorter (first stride
// of a phase compares i with its mirror image i ^ (k - 1) inside the block of k, the remaining
// strides are plain half-cleaners), so the padding keys (~0) never leave the tail [n, np) and every
// compare-exchange whose lower index is >= n is skipped: the work follows n, not the next power
// of two (4 450 candidates per EuRoC image used to cost a full 8192 network).  A thread keeps 16
// keys in registers and runs up to FOUR strides on them between two LDS round trips (24 passes
// for 8192 keys instead of 49); phases 1..4 run on the 16 keys a thread loads from the candidate
// list before anything is written to LDS.  LDS index i lives at slot i + (i >> 4): with one pad
// slot per 16 keys all access patterns of the passes (16 keys per lane at strides 1, 2, 4 ... ) are
// bank-conflict free.  LDS-bandwidth / VALU bound, two workgroups per CU.
constexpr int kRbThreads = 512;
constexpr int kRbKeys = 16;  // per thread and pass
__host__ __device__ __forceinline__ int rb_slot(int i) { return i + (i >> 4); }
// Keys travel through the network as FP64 bit patterns: a compare-exchange is then v_min_f64 +
// v_max_f64 (2 instructions) instead of two 64-bit integer compares and four selects (~14 with the
// SGPR hazards).  A key K = (0x7FFFFFFF - score) << 32 | y << 16 | x is below 2^63; D = K - 2^62 in
// sign-magnitude form is a finite double (|D| <= 2^62 < 0x7FF0...: never Inf / NaN) whose IEEE order
// is the order of K.  Denormal patterns are ordinary values here (FP64 denormals are never flushed
// on this target) and no arithmetic touches the bits.
using RbKey = double;
__device__ __forceinline__ RbKey rb_encode(uint64_t k) {
  const int64_t d = (int64_t)k - (int64_t)(1ull << 62);
  const uint64_t bits = d >= 0 ? (uint64_t)d : (0x8000000000000000ull | (uint64_t)(-d));
  return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ uint64_t rb_decode(RbKey v) {
  const uint64_t bits = (uint64_t)__double_as_longlong(v);
  const int64_t mag = (int64_t)(bits & 0x7FFFFFFFFFFFFFFFull);
  const int64_t d = (bits >> 63) ? -mag : mag;
  return (uint64_t)(d + (int64_t)(1ull << 62));
}
constexpr uint64_t kRbPadKey = 0x7FFFFFFFFFFFFFFFull;  // above every real key (score >= 1)
__device__ __forceinline__ void rb_cswap(RbKey& a, RbKey& b) {  // ascending
  RbKey lo, hi;
  asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
  a = lo;
  b = hi;
}
// R strides (bits R-1 .. 0 of the element number m) on 2^R keys; MIRROR: the first one pairs m
// with its complement (the keys of the upper half were fetched with mirrored low index bits)
template <int R, bool MIRROR>
__device__ __forceinline__ void rb_network(RbKey (&key)[kRbKeys]) {
  constexpr int N = 1 << R;
  if (MIRROR) {
#pragma unroll
    for (int m = 0; m < N / 2; ++m) rb_cswap(key[m], key[(N - 1) ^ m]);
  }
#pragma unroll
  for (int b = MIRROR ? R - 2 : R - 1; b >= 0; --b) {
#pragma unroll
    for (int m = 0; m < N; ++m)
      if ((m & (1 << b)) == 0) rb_cswap(key[m], key[m | (1 << b)]);
  }
}
// one LDS pass: strides 2^(q+R-1) .. 2^q of the network on np keys
template <int R, bool MIRROR, int THREADS>
__device__ __forceinline__ void rb_pass(RbKey* lds, int np, int n, int q, int tid) {
  constexpr int N = 1 << R;
  const int low_mask = (1 << q) - 1;
  for (int g = tid; g < (np >> R); g += THREADS) {
    const int low = g & low_mask;
    const int base = ((g >> q) << (q + R)) | low;
    if (base >= n) continue;  // smallest index of the group: all of its keys are padding
    const int base_hi = MIRROR ? base ^ low_mask : base;  // upper half: mirrored low bits
    RbKey key[kRbKeys];
#pragma unroll
    for (int m = 0; m < N; ++m)
      key[m] = lds[rb_slot(((MIRROR && m >= N / 2) ? base_hi : base) | (m << q))];
    rb_network<R, MIRROR>(key);
#pragma unroll
    for (int m = 0; m < N; ++m)
      lds[rb_slot(((MIRROR && m >= N / 2) ? base_hi : base) | (m << q))] = key[m];
  }
}

// the network on one image's list of n <= 2^lnp keys, THREADS threads, 2^lnp + 2^(lnp-4) LDS slots
template <int THREADS>
__device__ __forceinline__ void sort_rb_body(RbKey* lds, const Candidate* __restrict__ c, int n, int lnp,
                                             uint64_t* __restrict__ ws) {
  const int np = 1 << lnp;
  const int tid = threadIdx.x;
  // phases 1..4 on 16 consecutive keys straight from the candidate list
  for (int g = tid; g < (np >> 4); g += THREADS) {
    RbKey key[kRbKeys];
#pragma unroll
    for (int m = 0; m < kRbKeys; ++m) {
      const int i = g * kRbKeys + m;
      key[m] = rb_encode(i < n ? make_key(c[i]) : kRbPadKey);
    }
    if (g * kRbKeys < n) {
      // phase k = 2^p inside the thread: mirror within blocks of 2^p keys, then half-cleaners
#pragma unroll
      for (int p = 1; p <= 4; ++p) {
        const int blk = 1 << p;
#pragma unroll
        for (int m = 0; m < kRbKeys; ++m)
          if ((m & (blk - 1)) < blk / 2) rb_cswap(key[m], key[m ^ (blk - 1)]);
#pragma unroll
        for (int b = p - 2; b >= 0; --b)
#pragma unroll
          for (int m = 0; m < kRbKeys; ++m)
            if ((m & (1 << b)) == 0) rb_cswap(key[m], key[m | (1 << b)]);
      }
    }
#pragma unroll
    for (int m = 0; m < kRbKeys; ++m) lds[rb_slot(g * kRbKeys + m)] = key[m];
  }
  __syncthreads();
  for (int lk = 5; lk <= lnp; ++lk) {  // phase: blocks of 2^lk keys
    int top = lk - 1;                  // highest stride bit still to do
    bool mirror = true;
    while (top >= 0) {
      const int r = top + 1 < 4 ? top + 1 : 4;
      const int q = top - r + 1;
      if (mirror) {
        rb_pass<4, true, THREADS>(lds, np, n, q, tid);  // lk >= 5: the first pass always has 4 strides
      } else {
        switch (r) {
          case 4: rb_pass<4, false, THREADS>(lds, np, n, q, tid); break;
          case 3: rb_pass<3, false, THREADS>(lds, np, n, q, tid); break;
          case 2: rb_pass<2, false, THREADS>(lds, np, n, q, tid); break;
          default: rb_pass<1, false, THREADS>(lds, np, n, q, tid); break;
        }
      }
      __syncthreads();
      top -= r;
      mirror = false;
    }
  }
  for (int i = tid; i < n; i += THREADS) ws[i] = rb_decode(lds[rb_slot(i)]);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void sort_rb_kernel(const Candidate* __restrict__ cand,
                                                             int cand_cap,
                                                             const int32_t* __restrict__ cand_count,
                                                             uint64_t* __restrict__ sort_ws,
                                                             int ws_stride, int max_keys) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int img = blockIdx.x;
  int n = cand_count[img];
  n = n > cand_cap ? 0 : n;  // overflowed list: no keypoints (see sort_kernel)
  if (n == 0) return;
  int lnp = 4;
  while ((1 << lnp) < n) ++lnp;
  if ((1 << lnp) > max_keys) return;  // left to sort_kernel (second launch)
  sort_rb_body<THREADS>(reinterpret_cast<RbKey*>(smem_raw), cand + (size_t)img * cand_cap, n, lnp,
                        sort_ws + (size_t)img * ws_stride);
}

// The large-list launch (and the legacy path): lists of 8193 .. 16384 keys run the register-blocked
// network with 1024 threads in 136 KiB of LDS (13.5 k maxima per 1024 x 1024 image: 0.28 -> 0.17 ms per
// 1024 images with the two-stride network before), everything else the classic bodies above.
__global__ __launch_bounds__(kThreads) void sort_kernel(const Candidate* __restrict__ cand,
                                                        int cand_cap,
                                                        const int32_t* __restrict__ cand_count,
                                                        uint64_t* __restrict__ sort_ws,
                                                        int ws_stride, int lds_lo_keys,
                                                        int lds_keys, int rb_mid, int n_images) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // The large-list launch allocates 136 KiB of LDS per workgroup: one workgroup per CU, so a grid of
  // one block per image cost ~2.5 us per 256 images even when no image needs it (the usual case: 16 us
  // per 1536-image step).  The grid is at most one round of workgroups; each walks its images.
  for (int img = blockIdx.x; img < n_images; img += gridDim.x) {  // block-uniform
    bool done = false;
    if (rb_mid) {
      int n = cand_count[img];
      n = n > cand_cap ? 0 : n;
      int lnp = 4;
      while ((1 << lnp) < n) ++lnp;
      if ((1 << lnp) > lds_lo_keys && (1 << lnp) <= 2 * kLdsSortKeys) {  // block-uniform
        sort_rb_body<kThreads>(reinterpret_cast<RbKey*>(smem_raw), cand + (size_t)img * cand_cap, n, lnp,
                               sort_ws + (size_t)img * ws_stride);
        done = true;
      }
    }
    if (!done) sort_classic_body(cand, cand_cap, cand_count, sort_ws, ws_stride, lds_lo_keys, lds_keys, img);
    __syncthreads();  // the LDS is reused by the next image
  }
}

// 2-D quadratic sub-pixel refinement; mirrors the published BRISK Subpixel2D with 64-bit
// coefficients (Harris scores overflow 32-bit products) and double Hessian terms.
// The nine Harris scores around pixel (u, v), 2 <= u < w - 2, 2 <= v < h - 2, straight from the image: exactly
// what the score map of harris_kernel holds there (k_harris.hip; HarrisScoreCalculator of the brisk library):
// Scharr (3, 10, 3) gradients, products >> 14 (zero on the image rim), 3 x 3 binomial, det - (trace / 4)^2.

}  // namespace

void launch_sort(const Candidate* cand, int cand_cap, const int32_t* cand_count, int n_images,
                 float radius, uint64_t* sort_ws, hipStream_t stream) {
  if (n_images <= 0 || !(radius > 0.0f)) return;
  int ws_stride = 1;
  while (ws_stride < cand_cap) ws_stride <<= 1;
  const int sort_keys = ws_stride < kLdsSortKeys ? ws_stride : kLdsSortKeys;
  const bool two = ws_stride > kLdsSortKeys;
  static const bool legacy = lab_env("OKVFE_LEGACY_SORT") != nullptr;  // A/B knob
  if (legacy) {
    // first launch: up to 8192 keys in 64 KiB (when it is the only launch it also takes the rest)
    hipLaunchKernelGGL(sort_kernel, dim3(n_images), dim3(kThreads), (size_t)sort_keys * 8, stream,
                       cand, cand_cap, cand_count, sort_ws, ws_stride, 0,
                       two ? sort_keys : 2 * kLdsSortKeys, 0, n_images);
  } else {
    // up to 8192 keys: register-blocked network in 68 KiB (16 keys minimum: one thread's share)
    const int keys = sort_keys < kRbKeys ? kRbKeys : sort_keys;
    // a handful of images (the B = 1 seams) cannot fill the GPU: twice the threads per list shorten the
    // one chain there is (28 -> ~18 us per 4.4 k keys); batches keep two 512-thread workgroups per CU
    if (n_images <= 64)
      hipLaunchKernelGGL(sort_rb_kernel<2 * kRbThreads>, dim3(n_images), dim3(2 * kRbThreads),
                         (size_t)rb_slot(keys) * 8, stream, cand, cand_cap, cand_count, sort_ws, ws_stride, keys);
    else
      hipLaunchKernelGGL(sort_rb_kernel<kRbThreads>, dim3(n_images), dim3(kRbThreads), (size_t)rb_slot(keys) * 8,
                         stream, cand, cand_cap, cand_count, sort_ws, ws_stride, keys);
  }
  if (two) {  // second launch: 8193..16384 keys in 136 KiB, larger sets in the HBM workspace
    const size_t big_lds = (size_t)rb_slot(2 * kLdsSortKeys) * 8;  // >= the classic 128 KiB
    static PerDeviceOnce attr_once;  // (several host threads may launch through several contexts, on several devices)
    attr_once.run([&] {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)big_lds) != hipSuccess)
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL(sort_kernel, dim3(n_images < 256 ? n_images : 256), dim3(kThreads), big_lds, stream, cand,
                       cand_cap, cand_count, sort_ws, ws_stride, kLdsSortKeys, 2 * kLdsSortKeys, legacy ? 0 : 1,
                       n_images);
  }
}

}  // namespace okvfe
