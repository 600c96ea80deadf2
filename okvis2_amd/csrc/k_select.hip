// k_select.hip -- K3/K4: uniformity enforcement without an occupancy grid, cap, sub-pixel refinement.
//
// Replaces the tail of brisk::ScaleSpaceLayer::DetectScaleSpaceMaxima (sort by score,
// EnforceKeypointUniformity, stop at maxNumKpt, Subpixel2D, emit cv::KeyPoint(pt, 12*scale, -1, score,
// layer)) behind cv::FeatureDetector::detect (okvis_cv/include/okvis/implementation/Frame.hpp:152;
// parameters okvis_frontend/src/Frontend.cpp:2406-2409 = uniformityRadius, octaves, absoluteThreshold,
// maxNumKpt).
//
// Kernels, one workgroup per image:
//   select_lazy_kernel<SORTS>  production path: accepted points in LDS spatial bins, occupancy summed on
//                              demand; SORTS = the kernel orders its own candidates (no sort launch)
//   select_list_kernel         the same over linked lists (fine grids with few keypoints per bin)
// launch_select picks the form; sorts: k_sort.hip, grid fall-backs: k_select_grid.hip, BRISK scale
// refinement: k_brisk_refine.hip, shared device helpers: select_common_dev.h.
// Bound: latency / LDS (no HBM roofline); batches keep all CUs busy with independent images.
#include <mutex>
#include "describe_setup_dev.h"
#include <type_traits>

#include "select_common_dev.h"

namespace okvfe {
namespace {

// ---- lazy-occupancy selection: no grid, one workgroup of 4 waves per image ----------------------
// The occupancy grid of the reference is only ever READ at the cells of candidates: a candidate
// passes when its level is not below the grid value at its own cell, and that value is
// min(255, sum over the points accepted so far within 15 cells of ceil(weight(offset) * 0.99 *
// level(point))) -- every stamp adds a non-negative integer with u8 saturation, so the saturating
// adds collapse into one clamp of the plain sum, in any order.  This kernel therefore keeps no grid:
// accepted points go into spatial bins of 16 x 16 cells and a candidate's occupancy is evaluated on
// demand from the 3 x 3 bins around its cell.
//
// Bins are ARRAYS (round 4): kLazyBinCap slots of {cell code, level} per bin plus a count; a
// candidate reads all slots of a bin with two 16-byte loads that depend on nothing but its own cell,
// so an evaluation is two LDS round trips (slots, then weights) whatever the bins hold -- the
// round-3 form chased linked lists, one dependent round trip per point, and was bound by exactly that
// latency (profiles/round3_select_stats.txt: 1.5 k cycles per window in the walk).  Empty slots hold
// level 0 and contribute ceil(w * 0) = 0.  The uniformity rule itself keeps bins sparse (a point
// needs the summed weights of its accepted neighbours below ~1, which spaces equal-level points ~13
// cells apart); a bin that does fill up spills into a per-image list in the HBM workspace that every
// evaluation then scans as well (exact, practically never taken).
//
// Prefilter (round 4).  The greedy pass walks the candidates in rank order, but a candidate's test
// only ever gets HARDER (occupancy grows): one that already fails against the points accepted before
// a block of candidates started fails when its turn comes as well.  So the sorted list is cut into
// blocks of 64, 128, 256, ... kLazyBlockMax candidates; all of a block's candidates are first
// evaluated IN PARALLEL (lane = candidate, every lane reads its own nine bins, no barrier between
// them) against the points accepted before the block, and only the survivors -- in rank order, a
// compaction of a sorted list stays sorted -- go through the ordered windows of 64: EuRoC-like
// content, 4.4 k candidates, ~500 survivors: 11 ordered windows instead of 70; TUM-VI 17 instead of
// 119 (the blocks grow geometrically because early blocks reject little: nothing is accepted yet).
//
// Ordered window of 64 survivors (lane = candidate): the nine bins are split over the four waves
// (2 + 2 + 2 + 3), partial sums go to LDS; wave 0 accepts in ROUNDS: a passing lane is decided once
// every earlier passing lane within reach of it is (found through 128 bin-hashed lane masks), all
// such lanes together; a newly accepted point adds its weight to the undecided lanes it reaches --
// a window never restarts.  Two workgroup barriers per window.
//
// Ordering (round 4, select_lazy_kernel<true>, the default): the kernel takes the UNSORTED candidate
// records and orders only what the greedy pass consumes -- log buckets of the score, chunks of whole
// buckets scattered into bucket order, chunk prefilter in any order, survivors rank-sorted in LDS,
// oversized chunks split by key range (see the SORTS branch below).  select_lazy_kernel<false> reads
// keys sorted by launch_sort (lab knob OKVFE_SELECT_PRESORTED).
//
// K4 runs in the tail: 2-D sub-pixel fit of the kept keypoints, from the score map or -- map-free calls
// -- from nine scores recomputed out of 7 x 7 pixels (harris_scores_3x3), then the extractor's
// per-keypoint setup when detection and description are one call.
constexpr int kLazyTabBytes = 32 * 32 * 4;  // weight(|dx|, |dy|), zero beyond 15
constexpr int kLazyThreads = 256;
constexpr int kLazyBinCap = 4;
#ifndef OKVFE_LAZY_BLOCKMAX
#define OKVFE_LAZY_BLOCKMAX 1024
#endif
#ifndef OKVFE_LAZY_WAVES
#define OKVFE_LAZY_WAVES 6
#endif
constexpr int kLazyBlockMax = OKVFE_LAZY_BLOCKMAX;  // (A/B: 512 + seven waves per SIMD = seven images per CU, tools/lab)
constexpr int kLazySurvPerWave = kLazyBlockMax / 4;  // surviving KEYS, one list per wave (the ordered windows read no HBM)
__host__ __device__ inline size_t lazy_align16(size_t v) { return (v + 15) & ~(size_t)15; }
// table | counts of the bordered bin grid | bin slots | survivor lists | partial sums
__host__ __device__ inline size_t lazy_lds_bytes(int bins_x, int bins_y) {
  const size_t nbins = (size_t)(bins_x + 2) * (bins_y + 2);
  return (size_t)kLazyTabBytes + lazy_align16(nbins * 4) + nbins * kLazyBinCap * 8 + (size_t)kLazyBlockMax * 8 +
         4 * 64 * 4;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's global
// stores (s_waitcnt vmcnt(0)), and the acceptance writes a keypoint record per accepted point -- an
// HBM round trip per window on the critical path of the selection (4.2 us per window measured; the
// records are only read again after the loop, behind a full barrier).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef OKVFE_LAB
// lab build: where does an image's time go?  [0] launches, [1..4] 10-ns ticks (s_memrealtime) of image 0's
// phases init / blocks / tail / total, [5] prefilter (+ rank sort) ticks, [6] survivors, [7] candidates,
// [8..10] window walk / accept / insert, [11] rounds, [12..15] tables / bucket count / schedule / scatter
__device__ unsigned long long g_lazy_prof[16];
#define OKVFE_LAZY_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define OKVFE_LAZY_TICK(var)
#endif
// log buckets of the score for the kernel that orders its candidates itself: 16 per octave, bucket 0 = the
// highest scores (ascending with the sort key)
constexpr int kFuseBins = 496;
constexpr int kFuseUnroll = 8;  // candidate records in flight per thread in the scatter pass
constexpr int kFuseFirst = 20;   // scores per thread requested before the tables are set up (5120 candidates)
constexpr int kFuseSched = 40;  // chunk ends kept in LDS; what lies beyond them is ONE last chunk (split by key range)
__device__ __forceinline__ int fuse_bin(int32_t score) {
  if (score <= 0) return kFuseBins - 1;
  const uint32_t s = (uint32_t)score;
  const int e = 31 - __clz(s);
  const int b = e >= 4 ? (e << 4) | (int)((s >> (e - 4)) & 15u) : (int)s;
  return (kFuseBins - 1) - b;
}
// SORTS = true (round 4, default): the kernel takes the UNSORTED candidate records and orders only what the
// greedy pass consumes -- see "chunks" below; false: `sort_ws` holds the keys already sorted (launch_sort).
template <bool SORTS>
__global__ __launch_bounds__(kLazyThreads) __attribute__((amdgpu_waves_per_eu(OKVFE_LAZY_WAVES, 8))) void select_lazy_kernel(
    const int32_t* __restrict__ scores, ScoreLayout layout, int w, int h, const Candidate* __restrict__ cand, int cand_cap,
    const int32_t* __restrict__ cand_count, uint64_t* sort_ws, int ws_stride,
    float radius, int max_kpts, const float* __restrict__ lut, int bins_x, int bins_y, int cap,
    okvfe_keypoint* __restrict__ kps, int kp_cap, int32_t* __restrict__ kp_count, uint2* __restrict__ spill_ws,
    size_t spill_stride, int bin_cap, int round_cap, DescribeSetup setup, const uint8_t* __restrict__ images) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_kept;
  __shared__ int s_spill;       // points that did not fit their bin (HBM list)
  __shared__ int s_surv[2][4];  // survivor counts per wave, double-buffered by block parity
  // SORTS: ends of the chunks in the bucket-ordered key array, their number, the highest score, and the
  // scratch of the key-range split of an oversized chunk
  __shared__ uint32_t s_sched[kFuseSched + 2];
  __shared__ int s_nsched, s_max, s_cnt;
  __shared__ unsigned long long s_kmin, s_kmax;
  const int bpitch = bins_x + 2;  // bordered bin grid: the border bins stay empty, so no range checks
  const int nbins = bpitch * (bins_y + 2);
  float* tab = reinterpret_cast<float*>(smem_raw);
  unsigned char* q0 = smem_raw + kLazyTabBytes;
  uint32_t* head = reinterpret_cast<uint32_t*>(q0);  // bits 0..7: points in the bin (the acceptance keeps its lane masks in `part`)
  q0 += lazy_align16((size_t)nbins * 4);
  uint4* slot4 = reinterpret_cast<uint4*>(q0);  // [nbins][2]: {code, level, code, level}
  q0 += (size_t)nbins * kLazyBinCap * 8;
  uint64_t* surv = reinterpret_cast<uint64_t*>(q0);  // [4][kLazySurvPerWave]
  float* part = reinterpret_cast<float*>(q0 + (size_t)kLazyBlockMax * 8);  // [4][64]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int img = blockIdx.x;
  int n = cand_count[img];
  // overflowed candidate list: WHICH maxima were dropped depends on the order of the atomics, so
  // the image keeps no keypoints at all (deterministic) and okvfe_check_capacity reports it
  n = n > cand_cap ? 0 : n;
  uint64_t* keys = sort_ws + (size_t)img * ws_stride;
  okvfe_keypoint* out = kps + (size_t)img * kp_cap;
  uint2* spill = spill_ws + (size_t)img * spill_stride;  // {cy << 16 | cx, level}
  int kept = 0;
  OKVFE_LAZY_TICK(t_start);
#ifdef OKVFE_LAB
  unsigned long long t_pa = t_start, t_pb = t_start, t_p0 = t_start, t_init = t_start, t_blocks = t_start, t_pref = 0, t_mark = 0, t_walk = 0, t_acc = 0, t_ins = 0, t_w0 = 0, t_w1 = 0, t_w2 = 0;
  int n_surv = 0, n_rounds = 0;
#endif
  if (n > 0) {  // block-uniform
    // weight(|dx|, |dy|) at [|dy| << 5 | |dx|], zero outside the 31 x 31 stamp
    for (int i = tid; i < 32 * 32; i += kLazyThreads) {
      const int dx = i & 31, dy = i >> 5;
      tab[i] = (dx <= 15 && dy <= 15) ? lut[(dy + 15) * 31 + (dx + 15)] : 0.0f;
    }
    for (int i = tid; i < nbins; i += kLazyThreads) head[i] = 0u;
    // empty slot: level 0 (contributes nothing) at a VALID cell code, so its table index stays in range
    for (int i = tid; i < nbins * 2; i += kLazyThreads)
      slot4[i] = make_uint4((16u << 23) | (16u << 2), 0u, (16u << 23) | (16u << 2), 0u);
    if (tid == 0) {
      s_kept = 0;
      s_spill = 0;
      if constexpr (SORTS) s_max = INT_MIN;
    }
    if constexpr (SORTS) {  // bucket counts: in the survivor buffer, which is idle until the first chunk
      for (int i = tid; i <= kFuseBins; i += kLazyThreads) reinterpret_cast<uint32_t*>(surv)[i] = 0u;
    }
    // SORTS: the first 5120 candidate records are are read ONCE and STAY in registers as {score, y << 16 | x} for both bucket passes:
    // in a batch every pass over the records is 53 KB per image from HBM (81 MB per 1536 images: 16 us)
    const Candidate* crec = cand + (size_t)img * cand_cap;
    int32_t scv0[SORTS ? kFuseFirst : 1];
    uint32_t pyx0[SORTS ? kFuseFirst : 1];
    if constexpr (SORTS) {
#pragma unroll
      for (int u = 0; u < kFuseFirst; ++u) {
        const int i = tid + u * kLazyThreads;
        const Candidate c = i < n ? crec[i] : Candidate{0, 0, INT_MIN};
        scv0[u] = c.score;
        pyx0[u] = ((uint32_t)c.y << 16) | (uint32_t)c.x;
      }
    }
    const float scaling = (float)(15.0 / (double)radius);
    float max_score = 1.0f;  // the highest score of the image (SORTS: known after the bucket pass)
    if constexpr (!SORTS) max_score = (float)(0x7FFFFFFF - (int32_t)(keys[0] >> 32));
    // sorted key -> candidate record {cell, level, pixel, score}
    auto make_rec = [&](uint64_t k) {
      const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
      const int y = (int)((k >> 16) & 0xFFFF), x = (int)(k & 0xFFFF);
      const float fy = (float)y * scaling;
      const float fx = (float)x * scaling;
      const int cy = (int)(fy + 16.0f);
      const int cx = (int)(fx + 16.0f);
      const float q = (float)score / max_score;
      const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
      return make_uint4(((uint32_t)cy << 16) | (uint32_t)cx, __float_as_uint(nsc1), (uint32_t)(k & 0xFFFFFFFFu),
                        (uint32_t)score);
    };
    // One slot = {code, level}: code = (128 * (16 + y in bin)) << 16 | 4 * (16 + x in bin).  A candidate
    // at (lx, ly) of its bin looks at the bin at offset (ox, oy) with A = (128 * (ly + 16 (1 - oy))) << 16
    // | 4 * (lx + 16 (1 - ox)): v_sad_u16 adds the absolute differences of both halves, so
    // sad(A, code) + table address = the ADDRESS of weight(|dx|, |dy|) in one instruction.
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)tab;
    // (counters, batch of 1536 images: the LDS was the busiest unit of this kernel, 62 % of its cycles
    // and 71 % of those bank conflicts -- 64 lanes gathering weights at unrelated table addresses.  An
    // EMPTY slot, two thirds of them, now reads the table's first word instead: one address for all such
    // lanes is a broadcast, not a conflict.  Its level is 0, so the term stays 0.)
    auto waddr_of = [&](uint32_t A, uint32_t code, uint32_t levbits) {
      const uint32_t a = __builtin_amdgcn_sad_u16(A, code, tab_addr);
      return levbits != 0u ? a : tab_addr;
    };
    auto term = [&](uint32_t waddr, uint32_t levbits) {
      return ceilf(*reinterpret_cast<__attribute__((address_space(3))) const float*>((uintptr_t)waddr) *
                   __uint_as_float(levbits));  // 0 for an empty slot (level 0)
    };
    // three bins at once (all slot loads, then all weight loads, then the arithmetic: two LDS round
    // trips for the group); half = 0: slots 0, 1 of every bin, half = 1: slots 2, 3
    auto bins3 = [&](const int b[3], const uint32_t A[3], int half) {
      uint4 s3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) s3[k] = slot4[2 * b[k] + half];
      uint32_t wa[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // (bit 0 of a bin's first code word is its "more than two points" flag, not part of the code)
        wa[2 * k] = waddr_of(A[k], half == 0 ? s3[k].x & ~1u : s3[k].x, s3[k].y);
        wa[2 * k + 1] = waddr_of(A[k], s3[k].z, s3[k].w);
      }
      float f = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) f += term(wa[2 * k], s3[k].y) + term(wa[2 * k + 1], s3[k].w);
      return f;
    };
    // points that overflowed their bins (rare): every lane scans the whole list
    auto spill_terms = [&](int cx, int cy) {
      float f = 0.0f;
      const int ns = s_spill;
      for (int i = 0; i < ns; ++i) {  // block-uniform trip count
        const uint2 p = spill[i];
        const int dx = cx - (int)(p.x & 0xFFFF), dy = cy - (int)(p.x >> 16);
        const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        if (adx <= 15 && ady <= 15) f += ceilf(tab[(ady << 5) | adx] * __uint_as_float(p.y));
      }
      return f;
    };
    // the ordered windows split the nine bins of a candidate over the waves: bins c = first,
    // first + 4 (, first + 8) in row-major order of the 3 x 3 block
    const int first = (wave + 1) & 3;  // wave 0 (which also runs the acceptance) and waves 1, 2 take two bins
    const int nch = first == 0 ? 3 : 2;
    int hoff[3];          // offset of the bin from the candidate's bin in the bordered grid
    uint32_t aoff[3];     // what the bin's offset adds to the candidate's slot-address operand
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = first + 4 * k < 9 ? first + 4 * k : first;
      const int ox = (c % 3) - 1, oy = (c / 3) - 1;
      hoff[k] = oy * bpitch + ox;
      aoff[k] = ((uint32_t)(-16 * oy) << 23) + ((uint32_t)(-16 * ox) << 2);  // (wraps: the fields stay in range)
    }
    const int limit = min(min(max_kpts, kp_cap), cap);
    uint64_t* my_surv = surv + wave * kLazySurvPerWave;
    // one candidate against the points accepted so far, all nine bins from this lane: does it survive?
    // (occupancy only grows, so a failure is final)
    auto prefilter_key = [&](uint64_t kthis, bool valid) -> bool {
          uint4 rec = make_rec(kthis);
          if (!valid) rec.x = 0u;
          const int cx = (int)(rec.x & 0xFFFF), cy = (int)(rec.x >> 16);
          const float level = __uint_as_float(rec.y);
          const int bin = ((cy >> 4) + 1) * bpitch + ((cx >> 4) + 1);
          const uint32_t lx = (uint32_t)(cx & 15), ly = (uint32_t)(cy & 15);
          const uint32_t A0 = ((ly + 16u) << 23) | ((lx + 16u) << 2);  // the candidate's own bin
          float occf = 0.0f;  // sums of small integers: exact in float
          {
            // all nine count words and all nine first-half slot pairs are requested before anything is
            // used: one LDS round trip for them, one for the eighteen weights (the per-group form waited
            // ~50 times per candidate, alone on its SIMD at B = 1)
            // (two batches of five and four bins: all nine at once need more registers than six waves
            // per SIMD leave, and a spilled register costs a scratch round trip per candidate)
            bool more = false;
            auto batch = [&](auto first, auto count) {
              constexpr int B0 = decltype(first)::value, NB = decltype(count)::value;
              uint32_t A[NB];
              uint4 sb[NB];
#pragma unroll
              for (int q = 0; q < NB; ++q) {
                const int b = B0 + q;
                const int bi = bin + ((b / 3) - 1) * bpitch + (b % 3) - 1;
                sb[q] = slot4[2 * bi];
                A[q] = A0 + ((uint32_t)(16 * (1 - b / 3)) << 23) + ((uint32_t)(16 * (1 - b % 3)) << 2);
              }
              uint32_t wa[2 * NB];
#pragma unroll
              for (int q = 0; q < NB; ++q) {
                wa[2 * q] = waddr_of(A[q], sb[q].x & ~1u, sb[q].y);
                wa[2 * q + 1] = waddr_of(A[q], sb[q].z, sb[q].w);
              }
              float wv[2 * NB];
#pragma unroll
              for (int t = 0; t < 2 * NB; ++t)
                wv[t] = *reinterpret_cast<__attribute__((address_space(3))) const float*>((uintptr_t)wa[t]);
#pragma unroll
              for (int q = 0; q < NB; ++q) {
                occf += ceilf(wv[2 * q] * __uint_as_float(sb[q].y)) + ceilf(wv[2 * q + 1] * __uint_as_float(sb[q].w));
                more = more || (sb[q].x & 1u) != 0u;
              }
            };
            batch(std::integral_constant<int, 0>(), std::integral_constant<int, 5>());
            batch(std::integral_constant<int, 5>(), std::integral_constant<int, 4>());
            if (__any(more)) {  // some bin of some lane holds more than two points
#pragma unroll
              for (int g = 0; g < 3; ++g) {
                const int b0 = bin + (g - 1) * bpitch;
                const int b[3] = {b0 - 1, b0, b0 + 1};
                const uint32_t Ag = A0 + ((uint32_t)(16 * (1 - g)) << 23);
                const uint32_t A[3] = {Ag + (16u << 2), Ag, Ag - (16u << 2)};
                occf += bins3(b, A, 1);
              }
            }
          }
          if (s_spill != 0) occf += spill_terms(cx, cy);
          const int occ = (int)occf;
          const bool pass = valid && !(level < (float)(occ > 255 ? 255 : occ));
          return pass;
    };
    // ordered windows of 64 over `total` survivors; fetch(g) = the g-th of them in key order
    auto windows = [&](int total, auto fetch) {
      for (int pos = 0; pos < total; pos += 64) {  // block-uniform
#ifdef OKVFE_LAB
        t_w0 = __builtin_amdgcn_s_memrealtime();
#endif
        const int g = pos + lane;
        const bool valid = g < total;
        uint4 rec = make_rec(valid ? fetch(g) : 0ull);
        if (!valid) rec.x = 0u;  // a cell inside the grid; the lane never passes
        const int cx = (int)(rec.x & 0xFFFF), cy = (int)(rec.x >> 16);
        const float level = __uint_as_float(rec.y);
        const int bx = cx >> 4, by = cy >> 4;
        const int bin = (by + 1) * bpitch + (bx + 1);
        const uint32_t lx = (uint32_t)(cx & 15), ly = (uint32_t)(cy & 15);
        {
          const uint32_t A0 = ((ly + 16u) << 23) | ((lx + 16u) << 2);
          int b[3];
          uint32_t A[3];
          bool more = false;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            // a wave with two bins looks at bin 0 as its third: a border bin, always empty
            b[k] = k < nch ? bin + hoff[k] : 0;
            A[k] = A0 + aoff[k];
            more = more || (reinterpret_cast<const uint32_t*>(slot4)[8 * b[k]] & 1u) != 0u;
          }
          float f = bins3(b, A, 0);
          if (__any(more)) f += bins3(b, A, 1);
          if (wave == 3 && s_spill != 0) f += spill_terms(cx, cy);
          part[wave * 64 + lane] = f;
        }
        lds_barrier();
#ifdef OKVFE_LAB
        t_w1 = __builtin_amdgcn_s_memrealtime();
        t_walk += t_w1 - t_w0;
#endif
        if (wave == 0) {
          int occ = (int)(part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane]);
          bool pass = valid && !(level < (float)(occ > 255 ? 255 : occ));
          const float nsc = (float)(0.99 * (double)level);
          unsigned long long rem = __ballot(pass), accm = 0;
          int nacc = 0;
          if (rem != 0) {
            const int room = limit - kept;
            if (__popcll(rem) <= room) {
              // Acceptance in ROUNDS (round 4).  A passing lane is decided once every EARLIER passing
              // lane of the window within reach of it is: its occupancy is final then.  All such lanes
              // are decided together -- they cannot reach each other -- instead of one per iteration of
              // a scalar-vector ping-pong (~300 cycles per accepted point, half of the selection's
              // time): nb = the EARLIER passing lanes within the stamp's square around this lane; a round takes the lanes with no undecided earlier
              // neighbour, adds the weights of the newly accepted ones to the lanes they reach
              // (2 lane permutes + 1 table read each) and drops the lanes that now fail (a test only
              // ever gets harder).  3-5 rounds per window of survivors.
              // nb: the `part` array (free in this phase) becomes 128 lane masks hashed by bin; a passing
              // lane ORs its bit into its bin's mask, reads the nine masks around it (a superset: other
              // bins alias) and checks the few lanes it finds exactly
              unsigned long long* wm = reinterpret_cast<unsigned long long*>(part);
              wm[lane] = 0ull;
              wm[64 + lane] = 0ull;
              __builtin_amdgcn_wave_barrier();
              if (pass) atomicOr(&wm[bin & 127], 1ull << lane);
              __builtin_amdgcn_wave_barrier();
              const unsigned long long lower_ = (1ull << lane) - 1ull;
              unsigned long long cm = 0ull;
#pragma unroll
              for (int b = 0; b < 9; ++b) cm |= wm[(bin + ((b / 3) - 1) * bpitch + (b % 3) - 1) & 127];
              cm &= rem & lower_;
              if (!pass) cm = 0ull;
              unsigned long long nb = 0ull;
              while (__any(cm != 0ull)) {
                const int el = cm != 0ull ? (int)__ffsll((long long)cm) - 1 : lane;
                const uint32_t exy = (uint32_t)__builtin_amdgcn_ds_bpermute(el << 2, (int)rec.x);
                if (cm != 0ull) {
                  cm &= cm - 1ull;
                  const uint32_t ddx = (uint32_t)(cx - (int)(exy & 0xFFFF) + 15);
                  const uint32_t ddy = (uint32_t)(cy - (int)(exy >> 16) + 15);
                  nb |= (ddx <= 30u && ddy <= 30u) ? (1ull << el) : 0ull;
                }
              }
              while (rem != 0ull) {
                const bool in = ((rem >> lane) & 1ull) != 0ull;
                const bool ready = in && (nb & rem) == 0ull;  // (the lowest lane of rem always is)
                const unsigned long long rdy = __ballot(ready);
                const unsigned long long accn = __ballot(ready && pass);
                accm |= accn;
                rem &= ~rdy;
                // undecided lanes in reach of a point accepted in this round
                unsigned long long m = (in && !ready) ? (nb & accn) : 0ull;
                while (__any(m != 0ull)) {
                  const int jl = m != 0ull ? (int)__ffsll((long long)m) - 1 : lane;
                  const uint32_t jxy = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)rec.x);
                  const float jnsc = __int_as_float(__builtin_amdgcn_ds_bpermute(jl << 2, __float_as_int(nsc)));
                  if (m != 0ull) {
                    m &= m - 1ull;
                    const int dx = cx - (int)(jxy & 0xFFFF), dy = cy - (int)(jxy >> 16);
                    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
                    occ += (int)ceilf(tab[(ady << 5) | adx] * jnsc);
                  }
                }
                if (in && !ready) pass = !(level < (float)(occ > 255 ? 255 : occ));
                rem &= __ballot(pass);  // lanes that fail now fail at their turn as well: decided
              }
              nacc = __popcll(accm);
            } else {
              // the cap falls inside this window: one candidate at a time, in order
              while (rem != 0 && kept + nacc < limit) {
                const int f = (int)__ffsll((long long)rem) - 1;
                accm |= 1ull << f;
                ++nacc;
                const uint32_t wxy = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, f);
                const float wnsc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nsc), f));
                const int dx = cx - (int)(wxy & 0xFFFF), dy = cy - (int)(wxy >> 16);
                const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
                const bool near = pass && lane > f && adx <= 15 && ady <= 15;
                if (__any(near)) {  // the new point's stamp reaches later passing candidates of this window
                  if (near) {
                    occ += (int)ceilf(tab[(ady << 5) | adx] * wnsc);
                    pass = !(level < (float)(occ > 255 ? 255 : occ));
                  }
                  rem &= __ballot(pass) & ~(((2ull << f) - 1ull));
                } else {
                  rem &= rem - 1;
                }
              }
            }
#ifdef OKVFE_LAB
            t_w2 = __builtin_amdgcn_s_memrealtime();
            t_acc += t_w2 - t_w1;
#endif
            if ((accm >> lane) & 1) {
              const int slot = kept + __popcll(accm & ((1ull << lane) - 1ull));
              // into the bin: slot j of its array, or the spill list when the array is full
              const uint32_t j = atomicAdd(&head[bin], 1u) & 0xFFu;
              const uint32_t code = ((ly + 16u) << 23) | ((lx + 16u) << 2);
              if (j < (uint32_t)bin_cap) {  // (kLazyBinCap; the lab build can shrink it to exercise the spill list)
                reinterpret_cast<uint2*>(slot4)[bin * kLazyBinCap + (int)j] = make_uint2(code, __float_as_uint(nsc));
                // the bin's third point raises the "second half in use" flag in its first code word
                if (j == 2u) atomicOr(&reinterpret_cast<uint32_t*>(slot4)[8 * bin], 1u);
              } else {
                atomicSub(&head[bin], 1u);  // the count stays at the capacity
                const int at = atomicAdd(&s_spill, 1);
                spill[at] = make_uint2(rec.x, __float_as_uint(nsc));
                __threadfence_block();  // the other waves read the list right after the window's barrier
              }
              // pixel position and score wait in the output record for the sub-pixel pass
              okvfe_keypoint kp;
              kp.x = (float)(int)(rec.z & 0xFFFF);
              kp.y = (float)(int)(rec.z >> 16);
              kp.size = 12.0f;
              kp.angle = -1.0f;
              kp.response = (float)(int32_t)rec.w;
              kp.octave = 0;
              kp.class_id = (int32_t)rec.w;  // the exact score (response is its float image)
              out[slot] = kp;
            }
            kept += nacc;
            if (lane == 0) s_kept = kept;
          }
        }
        lds_barrier();
#ifdef OKVFE_LAB
        t_ins += __builtin_amdgcn_s_memrealtime() - (t_w2 > t_w1 ? t_w2 : t_w1);
#endif
        kept = s_kept;
        if (kept >= limit) break;  // block-uniform
      }
    };
    if constexpr (!SORTS) {
      __syncthreads();
#ifdef OKVFE_LAB
      t_init = __builtin_amdgcn_s_memrealtime();
#endif
      // this wave's quarter of a block: keys [lo, hi), up to kLazySurvPerWave = 4 x 64 of them, held in
      // registers; the NEXT block's are requested before the current block is worked on, so their HBM /
      // L2 round trip hides behind it
      auto load_quarter = [&](int a_, int blen_, uint64_t kr[4]) {
        const int q = blen_ >> 2;
        const int lo = a_ + wave * q, hi = min(lo + q, min(a_ + blen_, n));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = lo + 64 * t + lane;
          kr[t] = i < hi ? keys[i] : 0ull;
        }
      };
      uint64_t kcur[4] = {0ull, 0ull, 0ull, 0ull}, knxt[4];
      int blen = 64, par = 0;
      for (int a = 0; a < n; a += blen, blen = min(2 * blen, kLazyBlockMax), par ^= 1) {  // block-uniform
        const int e = min(a + blen, n);
#ifdef OKVFE_LAB
        t_mark = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- prefilter: this wave's quarter of the block, against the points accepted before the block
        int my_cnt = 0;
        if (a == 0) {
          // nothing is accepted yet: every candidate of the first block survives
          const int q = (e - a + 3) >> 2;
          const int lo = wave * q, hi = min(lo + q, e - a);
          for (int c = lo + lane; c < hi; c += 64) my_surv[c - lo] = keys[c];
          my_cnt = hi > lo ? hi - lo : 0;
        } else {
          const int q = blen >> 2;  // multiple of 16
          const int lo = a + wave * q, hi = min(lo + q, e);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c0 = lo + 64 * t;
            if (c0 >= hi) break;  // wave-uniform
            const uint64_t kthis = kcur[t];
            const bool pass = prefilter_key(kthis, c0 + lane < hi);
            const unsigned long long m = __ballot(pass);
            if (pass) my_surv[my_cnt + __popcll(m & ((1ull << lane) - 1ull))] = kthis;
            my_cnt += __popcll(m);
          }
        }
        // the next block's keys: requested now, used after this block's ordered windows
        load_quarter(a + blen, min(2 * blen, kLazyBlockMax), knxt);
        // (a wave may run ahead into the next block's prefilter while another still reads these counts:
        // the buffer written two blocks later is safe, that wave has passed the barrier in between)
        if (lane == 0) s_surv[par][wave] = my_cnt;
        lds_barrier();
        const int c0n = s_surv[par][0], c1n = c0n + s_surv[par][1], c2n = c1n + s_surv[par][2],
                  total = c2n + s_surv[par][3];
#ifdef OKVFE_LAB
        n_surv += total;
        t_pref += __builtin_amdgcn_s_memrealtime() - t_mark;
#endif
        // ---- ordered windows over the block's survivors (rank order: wave lists in turn)
        windows(total, [&](int g) {
          int li = 0, lo = g;
          if (g >= c2n) { li = 3; lo = g - c2n; }
          else if (g >= c1n) { li = 2; lo = g - c1n; }
          else if (g >= c0n) { li = 1; lo = g - c0n; }
          return surv[li * kLazySurvPerWave + lo];
        });
        if (kept >= limit) break;  // block-uniform
#pragma unroll
        for (int t = 0; t < 4; ++t) kcur[t] = knxt[t];
      }
    } else {
      // ---- the kernel orders its candidates itself (round 4) --------------------------------------
      // Sorting all ~4.4 k candidates of an image cost as much as selecting among them, and most of the
      // order is never used: a candidate that fails against the points accepted from HIGHER scores fails
      // whatever its rank among its peers.  So: (A) count the candidates into log buckets of the score,
      // (B) cut the bucket sequence into CHUNKS of about 64, 128, ... 1024 candidates (whole buckets), (C)
      // scatter the keys into bucket order (HBM workspace, unordered inside a bucket); then per chunk:
      // prefilter its keys in any order against the points accepted from the earlier chunks, rank-sort
      // the SURVIVORS (a hundred or so) in LDS, and run the ordered windows over them.  A chunk larger
      // than the buffer (a bucket of a thousand equal scores) is split by key range: histogram of the key
      // over the bucket's own [min, max], narrowed until a prefix of it fits.
      uint32_t* hist = reinterpret_cast<uint32_t*>(surv);  // [kFuseBins + 1], until the first chunk is worked on
      __syncthreads();
#ifdef OKVFE_LAB
      t_p0 = __builtin_amdgcn_s_memrealtime();
#endif
      {
        // (many records per thread in flight together: one at a time, each pass over the records was a
        // chain of ~17 HBM round trips)
        int mx = INT_MIN;
#pragma unroll
        for (int u = 0; u < kFuseFirst; ++u) {
          if (tid + u * kLazyThreads < n) {
            atomicAdd(&hist[fuse_bin(scv0[u])], 1u);
            mx = max(mx, scv0[u]);
          }
        }
        for (int base = tid + kFuseFirst * kLazyThreads; base < n; base += 8 * kLazyThreads) {  // (more than 5120)
          int32_t scv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = base + u * kLazyThreads;
            scv[u] = i < n ? crec[i].score : INT_MIN;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (base + u * kLazyThreads < n) {
              atomicAdd(&hist[fuse_bin(scv[u])], 1u);
              mx = max(mx, scv[u]);
            }
          }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
        if (lane == 0) atomicMax(&s_max, mx);
      }
      __syncthreads();
#ifdef OKVFE_LAB
      t_pa = __builtin_amdgcn_s_memrealtime();
#endif
      max_score = (float)s_max;
      if (wave == 0) {
        // bucket starts: exclusive prefix (lane = 8 buckets), hist[kFuseBins] = n
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int idx = lane * 8 + k;
          v[k] = idx < kFuseBins ? hist[idx] : 0u;
          sum += v[k];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(inc, d);
          if (lane >= d) inc += t;
        }
        uint32_t run = inc - sum;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int idx = lane * 8 + k;
          if (idx <= kFuseBins) hist[idx] = run;
          run += v[k];
        }
        __builtin_amdgcn_wave_barrier();
        // chunk schedule: whole buckets, about 64, 128, ... kLazyBlockMax candidates each
        int pos = 0, b = 0, k = 0;
        uint32_t target = min(64u, (uint32_t)round_cap);
        while (pos < n && k < kFuseSched) {  // wave-uniform
          int x_last = b - 1;
          for (int base = b; base < kFuseBins; base += 64) {
            const int idx = base + lane;
            const uint32_t ve = idx < kFuseBins ? hist[idx + 1] : 0xFFFFFFFFu;  // end of bucket idx
            const bool ok = idx < kFuseBins && ve - (uint32_t)pos <= target;     // (a prefix of the lanes: ends ascend)
            const int c = __popcll(__ballot(ok));
            x_last += c;
            if (c < 64) break;
          }
          int e = x_last >= b ? (int)hist[x_last + 1] : pos;
          if (e == pos) {  // the next non-empty bucket alone exceeds the target: it is the chunk
            for (int base = b; base < kFuseBins; base += 64) {
              const int idx = base + lane;
              const bool gt = idx < kFuseBins && (int)hist[idx + 1] > pos;
              const unsigned long long m = __ballot(gt);
              if (m != 0ull) {
                x_last = base + (int)__ffsll((long long)m) - 1;
                break;
              }
            }
            e = (int)hist[x_last + 1];
          }
          if (lane == 0) s_sched[k] = (uint32_t)e;
          ++k;
          pos = e;
          b = x_last + 1;
          target = min(2u * target, (uint32_t)round_cap);
        }
        if (pos < n) {  // what the table cannot hold: one last chunk
          if (lane == 0) s_sched[k] = (uint32_t)n;
          ++k;
        }
        if (lane == 0) s_nsched = k;
      }
      __syncthreads();
#ifdef OKVFE_LAB
      t_pb = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
      for (int u = 0; u < kFuseFirst; ++u) {
        if (tid + u * kLazyThreads < n) {
          const uint32_t p = atomicAdd(&hist[fuse_bin(scv0[u])], 1u);
          keys[p] = ((uint64_t)(uint32_t)(0x7FFFFFFF - scv0[u]) << 32) | pyx0[u];  // = make_key
        }
      }
      for (int base = tid + kFuseFirst * kLazyThreads; base < n; base += kFuseUnroll * kLazyThreads) {  // (more than 5120)
        Candidate cv[kFuseUnroll];
#pragma unroll
        for (int u = 0; u < kFuseUnroll; ++u) {
          const int i = base + u * kLazyThreads;
          cv[u] = i < n ? crec[i] : Candidate{0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < kFuseUnroll; ++u) {
          if (base + u * kLazyThreads < n) {
            const uint32_t p = atomicAdd(&hist[fuse_bin(cv[u].score)], 1u);
            keys[p] = make_key(cv[u]);
          }
        }
      }
      __syncthreads();
#ifdef OKVFE_LAB
      t_init = __builtin_amdgcn_s_memrealtime();
#endif
      const int nsched = s_nsched;
      uint64_t kcur[4] = {0ull, 0ull, 0ull, 0ull}, knxt[4] = {0ull, 0ull, 0ull, 0ull};
      // this wave's quarter of keys[c0 .. c0 + m)
      auto load_chunk = [&](int c0, int m, uint64_t kr[4]) {
        const int q = (m + 3) >> 2;
        const int lo = c0 + wave * q, hi = min(lo + q, c0 + m);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = lo + 64 * t + lane;
          kr[t] = i < hi ? keys[i] : 0ull;
        }
      };
      int cpos = 0, j = 0, par = 0;
      bool have_next = false;
      bool refining = false;  // inside an oversized chunk keys[r_base .. r_base + r_m): keys below r_lo are done
      int r_base = 0, r_m = 0, r_rem = 0;
      uint64_t r_lo = 0ull;
      while (true) {  // one round = up to kLazyBlockMax keys, this wave's quarter of them in kcur (all block-uniform)
        int cnt = 0;
#ifdef OKVFE_LAB
        t_mark = __builtin_amdgcn_s_memrealtime();
#endif
        if (!refining) {
          if (j >= nsched) break;
          const int e = (int)s_sched[j];
          const int m = e - cpos;
          ++j;
          if (m > round_cap) {  // (kLazyBlockMax; the lab build can shrink it to exercise the split)
            refining = true;
            r_base = cpos; r_m = m; r_rem = m; r_lo = 0ull;
            cpos = e;
            have_next = false;
            continue;
          }
          if (have_next) {
#pragma unroll
            for (int t = 0; t < 4; ++t) kcur[t] = knxt[t];
          } else {
            load_chunk(cpos, m, kcur);
          }
          cnt = m;
          cpos = e;
          have_next = false;
        } else {
          uint64_t hi = ~0ull;
          if (r_rem > round_cap) {
            // [min, max] of the keys still to do
            if (tid == 0) {
              s_kmin = ~0ull;
              s_kmax = 0ull;
            }
            __syncthreads();
            unsigned long long mn = ~0ull, mxk = 0ull;
            for (int i = tid; i < r_m; i += kLazyThreads) {
              const unsigned long long k = keys[r_base + i];
              if (k >= r_lo) {
                mn = k < mn ? k : mn;
                mxk = k > mxk ? k : mxk;
              }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
              const unsigned long long a1 = __shfl_xor(mn, d), a2 = __shfl_xor(mxk, d);
              mn = a1 < mn ? a1 : mn;
              mxk = a2 > mxk ? a2 : mxk;
            }
            if (lane == 0) {
              atomicMin(&s_kmin, mn);
              atomicMax(&s_kmax, mxk);
            }
            __syncthreads();
            const unsigned long long kmin = s_kmin, range = s_kmax - kmin;
            int sh = range < 256ull ? 0 : (64 - __clzll((long long)range)) - 8;  // (range >> sh) < 256
            uint32_t* sub = reinterpret_cast<uint32_t*>(part);  // [256] counts of (key - kmin) >> sh
            while (true) {  // block-uniform
              sub[tid] = 0u;
              __syncthreads();
              for (int i = tid; i < r_m; i += kLazyThreads) {
                const unsigned long long k = keys[r_base + i];
                if (k >= r_lo) {
                  const unsigned long long d = (k - kmin) >> sh;
                  if (d < 256ull) atomicAdd(&sub[(int)d], 1u);
                }
              }
              __syncthreads();
              // how many leading sub-buckets fit the buffer together (every wave computes it: lane = 4 of them)
              uint32_t v4[4], sum = 0;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v4[k] = sub[lane * 4 + k];
                sum += v4[k];
              }
              uint32_t inc = sum;
#pragma unroll
              for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d);
                if (lane >= d) inc += t;
              }
              uint32_t run = inc - sum;
              int nfit = 0;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                run += v4[k];
                nfit += __popcll(__ballot(run <= (uint32_t)round_cap));
              }
              __syncthreads();  // (the counts are zeroed again if the range narrows)
              if (nfit == 0 && sh > 0) {  // the first sub-bucket alone is too large: look inside it
                sh = sh >= 8 ? sh - 8 : 0;
                continue;
              }
              // (nfit == 0 at sh == 0 would be > kLazyBlockMax EQUAL keys; keys are unique.  The gather
              // below drops what does not fit, so even that cannot overrun the buffer.)
              hi = kmin + ((unsigned long long)(nfit > 0 ? nfit : 1) << sh);
              break;
            }
          }
          // gather the keys in [r_lo, hi) into the LDS buffer (any order)
          if (tid == 0) s_cnt = 0;
          __syncthreads();
          for (int i0 = 0; i0 < r_m; i0 += kLazyThreads) {  // block-uniform
            const int i = i0 + tid;
            const unsigned long long k = i < r_m ? keys[r_base + i] : 0ull;
            const bool pred = i < r_m && k >= r_lo && k < hi;
            const unsigned long long m = __ballot(pred);
            if (m != 0ull) {
              int base = 0;
              if (lane == 0) base = atomicAdd(&s_cnt, __popcll(m));
              base = __builtin_amdgcn_readfirstlane(base);
              const int at = base + __popcll(m & ((1ull << lane) - 1ull));
              if (pred && at < kLazyBlockMax) surv[at] = k;
            }
          }
          __syncthreads();
          const int got = s_cnt;
          cnt = got < kLazyBlockMax ? got : kLazyBlockMax;
          {
            const int q = (cnt + 3) >> 2;
            const int lo = wave * q, hi_i = min(lo + q, cnt);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int i = lo + 64 * t + lane;
              kcur[t] = i < hi_i ? surv[i] : 0ull;
            }
          }
          __syncthreads();  // the keys are in registers: the survivor lists may overwrite the buffer
          r_rem -= got;
          r_lo = hi;
          if (r_rem <= 0 || got == 0) refining = false;
        }
        // ---- prefilter: this wave's quarter of the round, against the points accepted before it
        int my_cnt = 0;
        {
          const int q = (cnt + 3) >> 2;
          const int lo = wave * q, hi_i = min(lo + q, cnt);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c0 = lo + 64 * t;
            if (c0 >= hi_i) break;  // wave-uniform
            const uint64_t kthis = kcur[t];
            const bool valid = c0 + lane < hi_i;
            // (nothing accepted yet: every candidate survives)
            const bool pass = kept > 0 ? prefilter_key(kthis, valid) : valid;
            const unsigned long long m = __ballot(pass);
            if (pass) my_surv[my_cnt + __popcll(m & ((1ull << lane) - 1ull))] = kthis;
            my_cnt += __popcll(m);
          }
        }
        // the next chunk's keys: requested now, used after this chunk's ordered windows
        if (!refining && j < nsched) {
          const int m2 = (int)s_sched[j] - cpos;
          if (m2 <= round_cap) {
            load_chunk(cpos, m2, knxt);
            have_next = true;
          }
        }
        int* racc = reinterpret_cast<int*>(part);  // rank accumulators of the survivor sort (the windows are not running)
        racc[tid] = 0;
        if (lane == 0) s_surv[par][wave] = my_cnt;
        lds_barrier();
        const int c0n = s_surv[par][0], c1n = c0n + s_surv[par][1], c2n = c1n + s_surv[par][2],
                  total = c2n + s_surv[par][3];
        par ^= 1;
#ifdef OKVFE_LAB
        n_surv += total;
        ++n_rounds;
#endif
        // ---- the survivors in key order: rank = number of smaller keys (keys are unique), in place
        auto list_key = [&](int g) {
          int li = 0, lo = g;
          if (g >= c2n) { li = 3; lo = g - c2n; }
          else if (g >= c1n) { li = 2; lo = g - c1n; }
          else if (g >= c0n) { li = 1; lo = g - c0n; }
          return surv[li * kLazySurvPerWave + lo];
        };
        auto list_len = [&](int w2) { return w2 == 0 ? c0n : (w2 == 1 ? c1n - c0n : (w2 == 2 ? c2n - c1n : total - c2n)); };
        if (total <= kLazyThreads / 2) {
          // the usual case, a hundred survivors or fewer: 2 or 4 threads per key, each counting in one or two
          // of the four lists (whole waves: no divergence), partial ranks added up in LDS
          const int per = total <= 64 ? 1 : 2;                  // lists per thread
          const int g = total <= 64 ? lane : (tid & 127);       // this thread's key
          const int w0 = total <= 64 ? wave : 2 * (wave >> 1);  // its first list
          const uint64_t mykey = g < total ? list_key(g) : ~0ull;
          int r = 0;
          for (int w2 = w0; w2 < w0 + per; ++w2) {  // wave-uniform
            const uint64_t* lst = surv + w2 * kLazySurvPerWave;
            const int c = list_len(w2);
#pragma unroll 8
            for (int jj = 0; jj < c; ++jj) r += lst[jj] < mykey ? 1 : 0;
          }
          if (g < total && r != 0) atomicAdd(&racc[g], r);
          lds_barrier();  // every list has been read, every partial rank added
          if (w0 == 0 && g < total) surv[racc[g]] = mykey;
          lds_barrier();
        } else {
          const int nu = (total + kLazyThreads - 1) / kLazyThreads;  // keys per thread, <= 4 (block-uniform)
          uint64_t mykey[4];
          int rank[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int g = tid + u * kLazyThreads;
            mykey[u] = (u < nu && g < total) ? list_key(g) : ~0ull;
            rank[u] = 0;
          }
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            const uint64_t* lst = surv + w2 * kLazySurvPerWave;
            const int c = list_len(w2);
            if (nu <= 1) {
#pragma unroll 8
              for (int jj = 0; jj < c; ++jj) rank[0] += lst[jj] < mykey[0] ? 1 : 0;
            } else {
              for (int jj = 0; jj < c; ++jj) {
                const uint64_t k2 = lst[jj];
#pragma unroll
                for (int u = 0; u < 4; ++u) rank[u] += k2 < mykey[u] ? 1 : 0;
              }
            }
          }
          lds_barrier();  // every list has been read
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (u < nu && tid + u * kLazyThreads < total) surv[rank[u]] = mykey[u];
          lds_barrier();
        }
#ifdef OKVFE_LAB
        t_pref += __builtin_amdgcn_s_memrealtime() - t_mark;
#endif
        windows(total, [&](int g) { return surv[g]; });
        if (kept >= limit) break;  // block-uniform
      }
    }
  }
  // ---- K4: sub-pixel refinement and keypoint emission (all four waves)
  __syncthreads();
#ifdef OKVFE_LAB
  t_blocks = __builtin_amdgcn_s_memrealtime();
#endif
  for (int i = tid; i < kept; i += kLazyThreads) {
    okvfe_keypoint kp = out[i];
    const int u = (int)kp.x, v = (int)kp.y;
    int32_t patch[9];
    if (images) {  // map-free call: the nine scores from the pixels (block-uniform)
      harris_scores_3x3(images + (size_t)img * w * h, w, h, u, v, patch);
    } else {
      const int32_t* sc = scores + (size_t)img * layout.pitch * h;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          patch[(dy + 1) * 3 + (dx + 1)] = sc[score_index(layout, u + dx, v + dy)];
    }
    float ddx, ddy;
    subpixel2d(patch, &ddx, &ddy);
    kp.x = (float)u + ddx;
    kp.y = (float)v + ddy;
    kp.response = (float)kp.class_id;
    kp.class_id = -1;
    out[i] = kp;
    // detection and description in one call: the extractor's per-keypoint preparation right here
    // (describe_setup_dev.h) instead of a launch of its own
    if (setup.pat) describe_setup_one(setup, w, h, img, (size_t)img * kp_cap + i, kp);
  }
  if (tid == 0) kp_count[img] = kept;
#ifdef OKVFE_LAB
  __syncthreads();
  if (tid == 0 && img == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    atomicAdd(&g_lazy_prof[0], 1ull);
    atomicAdd(&g_lazy_prof[1], t_init - t_start);
    atomicAdd(&g_lazy_prof[2], t_blocks - t_init);
    atomicAdd(&g_lazy_prof[3], t_end - t_blocks);
    atomicAdd(&g_lazy_prof[4], t_end - t_start);
    atomicAdd(&g_lazy_prof[5], t_pref);
    atomicAdd(&g_lazy_prof[6], (unsigned long long)n_surv);
    atomicAdd(&g_lazy_prof[7], (unsigned long long)n);
    atomicAdd(&g_lazy_prof[8], t_walk);
    atomicAdd(&g_lazy_prof[9], t_acc);
    atomicAdd(&g_lazy_prof[10], t_ins);
    atomicAdd(&g_lazy_prof[11], (unsigned long long)n_rounds);
    atomicAdd(&g_lazy_prof[12], t_p0 - t_start);
    atomicAdd(&g_lazy_prof[13], t_pa - t_p0);
    atomicAdd(&g_lazy_prof[14], t_pb - t_pa);
    atomicAdd(&g_lazy_prof[15], t_init - t_pb);
  }
#endif
}

}  // namespace


// ---- lazy-occupancy selection over LINKED LISTS (round 3): the form for FINE grids ----------------
// The array-bin kernel below spends 32 bytes of LDS per bin whatever the bin holds; a small
// uniformity radius on a large image (640x480 at radius 10: 64 x 49 bins for ~800 keypoints) would
// need 126 KB per image.  This kernel keeps 4 bytes per bin + 8 per keypoint slot (singly linked
// lists, heads swapped in with one LDS atomic; one dependent LDS round trip per point walked) and
// takes such configurations: same results, chosen by launch_select from the LDS each form needs.
constexpr int kListChunk = 128;
constexpr int kListLutBytes = 64 * 64 * 4;
__device__ __forceinline__ uint32_t lds_u32(uint32_t addr) {
  return *reinterpret_cast<__attribute__((address_space(3))) const uint32_t*>((uintptr_t)addr);
}
__device__ __forceinline__ float lds_f32(uint32_t addr) {
  return *reinterpret_cast<__attribute__((address_space(3))) const float*>((uintptr_t)addr);
}
// table | heads of the bordered bin grid | link, level (cap + 1 slots each) | record chunk | partial sums
__host__ __device__ inline size_t list_lds_bytes(int bins_x, int bins_y, int cap) {
  return (size_t)kListLutBytes + lazy_align16((size_t)(bins_x + 2) * (bins_y + 2) * 4) +
         2 * lazy_align16((size_t)(cap + 1) * 4) + (size_t)kListChunk * 16 + 4 * 64 * 4;
}

__global__ __launch_bounds__(kLazyThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void select_list_kernel(
    const int32_t* __restrict__ scores, ScoreLayout layout, int w, int h, int cand_cap,
    const int32_t* __restrict__ cand_count, const uint64_t* __restrict__ sort_ws, int ws_stride,
    float radius, int max_kpts, const float* __restrict__ lut, int bins_x, int bins_y, int cap,
    okvfe_keypoint* __restrict__ kps, int kp_cap, int32_t* __restrict__ kp_count, DescribeSetup setup,
    const uint8_t* __restrict__ images) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_kept;
  const int bpitch = bins_x + 2;  // bordered bin grid: the border bins stay empty, so no range checks
  const int nbins = bpitch * (bins_y + 2);
  const size_t cap4 = lazy_align16((size_t)(cap + 1) * 4);
  float* lut_s = reinterpret_cast<float*>(smem_raw);
  unsigned char* q0 = smem_raw + kListLutBytes;
  uint32_t* head = reinterpret_cast<uint32_t*>(q0);
  q0 += lazy_align16((size_t)nbins * 4);
  uint32_t* link = reinterpret_cast<uint32_t*>(q0);
  float* pnsc = reinterpret_cast<float*>(q0 + cap4);
  uint4* recs = reinterpret_cast<uint4*>(q0 + 2 * cap4);
  float* part = reinterpret_cast<float*>(q0 + 2 * cap4 + (size_t)kListChunk * 16);  // [4][64]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int img = blockIdx.x;
  int n = cand_count[img];
  // overflowed candidate list: WHICH maxima were dropped depends on the order of the atomics, so
  // the image keeps no keypoints at all (deterministic) and okvfe_check_capacity reports it
  n = n > cand_cap ? 0 : n;
  const uint64_t* keys = sort_ws + (size_t)img * ws_stride;
  okvfe_keypoint* out = kps + (size_t)img * kp_cap;
  int kept = 0;
  if (n > 0) {  // block-uniform
    // weight(dx, dy) at [(dy + 32) << 6 | (dx + 32)], zero outside the 31 x 31 stamp
    for (int i = tid; i < 64 * 64; i += kLazyThreads) {
      const int dx = (i & 63) - 32, dy = (i >> 6) - 32;
      const bool in = dx >= -15 && dx <= 15 && dy >= -15 && dy <= 15;
      lut_s[i] = in ? lut[(dy + 15) * 31 + (dx + 15)] : 0.0f;
    }
    const uint32_t link_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)link;
    const uint32_t end_addr = link_addr + 4u * (uint32_t)cap;
    const uint32_t nsc_delta = (uint32_t)cap4;
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lut_s;
    for (int i = tid; i < nbins; i += kLazyThreads) head[i] = end_addr;
    if (tid == 0) {
      pnsc[cap] = 0.0f;
      link[cap] = (end_addr >> 2) << 16;
      s_kept = 0;
    }
    const float scaling = (float)(15.0 / (double)radius);
    const float max_score = (float)(0x7FFFFFFF - (int32_t)(keys[0] >> 32));
    // candidates [base, base + kListChunk): keys are requested one chunk ahead (threads 64..191, one
    // key each), converted into records {cell, level, pixel, score} when their chunk is next
    uint64_t kq = 0;
    const bool converter = tid >= 64 && tid < 64 + kListChunk;
    auto request = [&](int base) {
      const int i = base + tid - 64;
      kq = (converter && i < n) ? keys[i] : 0ull;
    };
    auto convert = [&]() {
      if (converter) {
        const uint64_t k = kq;
        const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
        const int y = (int)((k >> 16) & 0xFFFF), x = (int)(k & 0xFFFF);
        const float fy = (float)y * scaling;
        const float fx = (float)x * scaling;
        const int cy = (int)(fy + 16.0f);
        const int cx = (int)(fx + 16.0f);
        const float q = (float)score / max_score;
        const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
        recs[tid - 64] = make_uint4(((uint32_t)cy << 16) | (uint32_t)cx, __float_as_uint(nsc1),
                                    (uint32_t)(k & 0xFFFFFFFFu), (uint32_t)score);
      }
    };
    request(0);
    convert();
    request(kListChunk);
    // this wave's share of the nine bin lists: chains c = first, first + 4 (, first + 8)
    const int first = (wave + 1) & 3;  // wave 0 (which also runs the acceptance) and waves 1, 2 walk two lists
    const int nch = first == 0 ? 3 : 2;
    int hoff[3];       // offset of the chain's bin from the candidate's bin in the bordered grid
    uint32_t boff4[3];  // table offset of the chain's bin: 4 * ((16 oy) << 6 + 16 ox)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = first + 4 * k < 9 ? first + 4 * k : first;
      const int ox = (c % 3) - 1, oy = (c / 3) - 1;
      hoff[k] = oy * bpitch + ox;
      boff4[k] = (uint32_t)(((16 * oy) << 6) + 16 * ox) << 2;
    }
    const int limit = min(min(max_kpts, kp_cap), cap);
    __syncthreads();
    for (int pos = 0; pos < n; pos += 64) {  // block-uniform
      const int idx = pos + lane;
      const bool valid = idx < n;
      uint4 rec = recs[idx & (kListChunk - 1)];
      if (!valid) rec.x = 0u;  // a cell inside the grid; the lane never passes
      const int cx = (int)(rec.x & 0xFFFF), cy = (int)(rec.x >> 16);
      const float level = __uint_as_float(rec.y);
      const int bx = cx >> 4, by = cy >> 4;
      const int bin = (by + 1) * bpitch + (bx + 1);
      // The weight table is indexed by (dy + 32) << 6 | (dx + 32): for the bin at offset (ox, oy)
      // that is cc - (link & 0xFFC) / 4 with cc = (cy & 15 + 32 - 16 oy) << 6 | (cx & 15 + 32 - 16 ox)
      // -- both fields stay in [1, 63], so there is no borrow between them.
      const uint32_t lcode = ((uint32_t)(cy & 15) << 6) | (uint32_t)(cx & 15);
      const uint32_t cc4 = lut_addr + ((lcode + ((32u << 6) | 32u)) << 2);
      uint32_t hp[3], lk[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) hp[k] = (k < nch) ? head[bin + hoff[k]] : end_addr;
#pragma unroll
      for (int k = 0; k < 3; ++k) lk[k] = lds_u32(hp[k]);
      float occf = 0.0f;  // sums of small integers: exact in float
      while (true) {
        const bool alive = hp[0] != end_addr || hp[1] != end_addr || hp[2] != end_addr;
        if (!__any(alive)) break;
        float wgt[3], lev[3];
        uint32_t nlk[3], nhp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          wgt[k] = lds_f32(cc4 - boff4[k] - (lk[k] & 0xFFCu));
          lev[k] = lds_f32(hp[k] + nsc_delta);
          nhp[k] = lk[k] >> 14;
          nlk[k] = lds_u32(nhp[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          occf += ceilf(wgt[k] * lev[k]);  // 0 for finished walks (level 0) and points out of reach (weight 0)
          hp[k] = nhp[k];
          lk[k] = nlk[k];
        }
      }
      part[wave * 64 + lane] = occf;
      lds_barrier();  // (LDS traffic only: see select_lazy_kernel)
      if (wave == 0) {
        int occ = (int)(part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane]);
        bool pass = valid && !(level < (float)(occ > 255 ? 255 : occ));
        const float nsc = (float)(0.99 * (double)level);
        unsigned long long rem = __ballot(pass), accm = 0;
        int nacc = 0;
        if (rem != 0) {
          const int room = limit - kept;
          if (__popcll(rem) <= room) {
            // Acceptance in ROUNDS, as in select_lazy_kernel (round 4): a passing lane is decided once every
            // EARLIER passing lane within reach of it is; all such lanes are decided together.  nb = the
            // earlier passing lanes within the stamp's square around this lane, found through 128 lane masks
            // hashed by bin in the `part` array (idle in this phase) and checked exactly.
            unsigned long long* wm = reinterpret_cast<unsigned long long*>(part);
            wm[lane] = 0ull;
            wm[64 + lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            if (pass) atomicOr(&wm[bin & 127], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long lower_ = (1ull << lane) - 1ull;
            unsigned long long cm = 0ull;
#pragma unroll
            for (int b = 0; b < 9; ++b) cm |= wm[(bin + ((b / 3) - 1) * bpitch + (b % 3) - 1) & 127];
            cm &= rem & lower_;
            if (!pass) cm = 0ull;
            unsigned long long nb = 0ull;
            while (__any(cm != 0ull)) {
              const int el = cm != 0ull ? (int)__ffsll((long long)cm) - 1 : lane;
              const uint32_t exy = (uint32_t)__builtin_amdgcn_ds_bpermute(el << 2, (int)rec.x);
              if (cm != 0ull) {
                cm &= cm - 1ull;
                const uint32_t ddx = (uint32_t)(cx - (int)(exy & 0xFFFF) + 15);
                const uint32_t ddy = (uint32_t)(cy - (int)(exy >> 16) + 15);
                nb |= (ddx <= 30u && ddy <= 30u) ? (1ull << el) : 0ull;
              }
            }
            while (rem != 0ull) {
              const bool in = ((rem >> lane) & 1ull) != 0ull;
              const bool ready = in && (nb & rem) == 0ull;  // (the lowest lane of rem always is)
              const unsigned long long rdy = __ballot(ready);
              const unsigned long long accn = __ballot(ready && pass);
              accm |= accn;
              rem &= ~rdy;
              unsigned long long m = (in && !ready) ? (nb & accn) : 0ull;  // undecided lanes a new point reaches
              while (__any(m != 0ull)) {
                const int jl = m != 0ull ? (int)__ffsll((long long)m) - 1 : lane;
                const uint32_t jxy = (uint32_t)__builtin_amdgcn_ds_bpermute(jl << 2, (int)rec.x);
                const float jnsc = __int_as_float(__builtin_amdgcn_ds_bpermute(jl << 2, __float_as_int(nsc)));
                if (m != 0ull) {
                  m &= m - 1ull;
                  const int dx = cx - (int)(jxy & 0xFFFF), dy = cy - (int)(jxy >> 16);
                  occ += (int)ceilf(lut_s[((dy + 32) << 6) | (dx + 32)] * jnsc);
                }
              }
              if (in && !ready) pass = !(level < (float)(occ > 255 ? 255 : occ));
              rem &= __ballot(pass);  // lanes that fail now fail at their turn as well: decided
            }
            nacc = __popcll(accm);
          } else {
            // the cap falls inside this window: one candidate at a time, in order
            while (rem != 0 && kept + nacc < limit) {
              const int f = (int)__ffsll((long long)rem) - 1;
              accm |= 1ull << f;
              ++nacc;
              const uint32_t wxy = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, f);
              const float wnsc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nsc), f));
              const int dx = cx - (int)(wxy & 0xFFFF), dy = cy - (int)(wxy >> 16);
              const bool near = pass && lane > f && (dx < 0 ? -dx : dx) <= 15 && (dy < 0 ? -dy : dy) <= 15;
              if (__any(near)) {  // the new point's stamp reaches later passing candidates of this window
                if (near) {
                  occ += (int)ceilf(lut_s[((dy + 32) << 6) | (dx + 32)] * wnsc);
                  pass = !(level < (float)(occ > 255 ? 255 : occ));
                }
                rem &= __ballot(pass) & ~(((2ull << f) - 1ull));
              } else {
                rem &= rem - 1;
              }
            }
          }
          if ((accm >> lane) & 1) {
            const int slot = kept + __popcll(accm & ((1ull << lane) - 1ull));
            const uint32_t addr = link_addr + 4u * (uint32_t)slot;
            const uint32_t prev = atomicExch(&head[bin], addr);
            link[slot] = (lcode << 2) | ((prev >> 2) << 16);
            pnsc[slot] = nsc;
            // pixel position and score wait in the output record for the sub-pixel pass
            okvfe_keypoint kp;
            kp.x = (float)(int)(rec.z & 0xFFFF);
            kp.y = (float)(int)(rec.z >> 16);
            kp.size = 12.0f;
            kp.angle = -1.0f;
            kp.response = (float)(int32_t)rec.w;
            kp.octave = 0;
            kp.class_id = (int32_t)rec.w;  // the exact score (response is its float image)
            out[slot] = kp;
          }
          kept += nacc;
          if (lane == 0) s_kept = kept;
        }
      } else if (((pos + 64) & (kListChunk - 1)) == 0) {
        // the next window starts a new chunk: every wave holds this window's records in registers
        convert();
        request(pos + 64 + kListChunk);
      }
      lds_barrier();  // (the keypoint records written above are read only after the loop, behind a full barrier)
      kept = s_kept;
      if (kept >= limit) break;  // block-uniform
    }
  }
  // ---- K4: sub-pixel refinement and keypoint emission (all four waves)
  __syncthreads();
  for (int i = tid; i < kept; i += kLazyThreads) {
    okvfe_keypoint kp = out[i];
    const int u = (int)kp.x, v = (int)kp.y;
    int32_t patch[9];
    if (images) {  // map-free call: the nine scores from the pixels (block-uniform)
      harris_scores_3x3(images + (size_t)img * w * h, w, h, u, v, patch);
    } else {
      const int32_t* sc = scores + (size_t)img * layout.pitch * h;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          patch[(dy + 1) * 3 + (dx + 1)] = sc[score_index(layout, u + dx, v + dy)];
    }
    float ddx, ddy;
    subpixel2d(patch, &ddx, &ddy);
    kp.x = (float)u + ddx;
    kp.y = (float)v + ddy;
    kp.response = (float)kp.class_id;
    kp.class_id = -1;
    out[i] = kp;
    // detection and description in one call: the extractor's per-keypoint preparation right here
    // (describe_setup_dev.h) instead of a launch of its own
    if (setup.pat) describe_setup_one(setup, w, h, img, (size_t)img * kp_cap + i, kp);
  }
  if (tid == 0) kp_count[img] = kept;
}

namespace {
struct LazyPlan {
  bool list, array;  // which lazy-occupancy kernel runs (neither: the grid kernels)
  int bins_x, bins_y, cap;
  size_t lds, lds_list;
};
constexpr size_t kLazyMaxLds = 159 * 1024;  // the kernels also have a few bytes of static LDS
LazyPlan lazy_plan(float radius, int max_kpts, int kp_cap, const uint8_t* occupancy, size_t occ_image_bytes,
                   int occ_rows, int occ_cols) {
  LazyPlan p{};
  static const bool legacy = lab_env("OKVFE_LEGACY_SELECT") != nullptr;  // A/B knob
  static const bool grid = lab_env("OKVFE_SELECT_GRID") != nullptr;      // A/B knob: the occupancy-grid kernels
  if (!(radius > 0.0f) || legacy || grid || occ_cols > 65535 || occ_rows > 65535) return p;
  // lazy occupancy (no grid): any image size / radius whose bin heads and keypoint slots fit in LDS
  p.bins_x = (occ_cols + 15) >> 4;
  p.bins_y = (occ_rows + 15) >> 4;
  p.cap = max_kpts < kp_cap ? max_kpts : kp_cap;
  p.lds = lazy_lds_bytes(p.bins_x, p.bins_y);
  p.lds_list = list_lds_bytes(p.bins_x, p.bins_y, p.cap);
  // Array bins cost 32 B of LDS per bin, linked lists 4 B per bin + 8 B per keypoint slot: a fine
  // grid with few keypoints per bin (640x480 at radius 10: 3136 bins, ~800 keypoints) takes the list
  // form when that keeps at least one more image on a CU
  // (the occupancy workspace doubles as the spill list of full bins: 8 bytes per point at worst)
  const bool array_ok = p.lds <= kLazyMaxLds && occupancy != nullptr && occ_image_bytes >= (size_t)p.cap * 8;
  const bool prefer_list = !array_ok || (p.lds > 48 * 1024 && kLazyMaxLds / p.lds_list > kLazyMaxLds / p.lds);
  p.list = prefer_list && p.lds_list <= kLazyMaxLds;
  p.array = !p.list && array_ok;
  return p;
}
// lab knob: the array-bin kernel on keys sorted by launch_sort (the round-3 / early round-4 form)
bool lazy_presorted() {
  static const bool v = lab_env("OKVFE_SELECT_PRESORTED") != nullptr;
  return v;
}
}  // namespace

// true: launch_select runs a kernel that can recompute the sub-pixel scores from the image (`images` != null):
// the score map need not exist for this configuration
bool select_recomputes_scores(float radius, int max_kpts, int kp_cap, const uint8_t* occupancy, size_t occ_image_bytes,
                              int occ_rows, int occ_cols) {
  const LazyPlan p = lazy_plan(radius, max_kpts, kp_cap, occupancy, occ_image_bytes, occ_rows, occ_cols);
  return p.array || p.list;
}
// true: launch_select orders the candidates itself for this configuration -- no launch_sort before it
bool select_sorts_candidates(float radius, int max_kpts, int kp_cap, const uint8_t* occupancy, size_t occ_image_bytes,
                             int occ_rows, int occ_cols) {
  return lazy_plan(radius, max_kpts, kp_cap, occupancy, occ_image_bytes, occ_rows, occ_cols).array && !lazy_presorted();
}

bool launch_select(const int32_t* score, ScoreLayout layout, int w, int h, int n_images, Candidate* cand,
                   int cand_cap, const int32_t* cand_count, float radius, int max_kpts,
                   const float* lut, uint8_t* occupancy, size_t occ_image_bytes, int occ_rows,
                   int occ_cols, okvfe_keypoint* kps, int kp_cap, int32_t* kp_count,
                   uint64_t* sort_ws, hipStream_t stream, const DescribeSetup* setup, const uint8_t* images) {
  if (n_images <= 0) return false;
  int ws_stride = 1;
  while (ws_stride < cand_cap) ws_stride <<= 1;
  const LazyPlan lp = lazy_plan(radius, max_kpts, kp_cap, occupancy, occ_image_bytes, occ_rows, occ_cols);
  if (lp.list) {
    static PerDeviceOnce attr_once_l;
    attr_once_l.run([] {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(select_list_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLazyMaxLds) != hipSuccess)
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL(select_list_kernel, dim3(n_images), dim3(kLazyThreads), lp.lds_list, stream, score, layout, w, h,
                       cand_cap, cand_count, sort_ws, ws_stride, radius, max_kpts, lut, lp.bins_x, lp.bins_y, lp.cap, kps,
                       kp_cap, kp_count, setup ? *setup : DescribeSetup{}, images);
    return setup != nullptr;
  }
  if (lp.array) {
    static const int bin_cap = [] {  // lab knob: 1..kLazyBinCap slots per bin (fewer = more spills)
      const char* e = lab_env("OKVFE_LAZY_BINCAP");
      const int v = e ? atoi(e) : kLazyBinCap;
      return v < 1 ? 1 : (v > kLazyBinCap ? kLazyBinCap : v);
    }();
    static const int round_cap = [] {  // lab knob: keys per round of the self-ordering kernel (smaller = more key-range splits)
      const char* e = lab_env("OKVFE_LAZY_ROUNDCAP");
      const int v = e ? atoi(e) : kLazyBlockMax;
      return v < 4 ? 4 : (v > kLazyBlockMax ? kLazyBlockMax : v);
    }();
    static PerDeviceOnce attr_once;
    attr_once.run([] {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(select_lazy_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLazyMaxLds) != hipSuccess)
        (void)hipGetLastError();  // launches above 64 KiB will then fail loudly on their own
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(select_lazy_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLazyMaxLds) != hipSuccess)
        (void)hipGetLastError();
    });
#define OKVFE_LAZY_LAUNCH(SORTS)                                                                                     \
  hipLaunchKernelGGL(select_lazy_kernel<SORTS>, dim3(n_images), dim3(kLazyThreads), lp.lds, stream, score, layout, w, \
                     h, cand, cand_cap, cand_count, sort_ws, ws_stride, radius, max_kpts, lut, lp.bins_x, lp.bins_y,  \
                     lp.cap, kps, kp_cap, kp_count, reinterpret_cast<uint2*>(occupancy), occ_image_bytes / 8, bin_cap, round_cap, \
                     setup ? *setup : DescribeSetup{}, images)
    if (lazy_presorted())
      OKVFE_LAZY_LAUNCH(false);
    else
      OKVFE_LAZY_LAUNCH(true);
#undef OKVFE_LAZY_LAUNCH
    return setup != nullptr;  // the extractor's setup ran with the emission
  }
  launch_select_grid(score, layout, w, h, n_images, cand, cand_cap, cand_count, radius, max_kpts, lut, occupancy,
                     occ_image_bytes, occ_rows, occ_cols, kps, kp_cap, kp_count, sort_ws, stream);
  return false;
}

#ifdef OKVFE_LAB
extern "C" int okvfe_lab_lazy_prof(unsigned long long out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lazy_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lazy_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

}  // namespace okvfe
