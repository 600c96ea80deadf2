// k_nms.hip -- K2: 8-neighbour non-max suppression + compaction of the Harris score map.
//
// Replaces HarrisScoreCalculator::Get2dMaxima of the brisk library (behind
// cv::FeatureDetector::detect, okvis_cv/include/okvis/implementation/Frame.hpp:152).
// A centre (2 <= x < w-2, 2 <= y < h-2) is a maximum when score >= absoluteThreshold and no
// neighbour is strictly greater; of a horizontal run of equal passing pixels every second one,
// starting with the leftmost, is kept (the raster scan skips the pixel after a hit).
//
// In the batch pipeline the NMS is FUSED into the score kernel (k_harris.hip,
// harris_kernel<TH, true>): a stand-alone pass has to read the whole score map back (4 B/px at
// the ~3.8 TB/s read-only kernels reach here), which costs as much as computing it.  This file
// keeps
//   nms_fixup_kernel    settles the candidates the fused kernel flagged (runs of equal maxima);
//   nms_kernel          the stand-alone fast path (w % 4 == 0; OKVFE_NO_FUSED_NMS, tiles of more
//                       than 32 rows): same strip mapping as K1 -- lane = one 16-byte column group
//                       (4 px), 62 valid lanes per wave, rolling window of three rows (fully
//                       unrolled), horizontal neighbours through DPP wave shifts, hits recorded
//                       as one bit mask per column, ONE slot reservation per wave after the loop;
//   nms_generic_kernel  any width.
// Roofline of the stand-alone kernels: HBM, 4 B/px read + 12 B per maximum written.
#include "okvfe_internal.h"

namespace okvfe {
namespace {

// score-map accessors: Dense = pitch w, Slotted = the fused kernel's layout (okvfe_internal.h)
struct DenseMap {
  const int32_t* __restrict__ s;
  int w;
  __device__ __forceinline__ int at(int x, int y) const { return s[(size_t)y * w + x]; }
};
struct LayoutMap {
  const int32_t* __restrict__ s;
  ScoreLayout L;
  __device__ __forceinline__ int at(int x, int y) const { return s[score_index(L, x, y)]; }
};

template <class Map>
__device__ __forceinline__ bool passes(const Map& m, int x, int y, int thr) {
  const int v = m.at(x, y);
  if (v < thr) return false;
  if (m.at(x + 1, y) > v || m.at(x - 1, y) > v) return false;
  if (m.at(x, y + 1) > v || m.at(x, y - 1) > v) return false;
  if (m.at(x + 1, y + 1) > v || m.at(x - 1, y + 1) > v || m.at(x + 1, y - 1) > v || m.at(x - 1, y - 1) > v)
    return false;
  return true;
}

// accepted(x) of the raster scan for a pixel that passes: parity of the run of passing pixels
// immediately to its left
template <class Map>
__device__ __forceinline__ bool accepted_slow(const Map& m, int w, int x, int y, int thr) {
  if (x < 2 || x >= w - 2 || !passes(m, x, y, thr)) return false;
  int run = 0;
  int xx = x - 1;
  while (xx >= 2 && passes(m, xx, y, thr)) {
    ++run;
    --xx;
  }
  return (run & 1) == 0;
}
__device__ __forceinline__ bool accepted_slow(const int32_t* __restrict__ s, int w, int x, int y, int thr) {
  return accepted_slow(DenseMap{s, w}, w, x, y, thr);
}

// The raster rule over ONE WAVE'S consecutive pixels, in registers.  Lane l holds the pass bits of its four pixels
// (bit i = pixel 4 l + i passes); accepted[p] = pass[p] & !accepted[p - 1] (the scan skips the pixel after a hit), which
// is the parity rule of accepted_slow.  A lane's nibble is a map carry-in -> carry-out (two bits); the maps are composed
// by a six-step wave prefix, then every lane evaluates its nibble with its carry-in.  `carry0` = accepted state of the
// pixel left of lane 0's first one (0 at the image's left edge).  AGAST / FAST score maps are small integers: runs of
// equal maxima are the rule there, and walking them pixel by pixel through global memory (accepted_slow) was most of
// the stand-alone NMS on those maps.
__device__ __forceinline__ uint32_t nibble_accept(uint32_t p, uint32_t c, uint32_t* carry_out) {
  const uint32_t a0 = p & ~c & 1u;
  const uint32_t a1 = (p >> 1) & ~a0 & 1u;
  const uint32_t a2 = (p >> 2) & ~a1 & 1u;
  const uint32_t a3 = (p >> 3) & ~a2 & 1u;
  *carry_out = a3;
  return a0 | (a1 << 1) | (a2 << 2) | (a3 << 3);
}
__device__ __forceinline__ uint32_t raster_accept_wave(uint32_t p, int lane, uint32_t carry0) {
  uint32_t t0, t1;
  (void)nibble_accept(p, 0u, &t0);
  (void)nibble_accept(p, 1u, &t1);
  uint32_t inc = t0 | (t1 << 1);  // this lane's map; after the scan: the map of lanes 0 .. l together
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const uint32_t prev = (uint32_t)__shfl_up((int)inc, dd);  // lanes l - 2 dd + 1 .. l - dd, applied first
    if (lane >= dd) inc = ((inc >> (prev & 1u)) & 1u) | (((inc >> ((prev >> 1) & 1u)) & 1u) << 1);
  }
  const uint32_t before = (uint32_t)__shfl_up((int)inc, 1);
  const uint32_t cin = lane == 0 ? carry0 : ((before >> carry0) & 1u);
  uint32_t unused;
  return nibble_accept(p, cin, &unused);
}

// ---- generic kernel (any width): one lane = 4 pixels of one row ---------------------------------
// A block = 64 lanes x 4 waves walks kGenIters groups of four rows (wave = row), keeps the accepted pixels as one nibble
// per group in a register, and reserves its slots in the image's list with ONE atomic: an image's rows run on all eight
// XCDs, so a returning atomic on its counter is a round trip to memory -- one per ROW (round 5: 156 per 250 x 160 layer)
// made this kernel 0.24 ms on a layer whose arithmetic takes 0.01 ms.
constexpr int kGenIters = 8;
__global__ __launch_bounds__(256) void nms_generic_kernel(const int32_t* __restrict__ scores, int w,
                                                          int h, int thr,
                                                          Candidate* __restrict__ cand,
                                                          int cand_cap,
                                                          int32_t* __restrict__ cand_count) {
  __shared__ int wtot[4];
  __shared__ int base_s;
  const int img = blockIdx.z;
  const int32_t* s = scores + (size_t)img * w * h;
  const int lane = threadIdx.x, wave = threadIdx.y;
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  uint32_t bits = 0u;  // nibble `it` = accepted pixels of row (blockIdx.y * kGenIters + it) * 4 + wave
  int cnt = 0;
#pragma unroll 1
  for (int it = 0; it < kGenIters; ++it) {
    const int y = (blockIdx.y * kGenIters + it) * 4 + wave;
    bool acc[4] = {false, false, false, false};
    if (y >= 2 && y < h - 2 && x0 < w) {
      // the 3 x 6 window of this lane's four pixels, loaded up front (clamped columns only ever stand in for the
      // neighbours of pixels that cannot be maxima): one memory round trip instead of a dependent chain per pixel
      int v[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int32_t* row = s + (size_t)(y - 1 + r) * w;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          int x = x0 - 1 + c;
          x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
          v[r][c] = row[x];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        int nb = thr;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (r != 1 || c != 1) nb = max(nb, v[r][i + c]);
        acc[i] = x >= 2 && x < w - 2 && v[1][i + 1] >= nb;  // >= thr and no strictly greater neighbour = passes()
      }
      if (blockIdx.x != 0) {  // further right than 256 pixels the run to the left is walked through memory
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (acc[i]) acc[i] = accepted_slow(s, w, x0 + i, y, thr);
      }
    }
    uint32_t a = (acc[0] ? 1u : 0u) | (acc[1] ? 2u : 0u) | (acc[2] ? 4u : 0u) | (acc[3] ? 8u : 0u);
    if (blockIdx.x == 0) a = raster_accept_wave(a, lane, 0u);  // block-uniform: the leftmost 256 pixels, in registers
    bits |= a << (4 * it);
    cnt += __popc(a);
  }
  int incl = cnt;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int t = __shfl_up(incl, dd);
    if (lane >= dd) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  if (lane == 0 && wave == 0) {
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    base_s = total > 0 ? atomicAdd(&cand_count[img], total) : 0;
  }
  __syncthreads();
  int pos = base_s + incl - cnt;
  for (int w2 = 0; w2 < wave; ++w2) pos += wtot[w2];
  if (cnt == 0) return;
#ifdef OKVFE_NMS_TIMING_NOWRITE  // (timing experiment only: wrong results)
  return;
#endif
  Candidate* out = cand + (size_t)img * cand_cap;
#pragma unroll 1
  for (int it = 0; it < kGenIters; ++it) {
    const int y = (blockIdx.y * kGenIters + it) * 4 + wave;
    uint32_t a = (bits >> (4 * it)) & 15u;
    while (a) {
      const int i = __ffs((int)a) - 1;
      a &= a - 1u;
      Candidate c;
      c.x = x0 + i;
      c.y = y;
      c.score = s[(size_t)y * w + x0 + i];
      if (pos < cand_cap) out[pos] = c;
      ++pos;
    }
  }
}

// ---- fast kernel --------------------------------------------------------------------------------
constexpr int kStripLanes = 62;
constexpr int kRows = 32;   // centre rows per wave
constexpr int kWaves = 4;

__device__ __forceinline__ int from_left(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_right(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ int max3(int a, int b, int c) { return max(max(a, b), c); }

// One lane = 4 columns (one int4 per row), 62 owned lanes per strip + 2 halo lanes, kRows centre
// rows per wave with the 3-row window rolled through registers (fully unrolled: no moves, row
// numbers are immediates).  Inside the row loop a lane only records its hits as bits (one 32-bit
// mask per column, bit = row): no ballots, no atomics, almost no scalar work.  After the loop the
// wave counts its hits, takes ONE slot range from the image's counter and every lane writes its
// own candidates (the scores are re-read, they are still in L2).
__global__ __launch_bounds__(64 * kWaves) void nms_kernel(const int32_t* __restrict__ scores, int w,
                                                          int h, int thr,
                                                          Candidate* __restrict__ cand,
                                                          int cand_cap,
                                                          int32_t* __restrict__ cand_count,
                                                          int strips, int ytiles, int n_images) {
  int img, tile;
  xcd_tile(strips * ytiles, n_images, &img, &tile);
  const int ytile = tile / strips;
  const int strip = tile - ytile * strips;
  const int32_t* s = scores + (size_t)img * w * h;
  const int lane = threadIdx.x;
  const int nd = w >> 2;  // 16-byte groups per row
  const int d = strip * kStripLanes + lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ys = (ytile * kWaves + wave) * kRows;  // first centre row of this wave
  if (ys >= h) return;                             // wave-uniform; no block barriers below
  const bool last_strip = strip * kStripLanes + 64 >= nd;
  const bool own = d < nd && (strip == 0 || lane >= 1) && (last_strip || lane <= kStripLanes);
  const int dcl = d < nd ? d : nd - 1;
  const int x0 = dcl * 4;
  const int ye = ys + kRows < h ? ys + kRows : h;
  const int4* rows = reinterpret_cast<const int4*>(s) + dcl;
  auto load_row = [&](int row) -> int4 {
    row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
    return rows[(size_t)row * (size_t)nd];
  };
  // horizontal 3-max of a row at this lane's 4 columns
  auto hmax = [&](const int4& r, int hm[4], int& left, int& right) {
    left = from_left(r.w);
    right = from_right(r.x);
    hm[0] = max3(left, r.x, r.y);
    hm[1] = max3(r.x, r.y, r.z);
    hm[2] = max3(r.y, r.z, r.w);
    hm[3] = max3(r.z, r.w, right);
  };
  constexpr int kAhead = 3;  // rows in flight beyond the window
  int4 r[kRows + 2 + kAhead];  // fully unrolled below: every index is a compile-time constant
  int hm[kRows + 2][4], lf[kRows + 2], rg[kRows + 2];
#pragma unroll
  for (int u = 0; u < 2 + kAhead; ++u) r[u] = load_row(ys - 1 + u);
  hmax(r[0], hm[0], lf[0], rg[0]);
  hmax(r[1], hm[1], lf[1], rg[1]);
  uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < kRows; ++k) {  // centre row y = ys + k: window rows k, k+1, k+2 of r[]
    const int y = ys + k;
    if (y >= ye) break;  // wave-uniform
    r[k + 2 + kAhead] = load_row(y + 1 + kAhead);
    hmax(r[k + 2], hm[k + 2], lf[k + 2], rg[k + 2]);
    if (y >= 2 && y < h - 2) {  // wave-uniform
      const int4 rc = r[k + 1];
      const int c[4] = {rc.x, rc.y, rc.z, rc.w};
      const int lft[4] = {lf[k + 1], rc.x, rc.y, rc.z};
      const int rgt[4] = {rc.y, rc.z, rc.w, rg[k + 1]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // >= thr and no strictly greater neighbour  <=>  c >= max(all 8 neighbours, thr)
        const int nb = max3(max3(hm[k][i], hm[k + 2][i], lft[i]), rgt[i], thr);
        m[i] = c[i] >= nb ? (m[i] | (1u << k)) : m[i];
      }
    }
  }
  // columns 0, 1, w-2, w-1 are never maxima
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (x0 + i < 2 || x0 + i >= w - 2) m[i] = 0u;
  // The raster scan of the reference skips the pixel after a hit, which only matters where two
  // horizontally adjacent pixels both pass (equal scores): re-derive those rows exactly.
  const uint32_t adj = (m[0] & m[1]) | (m[1] & m[2]) | (m[2] & m[3]) |
                       (m[3] & (uint32_t)from_right((int)m[0]));
  if (__builtin_expect(__any(adj != 0u), 0)) {
    uint32_t rows_adj = adj;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) rows_adj |= (uint32_t)__shfl_xor((int)rows_adj, dd);
    while (rows_adj) {
      const int k = __ffs((int)rows_adj) - 1;
      rows_adj &= rows_adj - 1;
      // the row's pass bits of this wave's 256 pixels (lanes past the row end repeat the last dword: no pixels)
      uint32_t p = ((m[0] >> k) & 1u) | (((m[1] >> k) & 1u) << 1) | (((m[2] >> k) & 1u) << 2) | (((m[3] >> k) & 1u) << 3);
      if (d >= nd) p = 0u;
      // a strip that does not start at the image's edge knows the state left of its halo lane only if that lane holds
      // a pixel that does not pass (the run then starts inside it); four passing halo pixels: walk the runs in memory
      const bool ambiguous = strip > 0 && (uint32_t)__shfl((int)p, 0) == 15u;  // wave-uniform
      if (!ambiguous) {
        const uint32_t a = raster_accept_wave(p, lane, 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (m[i] & ~(1u << k)) | (((a >> i) & 1u) << k);
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((m[i] >> k) & 1u)
          if (!accepted_slow(s, w, x0 + i, ys + k, thr)) m[i] &= ~(1u << k);
    }
  }
  if (!own) m[0] = m[1] = m[2] = m[3] = 0u;  // halo lanes only fed the neighbours
  const int cnt = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
  if (!__any(cnt != 0)) return;
  int incl = cnt;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int t = __shfl_up(incl, dd);
    if (lane >= dd) incl += t;
  }
  const int total = __shfl(incl, 63);
  int base = 0;
  if (lane == 0) base = atomicAdd(&cand_count[img], total);
  int pos = __shfl(base, 0) + incl - cnt;
  Candidate* out = cand + (size_t)img * cand_cap;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t mm = m[i];
    while (mm) {
      const int k = __ffs((int)mm) - 1;
      mm &= mm - 1;
      Candidate cd;
      cd.x = x0 + i;
      cd.y = ys + k;
      cd.score = s[(size_t)(ys + k) * w + x0 + i];
      if (pos < cand_cap) out[pos] = cd;
      ++pos;
    }
  }
}

// Settles the candidates the fused score+NMS kernel flagged (runs of horizontally adjacent equal
// maxima): with the score map complete, accepted_slow applies the raster-scan rule exactly; the
// rejected ones are removed from the (unordered) list.  One block per image, idle unless flagged.
// Pass 1 looks at every record once (strided, no barriers), clears the flag of the accepted ones
// and collects the indices of the rejected ones; a few rejections (the common case on natural
// content: a handful of tied pairs per image) are then removed by moving records from the tail
// into the holes -- the list is unordered -- instead of compacting the whole list chunk by chunk
// (0.09 ms per 1024 TUM-VI images, where nearly every image has one flagged row).
constexpr int kFixupHoles = 128;
constexpr int kFixupThreads = 256;  // (eight workgroups per CU: the few images with flagged records overlap)
constexpr int kFixupBatch = 8;
// MAPFREE (round 4): the score map was not written.  A flagged record is kept iff the run of hits to its
// left has even length, and every hit of such a run is a flagged record itself (the fused kernel flags a
// whole row of a wave as soon as two of its hits touch, halo lanes included), so the record set answers
// "is (x - k, y) a hit?": up to 256 flagged records are looked up in a list in LDS, more of them (plateaus,
// checkerboards) in a bitmap of the image built in the idle score-map buffer.
constexpr int kFixupSmall = 256;
template <bool MAPFREE>
__global__ __launch_bounds__(kFixupThreads) void nms_fixup_kernel(const int32_t* __restrict__ scores,
                                                        ScoreLayout layout, int w,
                                                        int h, int thr, Candidate* __restrict__ cand,
                                                        int cand_cap,
                                                        int32_t* __restrict__ cand_count,
                                                        const int32_t* __restrict__ fix_count,
                                                        const int32_t* __restrict__ fix_list,
                                                        uint32_t* bitmap_ws, size_t bitmap_stride) {
  __shared__ int wave_cnt[kFixupThreads / 64];
  __shared__ int s_base;
  __shared__ int n_rej;
  __shared__ int holes[kFixupHoles];
  __shared__ uint32_t fkeys[MAPFREE ? kFixupSmall : 1];
  const int img = blockIdx.x;
  if (fix_count[img] == 0) return;
  const LayoutMap s{scores + (size_t)img * layout.pitch * h, layout};
  Candidate* c = cand + (size_t)img * cand_cap;
  const int total = cand_count[img];
  const int n = total < cand_cap ? total : cand_cap;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    s_base = 0;
    n_rej = 0;
  }
  __syncthreads();
  const int n_flagged = fix_count[img];
  auto keep_or_hole = [&](int i, int y, bool accepted) {
    if (accepted) {
      c[i].y = y & ~kCandidateFixupFlag;
    } else {  // stays flagged = to be removed
      const int k = atomicAdd(&n_rej, 1);
      if (k < kFixupHoles) holes[k] = i;
    }
  };
  if constexpr (MAPFREE) {
    const bool listed = fix_list && n_flagged <= kFixListCap;
    if (listed && n_flagged <= kFixupSmall) {
      int i = -1, y = 0;
      uint32_t key = 0xFFFFFFFFu;
      if (tid < n_flagged) {
        i = fix_list[(size_t)img * kFixListCap + tid];
        if (i < n) {
          y = c[i].y;
          key = ((uint32_t)(y & ~kCandidateFixupFlag) << 16) | (uint32_t)c[i].x;
        } else {
          i = -1;
        }
      }
      fkeys[tid] = key;
      __syncthreads();
      if (i >= 0) {
        int run = 0;
        for (;;) {  // (x >= 2 for every hit: the subtraction never borrows from y before the run ends)
          const uint32_t target = key - (uint32_t)(run + 1);
          bool found = false;
          for (int j = 0; j < n_flagged; ++j) found = found || fkeys[j] == target;
          if (!found) break;
          ++run;
        }
        keep_or_hole(i, y, (run & 1) == 0);
      }
    } else {
      uint32_t* bm = bitmap_ws + (size_t)img * bitmap_stride;
      const int words = (w * h + 31) >> 5;
      for (int k = tid; k < words; k += kFixupThreads) bm[k] = 0u;
      __syncthreads();
      auto for_each_flagged = [&](auto f) {
        if (listed) {
          for (int t = tid; t < n_flagged; t += kFixupThreads) {
            const int i = fix_list[(size_t)img * kFixListCap + t];
            if (i < n) f(i, c[i].y);
          }
        } else {
          for (int i = tid; i < n; i += kFixupThreads) {
            const int y = c[i].y;
            if (y & kCandidateFixupFlag) f(i, y);
          }
        }
      };
      for_each_flagged([&](int i, int y) {
        const int p = (y & ~kCandidateFixupFlag) * w + c[i].x;
        atomicOr(&bm[p >> 5], 1u << (p & 31));
      });
      __threadfence_block();
      __syncthreads();
      for_each_flagged([&](int i, int y) {
        const int x = c[i].x, p0 = (y & ~kCandidateFixupFlag) * w;
        int run = 0;
        while (x - 1 - run >= 2) {
          const int p = p0 + x - 1 - run;
          if (!((__atomic_load_n(&bm[p >> 5], __ATOMIC_RELAXED) >> (p & 31)) & 1u)) break;
          ++run;
        }
        keep_or_hole(i, y, (run & 1) == 0);
      });
    }
  } else {
  auto settle = [&](int i, int y) {
    keep_or_hole(i, y, accepted_slow(s, w, c[i].x, y & ~kCandidateFixupFlag, thr));
  };
  if (fix_list && n_flagged <= kFixListCap) {
    // the fused kernel listed where its flagged records are: no pass over the whole list (reading
    // 13 k records per 1024 x 1024 image cost 0.04 ms per launch for a handful of ties)
    for (int t = tid; t < n_flagged; t += kFixupThreads) {
      const int i = fix_list[(size_t)img * kFixListCap + t];
      if (i < n) settle(i, c[i].y);
    }
  } else
  // (the y fields of a batch are requested before any of them is looked at: one memory round trip
  // per batch instead of one per record)
  for (int i0 = tid; i0 < n; i0 += kFixupThreads * kFixupBatch) {
    int ys[kFixupBatch];
#pragma unroll
    for (int u = 0; u < kFixupBatch; ++u) {
      const int i = i0 + u * kFixupThreads;
      ys[u] = i < n ? c[i].y : 0;
    }
#pragma unroll
    for (int u = 0; u < kFixupBatch; ++u) {
      const int i = i0 + u * kFixupThreads, y = ys[u];
      if (y & kCandidateFixupFlag) settle(i, y);
    }
  }
  }
  // (workgroup scope: an agent-scope fence writes the XCD's L2 back -- 0.1 ms per launch right after
  // the score kernel; thread 0 reads the flags back past the L1 instead)
  __threadfence_block();
  __syncthreads();
  const int rejected = n_rej;
  if (rejected == 0) return;
  if (rejected <= kFixupHoles) {
    // new length n - rejected: every rejected record below it is a hole, filled from the records at
    // or above it that are not rejected themselves (there are exactly as many)
    if (tid == 0) {
      const int new_n = n - rejected;
      int j = n - 1;
      for (int k = 0; k < rejected; ++k) {
        const int hole = holes[k];
        if (hole >= new_n) continue;
        int yj;
        while ((yj = __atomic_load_n(&c[j].y, __ATOMIC_RELAXED)) & kCandidateFixupFlag) --j;
        Candidate cd = c[j];
        cd.y = yj;
        c[hole] = cd;
        --j;
      }
      cand_count[img] = total > cand_cap ? total : new_n;  // an overflowing list stays marked
    }
    return;
  }
  // many rejections (plateaus, synthetic ties): in-place forward compaction in chunks of 256 -- a
  // chunk is read completely before its survivors are written at or before their old positions
  for (int i0 = 0; i0 < n; i0 += kFixupThreads) {
    const int i = i0 + tid;
    Candidate cd;
    bool keep = false;
    if (i < n) {
      cd = c[i];
      keep = !(cd.y & kCandidateFixupFlag);
    }
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wave_cnt[wv] = __popcll(b);
    __syncthreads();
    int pos = s_base;
    for (int k = 0; k < wv; ++k) pos += wave_cnt[k];
    pos += __popcll(b & ((1ull << lane) - 1ull));
    if (keep) c[pos] = cd;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < kFixupThreads / 64; ++k) t += wave_cnt[k];
      s_base += t;
    }
    __syncthreads();
  }
  if (tid == 0) cand_count[img] = total > cand_cap ? total : s_base;  // an overflowing list stays marked
}

}  // namespace

void launch_nms_fixup(const int32_t* score, ScoreLayout layout, int w, int h, int n_images,
                      int abs_threshold, Candidate* cand, int cand_cap, int32_t* cand_count,
                      const int32_t* fix_count, const int32_t* fix_list, hipStream_t stream, bool map_free,
                      int32_t* idle_score_buffer) {
  if (n_images <= 0) return;
  if (map_free) {
    // the score map was not written by this call: its buffer is free, and serves as bitmap scratch
    hipLaunchKernelGGL(nms_fixup_kernel<true>, dim3(n_images), dim3(kFixupThreads), 0, stream, score, layout, w, h,
                       abs_threshold, cand, cand_cap, cand_count, fix_count, fix_list,
                       reinterpret_cast<uint32_t*>(idle_score_buffer), (size_t)layout.pitch * h);
    return;
  }
  hipLaunchKernelGGL(nms_fixup_kernel<false>, dim3(n_images), dim3(kFixupThreads), 0, stream, score, layout, w, h,
                     abs_threshold, cand, cand_cap, cand_count, fix_count, fix_list, (uint32_t*)nullptr, (size_t)0);
}

void launch_nms(const int32_t* score, int w, int h, int n_images, int abs_threshold,
                Candidate* cand, int cand_cap, int32_t* cand_count, hipStream_t stream) {
  if (n_images <= 0) return;
  const bool aligned = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(score) & 15) == 0);
  if (aligned) {
    const int nd = w >> 2;
    int strips = 1;
    while ((strips - 1) * kStripLanes + 64 < nd) ++strips;
    const int ytiles = (h + kRows * kWaves - 1) / (kRows * kWaves);
    hipLaunchKernelGGL(nms_kernel, dim3(strips * ytiles * n_images), dim3(64, kWaves, 1), 0, stream,
                       score, w, h, abs_threshold, cand, cand_cap, cand_count, strips, ytiles,
                       n_images);
  } else {
    const dim3 grid((w + 255) / 256, (h + 4 * kGenIters - 1) / (4 * kGenIters), n_images);
    hipLaunchKernelGGL(nms_generic_kernel, grid, dim3(64, 4, 1), 0, stream, score, w, h,
                       abs_threshold, cand, cand_cap, cand_count);
  }
}

}  // namespace okvfe
