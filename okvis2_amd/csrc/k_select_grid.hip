// k_select_grid.hip -- uniformity selection on an explicit occupancy grid (the fall-backs of launch_select:
// lab knobs OKVFE_LEGACY_SELECT / OKVFE_SELECT_GRID, grids whose lazy form does not fit, radius 0).
//
//   select_greedy_kernel  occupancy grid in LDS (or HBM), one workgroup of 4 waves per image: the greedy is
//                         serial in its accepted points only -- occupancy only grows, so a candidate that
//                         fails its test once is dead for good; wave 0 decides 64-candidate windows, 4 waves stamp
//   select_kernel<>       1024 threads test 1024 candidates per round, the first that passes is accepted,
//                         961 threads add its 31 x 31 stamp
// Replaces EnforceKeypointUniformity + Subpixel2D behind Frame.hpp:152 (parameters Frontend.cpp:2406-2409).
#include <type_traits>

#include "select_common_dev.h"

namespace okvfe {
namespace {

template <bool OCC_LDS>
__global__ __launch_bounds__(kThreads) void select_kernel(
    const int32_t* __restrict__ scores, ScoreLayout layout, int w, int h, const Candidate* __restrict__ cand,
    int cand_cap, const int32_t* __restrict__ cand_count, const uint64_t* __restrict__ sort_ws,
    int ws_stride, float radius, int max_kpts, const float* __restrict__ lut,
    uint8_t* __restrict__ occ_ws, size_t occ_image_bytes, int occ_rows, int occ_cols,
    okvfe_keypoint* __restrict__ kps, int kp_cap, int32_t* __restrict__ kp_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int wave_first[16];
  __shared__ uint32_t accepted_xy[kMaxKp];  // (y << 16) | x of accepted points, <= kp_cap
  __shared__ int32_t accepted_score[kMaxKp];
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  int n = cand_count[img];
  // overflowed candidate list: WHICH maxima were dropped depends on the order of the atomics, so
  // the image keeps no keypoints at all (deterministic) and okvfe_check_capacity reports it
  n = n > cand_cap ? 0 : n;
  const uint64_t* keys = sort_ws + (size_t)img * ws_stride;
  const int32_t* sc = scores + (size_t)img * layout.pitch * h;
  okvfe_keypoint* out = kps + (size_t)img * kp_cap;
  int kept = 0;

  if (!(radius > 0.0f)) {
    // uniformity disabled: every maximum is a keypoint, in raster order, not capped by max_kpts
    // (sorted by (y, x) here: keys carry score in the high half, so re-sort is avoided by
    // ranking each candidate directly -- O(n^2/threads), only for this rarely used mode)
    const Candidate* c = cand + (size_t)img * cand_cap;
    for (int i = tid; i < n; i += kThreads) {
      const uint32_t me = ((uint32_t)c[i].y << 16) | (uint32_t)c[i].x;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += ((((uint32_t)c[j].y << 16) | (uint32_t)c[j].x) < me);
      if (rank < kp_cap && rank < kMaxKp) {
        accepted_xy[rank] = me;
        accepted_score[rank] = c[i].score;
      }
    }
    kept = n < kp_cap ? n : kp_cap;
    kept = kept < kMaxKp ? kept : kMaxKp;
    __syncthreads();
  } else if (n > 0) {
    uint8_t* occ = OCC_LDS ? smem_raw : occ_ws + (size_t)img * occ_image_bytes;
    if (OCC_LDS) {
      uint32_t* z = reinterpret_cast<uint32_t*>(smem_raw);
      const int nz = (occ_rows * occ_cols + 3) >> 2;
      for (int i = tid; i < nz; i += kThreads) z[i] = 0u;
    }
    const float scaling = (float)(15.0 / (double)radius);
    const float max_score = (float)(0x7FFFFFFF - (int32_t)(keys[0] >> 32));
    const int limit = max_kpts < kp_cap ? max_kpts : kp_cap;
    __syncthreads();
    int pos = 0;
    while (pos < n && kept < limit) {
      // ---- test the window [pos, pos + 1024) against the current occupancy
      const int idx = pos + tid;
      bool pass = false;
      if (idx < n) {
        const uint64_t k = keys[idx];
        const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
        const int y = (int)((k >> 16) & 0xFFFF), x = (int)(k & 0xFFFF);
        const float fy = (float)y * scaling;
        const float fx = (float)x * scaling;
        const int cy = (int)(fy + 16.0f);
        const int cx = (int)(fx + 16.0f);
        const float s0 = (float)occ[(size_t)cy * occ_cols + cx];
        const float q = (float)score / max_score;
        const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
        pass = !(nsc1 < s0);
      }
      const unsigned long long b = __ballot(pass);
      if ((tid & 63) == 0) wave_first[tid >> 6] = b ? (int)__ffsll((long long)b) - 1 : -1;
      __syncthreads();
      int first = -1;
#pragma unroll
      for (int wv = 15; wv >= 0; --wv)
        if (wave_first[wv] >= 0) first = wv * 64 + wave_first[wv];
      if (first < 0) {
        pos += kThreads;
        __syncthreads();
        continue;
      }
      // ---- accept candidate pos + first: stamp its 31x31 patch
      const uint64_t k = keys[pos + first];
      const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
      const int y = (int)((k >> 16) & 0xFFFF), x = (int)(k & 0xFFFF);
      if (tid < 961) {
        const float fy = (float)y * scaling;
        const float fx = (float)x * scaling;
        const int cy = (int)(fy + 16.0f);
        const int cx = (int)(fx + 16.0f);
        const float q = (float)score / max_score;
        const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
        const float nsc = (float)(0.99 * (double)nsc1);
        const int ry = tid / 31, rx = tid - ry * 31;
        const float m = lut[tid] * nsc;
        const int add = (int)ceilf(m);
        uint8_t* cell = occ + (size_t)(cy + ry - 15) * occ_cols + (cx + rx - 15);
        const int v = (int)(*cell) + add;
        *cell = (uint8_t)(v > 255 ? 255 : v);
      }
      if (tid == 0) {
        accepted_xy[kept] = (uint32_t)(k & 0xFFFFFFFFu);
        accepted_score[kept] = score;
      }
      ++kept;
      pos += first + 1;
      __syncthreads();
    }
  }

  // ---- K4: sub-pixel refinement and keypoint emission
  for (int i = tid; i < kept; i += kThreads) {
    const int u = (int)(accepted_xy[i] & 0xFFFF), v = (int)(accepted_xy[i] >> 16);
    int32_t patch[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        patch[(dy + 1) * 3 + (dx + 1)] = sc[score_index(layout, u + dx, v + dy)];
    float ddx, ddy;
    subpixel2d(patch, &ddx, &ddy);
    okvfe_keypoint kp;
    kp.x = (float)u + ddx;
    kp.y = (float)v + ddy;
    kp.size = 12.0f;
    kp.angle = -1.0f;
    kp.response = (float)accepted_score[i];
    kp.octave = 0;
    kp.class_id = -1;
    out[i] = kp;
  }
  if (tid == 0) kp_count[img] = kept;
}

// ---- greedy selection, one workgroup of 4 waves per image (the production path when the
// occupancy grid fits in LDS) ----------------------------------------------------------------
// LDS holds the occupancy grid, the indices of the accepted candidates (u16), a sliding chunk of
// per-candidate records {cell (cy << 16 | cx), level nsc1 (float)} refilled from the sorted keys,
// and the accept list of the current round -- under half a CU's LDS for EuRoC-sized grids, so two
// images run per CU.
//   decide (wave 0): tests 64 consecutive candidates against the occupancy and accepts, in order,
//     every passing one that is more than 30 cells (on either axis) away from all points accepted
//     before it in the same round: its occupancy value is then unchanged, so the sequential test
//     of the reference would pass as well, and its stamp is disjoint from theirs.  The first
//     passing candidate closer than that ends the round and is re-tested in the next one.
//     Windows without a passing candidate are skipped without leaving the wave.
//   stamp (all 4 waves): every thread takes 3 of the 697 non-zero cells of the 31x31 weight table
//     for each accepted point of the round; the stamps of a round touch disjoint cells, so no
//     ordering between them is needed.
// Two workgroup barriers per round; sub-pixel refinement of the accepted points runs at the end.
constexpr int kSelThreads = 256;
constexpr int kStampIts = (kStampCells + kSelThreads - 1) / kSelThreads;
constexpr int kRoundCap = 64;

// OCC_LDS = false: the occupancy grid does not fit in LDS (small uniformity radius or large images)
// and lives in the context's HBM workspace (zeroed by the launcher); same algorithm, the byte
// reads / read-modify-writes go to L2 and the workgroup barriers order them.
// AccT: type of the accepted-candidate indices kept in LDS (u16 while the candidate capacity allows).
template <bool OCC_LDS, typename AccT>
__global__ __launch_bounds__(kSelThreads) void select_greedy_kernel(
    const int32_t* __restrict__ scores, ScoreLayout layout, int w, int h, int cand_cap,
    const int32_t* __restrict__ cand_count, const uint64_t* __restrict__ sort_ws, int ws_stride,
    float radius, int max_kpts, const float* __restrict__ lut, int occ_cols, int occ_bytes16,
    int acc_bytes16, int chunk_cap, okvfe_keypoint* __restrict__ kps, int kp_cap,
    int32_t* __restrict__ kp_count, uint8_t* __restrict__ occ_hbm, size_t occ_hbm_pitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int2 round_list[kRoundCap];  // {cell, 0.99 * level} of the points accepted this round
  __shared__ int s_round, s_pos, s_kept;
  // serial dependency chain: when other streams' kernels share the SIMD, these waves should win
  // arbitration, the throughput kernels fill the gaps
  __builtin_amdgcn_s_setprio(3);
  const int img = blockIdx.x;
  const int lds_occ = OCC_LDS ? occ_bytes16 : 0;  // LDS bytes taken by the grid
  uint8_t* occ = OCC_LDS ? smem_raw : occ_hbm + (size_t)img * occ_hbm_pitch;
  AccT* acc_idx = reinterpret_cast<AccT*>(smem_raw + lds_occ);
  uint2* recs = reinterpret_cast<uint2*>(smem_raw + lds_occ + acc_bytes16);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const bool decider = tid < 64;  // wave 0
  int n = cand_count[img];
  // overflowed candidate list: WHICH maxima were dropped depends on the order of the atomics, so
  // the image keeps no keypoints at all (deterministic) and okvfe_check_capacity reports it
  n = n > cand_cap ? 0 : n;
  const uint64_t* keys = sort_ws + (size_t)img * ws_stride;
  const int32_t* sc = scores + (size_t)img * layout.pitch * h;
  okvfe_keypoint* out = kps + (size_t)img * kp_cap;
  int kept = 0;
  if (n > 0) {  // block-uniform
    if (OCC_LDS) {
      uint4* z = reinterpret_cast<uint4*>(smem_raw);
      const uint4 zero = make_uint4(0, 0, 0, 0);
      for (int i = tid; i < (occ_bytes16 >> 4); i += kSelThreads) z[i] = zero;
    }
    const float scaling = (float)(15.0 / (double)radius);
    const float max_score = (float)(0x7FFFFFFF - (int32_t)(keys[0] >> 32));
    // records of candidates [base, base + chunk_cap) -> recs[], by the whole workgroup (a handful
    // of independent loads per thread, all in flight together)
    auto fill_chunk = [&](int base) {
      const int cnt = min(chunk_cap, n - base);
      for (int i = tid; i < cnt; i += kSelThreads) {
        const uint64_t k = keys[base + i];
        const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
        const int y = (int)((k >> 16) & 0xFFFF), x = (int)(k & 0xFFFF);
        const float fy = (float)y * scaling;
        const float fx = (float)x * scaling;
        const int cy = (int)(fy + 16.0f);
        const int cx = (int)(fx + 16.0f);
        const float q = (float)score / max_score;
        const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
        recs[i] = make_uint2(((uint32_t)cy << 16) | (uint32_t)cx, __float_as_uint(nsc1));
      }
    };
    int chunk_base = 0;
    fill_chunk(0);
    // per-thread stamp geometry: slots j = it*256 + tid of the compacted table
    float lutv[kStampIts];
    int off[kStampIts];
    const uint2* stamp = reinterpret_cast<const uint2*>(lut + kStampTableOffset);
#pragma unroll
    for (int it = 0; it < kStampIts; ++it) {
      const int j = it * kSelThreads + tid;
      const uint2 e = stamp[j < kStampSlots ? j : kStampSlots - 1];  // padding slots: weight 0
      lutv[it] = __uint_as_float(e.y);
      off[it] = ((int)(e.x >> 8) - 15) * occ_cols + ((int)(e.x & 0xFF) - 15);
    }
    const bool last_writes = (kStampIts - 1) * kSelThreads + tid < kStampCells;
    const int limit = max_kpts < kp_cap ? max_kpts : kp_cap;
    int pos = 0;
    __syncthreads();
    while (true) {
      // the 64-candidate window would run past the resident chunk: slide it (block-uniform)
      if (pos + 64 > chunk_base + chunk_cap && chunk_base + chunk_cap < n) {
        chunk_base = pos;
        fill_chunk(pos);
        __syncthreads();
      }
      if (decider) {
        int nacc = 0;
        bool refill = false;
        while (pos < n && kept < limit) {
          if (pos + 64 > chunk_base + chunk_cap && chunk_base + chunk_cap < n) {
            refill = true;  // skipped past the chunk through windows without a passing candidate
            break;
          }
          const int idx = pos + lane;
          uint2 rec = make_uint2(0, 0);
          if (idx < n) rec = recs[idx - chunk_base];
          const int cx = (int)(rec.x & 0xFFFF), cy = (int)(rec.x >> 16);
          const int cell = cy * occ_cols + cx;
          const float s0 = (float)occ[cell];  // idx >= n reads cell 0: in range, result unused
          const bool pass = idx < n && !(__uint_as_float(rec.y) < s0);
          unsigned long long m = __ballot(pass);
          if (m == 0) {
            pos += 64;
            continue;
          }
          // One backward branch per accepted point.  `blocked` collects the candidates whose cell
          // lies within 30 cells (both axes) of a point accepted in this round: the first such
          // passing candidate ends the round and is re-tested in the next one.  Inside 15 cells
          // its occupancy value changes; between 16 and 30 it would still pass, but its stamp
          // would overlap the other one -- ending the round there keeps all stamps of a round
          // disjoint, so the four waves can apply them without any ordering between them.
          unsigned long long blocked = 0, accm = 0, rem = m, cand;
          int first = (int)__ffsll((long long)rem) - 1;
          bool go;
          do {
            const int wcx = __builtin_amdgcn_readlane(cx, first);
            const int wcy = __builtin_amdgcn_readlane(cy, first);
            const int ax = cx - wcx, ay = cy - wcy;
            blocked |= __ballot((ax < 0 ? -ax : ax) <= 30 && (ay < 0 ? -ay : ay) <= 30);
            accm |= 1ull << first;
            ++nacc;
            rem &= rem - 1;
            cand = kept + nacc < limit ? rem : 0ull;
            first = ((int)__ffsll((long long)cand) - 1) & 63;
            go = cand != 0 && !((blocked >> first) & 1);
          } while (go);
          const int adv = cand != 0 ? first : 64;  // limit reached: the outer loop ends anyway
          // accepted lanes publish themselves in order: rank = accepted lanes below this one
          if ((accm >> lane) & 1) {
            const int rank = __popcll(accm & ((1ull << lane) - 1ull));
            const float nsc = (float)(0.99 * (double)__uint_as_float(rec.y));
            round_list[rank] = make_int2(cell, __float_as_int(nsc));
            acc_idx[kept + rank] = (AccT)idx;
          }
          kept += nacc;
          pos += adv;
          break;
        }
        if (lane == 0) {
          s_round = refill ? -1 : nacc;  // refill is only set with nacc == 0
          s_pos = pos;
          s_kept = kept;
        }
      }
      __syncthreads();
      const int nacc = s_round;
      pos = s_pos;
      kept = s_kept;
      if (nacc == 0) break;  // block-uniform: candidates exhausted or limit reached
      if (nacc < 0) continue;  // chunk refill requested: back to the top
      // stamps of one round are disjoint: up to 4 are in flight together (all reads, then the
      // arithmetic and the writes)
      for (int a0 = 0; a0 < nacc; a0 += 4) {
        int2 e[4];
        int v[4][kStampIts];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = round_list[min(a0 + u, nacc - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int it = 0; it < kStampIts; ++it) v[u][it] = occ[e[u].x + off[it]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (a0 + u >= nacc) break;  // block-uniform
          const float nsc = __int_as_float(e[u].y);
#pragma unroll
          for (int it = 0; it < kStampIts; ++it) {
            const float mm = lutv[it] * nsc;
            const int nv = v[u][it] + (int)ceilf(mm);
            if (it < kStampIts - 1 || last_writes)
              occ[e[u].x + off[it]] = (uint8_t)(nv > 255 ? 255 : nv);
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < kept; i += kSelThreads) {
    const uint64_t k = keys[acc_idx[i]];
    const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
    const int v = (int)((k >> 16) & 0xFFFF), u = (int)(k & 0xFFFF);
    int32_t patch[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        patch[(dy + 1) * 3 + (dx + 1)] = sc[score_index(layout, u + dx, v + dy)];
    float ddx, ddy;
    subpixel2d(patch, &ddx, &ddy);
    okvfe_keypoint kp;
    kp.x = (float)u + ddx;
    kp.y = (float)v + ddy;
    kp.size = 12.0f;
    kp.angle = -1.0f;
    kp.response = (float)score;
    kp.octave = 0;
    kp.class_id = -1;
    out[i] = kp;
  }
  if (tid == 0) kp_count[img] = kept;
}

}  // namespace

// the grid paths of launch_select (k_select.hip decides; sorted keys are in sort_ws)
void launch_select_grid(const int32_t* score, ScoreLayout layout, int w, int h, int n_images, Candidate* cand,
                        int cand_cap, const int32_t* cand_count, float radius, int max_kpts, const float* lut,
                        uint8_t* occupancy, size_t occ_image_bytes, int occ_rows, int occ_cols, okvfe_keypoint* kps,
                        int kp_cap, int32_t* kp_count, uint64_t* sort_ws, hipStream_t stream) {
  int ws_stride = 1;
  while (ws_stride < cand_cap) ws_stride <<= 1;
  const size_t occ_bytes = ((size_t)occ_rows * occ_cols + 15) & ~(size_t)15;
  static const bool force_hbm = lab_env("OKVFE_SELECT_OCC_HBM") != nullptr;  // A/B knob
  const bool occ_lds = radius > 0.0f && occ_bytes <= 120 * 1024 && !force_hbm;
  // greedy kernel: occupancy + accepted indices (u16, u32 for capacities above 65536) + a sliding
  // chunk of candidate records.  Half a CU's LDS (two images per CU) when at least 128 records
  // fit, else the whole CU; grids that do not fit at all stay in the HBM workspace.
  const bool wide = cand_cap > 65536;
  const size_t acc_bytes = ((size_t)kp_cap * (wide ? 4 : 2) + 15) & ~(size_t)15;
  static const bool legacy = lab_env("OKVFE_LEGACY_SELECT") != nullptr;  // A/B knob
  if (occ_lds && !legacy) {
    const size_t fixed = occ_bytes + acc_bytes;
    const size_t half = 79 * 1024, full = 152 * 1024;  // + ~0.5 KiB static: two blocks per CU
    size_t budget = fixed + 128 * 8 <= half ? half : full;
    if (fixed + 128 * 8 <= budget) {
      size_t chunk = (budget - fixed) / 8 / 64 * 64;
      const size_t need = ((size_t)cand_cap + 63) / 64 * 64;
      if (chunk > need) chunk = need;
      const size_t lds = fixed + chunk * 8;
#define OKVFE_SELECT_LAUNCH(LDS, T, BYTES, OCC, PITCH)                                           \
  hipLaunchKernelGGL((select_greedy_kernel<LDS, T>), dim3(n_images), dim3(kSelThreads), BYTES,   \
                     stream, score, layout, w, h, cand_cap, cand_count, sort_ws, ws_stride, radius, \
                     max_kpts, lut, occ_cols, (int)occ_bytes, (int)acc_bytes, (int)chunk, kps,   \
                     kp_cap, kp_count, OCC, PITCH)
      if (wide)
        OKVFE_SELECT_LAUNCH(true, uint32_t, lds, (uint8_t*)nullptr, (size_t)0);
      else
        OKVFE_SELECT_LAUNCH(true, uint16_t, lds, (uint8_t*)nullptr, (size_t)0);
      return;
    }
  }
  if (radius > 0.0f && occupancy != nullptr && !legacy && acc_bytes + 128 * 8 <= 24 * 1024) {
    // grid in HBM: LDS only holds the accepted indices and the record chunk (24 KiB: 6 images / CU)
    size_t chunk = (24 * 1024 - acc_bytes) / 8 / 64 * 64;
    const size_t need = ((size_t)cand_cap + 63) / 64 * 64;
    if (chunk > need) chunk = need;
    (void)hipMemsetAsync(occupancy, 0, occ_image_bytes * (size_t)n_images, stream);
    if (wide)
      OKVFE_SELECT_LAUNCH(false, uint32_t, acc_bytes + chunk * 8, occupancy, occ_image_bytes);
    else
      OKVFE_SELECT_LAUNCH(false, uint16_t, acc_bytes + chunk * 8, occupancy, occ_image_bytes);
    return;
  }
#undef OKVFE_SELECT_LAUNCH
  if (occ_lds) {
    hipLaunchKernelGGL(select_kernel<true>, dim3(n_images), dim3(kThreads), occ_bytes, stream,
                       score, layout, w, h, cand, cand_cap, cand_count, sort_ws, ws_stride, radius,
                       max_kpts, lut, occupancy, occ_image_bytes, occ_rows, occ_cols, kps, kp_cap,
                       kp_count);
  } else {
    if (radius > 0.0f)
      (void)hipMemsetAsync(occupancy, 0, occ_image_bytes * (size_t)n_images, stream);
    hipLaunchKernelGGL(select_kernel<false>, dim3(n_images), dim3(kThreads), 0, stream, score, layout, w,
                       h, cand, cand_cap, cand_count, sort_ws, ws_stride, radius, max_kpts, lut,
                       occupancy, occ_image_bytes, occ_rows, occ_cols, kps, kp_cap, kp_count);
  }
}

}  // namespace okvfe
