// k_match.hip -- K7: brute-force 384-bit Hamming matching with the reference's FP64 gates.
//
//   match_stereo_kernel   the k0 x k1 loop of okvis::Frontend::matchStereo
//                         (okvis_frontend/src/Frontend.cpp:2016-2076) including
//                         triangulation::triangulateFast
//                         (okvis_frontend/src/stereo_triangulation.cpp:50-132).
//   hamming_argmin_kernel the ungated running minimum of verifyRecognisedPlace
//                         (Frontend.cpp:337-346).
//   hamming_count/emit    all pairs below a threshold in (i, j) order (host replay of the
//                         other gated loops: Frontend.cpp:1566-1587, 1651-1716, 1844-1893).
//   popcount primitive    = brisk::Hamming::PopcntofXORed(a, b, 3) (Frontend.cpp:2024).
//
//   match_motion_*        matchMotionStereo (Frontend.cpp:1789-1905), host arrays or gather blocks.
//   match_to_map_*        matchToMapByThread[Unitialised] (Frontend.cpp:1552-1589, 1616-1719).
//
// Mapping of the gated matchers: lane = one k0 with its 48-byte descriptor in 12 VGPRs; the other
// side is cut into 4 segments (one wave each) whose descriptors are staged in LDS by coalesced
// loads and read back as broadcasts.  The geometric gate of the reference does not depend on the
// running best, so "first k1 reaching the smallest gated distance" is found in rounds: a
// branch-free scan yields the two smallest admissible (dist, k1) keys per lane, the FP64 gate runs
// once per key for all lanes together, rejected lanes raise their floor and rescan; the segments
// are merged by (dist, segment).  Integer-ALU bound (v_xor + v_bcnt_u32_b32, 24 VALU per pair);
// inputs are 2 x 33.6 KB per EuRoC stereo frame, so HBM is not the limit.
#include "camera_dev.h"
#include <algorithm>

#include "okvfe_internal.h"

namespace okvfe {
namespace {

struct Desc12 {
  uint32_t w[12];
};

__device__ __forceinline__ Desc12 load_desc(const uint8_t* p) {
  Desc12 d;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2];
  d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w;
  d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
  d.w[8] = c.x; d.w[9] = c.y; d.w[10] = c.z; d.w[11] = c.w;
  return d;
}

__device__ __forceinline__ int hamming(const Desc12& a, const uint32_t* __restrict__ b) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) c += __popc(a.w[i] ^ b[i]);
  return c;
}

// Order of every 3-term FP64 sum (okvfe_set_fp64_reduction; oracle/orc_match.c states the reasoning):
// 1 = Eigen's unrolled non-vectorised redux x0 + (x1 + x2) (default), 0 = left to right.
__device__ int g_fp64_tree = 1;
__device__ __forceinline__ double sum3(double p0, double p1, double p2) {
  const bool tree = g_fp64_tree != 0;  // scalar load, uniform branch-free select
  const double u = tree ? p1 : p0, v = tree ? p2 : p1, w = tree ? p0 : p2;
  const double s = u + v;
  return tree ? w + s : s + w;  // (addition commutes bit-exactly: one add either way)
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) {
  const double p0 = a[0] * b[0];
  const double p1 = a[1] * b[1];
  const double p2 = a[2] * b[2];
  return sum3(p0, p1, p2);
}
__device__ __forceinline__ void normalize3(const double v[3], double out[3]) {
  const double n = sqrt(dot3(v, v));
  out[0] = v[0] / n;
  out[1] = v[1] / n;
  out[2] = v[2] / n;
}
__device__ __forceinline__ void rot(const double C[9], const double v[3], double out[3]) {
  out[0] = dot3(C, v);
  out[1] = dot3(C + 3, v);
  out[2] = dot3(C + 6, v);
}
__device__ __forceinline__ void rot_t(const double C[9], const double v[3], double out[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double p0 = C[i] * v[0];
    const double p1 = C[3 + i] * v[1];
    const double p2 = C[6 + i] * v[2];
    out[i] = sum3(p0, p1, p2);
  }
}
__device__ __forceinline__ void inv_transform_h(const double C[9], const double r[3],
                                                const double hp[4], double out[4]) {
  double cr[3], h[3];
  rot_t(C, r, cr);
  rot_t(C, hp, h);
  const double s = hp[3];
  out[0] = h[0] + (-cr[0]) * s;
  out[1] = h[1] + (-cr[1]) * s;
  out[2] = h[2] + (-cr[2]) * s;
  out[3] = s;
}

__device__ void midpoint_parallel(const double p1[3], const double e1[3], const double p2[3],
                                  const double e2[3], const double t12[3], double c26,
                                  double hp[4], bool* is_valid) {
  *is_valid = true;
  double mid[3], d[3], dn[3];
  const double tn = sqrt(dot3(t12, t12));
  const double f = 40.0 * (0.01 > tn ? 0.01 : tn);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double m = p1[i] + 0.5 * t12[i];
    mid[i] = m + f * (e1[i] + e2[i]);
  }
  hp[0] = mid[0]; hp[1] = mid[1]; hp[2] = mid[2]; hp[3] = 1.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) d[i] = mid[i] - p1[i];
  normalize3(d, dn);
  if (dot3(e1, dn) < c26) *is_valid = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) d[i] = mid[i] - p2[i];
  normalize3(d, dn);
  if (dot3(e2, dn) < c26) *is_valid = false;
}

// stereo_triangulation.cpp:50-132 with cos(2.6 sigma), cos(6 sigma) supplied by the host
__device__ void triangulate_fast(const double p1[3], const double e1[3], const double p2[3],
                                 const double e2[3], double c26, double c6, double hp[4],
                                 bool* is_valid, bool* is_parallel) {
  *is_parallel = false;
  *is_valid = true;
  const double t12[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double b0 = dot3(t12, e1), b1 = dot3(t12, e2);
  const double a00 = dot3(e1, e1);
  const double a10 = dot3(e1, e2);
  const double a01 = -a10;
  const double a11 = -dot3(e2, e2);
  const double det = a00 * a11 - a01 * a10;
  if (!(fabs(det) > 1.0e-12)) {
    *is_parallel = true;
    midpoint_parallel(p1, e1, p2, e2, t12, c26, hp, is_valid);
    return;
  }
  const double invdet = 1.0 / det;
  const double i00 = a11 * invdet, i10 = -a10 * invdet, i01 = -a01 * invdet, i11 = a00 * invdet;
  const double l0 = i00 * b0 + i01 * b1;
  const double l1 = i10 * b0 + i11 * b1;
  if (l0 < 0.01 || l1 < 0.01) {
    *is_parallel = true;
    midpoint_parallel(p1, e1, p2, e2, t12, c26, hp, is_valid);
    return;
  }
  double mid[3], d1[3], d2[3], n1[3], n2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xm = l0 * e1[i] + p1[i];
    const double xn = l1 * e2[i] + p2[i];
    mid[i] = (xm + xn) / 2.0;
    d1[i] = mid[i] - p1[i];
    d2[i] = mid[i] - p2[i];
  }
  normalize3(d1, n1);
  normalize3(d2, n2);
  if (dot3(e1, n1) < c26) *is_valid = false;
  if (dot3(e2, n2) < c26) *is_valid = false;
  if (dot3(n2, n1) > c6) *is_parallel = true;
  hp[0] = mid[0]; hp[1] = mid[1]; hp[2] = mid[2]; hp[3] = 1.0;
}

struct BlockView {  // per-image arrays as the matcher sees them
  const uint8_t* desc;
  const double* bp;
  const uint8_t* bpv;
  int n;
  const okvfe_keypoint* kps = nullptr;  // size classes (octave field); only read when P.cls != null
};
// cos(2.6 sigma), cos(6 sigma) for the size classes of the two keypoints
__device__ __forceinline__ void gate_cos(const PairParams& P, const okvfe_keypoint* kps0, int k0,
                                         const okvfe_keypoint* kps1, int k1, double* c26, double* c6) {
  *c26 = P.cos26;
  *c6 = P.cos6;
  if (P.cls != nullptr && kps0 != nullptr && kps1 != nullptr) {
    const int i = (kps0[k0].octave & (kSizeClasses - 1)) * kSizeClasses + (kps1[k1].octave & (kSizeClasses - 1));
    *c26 = P.cls[i];
    *c6 = P.cls[kSizeClasses * kSizeClasses + i];
  }
}

// The k1 range is cut into kStereoSegs contiguous segments, one wave (threadIdx.y) each, so that
// 4x as many waves hide the scalar-load and FP64 latency of the serial k1 loop.  The sequential
// rule of the reference (first k1 reaching the smallest gated distance) is a pure function of the
// candidate set, so merging the segments by (dist, segment) reproduces it exactly.
constexpr int kStereoSegs = 4;
#ifndef OKVFE_MATCH_CHUNK
#define OKVFE_MATCH_CHUNK 64   // (round 6: 128 -> 64 and five waves per SIMD: 262 -> 249 us per 3072 EuRoC pairs; six waves: 277)
#endif
#ifndef OKVFE_MATCH_WAVES
#define OKVFE_MATCH_WAVES 5
#endif
constexpr int kStereoChunk = OKVFE_MATCH_CHUNK;  // descriptors of one segment staged in LDS at a time (6 KiB)
struct SegBest {
  double hp[4];
  int best, k1, init, pad;
};

constexpr uint32_t kNoKey = 0xFFFFFFFFu;
#ifndef OKVFE_MATCH_HALF_REJECT
#define OKVFE_MATCH_HALF_REJECT 0  // measured SLOWER (280 vs 263 us: the scan is bound by LDS latency at four waves per SIMD, not by its 24 xor / bcnt; the branch stops the loads of the next descriptors from overlapping): A/B only
#endif
#ifndef OKVFE_MATCH_MORE
#define OKVFE_MATCH_MORE 6  // keys per re-scan of the gated matchers (the first scan keeps two)
#endif

// One scan of segment [k1_lo, k1_hi) of the other side's descriptors: the two smallest keys
// ((dist << 22) | k1) + 1 that exceed floor_key, have dist < threshold and are not flagged in
// skip1 (may be null).  The descriptors (and flags) are staged in LDS by coalesced vector loads
// and read back as broadcasts; a segment of at most kStereoChunk descriptors stays resident
// (`resident`: already loaded by load_scan_chunk before the first round).
struct ScanChunk {
  uint4* desc;     // kStereoChunk * 3
  uint8_t* skip;   // kStereoChunk
};
__device__ __forceinline__ void load_scan_chunk(const ScanChunk& C, const uint8_t* __restrict__ desc1,
                                                const uint8_t* __restrict__ skip1, int c0, int cnt) {
  const uint4* src = reinterpret_cast<const uint4*>(desc1 + (size_t)c0 * OKVFE_DESC_BYTES);
  __builtin_amdgcn_wave_barrier();
  for (int i = threadIdx.x; i < cnt * 3; i += 64) C.desc[i] = src[i];
  if (skip1)
    for (int i = threadIdx.x; i < cnt; i += 64) C.skip[i] = skip1[c0 + i];
  __builtin_amdgcn_wave_barrier();
}
// K smallest admissible keys, ascending in c[0 .. K)
template <int K, bool HAS_SKIP>
__device__ __forceinline__ void scan_top(const Desc12& d0, const ScanChunk& C,
                                         const uint8_t* __restrict__ desc1,
                                         const uint8_t* __restrict__ skip1, int k1_lo, int k1_hi,
                                         bool resident, uint32_t floor_key, uint32_t threshold,
                                         uint32_t (&c)[K]) {
#pragma unroll
  for (int t = 0; t < K; ++t) c[t] = kNoKey;
  for (int c0 = k1_lo; c0 < k1_hi; c0 += kStereoChunk) {
    const int cnt = min(kStereoChunk, k1_hi - c0);
    if (!resident) load_scan_chunk(C, desc1, HAS_SKIP ? skip1 : nullptr, c0, cnt);
    auto insert = [&](uint32_t dist, int jj) {
      uint32_t key = ((dist << 22) | (uint32_t)(c0 + jj)) + 1u;
      bool ok = key > floor_key && dist < threshold;
      if (HAS_SKIP) ok = ok && C.skip[jj] == 0;
      key = ok ? key : kNoKey;
#pragma unroll
      for (int t = K - 1; t > 0; --t) c[t] = min(c[t], max(c[t - 1], key));  // insertion into the sorted K
      c[0] = min(c[0], key);
    };
    int j = 0;
#if OKVFE_MATCH_HALF_REJECT
    // Round 6: the first 192 bits decide for almost every descriptor of the other side -- unrelated BRISK2 descriptors
    // differ in 96 +- 7 of them -- so four descriptors at a time are compared on six words first, and when NO lane of the
    // wave is below the threshold on any of the four (one v_min3 pair, one compare, a scalar branch) the other six
    // words cannot bring a lane below it either: the insertions would be no-ops and are skipped with the arithmetic
    for (; j + 4 <= cnt; j += 4) {
      uint32_t part[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t* bj = reinterpret_cast<const uint32_t*>(C.desc + 3 * (j + u));
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) d += __popc(d0.w[i] ^ bj[i]);
        part[u] = d;
      }
      const uint32_t pm = min(min(part[0], part[1]), min(part[2], part[3]));
      if (__builtin_amdgcn_ballot_w64(pm < threshold) == 0ull) continue;  // wave-uniform
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t* bj = reinterpret_cast<const uint32_t*>(C.desc + 3 * (j + u));
        uint32_t d = part[u];
#pragma unroll
        for (int i = 6; i < 12; ++i) d += __popc(d0.w[i] ^ bj[i]);
        insert(d, j + u);
      }
    }
#endif
#pragma unroll 4
    for (; j < cnt; ++j)
      insert((uint32_t)hamming(d0, reinterpret_cast<const uint32_t*>(C.desc + 3 * j)), j);
  }
}
template <bool HAS_SKIP>
__device__ __forceinline__ void scan_top2(const Desc12& d0, const ScanChunk& C,
                                          const uint8_t* __restrict__ desc1,
                                          const uint8_t* __restrict__ skip1, int k1_lo, int k1_hi,
                                          bool resident, uint32_t floor_key, uint32_t threshold,
                                          uint32_t* c1_out, uint32_t* c2_out) {
  uint32_t c[2];
  scan_top<2, HAS_SKIP>(d0, C, desc1, skip1, k1_lo, k1_hi, resident, floor_key, threshold, c);
  *c1_out = c[0];
  *c2_out = c[1];
}

__device__ void match_stereo_rows_r5(const PairParams& P, const BlockView& I0, const BlockView& I1,
                                  int threshold, okvfe_stereo_match* __restrict__ out) {
  __shared__ SegBest seg_best[kStereoSegs - 1][64];
  __shared__ uint4 seg_desc[kStereoSegs][kStereoChunk * 3];
  const int seg = threadIdx.y;
  const int per_seg = (I1.n + kStereoSegs - 1) / kStereoSegs;
  const int k1_lo = min(seg * per_seg, I1.n), k1_hi = min(k1_lo + per_seg, I1.n);
  if ((int)blockIdx.x * 64 >= I0.n) return;  // whole block past the last keypoint
  const int k0 = blockIdx.x * 64 + threadIdx.x;
  const bool active = k0 < I0.n;
  Desc12 d0 = {};
  if (active) d0 = load_desc(I0.desc + (size_t)k0 * OKVFE_DESC_BYTES);
  double e0_W[3] = {0, 0, 0};
  const bool v0 = active && I0.bpv[k0] != 0;
  if (v0) {
    double v[3];
    rot(P.C0, I0.bp + 3 * (size_t)k0, v);
    normalize3(v, e0_W);
  }
  int best = threshold;  // running "distances"
  int k1_match = 0;
  bool initialisable = false;
  double hps[4] = {0, 0, 0, 0};
  // The reference walks k1 upwards and runs the geometric gate whenever dist < best; as the gate
  // does not depend on `best`, its outcome is the gated candidate with the smallest (dist, k1).
  // Each round therefore scans the segment branch-free for the smallest key above the last
  // rejected one (wave-uniform descriptor loads, 24 VALU per k1) and runs the FP64 gate ONCE for
  // all lanes together, instead of once per k1 for the one or two lanes that improved there.
  uint32_t floor_key = 0;  // keys are ((dist << 22) | k1) + 1, so 0 admits everything
  bool done = !v0;         // without a back-projection the gate rejects every candidate
  const ScanChunk chunk{seg_desc[seg], nullptr};
  const bool resident = k1_hi - k1_lo <= kStereoChunk;
  if (resident) load_scan_chunk(chunk, I1.desc, nullptr, k1_lo, k1_hi - k1_lo);
  int n_scans = 0;
  while (__any(!done)) {
    // the two smallest admissible keys of the segment in one scan: a rejected best candidate
    // usually has its successor at hand, so the tail of the kernel is not set by re-scans.  A lane
    // that is still undecided after them sits in look-alike content (repetitive texture: 5-15
    // candidates below the threshold, most of them rejected by the gate): every further scan then
    // brings SIX keys (insertion costs 5 min/max pairs more per descriptor, a re-scan 24 + 7)
    constexpr int kMore = OKVFE_MATCH_MORE;
#ifndef OKVFE_MATCH_NARROW_SCANS
#define OKVFE_MATCH_NARROW_SCANS 1
#endif
    constexpr int kNarrowScans = OKVFE_MATCH_NARROW_SCANS;
    uint32_t cs[kMore];
#pragma unroll
    for (int u = 0; u < kMore; ++u) cs[u] = kNoKey;
    int n_c = 2;
    if (n_scans < kNarrowScans) {
      scan_top2<false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key,
                       (uint32_t)threshold, &cs[0], &cs[1]);
    } else {
      scan_top<kMore, false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, cs);
      n_c = kMore;
    }
    ++n_scans;
#pragma unroll 1
    for (int t = 0; t < n_c; ++t) {
      uint32_t cand = cs[0];  // cs[t] without a dynamically indexed register array
#pragma unroll
      for (int u = 1; u < kMore; ++u) cand = t == u ? cs[u] : cand;
      bool pending = !done;
      if (pending && cand == kNoKey) {  // nothing (more) above the floor in this segment
        done = true;
        pending = false;
      }
      if (!__any(pending)) break;  // wave-uniform
      if (!pending) continue;
      floor_key = cand;
      const int k1 = (int)((cand - 1u) & 0x3FFFFFu);
      const int dist = (int)((cand - 1u) >> 22);
      if (!I1.bpv[k1]) continue;
      double v[3], e1_W[3], hp_W[4], hp_C0[4], hp_C1[4];
      rot(P.C1, I1.bp + 3 * (size_t)k1, v);
      normalize3(v, e1_W);
      bool is_valid, is_parallel;
      double c26, c6;
      gate_cos(P, I0.kps, k0, I1.kps, k1, &c26, &c6);
      triangulate_fast(P.r0, e0_W, P.r1, e1_W, c26, c6, hp_W, &is_valid, &is_parallel);
      inv_transform_h(P.C0, P.r0, hp_W, hp_C0);
      inv_transform_h(P.C1, P.r1, hp_W, hp_C1);
      if (!is_parallel) {
        const double w4 = hp_W[3];
        hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
        if (hp_C0[2] / hp_C0[3] < 0.05) is_valid = false;
        if (hp_C1[2] / hp_C1[3] < 0.05) is_valid = false;
        if (dot3(e0_W, e1_W) < 0.8) is_valid = false;
      }
      if (is_valid) {
        best = dist;
        hps[0] = hp_W[0]; hps[1] = hp_W[1]; hps[2] = hp_W[2]; hps[3] = hp_W[3];
        k1_match = k1;
        initialisable = !is_parallel;
        done = true;
      }
    }
  }
  if (seg > 0) {
    SegBest& sb = seg_best[seg - 1][threadIdx.x];
    sb.best = best; sb.k1 = k1_match; sb.init = initialisable ? 1 : 0;
    sb.hp[0] = hps[0]; sb.hp[1] = hps[1]; sb.hp[2] = hps[2]; sb.hp[3] = hps[3];
  }
  __syncthreads();
  if (seg > 0) return;
#pragma unroll
  for (int s = 0; s < kStereoSegs - 1; ++s) {
    const SegBest& sb = seg_best[s][threadIdx.x];
    if (sb.best < best) {  // strict: ties stay with the lower segment = lower k1
      best = sb.best; k1_match = sb.k1; initialisable = sb.init != 0;
      hps[0] = sb.hp[0]; hps[1] = sb.hp[1]; hps[2] = sb.hp[2]; hps[3] = sb.hp[3];
    }
  }
  if (active) {
    okvfe_stereo_match m;
    const bool hit = best < threshold;
    m.k1 = hit ? k1_match : -1;
    m.dist = hit ? best : threshold;
    m.initialisable = hit ? (initialisable ? 1 : 0) : 0;
    m.pad = 0;
    m.hp_W[0] = hit ? hps[0] : 0.0;
    m.hp_W[1] = hit ? hps[1] : 0.0;
    m.hp_W[2] = hit ? hps[2] : 0.0;
    m.hp_W[3] = hit ? hps[3] : 0.0;
    out[k0] = m;
  }
}

__device__ void match_stereo_rows_bound(const PairParams& P, const BlockView& I0, const BlockView& I1,
                                  int threshold, okvfe_stereo_match* __restrict__ out) {
  __shared__ SegBest seg_best[kStereoSegs - 1][64];
  __shared__ uint4 seg_desc[kStereoSegs][kStereoChunk * 3];
  __shared__ uint32_t best_key[64];  // smallest key any segment has found VALID so far (kNoKey: none): larger keys cannot win
  __shared__ double e0_lds[3][64];
  const int seg = threadIdx.y;
  const int per_seg = (I1.n + kStereoSegs - 1) / kStereoSegs;
  const int k1_lo = min(seg * per_seg, I1.n), k1_hi = min(k1_lo + per_seg, I1.n);
  if ((int)blockIdx.x * 64 >= I0.n) return;  // whole block past the last keypoint
  const int k0 = blockIdx.x * 64 + threadIdx.x;
  const bool active = k0 < I0.n;
  Desc12 d0 = {};
  if (active) d0 = load_desc(I0.desc + (size_t)k0 * OKVFE_DESC_BYTES);
  double e0_W[3] = {0, 0, 0};
  const bool v0 = active && I0.bpv[k0] != 0;
  if (seg == 0) {  // the query rays once per block, not once per segment wave
    if (v0) {
      double v[3];
      rot(P.C0, I0.bp + 3 * (size_t)k0, v);
      normalize3(v, e0_W);
    }
    e0_lds[0][threadIdx.x] = e0_W[0]; e0_lds[1][threadIdx.x] = e0_W[1]; e0_lds[2][threadIdx.x] = e0_W[2];
    best_key[threadIdx.x] = kNoKey;
  }
  __syncthreads();
  if (seg != 0) {
    e0_W[0] = e0_lds[0][threadIdx.x]; e0_W[1] = e0_lds[1][threadIdx.x]; e0_W[2] = e0_lds[2][threadIdx.x];
  }
  int best = threshold;  // running "distances"
  int k1_match = 0;
  bool initialisable = false;
  double hps[4] = {0, 0, 0, 0};
  // The reference walks k1 upwards and runs the geometric gate whenever dist < best; as the gate
  // does not depend on `best`, its outcome is the gated candidate with the smallest (dist, k1).
  // Each round therefore scans the segment branch-free for the smallest key above the last
  // rejected one (wave-uniform descriptor loads, 24 VALU per k1) and runs the FP64 gate ONCE for
  // all lanes together, instead of once per k1 for the one or two lanes that improved there.
  uint32_t floor_key = 0;  // keys are ((dist << 22) | k1) + 1, so 0 admits everything
  bool done = !v0;         // without a back-projection the gate rejects every candidate
  const ScanChunk chunk{seg_desc[seg], nullptr};
  const bool resident = k1_hi - k1_lo <= kStereoChunk;
  if (resident) load_scan_chunk(chunk, I1.desc, nullptr, k1_lo, k1_hi - k1_lo);
  int n_scans = 0;
  while (__any(!done)) {
    // the two smallest admissible keys of the segment in one scan: a rejected best candidate
    // usually has its successor at hand, so the tail of the kernel is not set by re-scans.  A lane
    // that is still undecided after them sits in look-alike content (repetitive texture: 5-15
    // candidates below the threshold, most of them rejected by the gate): every further scan then
    // brings SIX keys (insertion costs 5 min/max pairs more per descriptor, a re-scan 24 + 7)
    constexpr int kMore = OKVFE_MATCH_MORE;
#ifndef OKVFE_MATCH_NARROW_SCANS
#define OKVFE_MATCH_NARROW_SCANS 1
#endif
    constexpr int kNarrowScans = OKVFE_MATCH_NARROW_SCANS;
    uint32_t cs[kMore];
#pragma unroll
    for (int u = 0; u < kMore; ++u) cs[u] = kNoKey;
    int n_c = 2;
    if (n_scans < kNarrowScans) {
      scan_top2<false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key,
                       (uint32_t)threshold, &cs[0], &cs[1]);
    } else {
      scan_top<kMore, false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, cs);
      n_c = kMore;
    }
    ++n_scans;
#pragma unroll 1
    for (int t = 0; t < n_c; ++t) {
      uint32_t cand = cs[0];  // cs[t] without a dynamically indexed register array
#pragma unroll
      for (int u = 1; u < kMore; ++u) cand = t == u ? cs[u] : cand;
      bool pending = !done;
      if (pending && cand == kNoKey) {  // nothing (more) above the floor in this segment
        done = true;
        pending = false;
      }
      // a candidate above a key that another segment has already found valid cannot win, nor can its successors
      // (a stale bound only costs work: the merge below decides)
      if (pending && cand > *reinterpret_cast<volatile uint32_t*>(&best_key[threadIdx.x])) {
        done = true;
        pending = false;
      }
      if (!__any(pending)) break;  // wave-uniform
      if (!pending) continue;
      floor_key = cand;
      const int k1 = (int)((cand - 1u) & 0x3FFFFFu);
      const int dist = (int)((cand - 1u) >> 22);
      if (!I1.bpv[k1]) continue;
      double v[3], e1_W[3], hp_W[4], hp_C0[4], hp_C1[4];
      rot(P.C1, I1.bp + 3 * (size_t)k1, v);
      normalize3(v, e1_W);
      bool is_valid, is_parallel;
      double c26, c6;
      gate_cos(P, I0.kps, k0, I1.kps, k1, &c26, &c6);
      triangulate_fast(P.r0, e0_W, P.r1, e1_W, c26, c6, hp_W, &is_valid, &is_parallel);
      inv_transform_h(P.C0, P.r0, hp_W, hp_C0);
      inv_transform_h(P.C1, P.r1, hp_W, hp_C1);
      if (!is_parallel) {
        const double w4 = hp_W[3];
        hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
        if (hp_C0[2] / hp_C0[3] < 0.05) is_valid = false;
        if (hp_C1[2] / hp_C1[3] < 0.05) is_valid = false;
        if (dot3(e0_W, e1_W) < 0.8) is_valid = false;
      }
      if (is_valid) {
        best = dist;
        hps[0] = hp_W[0]; hps[1] = hp_W[1]; hps[2] = hp_W[2]; hps[3] = hp_W[3];
        k1_match = k1;
        initialisable = !is_parallel;
        done = true;
        atomicMin(&best_key[threadIdx.x], cand);
      }
    }
  }
  if (seg > 0) {
    SegBest& sb = seg_best[seg - 1][threadIdx.x];
    sb.best = best; sb.k1 = k1_match; sb.init = initialisable ? 1 : 0;
    sb.hp[0] = hps[0]; sb.hp[1] = hps[1]; sb.hp[2] = hps[2]; sb.hp[3] = hps[3];
  }
  __syncthreads();
  if (seg > 0) return;
#pragma unroll
  for (int s = 0; s < kStereoSegs - 1; ++s) {
    const SegBest& sb = seg_best[s][threadIdx.x];
    if (sb.best < best) {  // strict: ties stay with the lower segment = lower k1
      best = sb.best; k1_match = sb.k1; initialisable = sb.init != 0;
      hps[0] = sb.hp[0]; hps[1] = sb.hp[1]; hps[2] = sb.hp[2]; hps[3] = sb.hp[3];
    }
  }
  if (active) {
    okvfe_stereo_match m;
    const bool hit = best < threshold;
    m.k1 = hit ? k1_match : -1;
    m.dist = hit ? best : threshold;
    m.initialisable = hit ? (initialisable ? 1 : 0) : 0;
    m.pad = 0;
    m.hp_W[0] = hit ? hps[0] : 0.0;
    m.hp_W[1] = hit ? hps[1] : 0.0;
    m.hp_W[2] = hit ? hps[2] : 0.0;
    m.hp_W[3] = hit ? hps[3] : 0.0;
    out[k0] = m;
  }
}

// Round 6, the kept form: POOLED gates.  The FP64 gate is ~700 instructions and runs wave-wide for whichever lanes hold a
// candidate -- a dozen of 64 in a typical segment wave, so the four waves of a block each paid a whole gate for a
// quarter-full wave, round after round.  Here every (segment, query) that holds a candidate pushes a job {segment, query,
// key} into one LDS queue per block; the jobs are gated DENSELY by the first ceil(J / 64) waves (one gate invocation
// for the block in the common case), results go back through LDS, and every (segment, query) carries on with its own
// sequence of keys exactly as before.  The segments also share, per query, the smallest key found valid so far: larger
// candidates cannot win and are not gated.  The query rays are rotated once per block.  Same decisions, same bytes.
__device__ void match_stereo_rows_pooled(const PairParams& P, const BlockView& I0, const BlockView& I1,
                                         int threshold, okvfe_stereo_match* __restrict__ out) {
  constexpr int kMore = OKVFE_MATCH_MORE;
  __shared__ uint4 seg_desc[kStereoSegs][kStereoChunk * 3];
  __shared__ uint32_t best_key[64];
  __shared__ double e0_lds[3][64];
  __shared__ uint32_t job_key[64 * kStereoSegs];
  __shared__ uint8_t job_who[64 * kStereoSegs];   // segment << 6 | query lane
  __shared__ uint8_t job_res[kStereoSegs][64];    // bit 0: valid, bit 1: parallel
  __shared__ double hp_lds[kStereoSegs][4][64];
  __shared__ int seg_out[kStereoSegs][3][64];     // best distance, k1, initialisable
  __shared__ int job_n[2];
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int lane = threadIdx.x;
  const int tid = seg * 64 + lane;
  const int per_seg = (I1.n + kStereoSegs - 1) / kStereoSegs;
  const int k1_lo = min(seg * per_seg, I1.n), k1_hi = min(k1_lo + per_seg, I1.n);
  if ((int)blockIdx.x * 64 >= I0.n) return;  // whole block past the last keypoint
  const int k0 = blockIdx.x * 64 + lane;
  const bool active = k0 < I0.n;
  Desc12 d0 = {};
  if (active) d0 = load_desc(I0.desc + (size_t)k0 * OKVFE_DESC_BYTES);
  const bool v0 = active && I0.bpv[k0] != 0;
  if (seg == 0) {
    double e[3] = {0, 0, 0};
    if (v0) {
      double v[3];
      rot(P.C0, I0.bp + 3 * (size_t)k0, v);
      normalize3(v, e);
    }
    e0_lds[0][lane] = e[0]; e0_lds[1][lane] = e[1]; e0_lds[2][lane] = e[2];
    best_key[lane] = kNoKey;
    if (lane < 2) job_n[lane] = 0;
  }
  const ScanChunk chunk{seg_desc[seg], nullptr};
  const bool resident = k1_hi - k1_lo <= kStereoChunk;
  if (resident) load_scan_chunk(chunk, I1.desc, nullptr, k1_lo, k1_hi - k1_lo);
  enum { kDone = 0, kHave = 1, kNeedScan = 2 };
  int state = v0 ? kNeedScan : kDone;  // without a back-projection the gate rejects every candidate
  uint32_t cs[kMore];
#pragma unroll
  for (int u = 0; u < kMore; ++u) cs[u] = kNoKey;
  int t = 0, n_c = 2, n_scans = 0;
  uint32_t floor_key = 0;  // keys are ((dist << 22) | k1) + 1, so 0 admits everything
  int best = threshold, k1_match = 0, initialisable = 0;
  int par = 0;
  __syncthreads();  // rays, bounds and counters are in place
  while (true) {
    // ---- scan: a wave scans when none of its lanes still holds a key (the lanes of a wave advance in step)
    if (!__any(state == kHave) && __any(state == kNeedScan)) {  // wave-uniform
      if (n_scans < OKVFE_MATCH_NARROW_SCANS) {
        scan_top2<false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, &cs[0], &cs[1]);
#pragma unroll
        for (int u = 2; u < kMore; ++u) cs[u] = kNoKey;
        n_c = 2;
      } else {
        scan_top<kMore, false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, cs);
        n_c = kMore;
      }
      ++n_scans;
      t = 0;
      if (state == kNeedScan) state = kHave;
    }
    // ---- this (segment, query)'s next key
    uint32_t cand = cs[0];
#pragma unroll
    for (int u = 1; u < kMore; ++u) cand = t == u ? cs[u] : cand;
    if (state == kHave && cand == kNoKey) state = kDone;  // nothing (more) above the floor in this segment
    // a candidate above a key that another segment has found valid cannot win, nor can its successors (a stale bound
    // only costs work: the merge at the end decides)
    if (state == kHave && cand > *reinterpret_cast<volatile uint32_t*>(&best_key[lane])) state = kDone;
    const bool pending = state == kHave;
    {
      const unsigned long long m = __ballot(pending);
      if (m != 0ull) {  // one counter update per wave
        int base = 0;
        if (lane == 0) base = atomicAdd(&job_n[par], __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (pending) {
          const int at = base + __popcll(m & ((1ull << lane) - 1ull));
          job_key[at] = cand;
          job_who[at] = (uint8_t)(seg << 6 | lane);
        }
      }
    }
    if (!__syncthreads_or(state != kDone)) break;  // (the jobs are published)
    const int J = job_n[par];
    if (J > 0) {  // block-uniform
      if (tid == 0) job_n[par ^ 1] = 0;  // next round's counter: nobody reads or pushes it before the barrier below
      if (seg * 64 < J) {  // the gates, densely: wave-uniform
        const bool on = tid < J;
        const uint32_t key = on ? job_key[tid] : 1u;
        const int who = on ? job_who[tid] : 0;
        const int jl = who & 63, js = who >> 6;
        const int k1 = (int)((key - 1u) & 0x3FFFFFu);
        bool is_valid = false, is_parallel = false;
        double hp_W[4] = {0, 0, 0, 0};
        if (on && I1.bpv[k1]) {
          const int jk0 = blockIdx.x * 64 + jl;
          const double e0_W[3] = {e0_lds[0][jl], e0_lds[1][jl], e0_lds[2][jl]};
          double v[3], e1_W[3], hp_C0[4], hp_C1[4];
          rot(P.C1, I1.bp + 3 * (size_t)k1, v);
          normalize3(v, e1_W);
          double c26, c6;
          gate_cos(P, I0.kps, jk0, I1.kps, k1, &c26, &c6);
          triangulate_fast(P.r0, e0_W, P.r1, e1_W, c26, c6, hp_W, &is_valid, &is_parallel);
          inv_transform_h(P.C0, P.r0, hp_W, hp_C0);
          inv_transform_h(P.C1, P.r1, hp_W, hp_C1);
          if (!is_parallel) {
            const double w4 = hp_W[3];
            hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
            if (hp_C0[2] / hp_C0[3] < 0.05) is_valid = false;
            if (hp_C1[2] / hp_C1[3] < 0.05) is_valid = false;
            if (dot3(e0_W, e1_W) < 0.8) is_valid = false;
          }
        }
        if (on) {
          job_res[js][jl] = (uint8_t)((is_valid ? 1 : 0) | (is_parallel ? 2 : 0));
          if (is_valid) {
            hp_lds[js][0][jl] = hp_W[0]; hp_lds[js][1][jl] = hp_W[1];
            hp_lds[js][2][jl] = hp_W[2]; hp_lds[js][3][jl] = hp_W[3];
            atomicMin(&best_key[jl], key);
          }
        }
      }
      __syncthreads();  // results are back
      if (pending) {
        const int r = job_res[seg][lane];
        floor_key = cand;
        if (r & 1) {
          best = (int)((cand - 1u) >> 22);
          k1_match = (int)((cand - 1u) & 0x3FFFFFu);
          initialisable = (r & 2) ? 0 : 1;
          state = kDone;
        } else {
          ++t;
          if (t >= n_c) state = kNeedScan;
        }
      }
      par ^= 1;
    }
  }
  seg_out[seg][0][lane] = best;
  seg_out[seg][1][lane] = k1_match;
  seg_out[seg][2][lane] = initialisable;
  __syncthreads();
  if (seg > 0 || !active) return;
  int win = 0;
#pragma unroll
  for (int s2 = 1; s2 < kStereoSegs; ++s2) {
    const int b2 = seg_out[s2][0][lane];
    if (b2 < best) {  // strict: ties stay with the lower segment = lower k1
      best = b2; k1_match = seg_out[s2][1][lane]; initialisable = seg_out[s2][2][lane];
      win = s2;
    }
  }
  okvfe_stereo_match m;
  const bool hit = best < threshold;
  m.k1 = hit ? k1_match : -1;
  m.dist = hit ? best : threshold;
  m.initialisable = hit ? initialisable : 0;
  m.pad = 0;
  m.hp_W[0] = hit ? hp_lds[win][0][lane] : 0.0;
  m.hp_W[1] = hit ? hp_lds[win][1][lane] : 0.0;
  m.hp_W[2] = hit ? hp_lds[win][2][lane] : 0.0;
  m.hp_W[3] = hit ? hp_lds[win][3][lane] : 0.0;
  out[k0] = m;
}

#ifndef OKVFE_MATCH_MERGED
#define OKVFE_MATCH_MERGED 3  // 0: per-segment gates (rounds 2-5), 1: one global candidate order (measured slower: LAB_NOTES), 2: per-segment + shared bound, 3: pooled gates (kept)
#endif
// Round 6: ONE candidate order per query.  The four segment waves of a query used to run the whole scheme each --
// the FP64 set-up of the query's ray, and a gate for the best candidates of THEIR segment even when another segment
// held a better, valid one (look-alike content: 5-15 candidates per query, most of them rejected).  Now:
//   * wave 0 rotates and normalises the query rays once, the others pick them up from LDS behind the first barrier;
//   * every wave scans its segment for its K smallest admissible keys above the query's floor (K = 2, then 6: as
//     before) and publishes them; all keys up to the HORIZON H = min over the segments of their K-th key are then
//     known to be complete (a segment that returned fewer than K keys is exhausted);
//   * the known keys are gated in GLOBAL (dist, k1) order, four per round, one per wave; the first valid one in that
//     order is the reference's match (the gate does not depend on the running best, Frontend.cpp:2027-2073) and the
//     wave that gated it writes the row from its registers; if all known keys are rejected the floor moves to H and
//     the segments are scanned again.
// Every wave derives the same per-query state from the same LDS words, so the loop conditions are block-uniform.
__device__ void match_stereo_rows_merged(const PairParams& P, const BlockView& I0, const BlockView& I1,
                                         int threshold, okvfe_stereo_match* __restrict__ out) {
  constexpr int kMore = OKVFE_MATCH_MORE;
  __shared__ uint4 seg_desc[kStereoSegs][kStereoChunk * 3];
  __shared__ uint32_t seg_keys[kStereoSegs][kMore][64];
  __shared__ double e0_lds[3][64];
  __shared__ uint8_t gate_flags[kStereoSegs][64];  // bit 0: valid, bit 1: initialisable
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int lane = threadIdx.x;
  const int per_seg = (I1.n + kStereoSegs - 1) / kStereoSegs;
  const int k1_lo = min(seg * per_seg, I1.n), k1_hi = min(k1_lo + per_seg, I1.n);
  if ((int)blockIdx.x * 64 >= I0.n) return;  // whole block past the last keypoint
  const int k0 = blockIdx.x * 64 + lane;
  const bool active = k0 < I0.n;
  Desc12 d0 = {};
  if (active) d0 = load_desc(I0.desc + (size_t)k0 * OKVFE_DESC_BYTES);
  const bool v0 = active && I0.bpv[k0] != 0;
  if (seg == 0) {
    double e[3] = {0, 0, 0};
    if (v0) {
      double v[3];
      rot(P.C0, I0.bp + 3 * (size_t)k0, v);
      normalize3(v, e);
    }
    e0_lds[0][lane] = e[0]; e0_lds[1][lane] = e[1]; e0_lds[2][lane] = e[2];
  }
  const ScanChunk chunk{seg_desc[seg], nullptr};
  const bool resident = k1_hi - k1_lo <= kStereoChunk;
  if (resident) load_scan_chunk(chunk, I1.desc, nullptr, k1_lo, k1_hi - k1_lo);
  double e0_W[3] = {0, 0, 0};
  bool have_e0 = false;
  uint32_t floor_key = 0;     // keys are ((dist << 22) | k1) + 1, so 0 admits everything
  uint32_t prev = 0;          // every known key <= prev has been gated and rejected
  uint32_t horizon = 0;       // keys <= horizon are known (kNoKey: every segment is exhausted)
  bool done = !v0;            // without a back-projection the gate rejects every candidate
  bool need_scan = !done;
  int n_keys = 2;             // keys per segment of the last scan (block-uniform)
  int n_scans = 0;
  bool wrote = false;         // this lane's row has been written by the wave that gated the winner
  while (true) {
    // ---- scan (block-uniform decision: every wave holds the same per-lane flags)
    const bool any_scan = __any(need_scan);
    if (any_scan) {
      uint32_t cs[kMore];
#pragma unroll
      for (int u = 0; u < kMore; ++u) cs[u] = kNoKey;
      // (lanes that need no scan ride along with a floor that admits nothing: their published keys stay)
      const uint32_t fl = need_scan ? floor_key : kNoKey - 1u;
      if (n_scans < OKVFE_MATCH_NARROW_SCANS) {
        scan_top2<false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, fl, (uint32_t)threshold, &cs[0], &cs[1]);
        n_keys = 2;
      } else {
        scan_top<kMore, false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, fl, (uint32_t)threshold, cs);
        n_keys = kMore;
      }
      ++n_scans;
      if (need_scan) {
#pragma unroll
        for (int u = 0; u < kMore; ++u) seg_keys[seg][u][lane] = cs[u];
      }
    }
    __syncthreads();  // keys (and, the first time, the query rays) are published; last round's flags have been read
    if (!have_e0) {
      e0_W[0] = e0_lds[0][lane]; e0_W[1] = e0_lds[1][lane]; e0_W[2] = e0_lds[2][lane];
      have_e0 = true;
    }
    if (need_scan) {
      uint32_t hz = kNoKey;
#pragma unroll
      for (int s2 = 0; s2 < kStereoSegs; ++s2) hz = min(hz, seg_keys[s2][n_keys - 1][lane]);
      horizon = hz;
      prev = floor_key;
      need_scan = false;
    }
    // ---- the next four known keys above prev, ascending (all waves compute the same four)
    uint32_t sel[kStereoSegs];
    {
      uint32_t lo = prev;
#pragma unroll
      for (int r = 0; r < kStereoSegs; ++r) {
        uint32_t m = kNoKey;
        if (!done) {
          for (int u = 0; u < n_keys; ++u) {  // block-uniform trip count
#pragma unroll
            for (int s2 = 0; s2 < kStereoSegs; ++s2) {
              const uint32_t k = seg_keys[s2][u][lane];
              m = (k > lo && k <= horizon && k < m) ? k : m;
            }
          }
        }
        sel[r] = m;
        lo = m == kNoKey ? lo : m;
      }
    }
    // ---- this wave gates candidate `seg` of the four
    uint32_t mine = sel[0];
#pragma unroll
    for (int r = 1; r < kStereoSegs; ++r) mine = seg == r ? sel[r] : mine;
    const bool pending = !done && mine != kNoKey;
    bool is_valid = false, is_parallel = false;
    double hp_W[4] = {0, 0, 0, 0};
    int k1 = 0, dist = 0;
    if (__any(pending)) {  // wave-uniform
      k1 = pending ? (int)((mine - 1u) & 0x3FFFFFu) : 0;
      dist = (int)((mine - 1u) >> 22);
      if (pending && I1.bpv[k1]) {
        double v[3], e1_W[3], hp_C0[4], hp_C1[4];
        rot(P.C1, I1.bp + 3 * (size_t)k1, v);
        normalize3(v, e1_W);
        double c26, c6;
        gate_cos(P, I0.kps, k0, I1.kps, k1, &c26, &c6);
        triangulate_fast(P.r0, e0_W, P.r1, e1_W, c26, c6, hp_W, &is_valid, &is_parallel);
        inv_transform_h(P.C0, P.r0, hp_W, hp_C0);
        inv_transform_h(P.C1, P.r1, hp_W, hp_C1);
        if (!is_parallel) {
          const double w4 = hp_W[3];
          hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
          if (hp_C0[2] / hp_C0[3] < 0.05) is_valid = false;
          if (hp_C1[2] / hp_C1[3] < 0.05) is_valid = false;
          if (dot3(e0_W, e1_W) < 0.8) is_valid = false;
        }
      }
    }
    gate_flags[seg][lane] = (uint8_t)((pending && is_valid) ? 1 : 0);
    __syncthreads();
    // ---- the first valid candidate in order wins (every wave reads the same four flags)
    int win = -1;
#pragma unroll
    for (int r = kStereoSegs - 1; r >= 0; --r) win = (!done && gate_flags[r][lane] != 0) ? r : win;
    if (win >= 0) {
      if (win == seg && active) {
        okvfe_stereo_match m;
        m.k1 = k1;
        m.dist = dist;
        m.initialisable = is_parallel ? 0 : 1;
        m.pad = 0;
        m.hp_W[0] = hp_W[0]; m.hp_W[1] = hp_W[1]; m.hp_W[2] = hp_W[2]; m.hp_W[3] = hp_W[3];
        out[k0] = m;
      }
      wrote = true;
      done = true;
    } else if (!done) {
      // all four (or fewer) rejected: move on among the known keys, or re-scan above the horizon, or give up
      uint32_t last = prev;
#pragma unroll
      for (int r = 0; r < kStereoSegs; ++r) last = sel[r] != kNoKey ? sel[r] : last;
      prev = last;
      if (sel[kStereoSegs - 1] == kNoKey) {  // the known keys are used up
        if (horizon == kNoKey) {
          done = true;  // every segment exhausted: no match
        } else {
          floor_key = horizon;
          need_scan = true;
        }
      }
    }
    if (!__any(!done)) break;  // block-uniform (identical state in the four waves)
  }
  if (seg == 0 && active && !wrote) {
    okvfe_stereo_match m;
    m.k1 = -1; m.dist = threshold; m.initialisable = 0; m.pad = 0;
    m.hp_W[0] = 0.0; m.hp_W[1] = 0.0; m.hp_W[2] = 0.0; m.hp_W[3] = 0.0;
    out[k0] = m;
  }
}

__device__ __forceinline__ void match_stereo_rows(const PairParams& P, const BlockView& I0, const BlockView& I1,
                                                  int threshold, okvfe_stereo_match* __restrict__ out) {
#if OKVFE_MATCH_MERGED == 3
  match_stereo_rows_pooled(P, I0, I1, threshold, out);
#elif OKVFE_MATCH_MERGED == 2
  match_stereo_rows_bound(P, I0, I1, threshold, out);
#elif OKVFE_MATCH_MERGED
  match_stereo_rows_merged(P, I0, I1, threshold, out);
#else
  match_stereo_rows_r5(P, I0, I1, threshold, out);  // (A/B: the per-segment form of rounds 2-5)
#endif
}

__global__ __launch_bounds__(64 * kStereoSegs) __attribute__((amdgpu_waves_per_eu(OKVFE_MATCH_WAVES, 8))) void match_stereo_kernel(
    const PairParams* __restrict__ pairs, const okvfe_keypoint* __restrict__ kps,
    const uint8_t* __restrict__ desc, const double* __restrict__ bp,
    const uint8_t* __restrict__ bpv, const int32_t* __restrict__ counts, int kp_cap,
    int threshold, okvfe_stereo_match* __restrict__ out) {
  const PairParams& P = pairs[blockIdx.y];
  BlockView I0, I1;
  const size_t o0 = (size_t)P.image0 * kp_cap, o1 = (size_t)P.image1 * kp_cap;
  I0.desc = desc + o0 * OKVFE_DESC_BYTES; I0.bp = bp + o0 * 3; I0.bpv = bpv + o0;
  I0.n = counts[P.image0]; I0.kps = kps + o0;
  I1.desc = desc + o1 * OKVFE_DESC_BYTES; I1.bp = bp + o1 * 3; I1.bpv = bpv + o1;
  I1.n = counts[P.image1]; I1.kps = kps + o1;
  match_stereo_rows(P, I0, I1, threshold, out + (size_t)blockIdx.y * kp_cap);
}

// explicit arrays (host-buffer API and gathered blocks)
__global__ __launch_bounds__(64 * kStereoSegs) __attribute__((amdgpu_waves_per_eu(4, 8))) void match_stereo_arrays_kernel(
    const PairParams* __restrict__ pair, const uint8_t* __restrict__ desc0,
    const double* __restrict__ bp0, const uint8_t* __restrict__ bpv0, const int32_t* n0p, int n0,
    const uint8_t* __restrict__ desc1, const double* __restrict__ bp1,
    const uint8_t* __restrict__ bpv1, const int32_t* n1p, int n1, int threshold,
    okvfe_stereo_match* __restrict__ out, const okvfe_keypoint* __restrict__ kp0,
    const okvfe_keypoint* __restrict__ kp1) {
  BlockView I0, I1;
  I0.desc = desc0; I0.bp = bp0; I0.bpv = bpv0; I0.n = n0p ? *n0p : n0; I0.kps = kp0;
  I1.desc = desc1; I1.bp = bp1; I1.bpv = bpv1; I1.n = n1p ? *n1p : n1; I1.kps = kp1;
  match_stereo_rows(*pair, I0, I1, threshold, out);
}

__global__ __launch_bounds__(64) void hamming_argmin_kernel(const uint8_t* __restrict__ A, int nA,
                                                            const uint8_t* __restrict__ B, int nB,
                                                            uint32_t thr,
                                                            int32_t* __restrict__ best_j,
                                                            uint32_t* __restrict__ best_d) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nA) return;
  const Desc12 a = load_desc(A + (size_t)i * OKVFE_DESC_BYTES);
  uint32_t dmin = thr;
  int jmin = -1;
  for (int j = 0; j < nB; ++j) {
    const uint32_t d =
        (uint32_t)hamming(a, reinterpret_cast<const uint32_t*>(B + (size_t)j * OKVFE_DESC_BYTES));
    if (d < dmin) {
      dmin = d;
      jmin = j;
    }
  }
  best_j[i] = jmin;
  best_d[i] = dmin;
}

__global__ __launch_bounds__(64) void hamming_count_kernel(const uint8_t* __restrict__ A, int nA,
                                                           const uint8_t* __restrict__ B, int nB,
                                                           int thr, int32_t* __restrict__ rows) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nA) return;
  const Desc12 a = load_desc(A + (size_t)i * OKVFE_DESC_BYTES);
  int c = 0;
  for (int j = 0; j < nB; ++j)
    c += hamming(a, reinterpret_cast<const uint32_t*>(B + (size_t)j * OKVFE_DESC_BYTES)) < thr;
  rows[i] = c;
}

__global__ __launch_bounds__(64) void hamming_emit_kernel(const uint8_t* __restrict__ A, int nA,
                                                          const uint8_t* __restrict__ B, int nB,
                                                          int thr,
                                                          const int32_t* __restrict__ offsets,
                                                          okvfe_candidate* __restrict__ out,
                                                          int cap) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nA) return;
  const Desc12 a = load_desc(A + (size_t)i * OKVFE_DESC_BYTES);
  int pos = offsets[i];
  for (int j = 0; j < nB; ++j) {
    const int d = hamming(a, reinterpret_cast<const uint32_t*>(B + (size_t)j * OKVFE_DESC_BYTES));
    if (d < thr) {
      if (pos < cap) {
        okvfe_candidate c;
        c.i = i;
        c.j = j;
        c.dist = d;
        out[pos] = c;
      }
      ++pos;
    }
  }
}

// ---- verifyRecognisedPlace, every landmark of one camera in ONE launch (Frontend.cpp:330-355) ----
// One wave per landmark (grid-stride): lanes stride over the K frame descriptors, each landmark
// descriptor is a wave-uniform row.  The reference's running minimum (strict <, descriptors outer,
// k inner) keeps the smallest distance and, among equals, the first in that scan order: the wave
// reduces the key (dist << 32 | scan position).  The frame descriptors are staged in LDS once per
// workgroup when they fit.
// Device-resident batches of the map matchers (okvfe_*_blocks_device): frame f = blockIdx.y takes its
// keypoints / descriptors / back-projections and its keypoint count from gather block f (the layout
// of okvfe_pack_gather_blocks_device), per-frame inputs and outputs advance by the strides below.
// blocks == nullptr is the classic single-frame call on plain arrays.
struct MapBatch {
  const uint8_t* blocks;
  int o_kps, o_desc, o_bp, block_bytes, kp_cap;
  size_t frame_stride;  // matchToMap: doubles per frame in `projections` (0 = one set for all frames)
  const int32_t* perm;  // matchToMap: [frames][kp_cap] keypoint order by image region (keypoint_order_kernel) or null
};
constexpr int kVerifyLdsRows = 1024;  // 48 KiB
__global__ __launch_bounds__(256) void verify_place_kernel(
    const uint8_t* __restrict__ pool, const int32_t* __restrict__ desc_begin, int n_landmarks,
    const uint8_t* __restrict__ frame_desc, int K, uint32_t threshold, int32_t* __restrict__ k_min,
    uint32_t* __restrict__ dist_min, MapBatch mb) {
  __shared__ uint4 lds_desc[kVerifyLdsRows * 3];
  if (mb.blocks) {  // frame blockIdx.y of a device-resident batch
    const uint8_t* base = mb.blocks + (size_t)blockIdx.y * mb.block_bytes;
    K = *reinterpret_cast<const int32_t*>(base);
    frame_desc = base + mb.o_desc;
    k_min += (size_t)blockIdx.y * n_landmarks;
    dist_min += (size_t)blockIdx.y * n_landmarks;
  }
  const bool in_lds = K <= kVerifyLdsRows;
  if (in_lds) {
    const uint4* src = reinterpret_cast<const uint4*>(frame_desc);
    for (int i = threadIdx.x; i < K * 3; i += 256) lds_desc[i] = src[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  for (int l = wave; l < n_landmarks; l += n_waves) {
    unsigned long long best = ~0ull;
    const int d0 = desc_begin[l], d1 = desc_begin[l + 1];
    for (int d = d0; d < d1; ++d) {
      const Desc12 a = load_desc(pool + (size_t)d * OKVFE_DESC_BYTES);  // same address in every lane
      for (int k = lane; k < K; k += 64) {
        const uint32_t* row = in_lds ? reinterpret_cast<const uint32_t*>(lds_desc + 3 * k)
                                     : reinterpret_cast<const uint32_t*>(frame_desc + (size_t)k * OKVFE_DESC_BYTES);
        const unsigned long long dist = (unsigned long long)hamming(a, row);
        const unsigned long long key = (dist << 32) | (unsigned long long)((d - d0) * K + k);
        best = key < best ? key : best;
      }
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
      const unsigned long long o = __shfl_xor(best, dd);
      best = o < best ? o : best;
    }
    if (lane == 0) {
      const uint32_t dist = (uint32_t)(best >> 32);
      const bool hit = best != ~0ull && dist < threshold;
      k_min[l] = hit ? (int32_t)((uint32_t)best % (uint32_t)K) : 0;
      dist_min[l] = hit ? dist : threshold;
    }
  }
}

// ---- DBoW2 vocabulary descent with the FBrisk trait (oracle: orc_voc_transform) ----------------
// Lane = one feature; the node descriptors (819 x 48 B for the shipped 9^3 vocabulary) sit in LDS
// when they fit.  At every level the child with the smallest Hamming distance wins, the first on
// ties.
constexpr int kVocLdsNodes = 1024;
__global__ __launch_bounds__(256) void voc_transform_kernel(
    const uint8_t* __restrict__ desc, int n, const uint8_t* __restrict__ node_desc, int n_nodes,
    const int32_t* __restrict__ child_begin, const int32_t* __restrict__ child_index,
    const int32_t* __restrict__ word, int32_t* __restrict__ word_out, int32_t* __restrict__ node_out) {
  __shared__ uint4 lds_nodes[kVocLdsNodes * 3];
  const bool in_lds = n_nodes <= kVocLdsNodes;
  if (in_lds) {
    const uint4* src = reinterpret_cast<const uint4*>(node_desc);
    for (int i = threadIdx.x; i < n_nodes * 3; i += 256) lds_nodes[i] = src[i];
    __syncthreads();
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Desc12 a = load_desc(desc + (size_t)i * OKVFE_DESC_BYTES);
  int node = 0;
  while (true) {
    const int c0 = child_begin[node], c1 = child_begin[node + 1];
    if (c1 <= c0) break;
    int best = -1, best_d = 0x7FFFFFFF;
    for (int c = c0; c < c1; ++c) {
      const int id = child_index[c];
      const uint32_t* row = in_lds ? reinterpret_cast<const uint32_t*>(lds_nodes + 3 * id)
                                   : reinterpret_cast<const uint32_t*>(node_desc + (size_t)id * OKVFE_DESC_BYTES);
      const int d = hamming(a, row);
      if (d < best_d) {
        best_d = d;
        best = id;
      }
    }
    node = best;
  }
  word_out[i] = word[node];
  node_out[i] = node;
}

// ---- matchMotionStereo (Frontend.cpp:1812-1905) -------------------------------------------------
// Frame 0 = older frame, frame 1 = current frame, same camera.  Lane = k0; k1 ascending over the
// current keypoints that do not carry a landmark yet (matched1[k1] == 0, the packed set of
// :1789-1802); skip0[k0] stands for the estimator-state tests at :1814-1841.  Gate order as in the
// reference: back-projection valid -> e0.e1 >= 0.5 -> triangulateFast valid -> e0.e1 >= 0.8 ->
// depths >= 0.2 unless parallel.  The winner is re-projected into the current camera and must
// land within 4 px (:1897-1905).  `cos_quality` is the cosine whose acos the reference stores
// as match quality (:1887-1889); the host takes the acos.
// One side of the motion-stereo matcher (explicit arrays or a gather block)
struct MotionView {
  const uint8_t* desc;
  const okvfe_keypoint* kps;
  const double* bp;
  const uint8_t* bpv;
  const uint8_t* flag;  // side 0: skip0 (k0 is not matched), side 1: matched1 (k1 left out); may be null
  int n;
};
struct SegBestMotion {
  double hp[4];
  double cosq;
  int best, k1, init, pad;
};

// Same scheme as match_stereo_rows (segments, LDS scan of the two best keys, converged FP64 gate):
// the gate of the reference loop (Frontend.cpp:1843-1895) does not depend on the running best, so
// its result is the gated candidate with the smallest (dist, k1).
__device__ void match_motion_rows(const PairParams& P, const DeviceCamera& camera, int w, int h,
                                  const MotionView& I0, const MotionView& I1, int threshold,
                                  okvfe_motion_match* __restrict__ out) {
  __shared__ SegBestMotion seg_best[kStereoSegs - 1][64];
  __shared__ uint4 seg_desc[kStereoSegs][kStereoChunk * 3];
  __shared__ uint8_t seg_skip[kStereoSegs][kStereoChunk];
  const int seg = threadIdx.y;
  const int per_seg = (I1.n + kStereoSegs - 1) / kStereoSegs;
  const int k1_lo = min(seg * per_seg, I1.n), k1_hi = min(k1_lo + per_seg, I1.n);
  if ((int)blockIdx.x * 64 >= I0.n) return;  // whole block past the last keypoint
  const int k0 = blockIdx.x * 64 + threadIdx.x;
  const bool in_range = k0 < I0.n;
  const bool active = in_range && !(I0.flag && I0.flag[k0]) && I0.bpv[k0] != 0;
  Desc12 d0 = {};
  double e0_W[3] = {0, 0, 0};
  if (active) {
    d0 = load_desc(I0.desc + (size_t)k0 * OKVFE_DESC_BYTES);
    double v[3];
    rot(P.C0, I0.bp + 3 * (size_t)k0, v);
    normalize3(v, e0_W);
  }
  int best = threshold;
  int k1_max = 0;
  bool initialisable = false;
  double cosq = 1.0;
  double hps[4] = {0, 0, 0, 0};
  uint32_t floor_key = 0;
  bool done = !active;
  const ScanChunk chunk{seg_desc[seg], seg_skip[seg]};
  const bool resident = k1_hi - k1_lo <= kStereoChunk;
  const bool has_skip = I1.flag != nullptr;  // kernel-uniform
  if (resident) load_scan_chunk(chunk, I1.desc, I1.flag, k1_lo, k1_hi - k1_lo);
  int n_scans = 0;
  while (__any(!done)) {
    // two keys from the first scan, six from every re-scan (see match_stereo_rows)
    constexpr int kMore = OKVFE_MATCH_MORE;
    uint32_t cs[kMore];
#pragma unroll
    for (int u = 0; u < kMore; ++u) cs[u] = kNoKey;
    int n_c = 2;
    if (n_scans < 1) {
      if (has_skip)
        scan_top2<true>(d0, chunk, I1.desc, I1.flag, k1_lo, k1_hi, resident, floor_key,
                        (uint32_t)threshold, &cs[0], &cs[1]);
      else
        scan_top2<false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key,
                         (uint32_t)threshold, &cs[0], &cs[1]);
    } else {
      if (has_skip)
        scan_top<kMore, true>(d0, chunk, I1.desc, I1.flag, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, cs);
      else
        scan_top<kMore, false>(d0, chunk, I1.desc, nullptr, k1_lo, k1_hi, resident, floor_key, (uint32_t)threshold, cs);
      n_c = kMore;
    }
    ++n_scans;
#pragma unroll 1
    for (int t = 0; t < n_c; ++t) {
      uint32_t cand = cs[0];
#pragma unroll
      for (int u = 1; u < kMore; ++u) cand = t == u ? cs[u] : cand;
      bool pending = !done;
      if (pending && cand == kNoKey) {
        done = true;
        pending = false;
      }
      if (!__any(pending)) break;  // wave-uniform
      if (!pending) continue;
      floor_key = cand;
      const int k1 = (int)((cand - 1u) & 0x3FFFFFu);
      const int dist = (int)((cand - 1u) >> 22);
      if (!I1.bpv[k1]) continue;
      double v[3], e1_W[3], hp_W[4], hp_C0[4], hp_C1[4];
      rot(P.C1, I1.bp + 3 * (size_t)k1, v);
      normalize3(v, e1_W);
      const double ee = dot3(e0_W, e1_W);
      if (ee < 0.5) continue;
      bool is_valid, is_parallel;
      double c26, c6;  // sigma = size0 / f0 * 0.125 (Frontend.cpp:1834): the table is constant in k1
      gate_cos(P, I0.kps, k0, I1.kps, k1, &c26, &c6);
      triangulate_fast(P.r0, e0_W, P.r1, e1_W, c26, c6, hp_W, &is_valid, &is_parallel);
      if (!is_valid) continue;
      inv_transform_h(P.C0, P.r0, hp_W, hp_C0);
      inv_transform_h(P.C1, P.r1, hp_W, hp_C1);
      if (ee < 0.8) is_valid = false;
      if (!is_parallel) {
        const double w4 = hp_W[3];
        hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
        if (hp_C0[2] / hp_C0[3] < 0.2) is_valid = false;
        if (hp_C1[2] / hp_C1[3] < 0.2) is_valid = false;
      }
      if (is_valid) {
        k1_max = k1;
        best = dist;
        double a[3], b[3], an[3], bn[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          a[i] = hp_W[i] - P.r0[i];
          b[i] = hp_W[i] - P.r1[i];
        }
        normalize3(a, an);
        normalize3(b, bn);
        cosq = dot3(an, bn);
        hps[0] = hp_W[0]; hps[1] = hp_W[1]; hps[2] = hp_W[2]; hps[3] = hp_W[3];
        initialisable = !is_parallel;
        done = true;
      }
    }
  }
  if (seg > 0) {
    SegBestMotion& sb = seg_best[seg - 1][threadIdx.x];
    sb.best = best; sb.k1 = k1_max; sb.init = initialisable ? 1 : 0; sb.cosq = cosq;
    sb.hp[0] = hps[0]; sb.hp[1] = hps[1]; sb.hp[2] = hps[2]; sb.hp[3] = hps[3];
  }
  __syncthreads();
  if (seg > 0) return;
#pragma unroll
  for (int sg = 0; sg < kStereoSegs - 1; ++sg) {
    const SegBestMotion& sb = seg_best[sg][threadIdx.x];
    if (sb.best < best) {  // strict: ties stay with the lower segment = lower k1
      best = sb.best; k1_max = sb.k1; initialisable = sb.init != 0; cosq = sb.cosq;
      hps[0] = sb.hp[0]; hps[1] = sb.hp[1]; hps[2] = sb.hp[2]; hps[3] = sb.hp[3];
    }
  }
  if (in_range) {
    okvfe_motion_match m;
    const bool hit = active && best < threshold;
    m.k1 = hit ? k1_max : -1;
    m.dist = hit ? best : threshold;
    m.initialisable = hit && initialisable ? 1 : 0;
    m.accepted = 0;
    m.cos_quality = hit ? cosq : 1.0;
    m.hp_W[0] = hit ? hps[0] : 0.0;
    m.hp_W[1] = hit ? hps[1] : 0.0;
    m.hp_W[2] = hit ? hps[2] : 0.0;
    m.hp_W[3] = hit ? hps[3] : 0.0;
    if (hit) {
      double hp_C1[4], head[3], pt1p[2];
      inv_transform_h(P.C1, P.r1, hps, hp_C1);
      // projectHomogeneous: PinholeCamera.hpp:493-503
      head[0] = hp_C1[3] < 0 ? -hp_C1[0] : hp_C1[0];
      head[1] = hp_C1[3] < 0 ? -hp_C1[1] : hp_C1[1];
      head[2] = hp_C1[3] < 0 ? -hp_C1[2] : hp_C1[2];
      const int status = cam::project(camera, w, h, head, pt1p);
      const double ex = (double)I1.kps[k1_max].x - pt1p[0], ey = (double)I1.kps[k1_max].y - pt1p[1];
      m.accepted = (status == 0 && sqrt(ex * ex + ey * ey) < 4.0) ? 1 : 0;
    }
    out[k0] = m;
  }
}

__global__ __launch_bounds__(64 * kStereoSegs) void match_motion_kernel(
    const PairParams* __restrict__ pair, const DeviceCamera* __restrict__ camera, int w, int h,
    const uint8_t* __restrict__ desc0, const okvfe_keypoint* __restrict__ kp0,
    const double* __restrict__ bp0, const uint8_t* __restrict__ bpv0,
    const uint8_t* __restrict__ skip0, int n0, const uint8_t* __restrict__ desc1,
    const okvfe_keypoint* __restrict__ kp1, const double* __restrict__ bp1,
    const uint8_t* __restrict__ bpv1, const uint8_t* __restrict__ matched1, int n1, int threshold,
    okvfe_motion_match* __restrict__ out) {
  const MotionView I0{desc0, kp0, bp0, bpv0, skip0, n0};
  const MotionView I1{desc1, kp1, bp1, bpv1, matched1, n1};
  match_motion_rows(*pair, *camera, w, h, I0, I1, threshold, out);
}

// ---- matchToMapByThread, 3-D landmarks (Frontend.cpp:1552-1589) ----------------------------------
// Lane = keypoint k; landmarks in the caller's order (ascending LandmarkId in the reference's
// std::map); per landmark the image-distance gate |projection - keypoint|^2 <= thr^2, then its
// <= 3 descriptors in order with the running minimum "dist < distances[k]" (strict, first-lowest
// wins).  Returns per keypoint the distance and the landmark INDEX (or -1).
// Keypoints of a frame ordered by image region (bands of 96 rows, then x): 64 consecutive entries of
// the order cover ~10 % of the image, so a wave of match_to_map_kernel can discard most landmarks by
// ONE vector test of 64 projections against its keypoints' bounding box instead of one scalar-loop
// iteration per landmark.  One workgroup per frame, bitonic network on <= 4096 keys in LDS; more
// keypoints (or none) keep the identity order.  The order only groups lanes: results do not depend on it.
constexpr int kOrderMax = 4096;
__global__ __launch_bounds__(256) void keypoint_order_kernel(MapBatch mb, int32_t* __restrict__ perm) {
  __shared__ uint64_t keys[kOrderMax];
  const uint8_t* base = mb.blocks + (size_t)blockIdx.x * mb.block_bytes;
  const int n = *reinterpret_cast<const int32_t*>(base);
  int32_t* out = perm + (size_t)blockIdx.x * mb.kp_cap;
  if (n <= 0 || n > kOrderMax) {
    for (int i = threadIdx.x; i < mb.kp_cap; i += 256) out[i] = i;
    return;
  }
  const okvfe_keypoint* kps = reinterpret_cast<const okvfe_keypoint*>(base + mb.o_kps);
  int np = 64;
  while (np < n) np <<= 1;
  for (int i = threadIdx.x; i < np; i += 256) {
    uint64_t key = ~0ull;
    if (i < n) {
      const float x = kps[i].x, y = kps[i].y;
      const uint32_t band = (uint32_t)(y > 0.0f ? (y < 65535.0f ? y : 65535.0f) : 0.0f) / 96u;
      const uint32_t xq = (uint32_t)(x > 0.0f ? (x < 65535.0f ? x : 65535.0f) : 0.0f);
      key = ((uint64_t)band << 40) | ((uint64_t)xq << 16) | (uint64_t)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < np / 2; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < mb.kp_cap; i += 256) out[i] = i < n ? (int32_t)(keys[i] & 0xFFFFu) : i;
}

// Lane = keypoint; the landmark list is cut into kMapSegs contiguous segments (one wave each) and
// walked in chunks of 64: lane j of the wave fetches landmark j's projection and descriptor range
// (coalesced), v_readlane broadcasts them, and the chunk's descriptors are staged in LDS.  The
// Hamming part runs only for landmarks that have at least one keypoint of the wave inside the
// reprojection radius.  Result = first landmark (ascending) reaching the smallest distance, as in
// the reference loop, merged over the segments by (dist, segment).
constexpr int kMapSegs = 4;
constexpr int kMapChunk = 64;
constexpr int kMapChunkDesc = 3 * kMapChunk;  // the reference keeps <= 3 descriptors per landmark

__global__ __launch_bounds__(64 * kMapSegs) void match_to_map_kernel(
    const uint8_t* __restrict__ desc_k, const okvfe_keypoint* __restrict__ kps,
    const uint8_t* __restrict__ use, int n_k, const double* __restrict__ projections,
    const int32_t* __restrict__ desc_begin, int n_lm, const uint8_t* __restrict__ pool,
    double thr_sq, int threshold, int32_t* __restrict__ best_lm, int32_t* __restrict__ best_d, MapBatch mb) {
  __shared__ uint4 seg_desc[kMapSegs][kMapChunkDesc * 3];
  __shared__ int2 seg_best[kMapSegs - 1][64];
  if (mb.blocks) {  // frame blockIdx.y of a device-resident batch
    const uint8_t* base = mb.blocks + (size_t)blockIdx.y * mb.block_bytes;
    n_k = *reinterpret_cast<const int32_t*>(base);
    desc_k = base + mb.o_desc;
    kps = reinterpret_cast<const okvfe_keypoint*>(base + mb.o_kps);
    if (use) use += (size_t)blockIdx.y * mb.kp_cap;
    projections += (size_t)blockIdx.y * mb.frame_stride;
    best_lm += (size_t)blockIdx.y * mb.kp_cap;
    best_d += (size_t)blockIdx.y * mb.kp_cap;
  }
  const int lane = threadIdx.x, seg = threadIdx.y;
  const int pos = blockIdx.x * 64 + lane;
  const bool in_range = pos < n_k;
  // region order of the frame's keypoints (device batches): lanes of a wave are image neighbours
  const int k = (in_range && mb.perm) ? mb.perm[(size_t)blockIdx.y * mb.kp_cap + pos] : pos;
  const bool active = in_range && (use == nullptr || use[k] != 0);
  Desc12 dk = {};
  double kx = 0.0, ky = 0.0;
  if (active) {
    dk = load_desc(desc_k + (size_t)k * OKVFE_DESC_BYTES);
    kx = (double)kps[k].x;
    ky = (double)kps[k].y;
  }
  // bounding box of the wave's keypoints, grown by the radius (+1 px for the float rounding): a
  // landmark outside it is near no keypoint of the wave.  NaN coordinates pass (as they pass the
  // reference's "!(dd > thr)").
  float bx0 = active ? (float)kx : INFINITY, bx1 = active ? (float)kx : -INFINITY;
  float by0 = active ? (float)ky : INFINITY, by1 = active ? (float)ky : -INFINITY;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    bx0 = fminf(bx0, __shfl_xor(bx0, d));
    bx1 = fmaxf(bx1, __shfl_xor(bx1, d));
    by0 = fminf(by0, __shfl_xor(by0, d));
    by1 = fmaxf(by1, __shfl_xor(by1, d));
  }
  const float rad = sqrtf((float)thr_sq) * 1.0001f + 1.0f;
  bx0 -= rad; by0 -= rad; bx1 += rad; by1 += rad;
  const int per_seg = (n_lm + kMapSegs - 1) / kMapSegs;
  const int l_lo = min(seg * per_seg, n_lm), l_hi = min(l_lo + per_seg, n_lm);
  uint4* chunk = seg_desc[seg];
  int best = threshold, lm = -1;
  for (int l0 = l_lo; l0 < l_hi; l0 += kMapChunk) {
    const int cnt = min(kMapChunk, l_hi - l0);
    // lane j holds landmark l0 + j
    double px = 0.0, py = 0.0;
    int b = 0, e = 0;
    if (lane < cnt) {
      px = projections[2 * (size_t)(l0 + lane)];
      py = projections[2 * (size_t)(l0 + lane) + 1];
      b = desc_begin[l0 + lane];
      e = desc_begin[l0 + lane + 1];
    }
    // ONE vector test for the 64 landmarks of the chunk: inside the wave's box?
    const float fpx = (float)px, fpy = (float)py;
    const unsigned long long cmask =
        __ballot(lane < cnt && !(fpx < bx0) && !(fpx > bx1) && !(fpy < by0) && !(fpy > by1));
    if (cmask == 0) continue;  // wave-uniform: nothing of this chunk is near the wave
    const int d_lo = __builtin_amdgcn_readfirstlane(b);
    const int d_hi = __builtin_amdgcn_readlane(e, cnt - 1);
    // descriptors of the chunk in LDS when they fit (always, with <= 3 per landmark)
    const bool staged = d_hi - d_lo <= kMapChunkDesc;
    __builtin_amdgcn_wave_barrier();
    if (staged) {
      const uint4* src = reinterpret_cast<const uint4*>(pool + (size_t)d_lo * OKVFE_DESC_BYTES);
      for (int i = lane; i < (d_hi - d_lo) * 3; i += 64) chunk[i] = src[i];
    }
    __builtin_amdgcn_wave_barrier();
    for (unsigned long long cm = cmask; cm != 0; cm &= cm - 1) {
      const int j = (int)__ffsll((long long)cm) - 1;
      const double lpx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(px), j),
                                          __builtin_amdgcn_readlane(__double2loint(px), j));
      const double lpy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(py), j),
                                          __builtin_amdgcn_readlane(__double2loint(py), j));
      const int lb = __builtin_amdgcn_readlane(b, j), le = __builtin_amdgcn_readlane(e, j);
      const double dx = lpx - kx, dy = lpy - ky;
      const double dd = dx * dx + dy * dy;
      const bool near = active && !(dd > thr_sq);
      if (!__any(near)) continue;  // wave-uniform: no keypoint of this wave near the landmark
      for (int d = lb; d < le; ++d) {
        const uint32_t* dp = staged
            ? reinterpret_cast<const uint32_t*>(chunk + 3 * (d - d_lo))
            : reinterpret_cast<const uint32_t*>(pool + (size_t)d * OKVFE_DESC_BYTES);
        const int dist = hamming(dk, dp);
        if (near && dist < best) {
          best = dist;
          lm = l0 + j;
        }
      }
    }
  }
  if (seg > 0) seg_best[seg - 1][lane] = make_int2(best, lm);
  __syncthreads();
  if (seg > 0) return;
#pragma unroll
  for (int sgm = 0; sgm < kMapSegs - 1; ++sgm) {
    const int2 o = seg_best[sgm][lane];
    if (o.x < best) {  // strict: ties stay with the lower segment = lower landmark index
      best = o.x;
      lm = o.y;
    }
  }
  if (in_range) {
    best_lm[k] = lm;
    best_d[k] = best;
  }
}


// ---- matchToMapByThreadUnitialised (Frontend.cpp:1616-1719) --------------------------------------
// Landmarks that are not 3-D yet: pooled descriptor d carries the observing ray e0_W[d] and camera
// centre r0_W[d].  Lane = keypoint k; landmark / descriptor loops are wave-uniform.  Gates in the
// reference's order: epipolar plane + divergence (unless nearly parallel) -> triangulateFast with
// sigma = 1/f -> not within 0.2 m of either centre; a hit on the landmark the keypoint already
// carries is counted and ends that landmark's descriptor loop; hp is only stored when the
// triangulation is not parallel.
__device__ __forceinline__ void cross3(const double a[3], const double b[3], double out[3]) {
  double t1, t2;
  t1 = a[1] * b[2]; t2 = a[2] * b[1]; out[0] = t1 - t2;
  t1 = a[2] * b[0]; t2 = a[0] * b[2]; out[1] = t1 - t2;
  t1 = a[0] * b[1]; t2 = a[1] * b[0]; out[2] = t1 - t2;
}

// Segments: the landmark list is cut into kUninitSegs contiguous ranges, one wave each, and every
// wave runs the reference loop over its range from best = threshold.  The gate is a pure function
// of (keypoint, pooled descriptor), so a range walked from a higher starting value accepts a
// superset: its accepted pairs have strictly decreasing distances, and the ones the sequential loop
// would have accepted with the incoming best B of the range are exactly those below B (a suffix).
// Per range and keypoint it is therefore enough to keep
//   * the last accepted pair (distance, landmark),
//   * the last accepted NON-parallel pair (distance, hp): it is the stored hp of the range iff its
//     distance is below B, otherwise the range stores none,
//   * for the landmark the keypoint already carries (never accepted, only counted): the smallest
//     gated distance of its descriptors below the range's running best at that point; the
//     sequential loop counts it iff that is below B as well,
// and to fold the ranges in order.
constexpr int kUninitSegs = 8;
struct UninitSegResult {
  int best, lm, np_dist, prev_min;
  double hp[4];
};

__global__ __launch_bounds__(64 * kUninitSegs) void match_to_map_uninit_kernel(
    const PairParams* __restrict__ pair, const uint8_t* __restrict__ desc_k,
    const double* __restrict__ bp, const uint8_t* __restrict__ use,
    const int32_t* __restrict__ previous, int n_k, const int32_t* __restrict__ desc_begin, int n_lm,
    const uint8_t* __restrict__ pool, const double* __restrict__ e0_W,
    const double* __restrict__ r0_W, int threshold, int32_t* __restrict__ best_lm,
    int32_t* __restrict__ best_d, double* __restrict__ hps_W, uint8_t* __restrict__ hp_set,
    int32_t* __restrict__ ctr_total, MapBatch mb) {
  __shared__ UninitSegResult seg_res[kUninitSegs - 1][64];
  if (mb.blocks) {  // frame blockIdx.y of a device-resident batch: its own pose record, block and rows
    const uint8_t* base = mb.blocks + (size_t)blockIdx.y * mb.block_bytes;
    pair += blockIdx.y;
    n_k = *reinterpret_cast<const int32_t*>(base);
    desc_k = base + mb.o_desc;
    bp = reinterpret_cast<const double*>(base + mb.o_bp);
    if (use) use += (size_t)blockIdx.y * mb.kp_cap;
    if (previous) previous += (size_t)blockIdx.y * mb.kp_cap;
    best_lm += (size_t)blockIdx.y * mb.kp_cap;
    best_d += (size_t)blockIdx.y * mb.kp_cap;
    hps_W += 4 * (size_t)blockIdx.y * mb.kp_cap;
    hp_set += (size_t)blockIdx.y * mb.kp_cap;
    ctr_total += blockIdx.y;
  }
  const PairParams& P = *pair;  // C1/r1 = T_WC1; cos26/cos6 for sigma = 1/f
  const int lane = threadIdx.x, seg = threadIdx.y;
  const int k = blockIdx.x * 64 + lane;
  const bool in_range = k < n_k;
  const bool active = in_range && (use == nullptr || use[k] != 0);
  Desc12 dk = {};
  double e1_W[3] = {0, 0, 0};
  int prev = -1;
  if (active) {
    dk = load_desc(desc_k + (size_t)k * OKVFE_DESC_BYTES);
    double en[3];
    normalize3(bp + 3 * (size_t)k, en);
    rot(P.C1, en, e1_W);
    prev = previous ? previous[k] : -1;
  }
  // gate of one (keypoint, pooled descriptor) pair in the reference's order; pure
  auto gate = [&](int d, double hp[4], bool* is_parallel) -> bool {
    const double* e0 = e0_W + 3 * (size_t)d;
    const double* r0 = r0_W + 3 * (size_t)d;
    const double e0v[3] = {e0[0], e0[1], e0[2]}, r0v[3] = {r0[0], r0[1], r0[2]};
    if (dot3(e0v, e1_W) < P.cos6) {
      double t[3], et[3], c0[3], c1[3], n0[3], n1[3], cx[3], nn[3], nnn[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = P.r1[i] - r0v[i];
      normalize3(t, et);
      cross3(e0v, et, c0);
      normalize3(c0, n0);
      cross3(e1_W, et, c1);
      normalize3(c1, n1);
      if (dot3(n0, n1) < P.cos6) return false;
      cross3(e0v, e1_W, cx);
#pragma unroll
      for (int i = 0; i < 3; ++i) nn[i] = n0[i] + n0[i];
      normalize3(nn, nnn);
      if (dot3(cx, nnn) > 0.0) return false;
    }
    bool is_valid;
    triangulate_fast(r0v, e0v, P.r1, e1_W, P.cos26, P.cos6, hp, &is_valid, is_parallel);
    if (!is_valid) return false;
    if (!*is_parallel) {
      double a[3], bb[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double p = hp[i] / hp[3];
        a[i] = p - r0v[i];
        bb[i] = p - P.r1[i];
      }
      if (sqrt(dot3(a, a)) < 0.2) is_valid = false;
      if (sqrt(dot3(bb, bb)) < 0.2) is_valid = false;
    }
    return is_valid;
  };
  const int per_seg = (n_lm + kUninitSegs - 1) / kUninitSegs;
  const int l_lo = min(seg * per_seg, n_lm), l_hi = min(l_lo + per_seg, n_lm);
  int best = threshold, lm = -1, np_dist = INT_MAX, prev_min = INT_MAX;
  double hps[4] = {0, 0, 0, 0};
  for (int l = l_lo; l < l_hi; ++l) {
    const int b = desc_begin[l], e = desc_begin[l + 1];
    for (int d = b; d < e; ++d) {
      const uint32_t* dd = reinterpret_cast<const uint32_t*>(pool + (size_t)d * OKVFE_DESC_BYTES);
      if (!active) continue;
      const int dist = hamming(dk, dd);
      if (dist < best) {
        double hp[4];
        bool is_parallel;
        if (!gate(d, hp, &is_parallel)) continue;
        if (l == prev) {  // counted, never accepted: `best` stays as it is for the whole landmark
          prev_min = dist < prev_min ? dist : prev_min;
          continue;
        }
        best = dist;
        lm = l;
        if (!is_parallel) {
          np_dist = dist;
          hps[0] = hp[0]; hps[1] = hp[1]; hps[2] = hp[2]; hps[3] = hp[3];
        }
      }
    }
  }
  if (seg > 0) {
    UninitSegResult& r = seg_res[seg - 1][lane];
    r.best = best; r.lm = lm; r.np_dist = np_dist; r.prev_min = prev_min;
    r.hp[0] = hps[0]; r.hp[1] = hps[1]; r.hp[2] = hps[2]; r.hp[3] = hps[3];
  }
  __syncthreads();
  if (seg > 0) return;
  // fold the ranges in landmark order; range 0 started from the threshold itself
  int ctr = prev_min < threshold ? 1 : 0;
  bool have_hp = np_dist != INT_MAX;
  for (int sg = 0; sg < kUninitSegs - 1; ++sg) {
    const UninitSegResult& r = seg_res[sg][lane];
    if (r.prev_min < best) ++ctr;
    if (r.best < best) {
      if (r.np_dist < best) {
        hps[0] = r.hp[0]; hps[1] = r.hp[1]; hps[2] = r.hp[2]; hps[3] = r.hp[3];
        have_hp = true;
      }
      best = r.best;
      lm = r.lm;
    }
  }
  if (in_range) {
    best_lm[k] = lm;
    best_d[k] = best;
    hp_set[k] = have_hp ? 1 : 0;
    hps_W[4 * (size_t)k + 0] = hps[0];
    hps_W[4 * (size_t)k + 1] = hps[1];
    hps_W[4 * (size_t)k + 2] = hps[2];
    hps_W[4 * (size_t)k + 3] = hps[3];
  }
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) ctr += __shfl_xor(ctr, dlt);
  if (lane == 0 && ctr) atomicAdd(ctr_total, ctr);
}

// ---- gather blocks (cross-camera exchange) ---------------------------------------
struct BlockOffsets {
  int o_count, o_kps, o_desc, o_bp, o_bpv, total;
};

// packs images first .. first+n-1 of the context's result arrays into n contiguous blocks
__global__ __launch_bounds__(256) void pack_blocks_kernel(BlockOffsets L, int first, int kp_cap,
                                                          const int32_t* __restrict__ counts,
                                                          const okvfe_keypoint* __restrict__ kps,
                                                          const uint8_t* __restrict__ desc,
                                                          const double* __restrict__ bp,
                                                          const uint8_t* __restrict__ bpv,
                                                          uint8_t* __restrict__ blocks) {
  const int img = first + blockIdx.x;
  uint8_t* b = blocks + (size_t)blockIdx.x * L.total;
  const size_t off = (size_t)img * kp_cap;
  const int tid = threadIdx.x;
  if (tid == 0) *reinterpret_cast<int32_t*>(b + L.o_count) = counts[img];
  const uint32_t* s32;
  uint32_t* d32;
  s32 = reinterpret_cast<const uint32_t*>(kps + off);
  d32 = reinterpret_cast<uint32_t*>(b + L.o_kps);
  for (int i = tid; i < kp_cap * 7; i += 256) d32[i] = s32[i];
  s32 = reinterpret_cast<const uint32_t*>(desc + off * OKVFE_DESC_BYTES);
  d32 = reinterpret_cast<uint32_t*>(b + L.o_desc);
  for (int i = tid; i < kp_cap * 12; i += 256) d32[i] = s32[i];
  s32 = reinterpret_cast<const uint32_t*>(bp + off * 3);
  d32 = reinterpret_cast<uint32_t*>(b + L.o_bp);
  for (int i = tid; i < kp_cap * 6; i += 256) d32[i] = s32[i];
  for (int i = tid; i < kp_cap; i += 256) b[L.o_bpv + i] = bpv[off + i];
}

// matches frame f of two gathered block arrays: grid (rows, frames)
__global__ __launch_bounds__(64 * kStereoSegs) void match_stereo_blocks_kernel(
    const PairParams pair, BlockOffsets L, const uint8_t* __restrict__ blocks0,
    const uint8_t* __restrict__ blocks1, int kp_cap, int threshold,
    okvfe_stereo_match* __restrict__ out) {
  const uint8_t* b0 = blocks0 + (size_t)blockIdx.y * L.total;
  const uint8_t* b1 = blocks1 + (size_t)blockIdx.y * L.total;
  BlockView I0, I1;
  I0.desc = b0 + L.o_desc; I0.bp = reinterpret_cast<const double*>(b0 + L.o_bp); I0.bpv = b0 + L.o_bpv;
  I0.n = *reinterpret_cast<const int32_t*>(b0 + L.o_count);
  I0.kps = reinterpret_cast<const okvfe_keypoint*>(b0 + L.o_kps);
  I1.desc = b1 + L.o_desc; I1.bp = reinterpret_cast<const double*>(b1 + L.o_bp); I1.bpv = b1 + L.o_bpv;
  I1.n = *reinterpret_cast<const int32_t*>(b1 + L.o_count);
  I1.kps = reinterpret_cast<const okvfe_keypoint*>(b1 + L.o_kps);
  match_stereo_rows(pair, I0, I1, threshold, out + (size_t)blockIdx.y * kp_cap);
}

// motion-stereo matcher on two gathered blocks (older frame, current frame) of one camera
__global__ __launch_bounds__(64 * kStereoSegs) void match_motion_blocks_kernel(
    const PairParams pair, const DeviceCamera* __restrict__ camera, int w, int h,
    BlockOffsets L, const uint8_t* __restrict__ b0, const uint8_t* __restrict__ b1,
    const uint8_t* __restrict__ skip0, const uint8_t* __restrict__ matched1, int threshold,
    okvfe_motion_match* __restrict__ out) {
  MotionView I0, I1;
  I0.desc = b0 + L.o_desc; I0.kps = reinterpret_cast<const okvfe_keypoint*>(b0 + L.o_kps);
  I0.bp = reinterpret_cast<const double*>(b0 + L.o_bp); I0.bpv = b0 + L.o_bpv; I0.flag = skip0;
  I0.n = *reinterpret_cast<const int32_t*>(b0 + L.o_count);
  I1.desc = b1 + L.o_desc; I1.kps = reinterpret_cast<const okvfe_keypoint*>(b1 + L.o_kps);
  I1.bp = reinterpret_cast<const double*>(b1 + L.o_bp); I1.bpv = b1 + L.o_bpv; I1.flag = matched1;
  I1.n = *reinterpret_cast<const int32_t*>(b1 + L.o_count);
  match_motion_rows(pair, *camera, w, h, I0, I1, threshold, out);
}


// ---- DBoW2 database query, L1 scoring (TemplatedDatabase::queryL1 behind Frontend.cpp:756-766) ----
// One thread per database entry: merge-join of the entry's BowVector with the query's (both in
// ascending word order), value += |q - d| - |q| - |d| over the common words in that order -- the
// order in which the reference's inverted-file walk reaches this entry -- then score = -value / 2.
// A few hundred words per vector, a few thousand entries: latency-bound, one small launch.
__global__ __launch_bounds__(256) void bow_query_l1_kernel(const int32_t* __restrict__ db_begin,
                                                           const int32_t* __restrict__ db_ids,
                                                           const double* __restrict__ db_values, int n_entries,
                                                           const int32_t* __restrict__ q_ids,
                                                           const double* __restrict__ q_values, int n_q,
                                                           double* __restrict__ scores) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_entries) return;
  int i = db_begin[e];
  const int i_end = db_begin[e + 1];
  int j = 0;
  double value = 0.0;
  bool any = false;
  while (i < i_end && j < n_q) {
    const int a = db_ids[i], b = q_ids[j];
    if (a == b) {
      const double d = db_values[i], q = q_values[j];
      double t = fabs(q - d);
      t = t - fabs(q);
      t = t - fabs(d);
      value = value + t;
      any = true;
      ++i;
      ++j;
    } else if (a < b) {
      ++i;
    } else {
      ++j;
    }
  }
  scores[e] = any ? -value / 2.0 : -1.0;
}
}  // namespace

void launch_match_motion_blocks(const PairParams& pair, const DeviceCamera* camera, int w, int h,
                                const int offs[6], const uint8_t* block0, const uint8_t* block1,
                                const uint8_t* skip0, const uint8_t* matched1, int kp_cap,
                                int threshold, okvfe_motion_match* out, hipStream_t stream) {
  const BlockOffsets L{offs[0], offs[1], offs[2], offs[3], offs[4], offs[5]};
  hipLaunchKernelGGL(match_motion_blocks_kernel, dim3((kp_cap + 63) / 64), dim3(64, kStereoSegs), 0,
                     stream, pair, camera, w, h, L, block0, block1, skip0, matched1, threshold, out);
}

void launch_pack_blocks(const int offs[6], int first, int n, int kp_cap, const int32_t* counts,
                        const okvfe_keypoint* kps, const uint8_t* desc, const double* bp,
                        const uint8_t* bpv, uint8_t* blocks, hipStream_t stream) {
  if (n <= 0) return;
  const BlockOffsets L{offs[0], offs[1], offs[2], offs[3], offs[4], offs[5]};
  hipLaunchKernelGGL(pack_blocks_kernel, dim3(n), dim3(256), 0, stream, L, first, kp_cap, counts, kps,
                     desc, bp, bpv, blocks);
}

void launch_match_stereo_blocks(const PairParams& pair, const int offs[6], const uint8_t* blocks0,
                                const uint8_t* blocks1, int n_frames, int kp_cap, int threshold,
                                okvfe_stereo_match* out, hipStream_t stream) {
  if (n_frames <= 0) return;
  const BlockOffsets L{offs[0], offs[1], offs[2], offs[3], offs[4], offs[5]};
  hipLaunchKernelGGL(match_stereo_blocks_kernel, dim3((kp_cap + 63) / 64, n_frames), dim3(64, kStereoSegs), 0,
                     stream, pair, L, blocks0, blocks1, kp_cap, threshold, out);
}

void launch_match_to_map_uninit(const PairParams* pair, const uint8_t* desc_k, const double* bp,
                                const uint8_t* use, const int32_t* previous, int n_k,
                                const int32_t* desc_begin, int n_lm, const uint8_t* pool,
                                const double* e0_W, const double* r0_W, int threshold,
                                int32_t* best_lm, int32_t* best_d, double* hps_W, uint8_t* hp_set,
                                int32_t* ctr_total, hipStream_t stream) {
  if (n_k <= 0) return;
  hipLaunchKernelGGL(match_to_map_uninit_kernel, dim3((n_k + 63) / 64), dim3(64, kUninitSegs), 0, stream, pair,
                     desc_k, bp, use, previous, n_k, desc_begin, n_lm, pool, e0_W, r0_W, threshold,
                     best_lm, best_d, hps_W, hp_set, ctr_total, MapBatch{});
}
void launch_match_to_map_uninit_blocks(const PairParams* pairs, const int offs[6], const uint8_t* blocks,
                                       int n_frames, int kp_cap, const uint8_t* use, const int32_t* previous,
                                       const int32_t* desc_begin, int n_lm, const uint8_t* pool,
                                       const double* e0_W, const double* r0_W, int threshold, int32_t* best_lm,
                                       int32_t* best_d, double* hps_W, uint8_t* hp_set, int32_t* ctr_total,
                                       hipStream_t stream) {
  if (n_frames <= 0 || kp_cap <= 0) return;
  const MapBatch mb{blocks, offs[1], offs[2], offs[3], offs[5], kp_cap, 0, nullptr};
  hipLaunchKernelGGL(match_to_map_uninit_kernel, dim3((kp_cap + 63) / 64, n_frames), dim3(64, kUninitSegs), 0,
                     stream, pairs, nullptr, nullptr, use, previous, 0, desc_begin, n_lm, pool, e0_W, r0_W,
                     threshold, best_lm, best_d, hps_W, hp_set, ctr_total, mb);
}

void launch_match_motion(const PairParams* pair, const DeviceCamera* camera, int w, int h,
                         const uint8_t* desc0, const okvfe_keypoint* kp0, const double* bp0,
                         const uint8_t* bpv0, const uint8_t* skip0, int n0, const uint8_t* desc1,
                         const okvfe_keypoint* kp1, const double* bp1, const uint8_t* bpv1,
                         const uint8_t* matched1, int n1, int threshold, okvfe_motion_match* out,
                         hipStream_t stream) {
  if (n0 <= 0) return;
  hipLaunchKernelGGL(match_motion_kernel, dim3((n0 + 63) / 64), dim3(64, kStereoSegs), 0, stream, pair, camera, w,
                     h, desc0, kp0, bp0, bpv0, skip0, n0, desc1, kp1, bp1, bpv1, matched1, n1,
                     threshold, out);
}

void launch_match_to_map(const uint8_t* desc_k, const okvfe_keypoint* kps, const uint8_t* use, int n_k,
                         const double* projections, const int32_t* desc_begin, int n_lm,
                         const uint8_t* pool, double thr_sq, int threshold, int32_t* best_lm,
                         int32_t* best_d, hipStream_t stream) {
  if (n_k <= 0) return;
  hipLaunchKernelGGL(match_to_map_kernel, dim3((n_k + 63) / 64), dim3(64, kMapSegs), 0, stream, desc_k, kps,
                     use, n_k, projections, desc_begin, n_lm, pool, thr_sq, threshold, best_lm,
                     best_d, MapBatch{});
}
void launch_match_to_map_blocks(const int offs[6], const uint8_t* blocks, int n_frames, int kp_cap,
                                const uint8_t* use, const double* projections, size_t proj_stride,
                                const int32_t* desc_begin, int n_lm, const uint8_t* pool, double thr_sq,
                                int threshold, int32_t* best_lm, int32_t* best_d, int32_t* perm_ws,
                                hipStream_t stream) {
  if (n_frames <= 0 || kp_cap <= 0) return;
  MapBatch mb{blocks, offs[1], offs[2], offs[3], offs[5], kp_cap, proj_stride, nullptr};
  if (perm_ws) {  // [n_frames][kp_cap] workspace: order the frames' keypoints by image region first
    hipLaunchKernelGGL(keypoint_order_kernel, dim3(n_frames), dim3(256), 0, stream, mb, perm_ws);
    mb.perm = perm_ws;
  }
  hipLaunchKernelGGL(match_to_map_kernel, dim3((kp_cap + 63) / 64, n_frames), dim3(64, kMapSegs), 0, stream,
                     nullptr, nullptr, use, 0, projections, desc_begin, n_lm, pool, thr_sq, threshold, best_lm,
                     best_d, mb);
}

void launch_match_stereo(const PairParams* pairs, int n_pairs, const okvfe_keypoint* kps,
                         const uint8_t* desc, const double* bp, const uint8_t* bpv,
                         const int32_t* counts, int kp_cap, int threshold,
                         okvfe_stereo_match* out, hipStream_t stream) {
  if (n_pairs <= 0) return;
  hipLaunchKernelGGL(match_stereo_kernel, dim3((kp_cap + 63) / 64, n_pairs), dim3(64, kStereoSegs), 0, stream,
                     pairs, kps, desc, bp, bpv, counts, kp_cap, threshold, out);
}

void launch_match_stereo_arrays(const PairParams* pair, const uint8_t* desc0, const double* bp0,
                                const uint8_t* bpv0, const int32_t* n0p, int n0,
                                const uint8_t* desc1, const double* bp1, const uint8_t* bpv1,
                                const int32_t* n1p, int n1, int max_rows, int threshold,
                                okvfe_stereo_match* out, hipStream_t stream, const okvfe_keypoint* kp0,
                                const okvfe_keypoint* kp1) {
  if (max_rows <= 0) return;
  hipLaunchKernelGGL(match_stereo_arrays_kernel, dim3((max_rows + 63) / 64), dim3(64, kStereoSegs), 0, stream,
                     pair, desc0, bp0, bpv0, n0p, n0, desc1, bp1, bpv1, n1p, n1, threshold, out, kp0, kp1);
}

void launch_verify_place(const uint8_t* pool, const int32_t* desc_begin, int n_landmarks,
                         const uint8_t* frame_desc, int K, uint32_t threshold, int32_t* k_min,
                         uint32_t* dist_min, hipStream_t stream) {
  if (n_landmarks <= 0) return;
  const int blocks = std::min((n_landmarks + 3) / 4, 2048);
  hipLaunchKernelGGL(verify_place_kernel, dim3(blocks), dim3(256), 0, stream, pool, desc_begin, n_landmarks,
                     frame_desc, K, threshold, k_min, dist_min, MapBatch{});
}
void launch_verify_place_blocks(const uint8_t* pool, const int32_t* desc_begin, int n_landmarks, const int offs[6],
                                const uint8_t* blocks, int n_frames, int kp_cap, uint32_t threshold,
                                int32_t* k_min, uint32_t* dist_min, hipStream_t stream) {
  if (n_landmarks <= 0 || n_frames <= 0) return;
  const int blocks_x = std::min((n_landmarks + 3) / 4, 2048);
  const MapBatch mb{blocks, offs[1], offs[2], offs[3], offs[5], kp_cap, 0, nullptr};
  hipLaunchKernelGGL(verify_place_kernel, dim3(blocks_x, n_frames), dim3(256), 0, stream, pool, desc_begin,
                     n_landmarks, nullptr, 0, threshold, k_min, dist_min, mb);
}
void launch_bow_query_l1(const int32_t* db_begin, const int32_t* db_ids, const double* db_values, int n_entries,
                         const int32_t* q_ids, const double* q_values, int n_q, double* scores,
                         hipStream_t stream) {
  if (n_entries <= 0) return;
  hipLaunchKernelGGL(bow_query_l1_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, stream, db_begin, db_ids,
                     db_values, n_entries, q_ids, q_values, n_q, scores);
}

void launch_voc_transform(const uint8_t* desc, int n, const uint8_t* node_desc, int n_nodes,
                          const int32_t* child_begin, const int32_t* child_index, const int32_t* word,
                          int32_t* word_out, int32_t* node_out, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(voc_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, desc, n, node_desc,
                     n_nodes, child_begin, child_index, word, word_out, node_out);
}

void launch_hamming_argmin(const uint8_t* A, int nA, const uint8_t* B, int nB, uint32_t thr,
                           int32_t* best_j, uint32_t* best_d, hipStream_t stream) {
  if (nA <= 0) return;
  hipLaunchKernelGGL(hamming_argmin_kernel, dim3((nA + 63) / 64), dim3(64), 0, stream, A, nA, B,
                     nB, thr, best_j, best_d);
}

void launch_hamming_count(const uint8_t* A, int nA, const uint8_t* B, int nB, int thr,
                          int32_t* row_counts, hipStream_t stream) {
  if (nA <= 0) return;
  hipLaunchKernelGGL(hamming_count_kernel, dim3((nA + 63) / 64), dim3(64), 0, stream, A, nA, B,
                     nB, thr, row_counts);
}

void launch_hamming_emit(const uint8_t* A, int nA, const uint8_t* B, int nB, int thr,
                         const int32_t* row_offsets, okvfe_candidate* out, int cap,
                         hipStream_t stream) {
  if (nA <= 0) return;
  hipLaunchKernelGGL(hamming_emit_kernel, dim3((nA + 63) / 64), dim3(64), 0, stream, A, nA, B, nB,
                     thr, row_offsets, out, cap);
}

// writes the 3-term-sum order into this translation unit's device flag (current device); synchronous
bool set_fp64_tree_match(int tree) {
  const int v = tree != 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fp64_tree), &v, sizeof(v)) == hipSuccess;
}

}  // namespace okvfe
