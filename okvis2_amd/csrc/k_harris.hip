// k_harris.hip -- K1: Harris corner score map, u8 image -> int32 score (gfx950).
//
// Replaces the whole-image passes of brisk::HarrisScoreCalculator (behind
// cv::FeatureDetector::detect, okvis_cv/include/okvis/implementation/Frame.hpp:152; detector
// built at okvis_frontend/src/Frontend.cpp:2406-2409) by ONE streaming pass:
//   Scharr (3,10,3) gradients -> gx^2, gy^2, gx*gy >> 14 (16-bit covariance entries, zero on the
//   image rim) -> 3x3 binomial sum -> det - (trace/4)^2, zero on the rim.
//
// Roofline: HBM.  Algorithmic bytes per pixel = 1 (u8 in) + 4 (int32 out) = 5.
// Mapping: one wave = a 256-pixel-wide column strip (4 px per lane, one aligned dword load per
// lane per row = 256 B coalesced per wave, one 16 B store per lane per row = 1 KiB per wave);
// the wave walks kTH rows down the strip keeping a rolling 3-row window of pixels and of
// horizontally smoothed covariance entries in registers, so every pixel is fetched once per
// strip (+4 halo rows per kTH) and nothing is staged through LDS.  The inner loop is branch-free:
// rim handling is an AND with per-lane column masks and a wave-uniform row mask.  All products
// fit 24 bits (|g| <= 4080, entries <= 16256), so the multiplies are the full-rate
// v_mul_i32_i24 / v_mad_i32_i24 (emitted explicitly: the compiler otherwise falls back to the
// quarter-rate v_mul_lo_u32 for loop-carried values whose range it cannot prove).
#include <cstdlib>

#include "okvfe_internal.h"

namespace okvfe {

namespace {

constexpr int kTH = 32;  // output rows per wave
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ int mul24(int a, int b) {
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mad24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int mad24_10(int a, int c) {  // a * 10 + c
  int d;
  asm("v_mad_i32_i24 %0, %1, 10, %2" : "=v"(d) : "v"(a), "v"(c));
  return d;
}
__device__ __forceinline__ int mul24_3(int a) {
  int d;
  asm("v_mul_i32_i24 %0, 3, %1" : "=v"(d) : "v"(a));
  return d;
}

// raw pixels of one row: ALIGNED = the three aligned dwords covering columns x0-4 .. x0+7,
// otherwise 8 clamped byte loads packed into two dwords (columns x0-2 .. x0+5)
template <bool ALIGNED>
__device__ __forceinline__ void load_raw(const uint8_t* __restrict__ img, int w, int h, int row,
                                         int x0, int dl, int dc, int dr, uint32_t raw[3]) {
  row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);  // wave-uniform
  if (ALIGNED) {
    const uint32_t* rq = reinterpret_cast<const uint32_t*>(img + (size_t)row * (size_t)w);
    raw[0] = rq[dl];
    raw[1] = rq[dc];
    raw[2] = rq[dr];
  } else {
    const uint8_t* rp = img + (size_t)row * (size_t)w;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int c0 = x0 - 2 + i, c1 = x0 + 2 + i;
      c0 = c0 < 0 ? 0 : (c0 > w - 1 ? w - 1 : c0);
      c1 = c1 < 0 ? 0 : (c1 > w - 1 ? w - 1 : c1);
      lo |= (uint32_t)rp[c0] << (8 * i);
      hi |= (uint32_t)rp[c1] << (8 * i);
    }
    raw[0] = lo;
    raw[1] = hi;
    raw[2] = 0;
  }
}

// raw -> pixels of columns x0-2 .. x0+5
template <bool ALIGNED>
__device__ __forceinline__ void unpack(const uint32_t raw[3], int p[8]) {
  if (ALIGNED) {
    const uint32_t L = raw[0], C = raw[1], R = raw[2];
    p[0] = (L >> 16) & 255;
    p[1] = (L >> 24);
    p[2] = C & 255;
    p[3] = (C >> 8) & 255;
    p[4] = (C >> 16) & 255;
    p[5] = (C >> 24);
    p[6] = R & 255;
    p[7] = (R >> 8) & 255;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[i] = (raw[0] >> (8 * i)) & 255;
      p[4 + i] = (raw[1] >> (8 * i)) & 255;
    }
  }
}

// Covariance entries of row g at columns x0-1 .. x0+4 from pixel rows a (g-1), b (g), c (g+1),
// then the horizontal binomial [1 2 1] for the 4 output columns -> hs[3][4].
// cmask[i] = -1 where column x0-1+i is inside 1..w-2 (else 0); rmask = -1 for 1 <= g <= h-2.
__device__ __forceinline__ void cov_row(const int a[8], const int b[8], const int c[8],
                                        const int cmask[6], int rmask, int hs[3][4]) {
  int vs[8], vd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    vs[i] = mad24_10(b[i], mul24_3(a[i] + c[i]));  // vertical (3,10,3)
    vd[i] = c[i] - a[i];                           // vertical difference
  }
  int gxx[6], gyy[6], gxy[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int m = cmask[i] & rmask;
    const int gx = (vs[i + 2] - vs[i]) & m;
    const int gy = mad24_10(vd[i + 1], mul24_3(vd[i] + vd[i + 2])) & m;
    gxx[i] = mul24(gx, gx) >> 14;
    gyy[i] = mul24(gy, gy) >> 14;
    gxy[i] = mul24(gx, gy) >> 14;  // arithmetic shift = floor
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hs[0][i] = gxx[i] + 2 * gxx[i + 1] + gxx[i + 2];
    hs[1][i] = gyy[i] + 2 * gyy[i + 1] + gyy[i + 2];
    hs[2][i] = gxy[i] + 2 * gxy[i + 1] + gxy[i + 2];
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(64 * kWavesPerBlock) void harris_generic_kernel(
    const uint8_t* __restrict__ images, int w, int h, int32_t* __restrict__ scores) {
  const int lane = threadIdx.x;
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ys = (blockIdx.y * kWavesPerBlock + wave) * kTH;
  if (ys >= h || x0 >= w) return;
  const int ye = ys + kTH < h ? ys + kTH : h;
  const size_t img_off = (size_t)blockIdx.z * (size_t)w * (size_t)h;
  const uint8_t* img = images + img_off;
  int32_t* out = scores + img_off + x0;

  // per-lane constants: clamped dword indices and rim masks
  const int nd = w >> 2;
  int dc = x0 >> 2;
  dc = dc > nd - 1 ? nd - 1 : dc;
  const int dl = dc > 0 ? dc - 1 : 0;
  const int dr = dc < nd - 1 ? dc + 1 : nd - 1;
  int cmask[6], smask[4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int col = x0 - 1 + i;
    cmask[i] = (col >= 1 && col <= w - 2) ? -1 : 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) smask[i] = cmask[i + 1];  // same predicate for output column x0+i

  int pr[3][8];        // rolling pixel rows
  int hsr[3][3][4];    // rolling horizontally smoothed entries
  uint32_t raw[3][3];  // pixel rows in flight: loads are issued two steps ahead of their use
  {
    uint32_t t0[3], t1[3];
    load_raw<ALIGNED>(img, w, h, ys - 2, x0, dl, dc, dr, t0);
    load_raw<ALIGNED>(img, w, h, ys - 1, x0, dl, dc, dr, t1);
    load_raw<ALIGNED>(img, w, h, ys, x0, dl, dc, dr, raw[0]);
    load_raw<ALIGNED>(img, w, h, ys + 1, x0, dl, dc, dr, raw[1]);
    unpack<ALIGNED>(t0, pr[0]);
    unpack<ALIGNED>(t1, pr[1]);
  }

  // step j: consume pixel row ys+j (loaded two steps ago), prefetch row ys+j+2, produce covariance
  // row g = ys+j-1; from j >= 2 on the score row y = g-1 is complete.  Unrolled by 3 so that all
  // rolling-buffer slots are compile-time.
  auto step = [&](int j, int s_new, int s_a, int s_b, int h_new, int h_a, int h_b) {
    const int r = ys + j;
    load_raw<ALIGNED>(img, w, h, r + 2, x0, dl, dc, dr, raw[s_new]);  // raw slot (j+2)%3: free
    unpack<ALIGNED>(raw[s_a], pr[s_new]);                           // raw slot of row r   == j%3
    const int g = r - 1;
    const int rmask = (g >= 1 && g <= h - 2) ? -1 : 0;  // wave-uniform
    cov_row(pr[s_a], pr[s_b], pr[s_new], cmask, rmask, hsr[h_new]);
    const int y = g - 1;
    if (j >= 2 && y < ye) {  // wave-uniform
      const int ymask = (y >= 1 && y <= h - 2) ? -1 : 0;
      int sc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int A = hsr[h_a][0][i] + 2 * hsr[h_b][0][i] + hsr[h_new][0][i];
        const int B = hsr[h_a][1][i] + 2 * hsr[h_b][1][i] + hsr[h_new][1][i];
        const int Cc = hsr[h_a][2][i] + 2 * hsr[h_b][2][i] + hsr[h_new][2][i];
        const int tq = ((A >> 1) + (B >> 1)) >> 1;
        const int det = mul24(A, B) - mul24(Cc, Cc);
        sc[i] = (det - mul24(tq, tq)) & (smask[i] & ymask);
      }
      int32_t* op = out + (size_t)y * (size_t)w;
      if (ALIGNED) {
        *reinterpret_cast<int4*>(op) = make_int4(sc[0], sc[1], sc[2], sc[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (x0 + i < w) op[i] = sc[i];
      }
    }
  };

  // slots: pixel row ys+j lives in pr slot (j+2)%3 and its raw dwords in raw slot j%3
  const int jn = (ye - ys) + 2;
  for (int j = 0; j < jn; j += 3) {
    step(j, 2, 0, 1, 0, 1, 2);
    if (j + 1 < jn) step(j + 1, 0, 1, 2, 1, 2, 0);
    if (j + 2 < jn) step(j + 2, 1, 2, 0, 2, 0, 1);
  }
}


// ---- fast path (w % 4 == 0): no redundant halo columns ------------------------------------------
// Lane l of strip s owns dword d = 62*s + l of every row (4 pixels).  Horizontal neighbours come
// from the adjacent lanes through DPP wave shifts, so each lane computes the vertical filters and
// the covariance entries of its own 4 columns only.  Lane 0 / lane 63 of interior strip borders
// have no neighbour on one side: they are halo lanes (strips overlap by 2 dwords) and do not
// store; at the image border no halo is needed (the rim is zero by definition).  752 px = 188
// dwords = 63 + 62 + 63 valid lanes in 3 waves.
__device__ __forceinline__ int mulhi24(int a, int b) {  // (a * b) >> 32 of the 48-bit product
  int d;
  asm("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// DPP reads of a VGPR written by a VALU op need 2 wait states; the compiler inserts them for
// its own instructions but cannot see through inline asm, so values produced by mulhi24 pass
// through this fence before they are shifted across lanes.
__device__ __forceinline__ void dpp_fence(int& a, int& b, int& c, int& d, int& e, int& f) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void dpp_fence2(int& a, int& b) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ int from_left(int v) {   // value of lane-1
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_right(int v) {  // value of lane+1
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}

constexpr int kStripLanes = 62;
#ifndef OKVFE_K1_WAVES
#define OKVFE_K1_WAVES __attribute__((amdgpu_waves_per_eu(6, 8)))
#endif

// NMS = true fuses the detector's non-maximum suppression (K2) into the same pass: the wave also
// computes the score rows ys-1 and ye (not stored), keeps the previous score row and the horizontal
// 3-maxima of the last two rows in registers, and tests every centre row against
// max(8 neighbours, thr) while it is still in registers -- the score map is then never read back
// from HBM by the detector (a pure-read pass over it costs as much as this whole kernel).  Hits are
// shifted into one 32-bit mask per column (v_cmp + v_addc_co), and written out after the row loop
// through one slot reservation per wave.  Where two horizontally adjacent pixels both pass (equal
// scores) the raster-scan rule of the reference needs the finished score row of the neighbouring
// strips, so those candidates are flagged and settled by nms_fixup_kernel (k_nms.hip).
struct NmsOut {
  int thr;
  Candidate* cand;
  int cand_cap;
  int32_t* cand_count;
  int32_t* fix_count;
};

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
// m = (m << 1) | (c >= nb)
__device__ __forceinline__ void push_hit(uint32_t& m, int c, int nb) {
  asm volatile("v_cmp_ge_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
               : "+v"(m)
               : "v"(c), "v"(nb)
               : "vcc");
}

template <int kTHF, bool NMS>
__global__ __launch_bounds__(64 * kWavesPerBlock) OKVFE_K1_WAVES void harris_kernel(
    const uint8_t* __restrict__ images, int w, int h, int32_t* __restrict__ scores, int strips,
    int ytiles, int n_images, NmsOut nms) {
  const int lane = threadIdx.x;
  int image, tile;
  xcd_tile(strips * ytiles, n_images, &image, &tile);
  const int ytile = tile / strips;
  const int strip = tile - ytile * strips;
  const int nd = w >> 2;
  const int d = strip * kStripLanes + lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ys_own = (ytile * kWavesPerBlock + wave) * kTHF;
  if (ys_own >= h) return;  // wave-uniform; all 64 lanes of a live wave stay active (DPP sources)
  const int ye_own = ys_own + kTHF < h ? ys_own + kTHF : h;
  // score rows computed by this wave: with NMS one more above and below the rows it owns
  const int ys = NMS ? ys_own - 1 : ys_own;
  const int ye = NMS ? ye_own + 1 : ye_own;
  const bool last_strip = strip * kStripLanes + 64 >= nd;
  const bool store = d < nd && (strip == 0 || lane >= 1) && (last_strip || lane <= kStripLanes);
  const int dcl = d < nd ? d : nd - 1;  // clamped dword index for loads
  const size_t img_off = (size_t)image * (size_t)w * (size_t)h;
  // buffer resources (SGPR base + 32-bit per-lane offset): no 64-bit VALU address arithmetic in
  // the row loop; the row offset rides in the scalar offset operand
  const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(images + img_off), 0, w * h, 0x00027000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      scores + img_off, 0, w * h * 4, 0x00027000);
  const int ld_off = dcl * 4;    // byte offset of this lane's dword within a pixel row
  const int st_off = dcl * 16;   // byte offset of this lane's 4 scores within a score row
  int m0 = d == 0 ? 0 : -1;        // column 0 is rim
  int m3 = d == nd - 1 ? 0 : -1;   // column w-1 is rim
  // keep the masks as opaque VGPR values: "x & m" then stays a 2-cycle v_and_b32 instead of being
  // rewritten into a v_cndmask_b32_e64 on a re-materialised compare
  asm volatile("" : "+v"(m0), "+v"(m3));
  const int k3 = 3 << 9, k10 = 10 << 9;  // gradients carry a factor 2^9: mulhi24 then yields >> 14

  auto load_row = [&](int row) -> uint32_t {
    row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(img_rsrc, ld_off, row * w, 0);
  };
  auto unpack4 = [](uint32_t c, int p[4]) {
    p[0] = c & 255;
    p[1] = (c >> 8) & 255;
    p[2] = (c >> 16) & 255;
    p[3] = c >> 24;
  };

  int pr[3][4];      // rolling pixel rows (own 4 columns)
  // stream 0 carries the xx and yy entries PACKED (xx | yy << 16: both are non-negative and every
  // partial sum stays below 2^16, so one 32-bit add smooths two channels); stream 1 carries xy
  int hs[2][2][4];   // horizontally smoothed entries: current / previous row
  int vp[2][2][4];   // vertical pair sums hs[g-1] + hs[g]
  uint32_t raw[3];   // pixel rows in flight (loaded two steps ahead)
  {
    const uint32_t t0 = load_row(ys - 2), t1 = load_row(ys - 1);
    raw[0] = load_row(ys);
    raw[1] = load_row(ys + 1);
    unpack4(t0, pr[0]);
    unpack4(t1, pr[1]);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) hs[q][c][i] = vp[q][c][i] = 0;

  // NMS state: scores + edge neighbours of the previous row, horizontal 3-max of the last two
  int nc[4], nh[2][4], nl = 0, nr = 0;
  uint32_t hits[4] = {0u, 0u, 0u, 0u};
  const int yt0 = ys_own > 2 ? ys_own : 2;                  // tested centre rows [yt0, yt1)
  const int yt1 = ye_own < h - 2 ? ye_own : h - 2;
  if (NMS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nc[i] = nh[0][i] = nh[1][i] = 0;
  }

  // step j: consume pixel row r = ys+j, covariance row g = r-1, score row y = g-1 (from j >= 2)
  auto step = [&](int j, int s_new, int s_a, int s_b, int q) {
    const int r = ys + j;
    raw[s_new] = load_row(r + 2);   // raw slot (j+2)%3 is free
    unpack4(raw[s_a], pr[s_new]);   // raw slot j%3 holds row r
    const int g = r - 1;
    const int* a = pr[s_a];
    const int* b = pr[s_b];
    const int* c = pr[s_new];
    int (*H)[4] = hs[q];
    int (*Hp)[4] = hs[q ^ 1];
    int (*V)[4] = vp[q];
    int (*Vp)[4] = vp[q ^ 1];
    if (g >= 1 && g <= h - 2) {  // wave-uniform
      int vs[4], vd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vs[i] = __mul24(b[i], k10) + __mul24(a[i] + c[i], k3);
        vd[i] = c[i] - a[i];
      }
      // NOTE: the subtrahend of gx[0] is shifted in NEGATED form and added: hipcc folds
      // "x - dpp(y)" into v_subrev_u32_dpp wave_shr:1, which does not shift on gfx950
      // (tools/ubench/dpp_test.hip); v_add_u32_dpp / v_sub_u32_dpp(dpp - x) are fine.
      const int vs_l_neg = from_left(-vs[3]), vs_r = from_right(vs[0]);
      const int vd_l = from_left(vd[3]), vd_r = from_right(vd[0]);
      int gx[4], gy[4];
      gx[0] = (vs[1] + vs_l_neg) & m0;
      gx[1] = vs[2] - vs[0];
      gx[2] = vs[3] - vs[1];
      gx[3] = (vs_r - vs[2]) & m3;
      // __mul24: 24-bit multiplies (v_mul_i32_i24 / v_mad_i32_i24, 4 cycles); a plain int
      // expression here is lowered to v_mad_u64_u32, which is several times slower
      gy[0] = (__mul24(vd[0], k10) + __mul24(vd_l + vd[1], k3)) & m0;
      gy[1] = __mul24(vd[1], k10) + __mul24(vd[0] + vd[2], k3);
      gy[2] = __mul24(vd[2], k10) + __mul24(vd[1] + vd[3], k3);
      gy[3] = (__mul24(vd[3], k10) + __mul24(vd[2] + vd_r, k3)) & m3;
      int G[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gxx = mulhi24(gx[i], gx[i]);
        const int gyy = mulhi24(gy[i], gy[i]);
        G[0][i] = gxx | (gyy << 16);
        G[1][i] = mulhi24(gx[i], gy[i]);
      }
      dpp_fence2(G[1][0], G[1][3]);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int gl = from_left(G[ch][3]), gr = from_right(G[ch][0]);
        const int pm = gl + G[ch][0];
        const int p0 = G[ch][0] + G[ch][1];
        const int p1 = G[ch][1] + G[ch][2];
        const int p2 = G[ch][2] + G[ch][3];
        const int p3 = G[ch][3] + gr;
        H[ch][0] = pm + p0;
        H[ch][1] = p0 + p1;
        H[ch][2] = p1 + p2;
        H[ch][3] = p2 + p3;
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) H[ch][i] = 0;
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) V[ch][i] = Hp[ch][i] + H[ch][i];
    const int y = g - 1;
    if (j >= 2 && y < ye) {  // wave-uniform
      int sc[4];
      if (y >= 1 && y <= h - 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned AB = (unsigned)(Vp[0][i] + V[0][i]);  // A | B << 16
          const int Cc = Vp[1][i] + V[1][i];
          const int tq = (int)((((AB >> 1) & 0x7FFFu) + (AB >> 17)) >> 1);  // ((A>>1)+(B>>1))>>1
          int ab;  // A * B straight from the packed halves (SDWA word selects): no unpacking
          asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 "
              "src1_sel:WORD_1"
              : "=v"(ab)
              : "v"(AB));
          sc[i] = ab - mad24(tq, tq, mul24(Cc, Cc));
        }
        sc[0] &= m0;
        sc[3] &= m3;
      } else {
        sc[0] = sc[1] = sc[2] = sc[3] = 0;
      }
      if (store && (!NMS || (y >= ys_own && y < ye_own))) {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i v = {sc[0], sc[1], sc[2], sc[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, st_off, y * w * 4, 0);
      }
      if (NMS) {
        // rows y-2 (nh[q]), y-1 (nc, nl, nr; nh[q^1]) and y (sc) -> test centre row y-1
        const int l = from_left(sc[3]), r2 = from_right(sc[0]);
        const int hn[4] = {max3i(l, sc[0], sc[1]), max3i(sc[0], sc[1], sc[2]),
                           max3i(sc[1], sc[2], sc[3]), max3i(sc[2], sc[3], r2)};
        const int yc = y - 1;
        if (yc >= yt0 && yc < yt1) {  // wave-uniform
          const int* c = nc;
          const int lft[4] = {nl, c[0], c[1], c[2]};
          const int rgt[4] = {c[1], c[2], c[3], nr};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            push_hit(hits[i], c[i], max3i(max3i(nh[q][i], hn[i], lft[i]), rgt[i], nms.thr));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nh[q][i] = hn[i];
          nc[i] = sc[i];
        }
        nl = l;
        nr = r2;
      }
    }
  };

  // pixel slots have period 3, the hs/vp ping-pong period 2 -> unroll by 6
  const int jn = (ye - ys) + 2;
  for (int j = 0; j < jn; j += 6) {
    step(j, 2, 0, 1, 0);
    if (j + 1 < jn) step(j + 1, 0, 1, 2, 1);
    if (j + 2 < jn) step(j + 2, 1, 2, 0, 0);
    if (j + 3 < jn) step(j + 3, 2, 0, 1, 1);
    if (j + 4 < jn) step(j + 4, 0, 1, 2, 0);
    if (j + 5 < jn) step(j + 5, 1, 2, 0, 1);
  }

  if (NMS) {
    const int tests = yt1 - yt0;  // bit b of hits[] <-> row yt0 + tests - 1 - b
    if (tests <= 0) return;
    const int x0 = dcl * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)  // columns 0, 1, w-2, w-1 are never maxima
      if (x0 + i < 2 || x0 + i >= w - 2) hits[i] = 0u;
    const uint32_t adj = (hits[0] & hits[1]) | (hits[1] & hits[2]) | (hits[2] & hits[3]) |
                         (hits[3] & (uint32_t)from_right((int)hits[0]));
    uint32_t rows_adj = 0u;
    if (__builtin_expect(__any(adj != 0u), 0)) {
      rows_adj = adj;
#pragma unroll
      for (int dd = 32; dd > 0; dd >>= 1) rows_adj |= (uint32_t)__shfl_xor((int)rows_adj, dd);
    }
    if (!store) hits[0] = hits[1] = hits[2] = hits[3] = 0u;  // halo lanes only fed the neighbours
    const int cnt = __popc(hits[0]) + __popc(hits[1]) + __popc(hits[2]) + __popc(hits[3]);
    if (!__any(cnt != 0)) return;
    int incl = cnt;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int t = __shfl_up(incl, dd);
      if (lane >= dd) incl += t;
    }
    const int total = __shfl(incl, 63);
    int base = 0;
    if (lane == 0) base = atomicAdd(&nms.cand_count[image], total);
    int pos = __shfl(base, 0) + incl - cnt;
    Candidate* outc = nms.cand + (size_t)image * nms.cand_cap;
    __builtin_amdgcn_s_waitcnt(0);  // this wave's own score stores have reached L2
    int flagged = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t mm = hits[i];
      while (mm) {
        const int b = __ffs((int)mm) - 1;
        mm &= mm - 1;
        const int y = yt0 + tests - 1 - b;
        Candidate cd;
        cd.x = x0 + i;
        cd.y = y;
        if ((rows_adj >> b) & 1u) {  // to be settled by nms_fixup_kernel
          cd.y |= kCandidateFixupFlag;
          ++flagged;
        }
        cd.score = __builtin_amdgcn_raw_buffer_load_b32(out_rsrc, st_off + 4 * i + y * w * 4, 0, 1);
        if (pos < nms.cand_cap) outc[pos] = cd;
        ++pos;
      }
    }
    if (rows_adj != 0u && __any(flagged != 0)) {
      if (flagged) atomicAdd(&nms.fix_count[image], flagged);
    }
  }
}

}  // namespace

static bool launch_harris_impl(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                               const NmsOut* nms, hipStream_t stream) {
  if (n_images <= 0) return true;
  const dim3 block(64, kWavesPerBlock, 1);
  const bool aligned = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0) &&
                       ((reinterpret_cast<uintptr_t>(score) & 15) == 0);
  if (aligned) {
    const int nd = w >> 2;
    int strips = 1;
    while ((strips - 1) * kStripLanes + 64 < nd) ++strips;  // last strip must reach dword nd-1
    static const int th_env = [] {
      const char* e = getenv("OKVFE_K1_TH");  // tuning knob: output rows per wave
      return e ? atoi(e) : 0;
    }();
#define OKVFE_K1_LAUNCH(TH)                                                                   \
  {                                                                                           \
    const int ytiles = (h + TH * kWavesPerBlock - 1) / (TH * kWavesPerBlock);                 \
    if (nms)                                                                                  \
      hipLaunchKernelGGL((harris_kernel<TH, true>), dim3(strips * ytiles * n_images), block, 0, \
                         stream, img, w, h, score, strips, ytiles, n_images, *nms);           \
    else                                                                                      \
      hipLaunchKernelGGL((harris_kernel<TH, false>), dim3(strips * ytiles * n_images), block, 0, \
                         stream, img, w, h, score, strips, ytiles, n_images, NmsOut{});       \
  }
    const int th = th_env ? th_env : (h % 30 == 0 ? 30 : 32);
    if (nms && th > 32) return false;  // one hit bit per row in a 32-bit mask
    switch (th) {
      case 16: OKVFE_K1_LAUNCH(16); break;
      case 24: OKVFE_K1_LAUNCH(24); break;
      case 30: OKVFE_K1_LAUNCH(30); break;
      case 40: OKVFE_K1_LAUNCH(40); break;
      case 60: OKVFE_K1_LAUNCH(60); break;
      case 80: OKVFE_K1_LAUNCH(80); break;
      case 120: OKVFE_K1_LAUNCH(120); break;
      case 160: OKVFE_K1_LAUNCH(160); break;
      case 240: OKVFE_K1_LAUNCH(240); break;
      default: OKVFE_K1_LAUNCH(32); break;
    }
#undef OKVFE_K1_LAUNCH
  } else {
    if (nms) return false;
    const dim3 grid((w + 255) / 256, (h + kTH * kWavesPerBlock - 1) / (kTH * kWavesPerBlock),
                    n_images);
    hipLaunchKernelGGL(harris_generic_kernel<false>, grid, block, 0, stream, img, w, h, score);
  }
  return true;
}

void launch_harris(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                   hipStream_t stream) {
  (void)launch_harris_impl(img, w, h, n_images, score, nullptr, stream);
}

bool launch_harris_nms(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                       int abs_threshold, Candidate* cand, int cand_cap, int32_t* cand_count,
                       int32_t* fix_count, hipStream_t stream) {
  static const bool off = getenv("OKVFE_NO_FUSED_NMS") != nullptr;  // A/B knob for profiling
  if (off) return false;
  const NmsOut nms{abs_threshold, cand, cand_cap, cand_count, fix_count};
  return launch_harris_impl(img, w, h, n_images, score, &nms, stream);
}

}  // namespace okvfe
