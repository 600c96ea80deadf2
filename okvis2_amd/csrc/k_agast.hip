// k_agast.hip -- AGAST / FAST 9-16 corner score map (okvfe_config.score_type = OKVFE_SCORE_AGAST_9_16) and the
// FAST 5-8 map of the published BRISK scale space's virtual first layer (OKVFE_SCORE_BRISK_SCALESPACE).
//
// Score calculator of brisk::BriskFeatureDetector, the detector the reference instantiates on ARM
// (okvis_cv/test/TestFrame.cpp:71-72: BriskFeatureDetector(34, 2)); the x86 path and every shipped
// configuration use the Harris calculator (k_harris.hip).  The brisk library is not vendored with
// the reference, so this follows the PUBLISHED predicate (FAST-9 on the 16-pixel Bresenham circle
// of radius 3; AGAST evaluates the same predicate with a different decision tree): corner at
// threshold t iff 9 contiguous circle pixels are all > p + t or all < p - t; score = the largest
// such t = max(max_s min_k (c - p), max_s min_k (p - c)) - 1, clamped at 0; 0 within 3 px of the
// border.  The rest of the detector (NMS, scale-space maxima, uniformity, cap, sub-pixel) is the
// shared pipeline; the oracle counterpart is orc_agast_score (oracle/orc_detect.c).
//
// One workgroup walks a column of 64 x 16-pixel tiles staged in LDS with a 3-pixel apron; a thread scores 4
// pixels of one image column per tile.  Three observations carry the kernel (round 4; 144 -> ~50 VALU
// instructions per pixel, 0.75 -> 0.38-0.40 ms per 512 EuRoC images):
//  * min over an arc of (c - p) = (min over the arc of c) - p: the centre is subtracted ONCE, after the
//    arc minima, so a circle pixel's operand is the same for every centre that reads it;
//  * gfx950 has 3-input packed minima / maxima for half floats only (v_pk_minimum3_f16 / v_pk_maximum3_f16).
//    A byte v written as the half 0x6400 | v is the number 1024 + v, exactly; 0xE400 | v is -(1024 + v).  The tile
//    is staged as one dword per pixel, (0x6400 | c) | (0xE400 | c) << 16 (ONE v_perm_b32 per pixel): a packed
//    minimum takes the arc minimum of c (bright test) in the low half and minus the arc MAXIMUM of c (dark
//    test) in the high half;
//  * the 16 windows of 9 are built from 3-windows: m3[i] = min3(c[i], c[i+1], c[i+2]), m9[i] = min3(m3[i],
//    m3[i+3], m3[i+6]); their maximum over the 16 starts by a max3 tree: 32 + 8 packed instructions per pixel
//    for both polarities.  best - centre (one packed subtraction) = [bright, dark].
// While a tile is scored out of one LDS buffer, the source dwords of the next one are in flight to registers and
// are expanded into the other buffer afterwards (one barrier per tile).  HBM traffic is the same 1 B in + 4 B
// out per pixel as the Harris kernel.
#include "okvfe_internal.h"

namespace okvfe {
namespace {

constexpr int kTileW = 64, kTileH = 16, kApron = 3;
constexpr int kMaxChunks = 8;  // tiles a workgroup walks at most (16 x 8 and 48 x 5 rows measured equal, 32 x 4 2 % slower)
constexpr int kLdsPitch = 72;  // dwords (one per pixel): columns x0 - 4 .. x0 + 67 (>= kTileW + 2 * kApron)
constexpr int kLdsX = 4;       // LDS column of image column x0: the row starts one aligned source dword to the left
constexpr int kStageRows = kTileH + 2 * kApron;
constexpr int kStageDwords = kStageRows * (kLdsPitch / 4);  // source dwords per tile
constexpr int kStageRounds = (kStageDwords + 255) / 256;
constexpr uint32_t kHalfBias = 0xe4006400u;   // byte-tile form: low half 1024 + c, high half -(1024 + c)
constexpr uint32_t kPermBias = 0xe4646464u;   // bytes 0..2 = 0x64, byte 3 = 0xe4: the exponent bytes v_perm_b32 picks

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// v_pk_minimum3_f16 / v_pk_maximum3_f16 (the compiler forms them from the nested two-input intrinsics)
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
  const half2_t x = __builtin_bit_cast(half2_t, a), y = __builtin_bit_cast(half2_t, b), z = __builtin_bit_cast(half2_t, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
  const half2_t x = __builtin_bit_cast(half2_t, a), y = __builtin_bit_cast(half2_t, b), z = __builtin_bit_cast(half2_t, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
// best - centre in both halves = [max_s min_k (c_k - p), max_s min_k (p - c_k)] (the high halves are negatives);
// the score is max(bright, dark, 1) - 1: small integers, exact in half precision
__device__ __forceinline__ int score_of(uint32_t best, uint32_t centre) {
  const half2_t bd = __builtin_bit_cast(half2_t, best) - __builtin_bit_cast(half2_t, centre);
  const _Float16 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(bd.x, bd.y), (_Float16)1.0f);
  return (int)(unsigned short)(short)(m - (_Float16)1.0f);
}

// rows y0 - 3 .. y0 + kTileH + 2 of the strip's columns into one LDS buffer, in two steps: the loads (issued
// before the previous tile is scored) and the expansion + LDS writes (after it).  Coordinates outside the image
// are clamped (those values only reach pixels whose score is 0 by the border rule).
struct StageGeom {
  int row[kStageRounds];       // tile row of the thread's dword in round r (kStageRows: none)
  uint32_t col4[kStageRounds]; // byte offset of the clamped source dword in its image row
  uint32_t lds[kStageRounds];  // dword index of the four packed pixels in a buffer
};

__device__ __forceinline__ void stage_geom(StageGeom& g, int w, int x0, int tid) {
  const int ndw = w >> 2;
#pragma unroll
  for (int r = 0; r < kStageRounds; ++r) {
    const int i = tid + 256 * r;
    const int row = i / (kLdsPitch / 4), c = i - row * (kLdsPitch / 4);
    int dq = (x0 >> 2) - 1 + c;
    dq = dq < 0 ? 0 : (dq > ndw - 1 ? ndw - 1 : dq);
    g.row[r] = i < kStageDwords ? row : -1;
    g.col4[r] = 4u * dq;
    g.lds[r] = row * kLdsPitch + 4 * c;
  }
}

__device__ __forceinline__ void stage_load(uint32_t (&px)[kStageRounds], const StageGeom& g,
                                           const uint8_t* __restrict__ img, int w, int h, int y0) {
#pragma unroll
  for (int r = 0; r < kStageRounds; ++r) {
    if (g.row[r] >= 0) {
      int yy = y0 - kApron + g.row[r];
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      px[r] = *reinterpret_cast<const uint32_t*>(img + ((uint32_t)yy * (uint32_t)w + g.col4[r]));
    }
  }
}

__device__ __forceinline__ void stage_store(const uint32_t (&px)[kStageRounds], const StageGeom& g, uint32_t* buf) {
#pragma unroll
  for (int r = 0; r < kStageRounds; ++r) {
    if (g.row[r] >= 0) {
      uint4 v;
      v.x = __builtin_amdgcn_perm(kPermBias, px[r], 0x07000400u);
      v.y = __builtin_amdgcn_perm(kPermBias, px[r], 0x07010401u);
      v.z = __builtin_amdgcn_perm(kPermBias, px[r], 0x07020402u);
      v.w = __builtin_amdgcn_perm(kPermBias, px[r], 0x07030403u);
      *reinterpret_cast<uint4*>(buf + g.lds[r]) = v;
    }
  }
}

// widths or bases that are not dword-aligned: byte by byte
__device__ __forceinline__ void stage_bytes(uint32_t* buf, const uint8_t* __restrict__ img, int w, int h, int x0, int y0,
                                            int tid) {
  for (int i = tid; i < kStageRows * (kTileW + 2 * kApron); i += 256) {
    const int row = i / (kTileW + 2 * kApron), c = i - row * (kTileW + 2 * kApron);
    int yy = y0 - kApron + row, xx = x0 - kApron + c;
    yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
    const uint32_t v = img[(size_t)yy * w + xx];
    buf[row * kLdsPitch + c + kLdsX - kApron] = v | (v << 16) | kHalfBias;
  }
}

// The thread's pixels of one tile: rows y, y + 4, ... of its column; c0 = the first one's centre in the LDS tile,
// off = its byte offset in the image's score map.
// FULL: all rows of the tile are inside the image -- one basic block (the border rule is a select, not a
// branch: every address read is inside the staged tile), so that the next pixel's LDS reads are issued under
// this pixel's minima.
template <bool FULL, bool F58>
__device__ __forceinline__ void score_rows(const uint32_t* c0, int32_t* __restrict__ out, uint32_t off, int w, int h,
                                           int y_first, bool x_inner) {
  constexpr int kB = F58 ? 1 : 3;  // pixels closer to the border score 0
#pragma unroll
  for (int k = 0; k < kTileH / 4; ++k) {
    const int y = y_first + 4 * k;  // uniform over the wave
    if (!FULL && y >= h) break;
    const uint32_t* c = c0 + 4 * k * kLdsPitch;
    uint32_t best;
    if (F58) {
      // ring of radius 1 in the oracle's order: (0,1) (1,1) (1,0) (1,-1) (0,-1) (-1,-1) (-1,0) (-1,1); arcs of 5:
      // m5[i] = min(m3[i], m3[i + 2])
      uint32_t d[8];
      d[0] = c[kLdsPitch];       d[1] = c[kLdsPitch + 1];   d[2] = c[1];   d[3] = c[-kLdsPitch + 1];
      d[4] = c[-kLdsPitch];      d[5] = c[-kLdsPitch - 1];  d[6] = c[-1];  d[7] = c[kLdsPitch - 1];
      uint32_t m3[8], m5[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) m3[i] = pk_min3(d[i], d[(i + 1) & 7], d[(i + 2) & 7]);
#pragma unroll
      for (int i = 0; i < 8; ++i) m5[i] = pk_min3(m3[i], m3[(i + 2) & 7], m3[(i + 2) & 7]);
      const uint32_t t0 = pk_max3(m5[0], m5[1], m5[2]), t1 = pk_max3(m5[3], m5[4], m5[5]);
      best = pk_max3(t0, t1, pk_max3(m5[6], m5[7], m5[7]));
    } else {
    // circle in the order of the oracle's table: (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3)
    // (0,-3) (-1,-3) (-2,-2) (-3,-1) (-3,0) (-3,1) (-2,2) (-1,3)
    uint32_t d[16];
    d[0] = c[3 * kLdsPitch + 0];    d[1] = c[3 * kLdsPitch + 1];
    d[2] = c[2 * kLdsPitch + 2];    d[3] = c[1 * kLdsPitch + 3];
    d[4] = c[3];                    d[5] = c[-1 * kLdsPitch + 3];
    d[6] = c[-2 * kLdsPitch + 2];   d[7] = c[-3 * kLdsPitch + 1];
    d[8] = c[-3 * kLdsPitch + 0];   d[9] = c[-3 * kLdsPitch - 1];
    d[10] = c[-2 * kLdsPitch - 2];  d[11] = c[-1 * kLdsPitch - 3];
    d[12] = c[-3];                  d[13] = c[1 * kLdsPitch - 3];
    d[14] = c[2 * kLdsPitch - 2];   d[15] = c[3 * kLdsPitch - 1];
    uint32_t m3[16], m9[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m3[i] = pk_min3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m9[i] = pk_min3(m3[i], m3[(i + 3) & 15], m3[(i + 6) & 15]);
    // low half: 1024 + max over arcs of the arc minimum of c; high half: -(1024 + min over arcs of the arc maximum)
    const uint32_t t0 = pk_max3(m9[0], m9[1], m9[2]), t1 = pk_max3(m9[3], m9[4], m9[5]);
    const uint32_t t2 = pk_max3(m9[6], m9[7], m9[8]), t3 = pk_max3(m9[9], m9[10], m9[11]);
    const uint32_t t4 = pk_max3(m9[12], m9[13], m9[14]);
    const uint32_t u0 = pk_max3(t0, t1, t2), u1 = pk_max3(t3, t4, m9[15]);
    best = __builtin_bit_cast(
        uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(half2_t, u0), __builtin_bit_cast(half2_t, u1)));
    }
    const int s = score_of(best, c[0]);
    // streamed: the map is 4 B per pixel of a batch that does not fit any cache; a 32-bit byte offset keeps the
    // store on the scalar-base form
    __builtin_nontemporal_store((x_inner && y >= kB && y < h - kB) ? s : 0,
                                reinterpret_cast<int32_t*>(reinterpret_cast<char*>(out) + off));
    off += 16u * (uint32_t)w;
  }
}

template <bool DWORDS, bool F58>
__global__ __launch_bounds__(256) void agast_score_kernel(const uint8_t* __restrict__ images, int w, int h,
                                                          int32_t* __restrict__ scores, int tiles_x,
                                                          int strips_y, int chunks, int n_images) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][kStageRows * kLdsPitch];
  int image, t;
  xcd_tile(tiles_x * strips_y, n_images, &image, &t);
  const int ty0 = t / tiles_x, tx0 = t - ty0 * tiles_x;
  const int x0 = tx0 * kTileW;
  const uint8_t* img = images + (size_t)image * w * h;
  int32_t* out = scores + (size_t)image * w * h;
  const int tid = threadIdx.x;
  const int tx = tid & 63;
  const int tyy = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's row phase: uniform
  const int x = x0 + tx;
  const bool x_inner = F58 ? (x >= 1 && x < w - 1) : (x >= 3 && x < w - 3);
  const int y_first = ty0 * chunks * kTileH;
  int n_chunks = (h - y_first + kTileH - 1) / kTileH;
  n_chunks = n_chunks < chunks ? n_chunks : chunks;
  StageGeom geom;
  uint32_t px[kStageRounds];
  if (DWORDS) {
    stage_geom(geom, w, x0, tid);
    stage_load(px, geom, img, w, h, y_first);
    stage_store(px, geom, tile[0]);
  } else {
    stage_bytes(tile[0], img, w, h, x0, y_first, tid);
  }
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int y0 = y_first + ch * kTileH;
    const bool more = ch + 1 < n_chunks;
    if (DWORDS && more) stage_load(px, geom, img, w, h, y0 + kTileH);
    const uint32_t* c0 = tile[ch & 1] + (tyy + kApron) * kLdsPitch + tx + kLdsX;
    if (x < w) {
      const uint32_t off = (uint32_t)(y0 + tyy) * (uint32_t)w + (uint32_t)x;
      if (y0 + kTileH <= h)
        score_rows<true, F58>(c0, out, 4u * off, w, h, y0 + tyy, x_inner);
      else
        score_rows<false, F58>(c0, out, 4u * off, w, h, y0 + tyy, x_inner);
    }
    if (more) {
      if (DWORDS)
        stage_store(px, geom, tile[(ch + 1) & 1]);
      else
        stage_bytes(tile[(ch + 1) & 1], img, w, h, x0, y0 + kTileH, tid);
    }
    __syncthreads();
  }
}

}  // namespace

template <bool F58>
void launch_ring_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream) {
  if (n_images <= 0) return;
  const int tiles_x = (w + kTileW - 1) / kTileW, tiles_y = (h + kTileH - 1) / kTileH;
  // tiles a workgroup walks: as many as leave >= 8192 workgroups, at most kMaxChunks (128 rows)
  int chunks = (int)(((long long)tiles_x * tiles_y * n_images) / 8192);
  chunks = chunks < 1 ? 1 : (chunks > kMaxChunks ? kMaxChunks : chunks);
  const int strips_y = (tiles_y + chunks - 1) / chunks;
  const dim3 grid(tiles_x * strips_y * n_images);
  if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0)
    hipLaunchKernelGGL((agast_score_kernel<true, F58>), grid, dim3(256), 0, stream, img, w, h, score, tiles_x,
                       strips_y, chunks, n_images);
  else
    hipLaunchKernelGGL((agast_score_kernel<false, F58>), grid, dim3(256), 0, stream, img, w, h, score, tiles_x,
                       strips_y, chunks, n_images);
}

// FAST 5-8 score of c0: the virtual intra-octave below the first octave of the published BRISK scale space
// (oracle: orc_fast58_score): the same kernel on the 8-pixel ring of radius 1, arcs of 5, 1-pixel border
void launch_fast58_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream) {
  launch_ring_score<true>(img, w, h, n_images, score, stream);
}

void launch_agast_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream) {
  launch_ring_score<false>(img, w, h, n_images, score, stream);
}

}  // namespace okvfe
