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
// fit 24 bits (|g| <= 4080, entries <= 16256), so the multiplies are v_mul_i32_i24 /
// v_mad_i32_i24 (emitted explicitly: a fused multiply-add is one issue slot where the compiler's
// v_mul_lo_u32 + add is two; the plain multiply itself issues at the same 4.4 cycles on this part,
// profiles/round2_valu_rate_ubench.txt).
//
// The production kernel is harris_kernel<61, true[, PACK]> further down: the same walk with the
// gradients on the matrix pipe (v_mfma_i32_4x4x4i8), the detector's non-maximum suppression fused in
// (hits pushed as {score, position} on per-lane LDS stacks, candidate records written through an LDS
// stage), whole-line stores through the slotted score layout, and its MEMONLY instantiation -- the
// byte mover that bench.py times next to it.  harris_generic_kernel / harris_kernel<30, false> are the
// plain score-map forms (unaligned widths, okvfe_harris_score_device).
#include <cstdlib>
#include <type_traits>

#include "okvfe_internal.h"

namespace okvfe {

namespace {

constexpr int kTH = 32;  // output rows per wave
#ifndef OKVFE_K1_WPB
#define OKVFE_K1_WPB 4
#endif
constexpr int kWavesPerBlock = OKVFE_K1_WPB;

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
__device__ __forceinline__ void dpp_fence4(int& a, int& b, int& c, int& d) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void dpp_fence2(int& a, int& b) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b));
}
// Neighbour reads with their wait states INSIDE the asm (round 6).  A DPP read of a VGPR needs two wait states behind the
// VALU write of that VGPR; the compiler keeps them by counting instructions, and it counts a v_mfma scheduled in between
// as one -- but on gfx950 the matrix instruction issues BESIDE the vector ALU when its pipe is free, so "v_xor; s_...;
// v_mfma; v_mov_dpp" can reach the DPP one state early.  Whether it does depends on what else runs on the SIMD: the
// map-writing PACK instantiation did it beside the generic score kernel of another scale-space layer (lanes 12..15 of
// every 16 -- DPP bank 3, the last quarter of a row to be written -- read the stale register: ~50 wrong score-map
// entries per call, tools/lab/ss_dump.py), never alone.  l = `a` of lane - 1, r = `b` of lane + 1, 0 beyond the wave.
__device__ __forceinline__ void lane_neighbours(int a, int b, int& l, int& r) {
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(l), "=&v"(r)
      : "v"(a), "v"(b));
}
__device__ __forceinline__ int from_left(int v) {   // value of lane-1
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_right(int v) {  // value of lane+1
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}

constexpr int kStripLanes = 62;
// Gradients run on the matrix pipe (v_mfma_i32_4x4x4i8): bit-exact, and with the slotted score layout
// 0.658 -> 0.635 ms per 1536 EuRoC images against the vector-ALU form (LAB_NOTES.md).
#ifndef OKVFE_K1_WAVES
#define OKVFE_K1_WAVES __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif

// NMS = true fuses the detector's non-maximum suppression (K2) into the same pass: the wave also
// computes the score rows ys-1 and ye (not stored), keeps the previous score row and the horizontal
// 3-maxima of the last two rows in registers, and tests every centre row against
// max(8 neighbours, thr) while it is still in registers -- the score map is then never read back
// from HBM by the detector (a pure-read pass over it costs as much as this whole kernel).  A lane
// that owns a hit pushes {score, row << 2 | column} on its LDS stack right at the test; after the row
// loop the stacks are copied out as candidate records through one slot reservation per wave.  Where
// two horizontally adjacent pixels both pass (equal scores) the raster-scan rule of the reference
// needs the finished score row of the neighbouring strips, so the candidates of such a row are
// flagged and settled by nms_fixup_kernel (k_nms.hip).
struct NmsOut {
  int thr;
  Candidate* cand;
  int cand_cap;
  int32_t* cand_count;
  int32_t* fix_count;
  int32_t* fix_list;  // [image][kFixListCap] list positions of the flagged records (fix_count = how many)
  int no_map;         // 1: the score map is NOT written (candidates only) -- see launch_harris_nms
};

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
// Hit stacks.  Re-reading candidate scores from the score map after the row loop missed the L2 (the
// tile's own stores had pushed them out) and cost one 64-128 B HBM fetch per candidate; hit BITS
// collected per column (v_addc_co into 32-bit masks, the round-2 form) cost four vector instructions
// per step plus an epilogue that walked eight masks per lane with a popcount chain per record -- 6 %
// of the kernel.  Now the hit itself carries everything: entry e of lane l of wave v = two dwords,
// the score at byte v * kSlots * 512 + e * 512 + l * 4 and the tag (row << 2 | column) 256 bytes
// behind it (one ds_write2_b32, bank-conflict-free).  A lane whose stack is full (flat or noisy
// content) hands the hit to emit_overflow() instead, which appends the record to the image's list
// directly (one atomic per record, flagged for the fix-up pass, which re-evaluates it exactly).
#ifndef OKVFE_K1_SLOTS
#define OKVFE_K1_SLOTS 8
#endif
constexpr int kScoreSlots = OKVFE_K1_SLOTS;
// mask = lanes with c >= nb (every lane: the adjacency test needs the halo lanes' hits too); the
// lanes of `own` among them push {c, tag}; ovf = owners that hit with a full stack
__device__ __forceinline__ void push_hit(uint64_t& mask, uint64_t& ovf, int c, int nb, uint64_t own, uint32_t tag,
                                         uint32_t& sp, uint32_t sp_end) {
  uint64_t sav, pm;
  uint32_t vtag;
  // about half of the (row, column) tests of a wave have no hit at all: the push is branched over
  asm volatile(
      "v_cmp_ge_i32_e64 %[mask], %[c], %[nb]\n\t"
      "s_mov_b64 %[ovf], 0\n\t"
      "s_and_b64 %[pm], %[mask], %[own]\n\t"
      "s_cbranch_scc0 .Lokvfe_nopush%=\n\t"
      "s_and_saveexec_b64 %[sav], %[pm]\n\t"
      "v_cmp_gt_u32_e32 vcc, %[end], %[sp]\n\t"
      "s_andn2_b64 %[ovf], exec, vcc\n\t"
      "s_and_b64 exec, exec, vcc\n\t"
      "v_mov_b32 %[vtag], %[tag]\n\t"
      "ds_write2_b32 %[sp], %[c], %[vtag] offset1:64\n\t"
      "v_add_u32 %[sp], %[stride], %[sp]\n\t"
      "s_mov_b64 exec, %[sav]\n"
      ".Lokvfe_nopush%=:"
      : [sp] "+v"(sp), [mask] "=&s"(mask), [ovf] "=&s"(ovf), [sav] "=&s"(sav), [pm] "=&s"(pm), [vtag] "=&v"(vtag)
      : [c] "v"(c), [nb] "v"(nb), [own] "s"(own), [tag] "s"(tag), [end] "s"(sp_end), [stride] "i"(512)
      : "vcc", "scc", "memory");
}

#ifndef OKVFE_K1_STAGE
#define OKVFE_K1_STAGE 248
#endif
constexpr int kCandStage = OKVFE_K1_STAGE;  // candidate records staged per wave (2976 B; 5 workgroups per CU leave 32 KB each)

// One wave = one strip x kTHF rows, ONE branch-free code path for every tile: a prologue,
// kMain / 6 groups of six identical steps (the rolling buffers have periods 2 and 3, so after six
// steps every value is back in the register it started in and the loop carries no copies) and, with
// the NMS fused, one final step.  The image rim costs no vector work:
//   * pixel rows outside the image: the scalar row offset is clamped;
//   * covariance rows 0 and h-1 (zero by definition): the filter constants k3 / k10 of that step are
//     selected to 0 on the scalar unit, which zeroes both gradients;
//   * score rows 0 and h-1: computed like any other row (they are never an NMS centre and never the
//     neighbour of a tested row) and overwritten with zeros after the loop by the two waves that
//     own them; rows past the image (partial last tile) are not stored;
//   * rows that may not carry maxima (y < 2, y >= h-2) skip their test on a scalar branch.
// MEMONLY: the byte mover of this kernel -- the same loads, the same stores on the same layout, no
// arithmetic (a stored row = the unpacked pixel row three steps ahead).  Its duration is the
// kernel's own memory floor (okvfe_harris_byte_mover_device; bench.py reports both).
template <int kTHF, bool NMS, bool PACK = false, bool MEMONLY = false, bool NOMAP = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) OKVFE_K1_WAVES void harris_kernel(
    const uint8_t* __restrict__ images, int w, int h, int32_t* __restrict__ scores, int pitch, int strips,
    int rtiles, int n_images, NmsOut nms, int pack_g, int pack_u, int main_blocks) {
  constexpr int kMain = NMS ? kTHF - 1 : kTHF;  // steps of the uniform main loop
  static_assert(kMain % 6 == 0 && kTHF <= 121, "rows per wave: 6k (+1 with the fused NMS), <= 121");
  constexpr int kSlots = kTHF > 64 ? kScoreSlots + 2 : kScoreSlots;  // stack depth per lane
  const int lane = threadIdx.x;
  const int nd = w >> 2;
  // PACK (narrow last strip, e.g. 7 dwords of a 1024-px row): `strips` counts the full strips
  // only; the last strips of pack_g consecutive images share one wave, pack_u lanes each (left
  // halo lane + store lanes), in the blocks behind the main_blocks ordinary ones.  Everything that
  // differs between the sub-strips is per-lane already (dword index, load / store offsets, store
  // predicate), so the row loop is the same code; only the candidate reservation of the epilogue
  // runs once per image.
  // Tiles are handed out per WAVE, not per block: wave v of block b takes tile 4 b + v of its stream, so
  // an image whose row tiles do not fill whole blocks of four (540 rows = 9 tiles of 61, 1024 rows = 17)
  // leaves no idle wave slots behind (a block with one live wave still held its 28 KB of LDS: Hilti
  // 720x540 ran 25 % more blocks than it had work for).  Streams keep the images on their XCDs: the
  // dispatcher places block L on XCD L % 8, stream x = the tiles of images x, x + 8, x + 16, ... in
  // order (row tile by row tile, the strips of a row tile consecutive); the last n % 8 images form one stream of their own.
  int image, tile, strip, d;
  int sub = 0;          // PACK: which of the wave's images this lane works on
  int group_images = 1;  // images addressed through this wave's buffer resources
  bool lane_on = true;  // PACK: lane belongs to an existing image
  const bool packed_block = PACK && (int)blockIdx.x >= main_blocks;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int row_tile;  // which kTHF-row tile of the image this wave owns
  if (packed_block) {
    const int p = ((int)blockIdx.x - main_blocks) * kWavesPerBlock + wave;
    const int group = p / rtiles;
    row_tile = p - group * rtiles;
    image = group * pack_g;  // first image of the group
    if (image >= n_images) return;  // wave-uniform: past the last group
    group_images = n_images - image < pack_g ? n_images - image : pack_g;
    strip = strips;  // index of the (packed) last strip
    sub = lane / pack_u;
    const int j = lane - sub * pack_u;
    lane_on = sub < group_images;
    if (!lane_on) sub = 0;
    d = lane_on ? strip * kStripLanes + j : nd;  // idle lanes behave like lanes past the row end
  } else {
    const int T = strips * rtiles;  // tiles per image
    const int n8 = n_images & ~7;
    const int S = (n8 >> 3) * T;    // tiles per XCD stream
    const int full = 8 * ((S + kWavesPerBlock - 1) / kWavesPerBlock);
    const int L = (int)blockIdx.x;
    if (L < full) {
      const int s = (L >> 3) * kWavesPerBlock + wave;
      if (s >= S) return;  // wave-uniform
      const int g = s / T;
      image = g * 8 + (L & 7);
      tile = s - g * T;
    } else {
      const int r = (L - full) * kWavesPerBlock + wave;
      const int i = r / T;
      if (n8 + i >= n_images) return;  // wave-uniform
      image = n8 + i;
      tile = r - i * T;
    }
    // strip-minor: a block's waves sit side by side in a row tile (their stores fill neighbouring slots of
    // the same score rows) rather than below each other: 0.600 vs 0.607 ms per 1536 EuRoC images
    row_tile = tile / strips;
    strip = tile - row_tile * strips;
    d = strip * kStripLanes + lane;
  }
  const int ys_own = row_tile * kTHF;
  if (ys_own >= h) return;  // wave-uniform; all 64 lanes of a live wave stay active (DPP sources)
  const int ye_own = ys_own + kTHF < h ? ys_own + kTHF : h;
  const int ys = NMS ? ys_own - 1 : ys_own;  // first score row computed (with NMS one above the tile)
  const bool last_strip = PACK ? packed_block : strip * kStripLanes + 64 >= nd;
  const bool first_lane_halo = packed_block ? d == strip * kStripLanes : (strip != 0 && lane < 1);
  const bool store = d < nd && !first_lane_halo && (last_strip || lane <= kStripLanes);
  const int dcl = d < nd ? d : nd - 1;  // clamped dword index for loads
  const size_t img_off = (size_t)image * (size_t)w * (size_t)h;
  // buffer resources (SGPR base + 32-bit per-lane offset): no 64-bit VALU address arithmetic in
  // the row loop; the row offset rides in the scalar offset operand
  const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(images + img_off), 0, w * h * group_images, 0x00027000);
  // score map: `pitch` ints per row, slotted layout (okvfe_internal.h, ScoreLayout): strip s of a row
  // has the 1024-byte slot s, lane l of the strip writes bytes [16 l, 16 l + 16) of it
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      scores + (size_t)image * (size_t)pitch * (size_t)h, 0, pitch * h * 4 * group_images, 0x00027000);
  const int ld_off = dcl * 4 + (PACK ? sub * w * h : 0);  // byte offset of this lane's dword within a pixel row
  // EVERY lane whose slot position lies inside the row stores -- halo lanes and lanes past the
  // image write pad columns nobody reads -- so that a store instruction covers whole 128-byte
  // lines; lanes outside the row (and idle lanes of a packed wave) get a per-lane offset outside the
  // resource, which the hardware range check drops (the scalar offset is not range-checked)
  // (the stand-alone score kernel, NMS = false, writes the dense map: pitch == w, owners only)
  // and so does the fused kernel under the OKVFE_K1_DENSE A/B knob (pitch == w)
  const bool slotted = NMS && pitch != w;
  const int slot_q = slotted ? d + 2 * strip : d;  // quad position within the (padded) row
  const bool st_on = slotted ? (lane_on && slot_q * 4 < pitch) : store;
  const int st_off = st_on ? slot_q * 16 + (PACK ? sub * pitch * h * 4 : 0) : 0x7FFFFFF0;
  int m0 = d == 0 ? 0 : -1;        // column 0 is rim
  int m3 = d == nd - 1 ? 0 : -1;   // column w-1 is rim
  // keep the masks as opaque VGPR values: "x & m" then stays a 2-cycle v_and_b32 instead of being
  // rewritten into a v_cndmask_b32_e64 on a re-materialised compare
  asm volatile("" : "+v"(m0), "+v"(m3));

  auto unpack4 = [](uint32_t c, int p[4]) {
    p[0] = c & 255;
    p[1] = (c >> 8) & 255;
    p[2] = (c >> 16) & 255;
    p[3] = c >> 24;
  };
  (void)unpack4;
  // Gradients on the matrix pipe.  v_mfma_i32_4x4x4i8 runs one 4x4x4 product per group of 4 lanes:
  // D[i] of a lane = sum_k A[i][k] * B_lane[k], with row i of A supplied by lane 4b + i -- i.e. FOUR
  // different 4-tap dot products of the lane's own B dword in one issue slot (4.2 cycles beside the
  // vector ALU, tools/ubench/mfma4_test.hip).  B = a 4-pixel window of one pixel row (bytes - 128:
  // both filters sum to zero, so the offset cancels exactly); rows 0/1 of A = the gx taps of the
  // window's two centre columns, rows 2/3 = the gy taps, times 8 (i8 range); A depends on the role
  // of the pixel row (above / centre / below).  Two windows per pixel row cover the lane's four
  // columns: [x-1 .. x+2] and [x+1 .. x+4].  Six MFMAs per covariance row replace ~30 multiply /
  // add / DPP instructions; the results are the same integers (x 2^9 after one shift).
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  int win[3][2];     // rolling windows of the last three pixel rows
  auto taps = [](int b0, int b1, int b2, int b3) {
    return (int)((uint32_t)(b0 & 255) | ((uint32_t)(b1 & 255) << 8) | ((uint32_t)(b2 & 255) << 16) |
                 ((uint32_t)(b3 & 255) << 24));
  };
  const int r4 = lane & 3;
  // gx = [3 10 3]^T (rows) x [-1 0 1] (columns); gy = [-1 0 1]^T x [3 10 3]
  int A_above = r4 == 0 ? taps(-24, 0, 24, 0) : r4 == 1 ? taps(0, -24, 0, 24)
              : r4 == 2 ? taps(-24, -80, -24, 0) : taps(0, -24, -80, -24);
  int A_centre = r4 == 0 ? taps(-80, 0, 80, 0) : r4 == 1 ? taps(0, -80, 0, 80) : 0;
  int A_below = r4 == 0 ? taps(-24, 0, 24, 0) : r4 == 1 ? taps(0, -24, 0, 24)
              : r4 == 2 ? taps(24, 80, 24, 0) : taps(0, 24, 80, 24);
  asm volatile("" : "+v"(A_above), "+v"(A_centre), "+v"(A_below));
  auto make_windows = [&](uint32_t dw, int wn[2]) {
    const int x = (int)(dw ^ 0x80808080u);
    int l, r;
    lane_neighbours(x, x, l, r);
    wn[0] = (int)__builtin_amdgcn_alignbyte((uint32_t)x, (uint32_t)l, 3);  // l.b3 x.b0 x.b1 x.b2
    wn[1] = (int)__builtin_amdgcn_alignbyte((uint32_t)r, (uint32_t)x, 1);  // x.b1 x.b2 x.b3 r.b0
  };
  // stream 0 carries the xx and yy entries PACKED (xx | yy << 16: both are non-negative and every
  // partial sum stays below 2^16, so one 32-bit add smooths two channels); stream 1 carries xy
  int hs[2][2][4];   // horizontally smoothed entries: current / previous row
  int vp[2][2][4];   // vertical pair sums hs[g-1] + hs[g]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) hs[q][c][i] = vp[q][c][i] = 0;

  // NMS state: scores + edge neighbours of the previous row, horizontal 3-max of the last two
  int nc[4] = {0, 0, 0, 0}, nh[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, nl = 0, nr = 0;
  __shared__ int32_t hit_stack[NMS ? kWavesPerBlock * kSlots * 128 : 1];  // [wave][slot][score | tag][lane]
  // candidate records of a wave are collected here and written out as ONE contiguous run: 12-byte
  // records stored straight from the lanes land in scattered 32-byte sectors (the WRITE_SIZE counter
  // showed ~110 KB per image of extra write traffic, 7 % of the score map) -- the common case (a
  // wave's candidates fit) goes through LDS, larger sets keep the direct stores
  constexpr int kStageCap = kCandStage;
  __shared__ Candidate cand_stage[NMS ? kWavesPerBlock : 1][NMS ? kCandStage : 1];
  const uint32_t sp0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)hit_stack +
                       (uint32_t)(wave * (kSlots * 512) + lane * 4);
  uint32_t sp = sp0;  // LDS byte address of this lane's next slot
  // end of the wave's stack as ONE scalar: slot e < kSlots of any lane lies below it
  const uint32_t sp_end = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)hit_stack +
                          (uint32_t)((wave + 1) * (kSlots * 512));
  // lanes that own column i of their quad as a possible maximum: store lanes, columns 2 .. w-3
  const uint64_t own_lo = __ballot(store && d != 0);       // columns 0, 1 of the quad
  const uint64_t own_hi = __ballot(store && d != nd - 1);  // columns 2, 3
  // lanes past the row end (and idle lanes of a packed wave) repeat the last dword: their hits are
  // no neighbours of anything (they used to flag whole rows for the fix-up pass on every width
  // whose last strip is not full: 640, 720, 1024 px)
  const uint64_t real_lanes = __ballot(d < nd);
  const int r_lo = ys_own > 2 ? ys_own : 2;                       // rows that may carry maxima:
  const int r_hi = ye_own < h - 2 ? ye_own : h - 2;               // [r_lo, r_hi) (scalars)
  uint64_t rows_adj[2] = {0ull, 0ull};  // bit t: tested row ys_own + t holds two adjacent hits
  // a hit of a lane whose stack is full: straight into the image's list (rare)
  auto emit_overflow = [&](uint64_t ovf, int c, int i, int r, bool flag) {
    if ((ovf >> lane) & 1ull) {
      int sub_o = 0;
      if (packed_block) {
        sub_o = lane / pack_u;
        if (sub_o >= group_images) sub_o = 0;
      }
      const int img = image + sub_o;
      const int p = atomicAdd(&nms.cand_count[img], 1);
      if (p < nms.cand_cap) {
        Candidate cd;
        cd.x = dcl * 4 + i;
        cd.y = flag ? r | kCandidateFixupFlag : r;
        cd.score = c;
        nms.cand[(size_t)img * nms.cand_cap + p] = cd;
      }
      if (flag) {
        const int k = atomicAdd(&nms.fix_count[img], 1);
        if (k < kFixListCap) nms.fix_list[(size_t)img * kFixListCap + k] = p;
      }
    }
  };

  // covariance row g from pixel rows a (g-1), b (g), c (g+1) -> H (horizontally smoothed); k3, k10 =
  // the (3, 10, 3) filter taps times 2^9 (mulhi24 of two such gradients then yields g*g >> 14), or
  // 0, 0 for a rim row
  auto cov_row = [&](const int* wa, const int* wb, const int* wc, int (*H)[4], bool inner) {
    v4i_t acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_above, wa[0], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_above, wa[1], acc2, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_centre, wb[0], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_centre, wb[1], acc2, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_below, wc[0], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_below, wc[1], acc2, 0, 0, 0);
    // One more wait state than hipcc (ROCm 7.2) leaves between the last MFMA and the first vector
    // instruction that reads its result: without it the shifts below read stale registers on
    // gfx950 (tools/ubench/mfma_grad_test.hip is correct stand-alone; in this kernel the schedule
    // is tighter).  Measured with 1 .. 16 wait states: results exact from 1 on.
    // Round 6: TWO wait states are not enough when another kernel shares the SIMD.  The map-writing PACK instantiation
    // (scale-space layers, okvfe_set_keep_score_map) read acc1 four wait states behind its MFMA (next MFMA, s_nop 1, a
    // scalar branch) and -- with the layers of a scale space on streams of their own -- delivered stale gradients in
    // lanes 12..15 of every 16 (tools/lab/ss_dump.py: ~50 wrong map entries per call, element 0 of a quad); alone, or
    // beside kernels of its own kind, it never did.  The operands being inline-asm outputs ("+v") hides the MFMA from
    // the compiler's hazard recogniser, so the distance is ours to keep: eight wait states (the table of the ISA guide
    // asks for at most seven behind a 4 x 4 MFMA).
#ifndef OKVFE_K1_MFMA_NOPS
#define OKVFE_K1_MFMA_NOPS 7
#endif
#define OKVFE_STR2(x) #x
#define OKVFE_STR(x) OKVFE_STR2(x)
    asm volatile("s_nop " OKVFE_STR(OKVFE_K1_MFMA_NOPS) : "+v"(acc1), "+v"(acc2));
    // The B operands stay LIVE past the matrix instructions: a window that dies at its MFMA (the first pixel rows of a
    // tile, in the prologue) was given a register INSIDE the MFMA's destination quad ("v_mfma v[8:11], v34, v8, ...") --
    // legal for the assembler, and exact alone, but beside another kernel the last quarter of every 16 lanes (DPP bank
    // 3) came back wrong for the tile's first row (tools/lab/ss_dump.py).  Live operands cannot share the destination.
    asm volatile("" ::"v"(wa[0]), "v"(wa[1]), "v"(wb[0]), "v"(wb[1]), "v"(wc[0]), "v"(wc[1]), "v"(A_above), "v"(A_centre),
                 "v"(A_below));
    if (__builtin_expect(!inner, 0)) {  // covariance rows 0 / h-1 are rim: scalar branch, two steps per strip
      asm volatile("" ::: "memory");
      acc1 = v4i_t{0, 0, 0, 0};
      acc2 = v4i_t{0, 0, 0, 0};
    }
    int gx[4], gy[4];
    // x 8 from the taps, x 64 here = the 2^9 scale of the products below
    gx[0] = (acc1[0] << 6) & m0;
    gx[1] = acc1[1] << 6;
    gx[2] = acc2[0] << 6;
    gx[3] = (acc2[1] << 6) & m3;
    gy[0] = (acc1[2] << 6) & m0;
    gy[1] = acc1[3] << 6;
    gy[2] = acc2[2] << 6;
    gy[3] = (acc2[3] << 6) & m3;
    int G[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // xx | yy << 16 without a separate pack: the second product is written into the upper half
      // of the register that already holds the first one (SDWA destination select; both products
      // are < 2^10)
      int g0 = mulhi24(gx[i], gx[i]);
      asm("v_mul_hi_i32_i24_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE "
          "src0_sel:DWORD src1_sel:DWORD"
          : "+v"(g0)
          : "v"(gy[i]));
      G[0][i] = g0;
      G[1][i] = mulhi24(gx[i], gy[i]);
    }
    dpp_fence4(G[0][0], G[0][3], G[1][0], G[1][3]);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      int gl, gr;
      lane_neighbours(G[ch][3], G[ch][0], gl, gr);
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
  };
  // score row from the vertical pair sums of the last two steps
  auto score_row = [&](int (*Vp)[4], int (*V)[4], int sc[4]) {
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
  };
  // (map-free calls, round 4: the selection recomputes the 3 x 3 scores of the few keypoints it keeps and
  // the fix-up works on the candidate records, so four of the five bytes per pixel are never written;
  // the map-free call takes the NOMAP instantiation, compiled without the store; the run-time flag keeps the
  // map-writing instantiation usable for it as well: lab knob OKVFE_K1_GENERIC_NOMAP)
  const bool store_map = !NOMAP && (MEMONLY || !NMS || nms.no_map == 0);  // NOMAP: the map-free instantiation carries no store
  auto store_row = [&](const int sc[4], int y) {
    if (!store_map) return;
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = {sc[0], sc[1], sc[2], sc[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, st_off, y * pitch * 4, 0);
    // Round 6: a 16-byte store reads its data registers AFTER it has issued, and a vector instruction that overwrites one
    // of them needs wait states behind it (the 12-dword-store hazard of the ISA).  The compiler keeps them inside a basic
    // block; here the store ends the block of the `y < h` branch and the next block began with "v_sub v0, ..." -- the first
    // data register -- zero states later.  With the GPU to itself the store always won; beside another kernel's memory
    // traffic the last quarter of every 16 lanes stored the NEW v0 (element 0 of the tile's first row = a score of the
    // row above: tools/lab/ss_analyse.py).  The data stay live, and untouched, for two states behind the store.
#ifndef OKVFE_K1_NO_STORE_NOP  // (A/B: the binary before the fix, to check that the stress tools still catch it)
    asm volatile("s_nop 1" ::"v"(v));
#endif
  };
  // rows y-2 (nh[q]), y-1 (nc, nl, nr; nh[q^1]) and y (sc): optionally test centre row y-1, then
  // roll the state
  auto nms_row = [&](const int sc[4], int q, bool test, int r) {  // r = the centre row (scalar)
    int l, r2;
    lane_neighbours(sc[3], sc[0], l, r2);
    const int hn[4] = {max3i(l, sc[0], sc[1]), max3i(sc[0], sc[1], sc[2]),
                       max3i(sc[1], sc[2], sc[3]), max3i(sc[2], sc[3], r2)};
#ifndef OKVFE_K1_SKIP
#define OKVFE_K1_SKIP 1
#endif
    // (OKVFE_K1_SKIP, round 6 A/B: a wave whose centre row has no score at the threshold skips its four tests -- the
    // row's maximum is at hand in the 3-maxima of the previous step)
    if (test && r >= r_lo && r < r_hi &&
        (!OKVFE_K1_SKIP || __builtin_amdgcn_ballot_w64(max(nh[q ^ 1][1], nh[q ^ 1][2]) >= nms.thr) != 0ull)) {
      const int lft[4] = {nl, nc[0], nc[1], nc[2]};
      const int rgt[4] = {nc[1], nc[2], nc[3], nr};
      uint64_t mk[4], ovf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        push_hit(mk[i], ovf[i], nc[i], max3i(max3i(nh[q][i], hn[i], lft[i]), rgt[i], nms.thr), i < 2 ? own_lo : own_hi,
                 (uint32_t)(r << 2 | i), sp, sp_end);
      // two horizontally adjacent hits anywhere in the row (lane l column 3 | lane l + 1 column 0)
      const uint64_t adj = ((mk[0] & mk[1]) | (mk[1] & mk[2]) | (mk[2] & mk[3]) | (mk[3] & (mk[0] >> 1))) & real_lanes;
      if (adj != 0ull) {
        const int t = r - ys_own;
        if (kTHF <= 64 || t < 64)
          rows_adj[0] |= 1ull << t;
        else
          rows_adj[1] |= 1ull << (t - 64);
      }
      if (__builtin_expect((ovf[0] | ovf[1] | ovf[2] | ovf[3]) != 0ull, 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) emit_overflow(ovf[i], nc[i], i, r, adj != 0ull);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      nh[q][i] = hn[i];
      nc[i] = sc[i];
    }
    nl = l;
    nr = r2;
  };

  // pixel rows are loaded kAhead steps before they are consumed (ring of 6 slots)
#ifndef OKVFE_K1_AHEAD
#define OKVFE_K1_AHEAD 3
#endif
  constexpr int kAhead = OKVFE_K1_AHEAD;
  static_assert(kAhead >= 1 && kAhead <= 5, "prefetch distance");
  int row_next = ys - 2;  // next pixel row to load (scalar)
  auto load_next = [&]() -> uint32_t {
    const int r = row_next < 0 ? 0 : (row_next > h - 1 ? h - 1 : row_next);
    ++row_next;
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(img_rsrc, ld_off, r * w, 0);
  };
  uint32_t ring[6];
  {
    const uint32_t t0 = load_next(), t1 = load_next();
#pragma unroll
    for (int i = 0; i < kAhead; ++i) ring[i] = load_next();
    make_windows(t0, win[0]);
    make_windows(t1, win[1]);
  }
  int y = ys - 2;  // score row completed by the current step (meaningful from step 2 on)
  // step j (compile-time phase PH = j % 6): pixel row ys+j, covariance row g = ys+j-1, score row
  // y = ys+j-2
  auto step = [&](auto ph, auto want_score, auto want_store, auto want_test) {
    constexpr int PH = decltype(ph)::value;
    constexpr int s_new = (PH + 2) % 3, s_a = PH % 3, s_b = (PH + 1) % 3, q = PH % 2;
    ring[(PH + kAhead) % 6] = load_next();
    const int g = y + 1;
    const bool inner = g >= 1 && g <= h - 2;  // scalar
    if constexpr (MEMONLY) {  // the byte mover: loads and stores, no arithmetic
      int sc[4];
      unpack4(ring[PH], sc);
      if (decltype(want_store)::value && y < h) store_row(sc, y);
      ++y;
      (void)inner;
      return;
    }
    make_windows(ring[PH], win[s_new]);
    cov_row(win[s_a], win[s_b], win[s_new], hs[q], inner);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) vp[q][ch][i] = hs[q ^ 1][ch][i] + hs[q][ch][i];
    if (decltype(want_score)::value) {
      int sc[4];
      score_row(vp[q ^ 1], vp[q], sc);
      if (decltype(want_store)::value) {
        if (y < h) store_row(sc, y);  // scalar branch around one instruction (partial last tile)
      }
      if (NMS) nms_row(sc, q, decltype(want_test)::value, y - 1);
    }
    ++y;
  };
  using T = std::true_type;
  using F = std::false_type;
#define OKVFE_PH(n) std::integral_constant<int, (n) % 6>()
  step(OKVFE_PH(0), F(), F(), F());
  step(OKVFE_PH(1), F(), F(), F());
  if (NMS) {
    step(OKVFE_PH(2), T(), F(), F());  // score row ys_own - 1: NMS state only
    step(OKVFE_PH(3), T(), T(), F());  // score row ys_own: stored, nothing to test yet
    for (int g = 0; g < kMain / 6; ++g) {
      step(OKVFE_PH(4), T(), T(), T());
      step(OKVFE_PH(5), T(), T(), T());
      step(OKVFE_PH(6), T(), T(), T());
      step(OKVFE_PH(7), T(), T(), T());
      step(OKVFE_PH(8), T(), T(), T());
      step(OKVFE_PH(9), T(), T(), T());
    }
    step(OKVFE_PH(kTHF + 3), T(), F(), T());  // score row ys_own + kTHF: last test, not stored
  } else {
    for (int g = 0; g < kMain / 6; ++g) {
      step(OKVFE_PH(2), T(), T(), F());
      step(OKVFE_PH(3), T(), T(), F());
      step(OKVFE_PH(4), T(), T(), F());
      step(OKVFE_PH(5), T(), T(), F());
      step(OKVFE_PH(6), T(), T(), F());
      step(OKVFE_PH(7), T(), T(), F());
    }
  }
#undef OKVFE_PH
  // rim score rows are zero by definition (same lanes, same addresses as the stores above, which a
  // wave's memory operations reach in order)
  if (ys_own == 0 || ye_own == h) {
    const int zero[4] = {0, 0, 0, 0};
    if (ys_own == 0) store_row(zero, 0);
    if (ye_own == h) store_row(zero, h - 1);
  }

  if (NMS) {
    // hit stacks -> candidate records
    const int cnt = (int)((sp - sp0) >> 9);
    if (!__any(cnt != 0)) return;
    // (the sub-strip index is recomputed here instead of being kept in a register across the
    // row loop: the PACK variant has to stay within the same register budget)
    int sub_e = 0;
    if (packed_block) {
      int l2 = lane;
      asm volatile("" : "+v"(l2));
      sub_e = l2 / pack_u;
      if (sub_e >= group_images) sub_e = 0;
    }
    const int image_l = image + sub_e;
    const int x0 = dcl * 4;
    Candidate* outc = nms.cand + (size_t)image_l * nms.cand_cap;
    const int32_t* mine = hit_stack + wave * (kSlots * 128) + lane;
    // a flagged record registers its list position for nms_fixup_kernel (rare)
    auto flag_record = [&](int at) {
      const int k = atomicAdd(&nms.fix_count[image_l], 1);
      if (k < kFixListCap) nms.fix_list[(size_t)image_l * kFixListCap + k] = at;
    };
    auto record = [&](int e, bool* flagged) {
      Candidate cd;
      const int tag = mine[e * 128 + 64];
      cd.score = mine[e * 128];
      cd.x = x0 + (tag & 3);
      cd.y = tag >> 2;
      const int t = cd.y - ys_own;
      const uint64_t bit = (kTHF <= 64 || t < 64) ? rows_adj[0] >> t : rows_adj[1] >> (t - 64);
      *flagged = (bit & 1ull) != 0ull;
      if (*flagged) cd.y |= kCandidateFixupFlag;  // to be settled by nms_fixup_kernel
      return cd;
    };
    // one reservation in the image's candidate list per wave (PACK: per image of the wave)
    auto scan = [&](int c, int* total) {
      int incl = c;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const int t = __shfl_up(incl, dd);
        if (lane >= dd) incl += t;
      }
      *total = __builtin_amdgcn_readlane(incl, 63);
      return incl - c;
    };
    if (packed_block) {
      int pos = 0;
      for (int g = 0; g < group_images; ++g) {  // wave-uniform
        const bool in_g = sub_e == g;
        if (!__any(in_g && cnt != 0)) continue;
        int total;
        const int first = scan(in_g ? cnt : 0, &total);
        int base = 0;
        if (lane == 0) base = atomicAdd(&nms.cand_count[image + g], total);
        base = __builtin_amdgcn_readlane(base, 0);
        if (in_g) pos = base + first;
      }
      for (int e = 0; __any(e < cnt); ++e)
        if (e < cnt) {
          bool fl;
          const Candidate cd = record(e, &fl);
          if (pos + e < nms.cand_cap) outc[pos + e] = cd;
          if (fl) flag_record(pos + e);
        }
    } else {
      // the reservation's round trip overlaps the record loop: the records go to LDS by their index
      // within the wave, the list position is only needed for the copy-out
      int total;
      const int first = scan(cnt, &total);
      int base_l0 = 0;
      if (lane == 0) base_l0 = atomicAdd(&nms.cand_count[image], total);
      constexpr int cap_stage = kStageCap;  // records past the stage's capacity are stored directly
      Candidate* stage = cand_stage[wave];
      for (int e = 0; __any(e < cnt); ++e)
        if (e < cnt) {
          bool fl;
          const Candidate cd = record(e, &fl);
          const int idx = first + e;
          if (idx < cap_stage) {
            stage[idx] = cd;
          } else {
            const int at = __builtin_amdgcn_readlane(base_l0, 0) + idx;
            if (at < nms.cand_cap) outc[at] = cd;
          }
          if (fl) flag_record(__builtin_amdgcn_readlane(base_l0, 0) + idx);
        }
      // the wave's records as one contiguous run: lane = record, 12 bytes each
      __builtin_amdgcn_wave_barrier();
      const int base = __builtin_amdgcn_readlane(base_l0, 0);
      const int n_staged = total < cap_stage ? total : cap_stage;
      for (int r = lane; r < n_staged; r += 64)
        if (base + r < nms.cand_cap) outc[base + r] = stage[r];
    }
  }
}

}  // namespace

static int harris_strips(int w) {  // strips of 62 quads (+ halo lanes) covering a row; the last may be narrow
  const int nd = w >> 2;
  int strips = 1;
  while ((strips - 1) * kStripLanes + 64 < nd) ++strips;  // last strip must reach dword nd-1
  return strips;
}

ScoreLayout harris_nms_layout(int w, int h) {
  (void)h;
  static const bool off = lab_env("OKVFE_NO_FUSED_NMS") != nullptr;  // A/B knob for profiling
  if (off || w % 4 != 0) return ScoreLayout{w, 0};
  const int strips = harris_strips(w);
  static const bool dense = lab_env("OKVFE_K1_DENSE") != nullptr;  // A/B knob: fused kernel, dense map
  if (strips == 1 || dense) return ScoreLayout{w, 1};
  const int pitch = ((w + 8 * (strips - 1)) + 31) & ~31;
  return ScoreLayout{pitch, strips};
}

static bool launch_harris_impl(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                               ScoreLayout layout, const NmsOut* nms, hipStream_t stream, bool byte_mover = false) {
  if (n_images <= 0) return true;
  const dim3 block(64, kWavesPerBlock, 1);
  const bool aligned = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0) &&
                       ((reinterpret_cast<uintptr_t>(score) & 15) == 0);
  if (aligned) {
    const int nd = w >> 2;
    const int strips = harris_strips(w);
    // the caller's map must be the fused kernel's layout (or dense: strips == 1, pitch == w)
    if (nms && !(layout.strips == strips || (layout.strips == 1 && layout.pitch == w))) return false;
    static const int th_env = [] {
      const char* e = lab_env("OKVFE_K1_TH");  // A/B knob: rows per wave of the fused kernel (25..61)
      return e ? atoi(e) : 0;
    }();
    // narrow last strip (1024-px rows: 7 of 64 lanes): the last strips of pack_g images share a wave
    const int last_first = (strips - 1) * kStripLanes;      // first dword (halo lane) of the last strip
    const int pack_u = nd - last_first;                     // halo lane + store lanes
    const int pack_g = strips >= 2 ? 64 / pack_u : 1;
#define OKVFE_K1_TILING(TH)                                                                     \
  const int rtiles = (h + TH - 1) / TH; /* row tiles per image; waves take tiles one by one */  \
  auto main_blocks_of = [&](int nstrips) {                                                       \
    const int T = nstrips * rtiles, n8 = n_images & ~7;                                          \
    return 8 * (((n8 >> 3) * T + kWavesPerBlock - 1) / kWavesPerBlock) +                         \
           ((n_images - n8) * T + kWavesPerBlock - 1) / kWavesPerBlock;                          \
  };
#define OKVFE_K1_NMS_LAUNCH_M(TH, MEM)                                                          \
  {                                                                                              \
    OKVFE_K1_TILING(TH)                                                                          \
    if (pack_g >= 2 && !no_pack) {                                                               \
      const int main_blocks = main_blocks_of(strips - 1);                                        \
      const int groups = (n_images + pack_g - 1) / pack_g;                                       \
      const int pblocks = (groups * rtiles + kWavesPerBlock - 1) / kWavesPerBlock;               \
      if (!MEM && nms->no_map && !generic_nomap)                                                 \
        hipLaunchKernelGGL((harris_kernel<TH, true, true, MEM, !MEM>), dim3(main_blocks + pblocks), \
                           block, 0, stream, img, w, h, score, layout.pitch, strips - 1, rtiles, n_images, *nms, \
                           pack_g, pack_u, main_blocks);                                         \
      else                                                                                       \
      hipLaunchKernelGGL((harris_kernel<TH, true, true, MEM>), dim3(main_blocks + pblocks),      \
                         block, 0, stream, img, w, h, score, layout.pitch, strips - 1, rtiles, n_images, *nms, \
                         pack_g, pack_u, main_blocks);                                           \
    } else {                                                                                     \
      if (!MEM && nms->no_map && !generic_nomap)                                                 \
        hipLaunchKernelGGL((harris_kernel<TH, true, false, MEM, !MEM>), dim3(main_blocks_of(strips)), block, 0, \
                           stream, img, w, h, score, layout.pitch, strips, rtiles, n_images, *nms, 1, 64, 0);  \
      else                                                                                       \
      hipLaunchKernelGGL((harris_kernel<TH, true, false, MEM>), dim3(main_blocks_of(strips)), block, 0,  \
                         stream, img, w, h, score, layout.pitch, strips, rtiles, n_images, *nms, 1, 64, 0);    \
    }                                                                                            \
  }
#define OKVFE_K1_NMS_LAUNCH(TH) OKVFE_K1_NMS_LAUNCH_M(TH, false)
    static const bool no_pack = lab_env("OKVFE_K1_NOPACK") != nullptr;  // A/B knob
    static const bool generic_nomap = lab_env("OKVFE_K1_GENERIC_NOMAP") != nullptr;  // A/B knob: map-free through the map-writing instantiation's branch
    if (nms && byte_mover) {
      OKVFE_K1_NMS_LAUNCH_M(61, true);
    } else if (nms) {
      // rows per wave (6k + 1): more rows = fewer halo rows per tile (6 per tile), but longer waves
      // (tail) and a later epilogue
      // A call that cannot fill the GPU anyway (a frame or two: the B = 1 seams) is latency-bound --
      // one wave's dependent row steps, pixel loads three rows ahead -- so it takes 25-row tiles:
      // 2.4 x as many, shorter waves (one 752x480 image: 36 -> ~16 us)
      int th = th_env;
      if (th == 0) {
        const int blocks61 = (int)(((long long)strips * ((h + 60) / 61) * n_images + kWavesPerBlock - 1) / kWavesPerBlock);
        th = blocks61 <= 128 ? 25 : 61;
      }
      switch (th) {
        case 25: OKVFE_K1_NMS_LAUNCH(25); break;
#ifdef OKVFE_LAB  // tile heights of the experiments in LAB_NOTES.md
        case 31: OKVFE_K1_NMS_LAUNCH(31); break;
        case 37: OKVFE_K1_NMS_LAUNCH(37); break;
        case 43: OKVFE_K1_NMS_LAUNCH(43); break;
        case 49: OKVFE_K1_NMS_LAUNCH(49); break;
        case 55: OKVFE_K1_NMS_LAUNCH(55); break;
        case 91: OKVFE_K1_NMS_LAUNCH(91); break;
        case 121: OKVFE_K1_NMS_LAUNCH(121); break;
#endif
        default: OKVFE_K1_NMS_LAUNCH(61); break;
      }
    } else {
      OKVFE_K1_TILING(30)
      hipLaunchKernelGGL((harris_kernel<30, false>), dim3(main_blocks_of(strips)), block, 0, stream,
                         img, w, h, score, w, strips, rtiles, n_images, NmsOut{}, 1, 64, 0);
    }
#undef OKVFE_K1_NMS_LAUNCH
#undef OKVFE_K1_NMS_LAUNCH_M
#undef OKVFE_K1_TILING
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
  (void)launch_harris_impl(img, w, h, n_images, score, ScoreLayout{w, 0}, nullptr, stream);
}

bool launch_harris_nms(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                       ScoreLayout layout, int abs_threshold, Candidate* cand, int cand_cap,
                       int32_t* cand_count, int32_t* fix_count, int32_t* fix_list, hipStream_t stream,
                       bool store_map) {
  if (layout.strips < 1) return false;
  const NmsOut nms{abs_threshold, cand, cand_cap, cand_count, fix_count, fix_list, store_map ? 0 : 1};
  return launch_harris_impl(img, w, h, n_images, score, layout, &nms, stream);
}

// the fused kernel's byte mover on the same layout (diagnostic: the score map receives pixel bytes)
bool launch_harris_byte_mover(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                              ScoreLayout layout, hipStream_t stream) {
  if (layout.strips < 1) return false;
  const NmsOut nms{1, nullptr, 0, nullptr, nullptr, nullptr, 0};
  return launch_harris_impl(img, w, h, n_images, score, layout, &nms, stream, true);
}

}  // namespace okvfe
