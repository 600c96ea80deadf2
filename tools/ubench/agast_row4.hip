// Lab prototype: AGAST 9-16 score with FOUR pixels of a row per thread (16-byte stores, shared LDS reads) against the
// library's one-pixel-per-thread column walker (k_agast.hip).  Aligned widths, whole images of 16-row tiles only.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../okvis2_amd/csrc agast_row4.hip -o agast_row4
#include "../../okvis2_amd/csrc/k_agast.hip"
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
namespace lab {
using namespace okvfe;
constexpr int kW = 128, kH = 16, kA = 3, kPitch = kW + 8, kRowsS = kH + 2 * kA;
constexpr int kStageDw = kRowsS * (kPitch / 4), kRounds = (kStageDw + 255) / 256;
constexpr uint32_t kBias = 0xe4646464u;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mn3(uint32_t a, uint32_t b, uint32_t c) {
  const h2 x = __builtin_bit_cast(h2, a), y = __builtin_bit_cast(h2, b), z = __builtin_bit_cast(h2, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));
}
__device__ __forceinline__ uint32_t mx3(uint32_t a, uint32_t b, uint32_t c) {
  const h2 x = __builtin_bit_cast(h2, a), y = __builtin_bit_cast(h2, b), z = __builtin_bit_cast(h2, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
__device__ __forceinline__ int score16(const uint32_t (&d)[16], uint32_t centre) {
  uint32_t m3[16], m9[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m3[i] = mn3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; ++i) m9[i] = mn3(m3[i], m3[(i + 3) & 15], m3[(i + 6) & 15]);
  const uint32_t t0 = mx3(m9[0], m9[1], m9[2]), t1 = mx3(m9[3], m9[4], m9[5]), t2 = mx3(m9[6], m9[7], m9[8]);
  const uint32_t t3 = mx3(m9[9], m9[10], m9[11]), t4 = mx3(m9[12], m9[13], m9[14]);
  const uint32_t best = mx3(mx3(t0, t1, t2), mx3(t3, t4, m9[15]), m9[15]);
  const h2 bd = __builtin_bit_cast(h2, best) - __builtin_bit_cast(h2, centre);
  const _Float16 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(bd.x, bd.y), (_Float16)1.0f);
  return (int)(unsigned short)(short)(m - (_Float16)1.0f);
}
__global__ __launch_bounds__(256) void row4_kernel(const uint8_t* __restrict__ images, int w, int h, int32_t* __restrict__ scores,
                                                   int tiles_x, int strips_y, int chunks, int n_images) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][kRowsS * kPitch];
  int image, t;
  xcd_tile(tiles_x * strips_y, n_images, &image, &t);
  const int ty0 = t / tiles_x, tx0 = t - ty0 * tiles_x, x0 = tx0 * kW;
  const uint8_t* img = images + (size_t)image * w * h;
  int32_t* out = scores + (size_t)image * w * h;
  const int tid = threadIdx.x, cg = tid & 31;
  const int ry = __builtin_amdgcn_readfirstlane(tid >> 6) * 2 + ((tid >> 5) & 1);  // 0..7 (two rows per wave)
  const int x = x0 + 4 * cg;
  const int y_first = ty0 * chunks * kH;
  int n_chunks = (h - y_first + kH - 1) / kH;
  n_chunks = n_chunks < chunks ? n_chunks : chunks;
  const int ndw = w >> 2;
  int srow[kRounds]; uint32_t scol[kRounds], slds[kRounds], px[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int i = tid + 256 * r, row = i / (kPitch / 4), c = i - row * (kPitch / 4);
    int dq = (x0 >> 2) - 1 + c;
    dq = dq < 0 ? 0 : (dq > ndw - 1 ? ndw - 1 : dq);
    srow[r] = i < kStageDw ? row : -1; scol[r] = 4u * dq; slds[r] = row * kPitch + 4 * c;
  }
  auto load = [&](int y0) {
#pragma unroll
    for (int r = 0; r < kRounds; ++r) if (srow[r] >= 0) {
      int yy = y0 - kA + srow[r]; yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      px[r] = *reinterpret_cast<const uint32_t*>(img + ((uint32_t)yy * (uint32_t)w + scol[r]));
    }
  };
  auto store = [&](uint32_t* buf) {
#pragma unroll
    for (int r = 0; r < kRounds; ++r) if (srow[r] >= 0) {
      uint4 v;
      v.x = __builtin_amdgcn_perm(kBias, px[r], 0x07000400u); v.y = __builtin_amdgcn_perm(kBias, px[r], 0x07010401u);
      v.z = __builtin_amdgcn_perm(kBias, px[r], 0x07020402u); v.w = __builtin_amdgcn_perm(kBias, px[r], 0x07030403u);
      *reinterpret_cast<uint4*>(buf + slds[r]) = v;
    }
  };
  load(y_first); store(tile[0]);
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int y0 = y_first + ch * kH;
    const bool more = ch + 1 < n_chunks;
    if (more) load(y0 + kH);
    if (x < w) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ly = ry + 8 * half, y = y0 + ly;
        if (y < h) {
          const uint32_t* c = tile[ch & 1] + (ly + kA) * kPitch + 4 * cg + 4;  // pixel x of row y
          // rows -3 .. 3 around y, columns x-3 .. x+6 as needed
          uint32_t a3[6], b3[6], a2[8], b2[8], a1[10], b1[10], z[10];
          auto rd6 = [&](const uint32_t* p, uint32_t (&o)[6]) {  // x-1 .. x+4
            o[0] = p[-1]; const uint4 q = *reinterpret_cast<const uint4*>(p); o[1] = q.x; o[2] = q.y; o[3] = q.z; o[4] = q.w; o[5] = p[4];
          };
          auto rd8 = [&](const uint32_t* p, uint32_t (&o)[8]) {  // x-2 .. x+5
            const uint2 l = *reinterpret_cast<const uint2*>(p - 2); const uint4 q = *reinterpret_cast<const uint4*>(p);
            const uint2 r = *reinterpret_cast<const uint2*>(p + 4);
            o[0] = l.x; o[1] = l.y; o[2] = q.x; o[3] = q.y; o[4] = q.z; o[5] = q.w; o[6] = r.x; o[7] = r.y;
          };
          auto rd10 = [&](const uint32_t* p, uint32_t (&o)[10]) {  // x-3 .. x+6
            o[0] = p[-3]; const uint2 l = *reinterpret_cast<const uint2*>(p - 2); const uint4 q = *reinterpret_cast<const uint4*>(p);
            const uint2 r = *reinterpret_cast<const uint2*>(p + 4); o[9] = p[6];
            o[1] = l.x; o[2] = l.y; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w; o[7] = r.x; o[8] = r.y;
          };
          rd6(c + 3 * kPitch, a3); rd6(c - 3 * kPitch, b3); rd8(c + 2 * kPitch, a2); rd8(c - 2 * kPitch, b2);
          rd10(c + kPitch, a1); rd10(c - kPitch, b1); rd10(c, z);
          int s[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // circle of pixel x+i: (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3) (0,-3) (-1,-3) (-2,-2) (-3,-1) (-3,0) (-3,1) (-2,2) (-1,3)
            // a3/b3 index = dx + 1 + i, a2/b2 = dx + 2 + i, a1/b1/z = dx + 3 + i
            const uint32_t d[16] = {a3[1 + i], a3[2 + i], a2[4 + i], a1[6 + i], z[6 + i], b1[6 + i], b2[4 + i], b3[2 + i],
                                    b3[1 + i], b3[0 + i], b2[0 + i], b1[0 + i], z[0 + i], a1[0 + i], a2[0 + i], a3[0 + i]};
            const int v = score16(d, z[3 + i]);
            const int xi = x + i;
            s[i] = (xi >= 3 && xi < w - 3 && y >= 3 && y < h - 3) ? v : 0;
          }
          typedef int v4i __attribute__((ext_vector_type(4)));
          const v4i sv = {s[0], s[1], s[2], s[3]};
          __builtin_nontemporal_store(sv, reinterpret_cast<v4i*>(out + (size_t)y * w + x));
        }
      }
    }
    if (more) store(tile[(ch + 1) & 1]);
    __syncthreads();
  }
}
}  // namespace lab
int main() {
  const int w = 752, h = 480, n = 512;
  std::vector<uint8_t> img((size_t)w * h * n);
  uint32_t s = 12345;
  for (auto& v : img) { s = s * 1664525u + 1013904223u; v = s >> 24; }
  uint8_t* d_img; int32_t *d_a, *d_b;
  hipMalloc(&d_img, img.size()); hipMalloc(&d_a, img.size() * 4); hipMalloc(&d_b, img.size() * 4);
  hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
  const int tiles_x = (w + lab::kW - 1) / lab::kW, tiles_y = h / lab::kH;
  int chunks = (int)(((long long)tiles_x * tiles_y * n) / 8192); chunks = chunks < 1 ? 1 : (chunks > 8 ? 8 : chunks);
  const int strips_y = (tiles_y + chunks - 1) / chunks;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 3; ++i) okvfe::launch_agast_score(d_img, w, h, n, d_a, 0);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) okvfe::launch_agast_score(d_img, w, h, n, d_a, 0);
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("library kernel: %.3f ms per %d images\n", ms / 10, n);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab::row4_kernel, dim3(tiles_x * strips_y * n), dim3(256), 0, 0, d_img, w, h, d_b, tiles_x, strips_y, chunks, n);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(lab::row4_kernel, dim3(tiles_x * strips_y * n), dim3(256), 0, 0, d_img, w, h, d_b, tiles_x, strips_y, chunks, n);
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("row4 prototype: %.3f ms per %d images\n", ms / 10, n);
  }
  std::vector<int32_t> ha((size_t)w * h * 8), hb((size_t)w * h * 8);
  hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
  printf("mismatching pixels in the first 8 images: %zu\n", bad);
  return bad != 0;
}
