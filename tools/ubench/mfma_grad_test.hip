// Stand-alone check of the MFMA gradient formulation of k_harris.hip (OKVFE_K1_MFMA): one wave, three
// pixel rows of 256 random bytes; gx / gy of the centre row against the direct formula.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int from_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__global__ void grad_kernel(const uint32_t* rows, int* out) {
  const int lane = threadIdx.x;
  auto taps = [](int b0, int b1, int b2, int b3) {
    return (int)((uint32_t)(b0 & 255) | ((uint32_t)(b1 & 255) << 8) | ((uint32_t)(b2 & 255) << 16) | ((uint32_t)(b3 & 255) << 24));
  };
  const int r4 = lane & 3;
  int A_above = r4 == 0 ? taps(-24, 0, 24, 0) : r4 == 1 ? taps(0, -24, 0, 24) : r4 == 2 ? taps(-24, -80, -24, 0) : taps(0, -24, -80, -24);
  int A_centre = r4 == 0 ? taps(-80, 0, 80, 0) : r4 == 1 ? taps(0, -80, 0, 80) : 0;
  int A_below = r4 == 0 ? taps(-24, 0, 24, 0) : r4 == 1 ? taps(0, -24, 0, 24) : r4 == 2 ? taps(24, 80, 24, 0) : taps(0, 24, 80, 24);
  int win[3][2];
  for (int r = 0; r < 3; ++r) {
    const int x = (int)(rows[r * 64 + lane] ^ 0x80808080u);
    const int l = from_left(x), rr = from_right(x);
    win[r][0] = (int)__builtin_amdgcn_alignbyte((uint32_t)x, (uint32_t)l, 3);
    win[r][1] = (int)__builtin_amdgcn_alignbyte((uint32_t)rr, (uint32_t)x, 1);
  }
  v4i acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_above, win[0][0], acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_above, win[0][1], acc2, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_centre, win[1][0], acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_centre, win[1][1], acc2, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_below, win[2][0], acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_i32_4x4x4i8(A_below, win[2][1], acc2, 0, 0, 0);
  out[lane * 8 + 0] = acc1[0]; out[lane * 8 + 1] = acc1[1]; out[lane * 8 + 2] = acc2[0]; out[lane * 8 + 3] = acc2[1];
  out[lane * 8 + 4] = acc1[2]; out[lane * 8 + 5] = acc1[3]; out[lane * 8 + 6] = acc2[2]; out[lane * 8 + 7] = acc2[3];
}
int main() {
  std::vector<uint8_t> px(3 * 256);
  unsigned st = 7;
  for (auto& p : px) { st = st * 1664525u + 1013904223u; p = (uint8_t)(st >> 24); }
  uint32_t* d_rows; int* d_out;
  hipMalloc(&d_rows, 768); hipMalloc(&d_out, 64 * 8 * 4);
  hipMemcpy(d_rows, px.data(), 768, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(grad_kernel, dim3(1), dim3(64), 0, 0, d_rows, d_out);
  std::vector<int> out(512);
  hipMemcpy(out.data(), d_out, 2048, hipMemcpyDeviceToHost);
  auto P = [&](int r, int c) { return c < 0 || c > 255 ? 128 : (int)px[r * 256 + c]; };
  int bad[8] = {0};
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 4; ++i) {
      const int c = lane * 4 + i;
      auto vs = [&](int cc) { return 10 * P(1, cc) + 3 * (P(0, cc) + P(2, cc)); };
      auto vd = [&](int cc) { return P(2, cc) - P(0, cc); };
      const int gx = 8 * (vs(c + 1) - vs(c - 1)), gy = 8 * (10 * vd(c) + 3 * (vd(c - 1) + vd(c + 1)));
      if (out[lane * 8 + i] != gx) { if (bad[i]++ < 2) printf("gx lane %d col %d: got %d want %d\n", lane, i, out[lane * 8 + i], gx); }
      if (out[lane * 8 + 4 + i] != gy) { if (bad[4 + i]++ < 2) printf("gy lane %d col %d: got %d want %d\n", lane, i, out[lane * 8 + 4 + i], gy); }
    }
  printf("bad gx per col: %d %d %d %d   gy per col: %d %d %d %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
  return 0;
}
