// Memory floor of K1's byte mix on gfx950: read 1 B, write 4 B per pixel with NO arithmetic, in the
// friendliest patterns (fully contiguous streams), for 1536 x 752 x 480 pixels (554 MB in, 2.2 GB out).
//   linear:   wave i handles 256-pixel chunk i, i + nwaves, ...  (one dword load, one 16 B store per lane)
//   blocked:  each workgroup walks its own contiguous range
//   rows8:    like linear, but each lane loads 8 B and stores 2 x 16 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void widen_linear(const uint32_t* __restrict__ in, v4i* __restrict__ out, size_t n_dw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dw; i += (size_t)gridDim.x * 256) {
    const uint32_t c = in[i];
    out[i] = v4i{(int)(c & 255), (int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24)};
  }
}
__global__ __launch_bounds__(256) void widen_blocked(const uint32_t* __restrict__ in, v4i* __restrict__ out, size_t n_dw, size_t per_block) {
  const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n_dw ? lo + per_block : n_dw;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t c = in[i];
    out[i] = v4i{(int)(c & 255), (int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24)};
  }
}
template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main() {
  const size_t px = (size_t)1536 * 752 * 480, n_dw = px / 4;
  uint32_t* in; v4i* out;
  hipMalloc(&in, px); hipMalloc(&out, px * 4);
  hipMemset(in, 7, px);
  const double gb = 5.0 * px / 1e9;
  for (int blocks : {1536, 3072, 6144, 12288, 24576}) {
    float ms = time_ms([&] { hipLaunchKernelGGL(widen_linear, dim3(blocks), dim3(256), 0, 0, in, out, n_dw); }, 5);
    printf("linear  %6d blocks: %.3f ms  %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
  }
  for (int blocks : {1536, 6144, 24576, 98304}) {
    const size_t per = (n_dw + blocks - 1) / blocks;
    float ms = time_ms([&] { hipLaunchKernelGGL(widen_blocked, dim3(blocks), dim3(256), 0, 0, in, out, n_dw, per); }, 5);
    printf("blocked %6d blocks (%zu KB out each): %.3f ms  %.0f GB/s\n", blocks, per * 16 / 1024, ms, gb / ms * 1e3);
  }
  return 0;
}
