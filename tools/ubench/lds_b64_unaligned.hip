// micro-benchmark: is a 4-byte-aligned (not 8-byte-aligned) ds_read_b64 correct and as fast as an aligned one on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_b64 tools/ubench/lds_b64_unaligned.hip ; run: /tmp/lds_b64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void k(const int* offs, uint64_t* out, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  const int o = offs[threadIdx.x & 63];  // dword index
  uint64_t acc = 0;
  uint32_t a = (uint32_t)(uintptr_t)lds + 4 * o;
  for (int it = 0; it < iters; ++it) {
    uint32_t addr = a + ((it & 15) << 8);
    if (mode == 0) {  // one b64
      uint64_t v;
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
      acc += v;
    } else {  // two b32
      uint32_t x, y;
      asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(x), "=v"(y) : "v"(addr));
      acc += ((uint64_t)y << 32) | x;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int nb = 1024, iters = 4096;
  int* d_off; uint64_t* d_out;
  hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, nb * 256 * 8);
  std::vector<uint64_t> r0(nb * 256), r1(nb * 256);
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<int> off(64);
    for (int l = 0; l < 64; ++l) {
      int row = (l * 7) % 23, col = (l * 5) % 13;
      int dw = row * 16 + col;                       // gather like the box sums: pitch 16 dwords
      if (variant == 0) dw &= ~1;                    // 8-byte aligned
      if (variant == 1) dw |= 1;                     // 4-byte aligned only
      if (variant == 2) dw = l * 2;                  // conflict-free aligned
      off[l] = dw;
    }
    hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d_off, d_out, iters, mode);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d_off, d_out, iters, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy((mode ? r1 : r0).data(), d_out, nb * 256 * 8, hipMemcpyDeviceToHost);
      printf("variant %d (%s) mode %s: %.3f ms\n", variant, variant == 0 ? "8B-aligned gather" : variant == 1 ? "4B-aligned gather" : "linear", mode ? "2 x b32" : "b64", ms);
    }
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("  b64 vs 2xb32 mismatches: %zu\n", bad);
  }
  return 0;
}
