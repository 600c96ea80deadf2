// FETCH_SIZE / WRITE_SIZE calibration for K1's access shapes on gfx950 (run under
// rocprofv3 --pmc FETCH_SIZE resp. WRITE_SIZE): every kernel moves a known number of bytes.
//   read4   one dword per lane per row, rows of 752 B, 256 B per wave-instruction (K1's loads)
//   read16  16 B per lane (the shape the guide calibrated: FETCH_SIZE = bytes / 2)
//   write16 16 B per lane per row of 3008 B (K1's stores)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read4(const uint32_t* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void read16(const uint4* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void write16(uint4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4(i, 1, 2, 3);
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: far beyond the 256 MiB Infinity Cache
  void *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  for (int r = 0; r < 3; ++r) {
    read4<<<4096, 256>>>((const uint32_t*)a, (uint32_t*)b, bytes / 4);
    read16<<<4096, 256>>>((const uint4*)a, (uint32_t*)b, bytes / 16);
    write16<<<4096, 256>>>((uint4*)b, bytes / 16);
  }
  (void)hipDeviceSynchronize();
  printf("each kernel moved %zu bytes per launch\n", bytes);
  return 0;
}
