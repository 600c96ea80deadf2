// v_mfma_i32_4x4x4i8 (16 independent 4x4x4 blocks, one per 4 lanes) on gfx950:
//  (1) operand layout: which lane supplies which row of A / column of B, which lane/VGPR holds D;
//  (2) cost when interleaved with VALU work at 6 waves per SIMD (does the matrix pipe run beside the
//      vector ALU, what does an MFMA cost in issue slots).
// Build: hipcc --offload-arch=gfx950 -O3 mfma4_test.hip -o mfma4_test ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void signed_kernel(const int* a, const int* b, int* d) {
  const int lane = threadIdx.x;
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[lane], b[lane], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
  d[256 + lane] = (int)__builtin_amdgcn_alignbyte(0x44332211u, 0xDDCCBBAAu, 3);
  d[320 + lane] = (int)__builtin_amdgcn_alignbyte(0x44332211u, 0xDDCCBBAAu, 1);
}
__global__ void layout_kernel(const int* a, const int* b, int* d) {
  const int lane = threadIdx.x;
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[lane], b[lane], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
}

#define ITERS 4000
// VALU only: 16 simple adds + 8 mul24 per iteration
__global__ void valu_kernel(int* out, int seed) {
  int x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, b = threadIdx.x | 1;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(b));
      asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x2) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b));
      asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x1) : "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
// the same VALU work + NM independent-chain MFMAs per iteration
template <int NM>
__global__ void mix_kernel(int* out, int seed) {
  int x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, b = threadIdx.x | 1;
  v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  const int av = 0x01020304 + seed, bv = 0x04030201 + (int)threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b));
      if (NM > r * 2) c0 = __builtin_amdgcn_mfma_i32_4x4x4i8(av, bv, c0, 0, 0, 0);
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(b));
      asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x2) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(b));
      if (NM > r * 2 + 1) c1 = __builtin_amdgcn_mfma_i32_4x4x4i8(bv, av, c1, 0, 0, 0);
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b));
      asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x1) : "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[3];
}
// MFMA only, NM per iteration (two accumulation chains)
template <int NM>
__global__ void mfma_kernel(int* out, int seed) {
  v4i c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  const int av = 0x01020304 + seed, bv = 0x04030201 + (int)threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < NM / 2; ++r) {
      c0 = __builtin_amdgcn_mfma_i32_4x4x4i8(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_4x4x4i8(bv, av, c1, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[3];
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  // ---- layout: A lane l holds bytes (l, k) = 16 l + k ... use small distinct primes instead
  std::vector<int> a(64), b(64), d(256);
  for (int l = 0; l < 64; ++l) {
    // A row held by lane l: bytes k=0..3 = 1+k + 4*(l%4)  (rows differ within a block only)
    // B column held by lane l: bytes k = one-hot at k = l%4 scaled by (1 + l/4)  (block id visible)
    int av = 0, bv = 0;
    for (int k = 0; k < 4; ++k) av |= ((1 + k + 4 * (l % 4)) & 0xFF) << (8 * k);
    bv = ((1 + l / 4) & 0x7F) << (8 * (l % 4));
    a[l] = av;
    b[l] = bv;
  }
  int *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  // expectation if D[i][j] (VGPR i of lane 4b+j) = sum_k A[i][k] (lane 4b+i, byte k) * B[k][j] (lane 4b+j, byte k):
  //   B[k][j] = (1+b) if k == j else 0  =>  D[i][j] = A[i][j] * (1+b) = (1 + j + 4 i) * (1 + b)
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int blk = l / 4, j = l % 4, want = (1 + j + 4 * i) * (1 + blk);
      if (d[l * 4 + i] != want) ++bad;
    }
  printf("layout D[i][j] in VGPR i of lane 4b+j, A row i from lane 4b+i, B col j from lane 4b+j: %s\n", bad ? "NO" : "yes");
  if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %d %d %d %d\n", l, d[l*4], d[l*4+1], d[l*4+2], d[l*4+3]);
  {  // signed operands: A rows random in [-128, 127], B columns random in [-128, 127]
    std::vector<int> sa(64), sb(64), sd(384);
    auto byte = [](int v, int k) { return (int)(signed char)((v >> (8 * k)) & 0xFF); };
    unsigned st = 12345;
    for (int l = 0; l < 64; ++l) { st = st * 1664525u + 1013904223u; sa[l] = (int)st; st = st * 1664525u + 1013904223u; sb[l] = (int)st; }
    int *ga, *gb, *gd;
    hipMalloc(&ga, 256); hipMalloc(&gb, 256); hipMalloc(&gd, 1536);
    hipMemcpy(ga, sa.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(gb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(signed_kernel, dim3(1), dim3(64), 0, 0, ga, gb, gd);
    hipMemcpy(sd.data(), gd, 1536, hipMemcpyDeviceToHost);
    int sbad = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        int want = 0;
        for (int k = 0; k < 4; ++k) want += byte(sa[(l & ~3) + i], k) * byte(sb[l], k);
        if (sd[l * 4 + i] != want) { if (sbad < 6) printf("  signed mismatch lane %d i %d: got %d want %d\n", l, i, sd[l*4+i], want); ++sbad; }
      }
    printf("signed i8 x i8 dot products: %s (%d bad)\n", sbad ? "NO" : "yes", sbad);
    printf("alignbyte({44332211, DDCCBBAA}, 3) = %08x (expect 332211DD), (.., 1) = %08x (expect 11DDCCBB)\n", (unsigned)sd[256], (unsigned)sd[320]);
  }
  // ---- rates: 1024 workgroups x 256 threads... 6 waves per SIMD on 256 CUs = 6144 waves
  int* out;
  hipMalloc(&out, 6144 * 64 * 4 * 4);
  const dim3 grid(6144 / 4 * 4), block(64 * 4);  // 4 waves per block, 6144 blocks => 4 rounds of 6 waves/SIMD
  const double clk = 2.4e6;  // cycles per ms at 2.4 GHz (nominal)
  auto rep = [&](const char* n, float ms, int valu, int mf) {
    // per SIMD: waves = grid*4/1024; cycles/iter/wave = ms*clk / (waves_per_simd * ITERS)
    const double wps = (double)grid.x * 4 / 1024.0;
    printf("%-28s %.3f ms  -> %.1f cycles per iteration per wave (VALU %d, MFMA %d)\n", n, ms, ms * clk / (wps * ITERS), valu, mf);
  };
  rep("valu only", time_ms([&] { hipLaunchKernelGGL(valu_kernel, grid, block, 0, 0, out, 1); }), 24, 0);
  rep("valu + 2 mfma", time_ms([&] { hipLaunchKernelGGL(mix_kernel<2>, grid, block, 0, 0, out, 1); }), 24, 2);
  rep("valu + 4 mfma", time_ms([&] { hipLaunchKernelGGL(mix_kernel<4>, grid, block, 0, 0, out, 1); }), 24, 4);
  rep("valu + 8 mfma", time_ms([&] { hipLaunchKernelGGL(mix_kernel<8>, grid, block, 0, 0, out, 1); }), 24, 8);
  rep("8 mfma only", time_ms([&] { hipLaunchKernelGGL(mfma_kernel<8>, grid, block, 0, 0, out, 1); }), 0, 8);
  rep("2 mfma only", time_ms([&] { hipLaunchKernelGGL(mfma_kernel<2>, grid, block, 0, 0, out, 1); }), 0, 2);
  return 0;
}
