#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) {
  int lane = threadIdx.x;
  int v = 100 + lane, w = 1000 * n;
  int r0, r1, r2, r3, r4;
  asm volatile("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r0) : "v"(v), "v"(w));
  asm volatile("v_add_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r1) : "v"(v), "v"(w));
  asm volatile("v_subrev_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r2) : "v"(v), "v"(w));
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r3) : "v"(v));
  int x = v * n;
  asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r4) : "v"(x));
  out[lane] = r0; out[64 + lane] = r1; out[128 + lane] = r2; out[192 + lane] = r3; out[256 + lane] = r4;
}
int main() {
  int* d; hipMalloc(&d, 320 * 4);
  k<<<1, 64>>>(d, 1);
  int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"add_dpp shr (1000+lane-1)", "add_dpp shl", "subrev_dpp shr (1000-(100+lane-1))", "mov shr", "mov shr fresh"};
  for (int r = 0; r < 5; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) if (i < 6 || i > 60 || (i>=14&&i<=18) || (i>=30&&i<=34)) printf(" %d", h[r * 64 + i]); printf("\n"); }
  return 0;
}
