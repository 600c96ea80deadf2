// Lab: semantics of v_msad_u8 on gfx950 -- bytes whose SECOND operand byte is 0 are skipped (verified: 0 mismatches).
// Build: hipcc --offload-arch=gfx950 -O3 msad_test.hip -o msad_test
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
__global__ void k(const unsigned* d, const unsigned* m, unsigned* out) {
  int i = threadIdx.x;
  out[i] = __builtin_amdgcn_msad_u8(d[i], m[i], 1000u);
  out[64 + i] = __builtin_amdgcn_msad_u8(m[i], d[i], 1000u);
}
int main() {
  unsigned hd[64], hm[64], ho[128];
  unsigned s = 777;
  for (int i = 0; i < 64; ++i) {
    s = s * 1664525u + 1013904223u; hd[i] = s;
    s = s * 1664525u + 1013904223u;
    unsigned mm = 0; for (int b = 0; b < 4; ++b) if ((s >> (8 + b)) & 1) mm |= 0xFFu << (8 * b);
    hm[i] = mm;
  }
  hd[0] = 0x00FF0000u;  // zero data bytes under a set mask
  unsigned *d, *m, *o; hipMalloc(&d, 256); hipMalloc(&m, 256); hipMalloc(&o, 512);
  hipMemcpy(d, hd, 256, hipMemcpyHostToDevice); hipMemcpy(m, hm, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, m, o); hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
  int badA = 0, badB = 0;
  for (int i = 0; i < 64; ++i) {
    unsigned eA = 1000, eB = 1000;  // A: skip where the SECOND operand's byte is 0; B: where the first's is
    for (int b = 0; b < 4; ++b) {
      int db = (hd[i] >> (8 * b)) & 255, mb = (hm[i] >> (8 * b)) & 255;
      if (mb) eA += abs(db - mb);
      if (db) eB += abs(db - mb);
    }
    badA += ho[i] != eA; badB += ho[64 + i] != eB;
  }
  printf("msad(d, m): mask on 2nd operand mismatches %d; msad(m, d): mask on 2nd operand (=d) mismatches %d; sample %u\n", badA, badB, ho[1]);
  return 0;
}
