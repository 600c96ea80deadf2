// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction of the ops K1 uses.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 2000

#define KERNEL(name, decl, body)                                            \
  __global__ void name(int* out, int seed) {                                \
    decl;                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                    \
      _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { body }          \
    }                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
  }

#define DECL_I int a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = threadIdx.x | 1
#define OP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MUL24(x) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x) : "v"(b));
#define MAD24(x) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MULHI24(x) asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(x) : "v"(b));
#define ASHR(x) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x));
#define ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define PKMAD(x) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define PKMUL(x) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define MADI16(x) asm volatile("v_mad_i32_i16 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define DOT2(x) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define DOT4(x) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define ALIGNBIT(x) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x) : "v"(b));
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define DPPMOV(x) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define SUBSDWA(x) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0" : "+v"(x) : "v"(b));

#define MAX3(x) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MAXI(x) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
#define BFE(x) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x));
#define LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(x) : "v"(b));
#define ANDLIT(x) asm volatile("v_and_b32 %0, 0x7fff, %0" : "+v"(x));
#define CMPADDC(x) asm volatile("v_cmp_ge_i32_e32 vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(b) : "vcc");
#define MULHISDWA(x) asm volatile("v_mul_hi_i32_i24_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(b));
#define MULU24SDWA(x) asm volatile("v_mul_u32_u24_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x));
#define PKLSHR(x) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(x));
#define ADDSDWA(x) asm volatile("v_add_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x));
#define DPPADD(x) asm volatile("v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
#define DPPROW(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define SADU8(x) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define LSHR(x) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
#define MADU24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define ADDMIX(x) asm volatile("v_add_u32 %0, %0, %1\n\tv_mul_i32_i24 %0, %0, %1" : "+v"(x) : "v"(b));
#define LSHRB16(x) asm volatile("v_lshrrev_b16 %0, 1, %0" : "+v"(x));
#define ADDU16(x) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define MULHI32(x) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x) : "v"(b));
#define DOT4I8(x) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define ADDSGPR(x) asm volatile("v_add_u32 %0, s4, %0" : "+v"(x));
#define CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
#define DECL_H int a0 = seed | 0x64006400, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = threadIdx.x | 0x64016401
#define PKMIN3H(x) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define PKMINH(x) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(b));
#define PKMINI16(x) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(b));
#define PKMAXU16(x) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define MIN3I16(x) asm volatile("v_min3_i16 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MIN3F16(x) asm volatile("v_min3_f16 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MIN3U32(x) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MINU32(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MINIMUM3F32(x) asm volatile("v_minimum3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MINU16(x) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x) : "v"(b));
KERNEL(k_pkmin3h, DECL_H, OP8(PKMIN3H))
KERNEL(k_pkminh, DECL_H, OP8(PKMINH))
KERNEL(k_pkmini16, DECL_H, OP8(PKMINI16))
KERNEL(k_pkmaxu16, DECL_H, OP8(PKMAXU16))
KERNEL(k_min3i16, DECL_H, OP8(MIN3I16))
KERNEL(k_min3f16, DECL_H, OP8(MIN3F16))
KERNEL(k_min3u32, DECL_H, OP8(MIN3U32))
KERNEL(k_minu32, DECL_H, OP8(MINU32))
KERNEL(k_minimum3f32, DECL_H, OP8(MINIMUM3F32))
KERNEL(k_minu16, DECL_H, OP8(MINU16))
KERNEL(k_lshrb16, DECL_I, OP8(LSHRB16))
KERNEL(k_addu16, DECL_I, OP8(ADDU16))
KERNEL(k_mulhi32, DECL_I, OP8(MULHI32))
KERNEL(k_dot4i8, DECL_I, OP8(DOT4I8))
KERNEL(k_xor, DECL_I, OP8(XOR))
KERNEL(k_cndmask, DECL_I, OP8(CNDMASK))
KERNEL(k_max3, DECL_I, OP8(MAX3))
KERNEL(k_maxi, DECL_I, OP8(MAXI))
KERNEL(k_bfe, DECL_I, OP8(BFE))
KERNEL(k_lshlor, DECL_I, OP8(LSHLOR))
KERNEL(k_andlit, DECL_I, OP8(ANDLIT))
KERNEL(k_cmpaddc2, DECL_I, OP8(CMPADDC))
KERNEL(k_mulhisdwa, DECL_I, OP8(MULHISDWA))
KERNEL(k_mulu24sdwa, DECL_I, OP8(MULU24SDWA))
KERNEL(k_pklshr, DECL_I, OP8(PKLSHR))
KERNEL(k_addsdwa, DECL_I, OP8(ADDSDWA))
KERNEL(k_dppadd, DECL_I, OP8(DPPADD))
KERNEL(k_dpprow, DECL_I, OP8(DPPROW))
KERNEL(k_sadu8, DECL_I, OP8(SADU8))
KERNEL(k_mov, DECL_I, OP8(MOV))
KERNEL(k_sub, DECL_I, OP8(SUB))
KERNEL(k_lshr, DECL_I, OP8(LSHR))
KERNEL(k_madu24, DECL_I, OP8(MADU24))
KERNEL(k_addmul2, DECL_I, OP8(ADDMIX))
KERNEL(k_add, DECL_I, OP8(ADD))
KERNEL(k_mul24, DECL_I, OP8(MUL24))
KERNEL(k_mad24, DECL_I, OP8(MAD24))
KERNEL(k_mullo, DECL_I, OP8(MULLO))
KERNEL(k_mulhi24, DECL_I, OP8(MULHI24))
KERNEL(k_ashr, DECL_I, OP8(ASHR))
KERNEL(k_add3, DECL_I, OP8(ADD3))
KERNEL(k_lshladd, DECL_I, OP8(LSHLADD))
KERNEL(k_pkadd, DECL_I, OP8(PKADD))
KERNEL(k_pkmad, DECL_I, OP8(PKMAD))
KERNEL(k_pkmul, DECL_I, OP8(PKMUL))
KERNEL(k_madi16, DECL_I, OP8(MADI16))
KERNEL(k_dot2, DECL_I, OP8(DOT2))
KERNEL(k_dot4, DECL_I, OP8(DOT4))
KERNEL(k_perm, DECL_I, OP8(PERM))
KERNEL(k_alignbit, DECL_I, OP8(ALIGNBIT))
KERNEL(k_fma, DECL_I, OP8(FMA))
KERNEL(k_and, DECL_I, OP8(AND))
KERNEL(k_dppmov, DECL_I, OP8(DPPMOV))
KERNEL(k_subsdwa, DECL_I, OP8(SUBSDWA))

int main() {
  int* d;
  const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  hipMalloc(&d, blocks * threads * sizeof(int));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
#define RUN(k)                                                                              \
  {                                                                                         \
    k<<<blocks, threads>>>(d, 1);                                                           \
    hipDeviceSynchronize();                                                                 \
    hipEventRecord(a);                                                                      \
    k<<<blocks, threads>>>(d, 1);                                                           \
    hipEventRecord(b);                                                                      \
    hipEventSynchronize(b);                                                                 \
    float ms;                                                                               \
    hipEventElapsedTime(&ms, a, b);                                                         \
    const double instr = (double)blocks * (threads / 64) * ITERS * REP;                     \
    const double per_simd = instr / (p.multiProcessorCount * 4);                            \
    printf("%-10s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", #k, ms, \
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);                                 \
  }
  RUN(k_add) RUN(k_mul24) RUN(k_mad24) RUN(k_mullo) RUN(k_mulhi24) RUN(k_ashr) RUN(k_add3) RUN(k_lshladd)
  RUN(k_pkadd) RUN(k_pkmad) RUN(k_pkmul) RUN(k_madi16) RUN(k_dot2) RUN(k_dot4) RUN(k_perm) RUN(k_alignbit)
  RUN(k_fma) RUN(k_and) RUN(k_dppmov) RUN(k_subsdwa)
  printf("-- round 2 additions (k_cmpaddc2 and k_addmul2 are TWO instructions per count)\n");
  RUN(k_max3) RUN(k_maxi) RUN(k_bfe) RUN(k_lshlor) RUN(k_andlit) RUN(k_cmpaddc2) RUN(k_mulhisdwa)
  RUN(k_mulu24sdwa) RUN(k_pklshr) RUN(k_addsdwa) RUN(k_dppadd) RUN(k_dpprow) RUN(k_sadu8) RUN(k_mov)
  RUN(k_sub) RUN(k_lshr) RUN(k_madu24) RUN(k_addmul2)
  RUN(k_lshrb16) RUN(k_addu16) RUN(k_mulhi32) RUN(k_dot4i8) RUN(k_xor) RUN(k_cndmask)
  printf("-- round 4: minima / maxima (AGAST kernel)\n");
  RUN(k_pkmin3h) RUN(k_pkminh) RUN(k_pkmini16) RUN(k_pkmaxu16) RUN(k_min3i16) RUN(k_min3f16) RUN(k_min3u32)
  RUN(k_minu32) RUN(k_minimum3f32) RUN(k_minu16)
  return 0;
}
