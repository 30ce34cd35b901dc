// Micro-benchmark (r02): issue cost per wave64 instruction of the integer VALU ops the reconstruction kernels use, one
// inline-asm instruction form per line (the compiler cannot fuse or re-encode them).  8 independent accumulators per
// loop body, 8 waves per SIMD: throughput, not latency.  Output: cycles per instruction per SIMD at 2.4 GHz (the part
// may clock lower under load: read the ratios).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {                      \
    uint32_t r[8], x = threadIdx.x * 2654435761u + seed, y = x ^ 0x5bd1e995u, z = (x >> 3) | 1u;              \
    unsigned long long m = 0x5555555555555555ull ^ seed;                                                       \
    for (int i = 0; i < 8; i++) r[i] = x + i * 77u;                                                           \
    for (int it = 0; it < iters; it++) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                         \
        REP8(ASM)                                                                                             \
      }                                                                                                       \
    }                                                                                                         \
    uint32_t acc = 0;                                                                                         \
    for (int i = 0; i < 8; i++) acc ^= r[i];                                                                  \
    if (acc == 0x12345678u) out[0] = acc + (uint32_t)m;                                                       \
  }

#define A_ADD(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_ADD64(i) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_AND(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_ANDK(i) asm volatile("v_and_b32_e32 %0, 0xfefefefe, %0" : "+v"(r[i]));
#define A_LSHR(i) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(r[i]));
#define A_ASHR(i) asm volatile("v_ashrrev_i32_e32 %0, 6, %0" : "+v"(r[i]));
#define A_MAX(i) asm volatile("v_max_i32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r[i]) : "v"(y));
#define A_CND32(i) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(y) : "vcc");
#define A_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r[i]) : "v"(y), "s"(m));
#define A_LERP(i) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_ALIGN(i) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_MAX3(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(r[i]) : "v"(y));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r[i]));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_SDWA(i) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r[i]) : "v"(y));
#define A_SDWAD(i) asm volatile("v_max_i32_sdwa %0, %1, %0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r[i]) : "v"(y));
#define A_MUL24(i) asm volatile("v_mul_i32_i24_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_MAD24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_PKADD(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_PKASHR(i) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(r[i]));
#define A_PKMAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_SUBREV(i) asm volatile("v_subrev_u32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_XOR(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(y));
#define A_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(y));
#define A_SAD(i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_CVTPK(i) asm volatile("v_cvt_pk_u8_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(z));
#define A_PACK(i) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(r[i]) : "v"(y));
#define A_CMP(i) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(r[i]), "v"(y) : "vcc");
#define A_CMP64(i) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(y));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(y));
#define A_MAD64(i) { unsigned long long t_; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "=v"(t_) : "v"(r[i]), "v"(y) : "vcc"); r[i] ^= (uint32_t)t_; }
#define A_MULU24(i) asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(r[i]) : "v"(y));
#define A_LSHLADD64(i) { unsigned long long t_ = m; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(t_) : "v"(m)); r[i] ^= (uint32_t)t_; }
#define A_LSHL64(i) { unsigned long long t_ = ((unsigned long long)r[i] << 32) | y; asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(t_) : "v"(z)); r[i] = (uint32_t)(t_ >> 32); }
#define A_FFBH(i) asm volatile("v_ffbh_u32_e32 %0, %0" : "+v"(r[i]));
#define A_SALU(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(seed));
#define A_RDLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(seed) : "v"(r[i]));

KERNEL(k_add, A_ADD) KERNEL(k_add64, A_ADD64) KERNEL(k_and, A_AND) KERNEL(k_andk, A_ANDK) KERNEL(k_lshr, A_LSHR) KERNEL(k_ashr, A_ASHR)
KERNEL(k_max, A_MAX) KERNEL(k_mov, A_MOV) KERNEL(k_cnd32, A_CND32) KERNEL(k_cnd64, A_CND64) KERNEL(k_lerp, A_LERP) KERNEL(k_perm, A_PERM)
KERNEL(k_align, A_ALIGN) KERNEL(k_med3, A_MED3) KERNEL(k_max3, A_MAX3) KERNEL(k_add3, A_ADD3) KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_andor, A_ANDOR)
KERNEL(k_bfe, A_BFE) KERNEL(k_bfi, A_BFI) KERNEL(k_sdwa, A_SDWA) KERNEL(k_sdwad, A_SDWAD) KERNEL(k_mul24, A_MUL24) KERNEL(k_mad24, A_MAD24)
KERNEL(k_pkadd, A_PKADD) KERNEL(k_pkashr, A_PKASHR) KERNEL(k_pkmax, A_PKMAX) KERNEL(k_subrev, A_SUBREV) KERNEL(k_xor, A_XOR) KERNEL(k_bcnt, A_BCNT)
KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_dpp, A_DPP) KERNEL(k_adddpp, A_ADDDPP) KERNEL(k_sad, A_SAD) KERNEL(k_cvtpk, A_CVTPK) KERNEL(k_pack, A_PACK)
KERNEL(k_mullo, A_MULLO) KERNEL(k_mulhi, A_MULHI) KERNEL(k_mulu24, A_MULU24)
KERNEL(k_lshl64, A_LSHL64) KERNEL(k_ffbh, A_FFBH)
KERNEL(k_lshlor, A_LSHLOR) KERNEL(k_cmp, A_CMP) KERNEL(k_cmp64, A_CMP64) KERNEL(k_salu, A_SALU) KERNEL(k_rdlane, A_RDLANE)

typedef void (*kern_t)(uint32_t *, int, uint32_t);
static void run(const char *tag, kern_t k, uint32_t *out, int waves_per_simd) {
  const int iters = 4000, blocks = 256 * waves_per_simd;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double instr_per_simd = (double)waves_per_simd * iters * 32;
  printf("%-34s %d waves/SIMD  %.3f ms  %.2f cycles/instr\n", tag, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main(int argc, char **argv) {
  uint32_t *out; (void)hipMalloc(&out, 64);
  const int w = argc > 1 ? atoi(argv[1]) : 4;
#define R(name, k) run(name, k, out, w);
  R("v_add_u32_e32", k_add) R("v_add_u32_e64", k_add64) R("v_and_b32_e32", k_and) R("v_and_b32_e32 literal", k_andk) R("v_lshrrev_b32_e32 imm", k_lshr)
  R("v_ashrrev_i32_e32 imm", k_ashr) R("v_max_i32_e32", k_max) R("v_mov_b32", k_mov) R("v_cndmask_b32_e32 (vcc)", k_cnd32) R("v_cndmask_b32_e64 (sgpr pair)", k_cnd64)
  R("v_subrev_u32_e32", k_subrev) R("v_xor_b32_e32", k_xor) R("v_mul_i32_i24_e32", k_mul24)
  R("v_lerp_u8", k_lerp) R("v_perm_b32", k_perm) R("v_alignbyte_b32", k_align) R("v_med3_i32", k_med3) R("v_max3_i32", k_max3) R("v_add3_u32", k_add3)
  R("v_lshl_add_u32", k_lshladd) R("v_and_or_b32", k_andor) R("v_lshl_or_b32", k_lshlor) R("v_bfe_u32", k_bfe) R("v_bfi_b32", k_bfi) R("v_mad_i32_i24", k_mad24) R("v_sad_u8", k_sad)
  R("v_cvt_pk_u8_f32", k_cvtpk) R("v_pack_b32_f16", k_pack) R("v_bcnt_u32_b32", k_bcnt) R("v_mbcnt_lo_u32_b32", k_mbcnt)
  R("v_add_u32_sdwa src byte", k_sdwa) R("v_max_i32_sdwa dst byte preserve", k_sdwad) R("v_mov_b32_dpp quad_perm", k_dpp) R("v_add_u32_dpp row_shr", k_adddpp)
  R("v_pk_add_i16", k_pkadd) R("v_pk_ashrrev_i16", k_pkashr) R("v_pk_max_i16", k_pkmax)
  R("v_mul_lo_u32", k_mullo) R("v_mul_hi_u32", k_mulhi) R("v_mul_u32_u24_e32", k_mulu24)
  R("v_lshlrev_b64 (+ 2 moves packing the operand)", k_lshl64) R("v_ffbh_u32", k_ffbh)
  R("v_cmp_lt_i32_e32 (vcc)", k_cmp) R("v_cmp_lt_i32_e64 (sgpr)", k_cmp64) R("s_add_u32", k_salu) R("v_readlane_b32", k_rdlane)
  return 0;
}
