// valu_rates2.hip -- round 5 instruction-rate probe for gfx950 (tooling, not part of the product): the operations of k_pb_half's row loop that round 1's
// valu_rates.hip did not time (double precision, DPP moves, SDWA adds, v_dot2_u32_u16, three-operand logic), each as 16 independent instances per loop trip,
// every SIMD holding W waves.  Prints ns per wave-instruction per SIMD from HIP events AND shader-clock cycles per wave-instruction from s_memtime deltas
// (cycles = (W waves * per-wave delta) / instructions), so that the figure does not depend on what the clock was.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY32(NAME, ASM)                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, unsigned long long *clk, int iters, unsigned seed) {   \
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    unsigned b = a0 ^ 0x5bd1e995u, c = a0 + 77;                                                 \
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555\ns_mov_b32 s10, 0x33333333\ns_mov_b32 s11, 0x33333333\n" ::: "vcc", "s10", "s11");   \
    const unsigned long long t0 = __builtin_readcyclecounter();                                 \
    for (int i = 0; i < iters; i++) {                                                           \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
    }                                                                                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                                 \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                \
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
  }
#define BODY64(NAME, ASM)                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, unsigned long long *clk, int iters, unsigned seed) {   \
    double a0 = 1.0 + threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    double b = 1.0000001, c = 0.9999999;                                                        \
    const unsigned long long t0 = __builtin_readcyclecounter();                                 \
    for (int i = 0; i < iters; i++) {                                                           \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }                                                                                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                                 \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);    \
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
  }

#define A_ADD(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define A_MULU24(i) "v_mul_u32_u24 %" #i ", %8, %" #i "\n"
#define A_DPPSHR(i) "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_DPPQUAD(i) "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_DPPROWSHR(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define A_DOT2U(i) "v_dot2_u32_u16 %" #i ", %8, %9, %" #i "\n"
#define A_DOT4U(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %8\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 8, %" #i "\n"
#define A_OR(i) "v_or_b32 %" #i ", %8, %" #i "\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDSGPR(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define A_CNDE64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define A_CMPCND(i) "v_cmp_gt_i32 vcc, %8, %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_CMPCNDS(i) "v_cmp_gt_i32 s[10:11], %8, %" #i "\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define A_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %8, %" #i ", %9\n"
#define A_MAXU(i) "v_max_u32 %" #i ", %8, %" #i "\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %8, %" #i "\n"
#define A_PKMAD(i) "v_pk_mad_u16 %" #i ", %8, %9, %" #i "\n"
#define A_PKLSHL(i) "v_pk_lshlrev_b16 %" #i ", 2, %" #i "\n"
#define A_MULSDWA(i) "v_mul_u32_u24_sdwa %" #i ", %8, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_1\n"
#define A_FMAF32(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define A_CVTF32U(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_RCPF32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %8, %" #i "\n"
#define A_MADU64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"

#define D_FMA(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n"
#define D_MUL(i) "v_mul_f64 %" #i ", %8, %" #i "\n"
#define D_ADD(i) "v_add_f64 %" #i ", %8, %" #i "\n"
#define D_RCP(i) "v_rcp_f64 %" #i ", %" #i "\n"

BODY32(add_u32, A_ADD) BODY32(mov_b32, A_MOV) BODY32(mad_u32_u24, A_MADU24) BODY32(mul_u32_u24, A_MULU24) BODY32(dpp_wave_shr, A_DPPSHR) BODY32(dpp_quad_perm, A_DPPQUAD)
BODY32(dpp_row_shr, A_DPPROWSHR) BODY32(add_u32_sdwa, A_ADDSDWA) BODY32(dot2_u32_u16, A_DOT2U) BODY32(dot4_u32_u8, A_DOT4U) BODY32(lshl_or, A_LSHLOR) BODY32(and_or, A_ANDOR)
BODY32(or3, A_OR3) BODY32(add3, A_ADD3) BODY32(add_lshl, A_ADDLSHL) BODY32(lshl_add, A_LSHLADD) BODY32(lshlrev, A_LSHL) BODY32(lshrrev, A_LSHR) BODY32(or_b32, A_OR)
BODY32(cndmask, A_CNDMASK) BODY32(cndmask_sgpr, A_CNDSGPR) BODY32(cndmask_e64_vcc, A_CNDE64VCC) BODY32(bfi, A_BFI) BODY32(med3_i32, A_MED3) BODY32(cmp_cndmask_vcc_pair, A_CMPCND) BODY32(cmp_cndmask_sgpr_pair, A_CMPCNDS) BODY32(ffbl, A_FFBL) BODY32(perm, A_PERM) BODY32(max_u32, A_MAXU) BODY32(pk_add_u16, A_PKADD) BODY32(pk_mad_u16, A_PKMAD) BODY32(pk_lshlrev_b16, A_PKLSHL)
BODY32(mul_u24_sdwa_preserve, A_MULSDWA) BODY32(fma_f32, A_FMAF32) BODY32(cvt_f32_u32, A_CVTF32U) BODY32(rcp_f32, A_RCPF32) BODY32(mul_hi_u32, A_MULHI)
BODY64(fma_f64, D_FMA) BODY64(mul_f64, D_MUL) BODY64(add_f64, D_ADD) BODY64(rcp_f64, D_RCP)

// conversions between 32- and 64-bit registers: written out (the operand widths differ)
__global__ __launch_bounds__(256) void k_cvt_f64_u32(unsigned *out, unsigned long long *clk, int iters, unsigned seed) {
  unsigned s0 = threadIdx.x + seed, s1 = s0 * 3, s2 = s0 * 5, s3 = s0 * 7;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++)
    asm volatile("v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11\nv_cvt_f64_u32 %4, %8\nv_cvt_f64_u32 %5, %9\nv_cvt_f64_u32 %6, %10\nv_cvt_f64_u32 %7, %11\n"
                 "v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11\nv_cvt_f64_u32 %4, %8\nv_cvt_f64_u32 %5, %9\nv_cvt_f64_u32 %6, %10\nv_cvt_f64_u32 %7, %11\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_cvt_i32_f64(unsigned *out, unsigned long long *clk, int iters, unsigned seed) {
  double s0 = 1.5 + threadIdx.x + seed, s1 = s0 * 3, s2 = s0 * 5, s3 = s0 * 7;
  unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++)
    asm volatile("v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\nv_cvt_i32_f64 %4, %8\nv_cvt_i32_f64 %5, %9\nv_cvt_i32_f64 %6, %10\nv_cvt_i32_f64 %7, %11\n"
                 "v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\nv_cvt_i32_f64 %4, %8\nv_cvt_i32_f64 %5, %9\nv_cvt_i32_f64 %6, %10\nv_cvt_i32_f64 %7, %11\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned *, unsigned long long *, int, unsigned);
struct Ent { const char *name; kern_t k; };

int main(int argc, char **argv) {
  std::vector<Ent> ents = {
#define E(n) {#n, k_##n}
      E(add_u32), E(mov_b32), E(mad_u32_u24), E(mul_u32_u24), E(dpp_wave_shr), E(dpp_quad_perm), E(dpp_row_shr), E(add_u32_sdwa), E(dot2_u32_u16), E(dot4_u32_u8),
      E(lshl_or), E(and_or), E(or3), E(add3), E(add_lshl), E(lshl_add), E(lshlrev), E(lshrrev), E(or_b32), E(cndmask), E(cndmask_sgpr), E(cndmask_e64_vcc), E(bfi), E(med3_i32), E(cmp_cndmask_vcc_pair), E(cmp_cndmask_sgpr_pair), E(ffbl), E(perm), E(max_u32), E(pk_add_u16), E(pk_mad_u16),
      E(pk_lshlrev_b16), E(mul_u24_sdwa_preserve), E(fma_f32), E(cvt_f32_u32), E(rcp_f32), E(mul_hi_u32), E(fma_f64), E(mul_f64), E(add_f64), E(rcp_f64), E(cvt_f64_u32), E(cvt_i32_f64)};
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int wps = argc > 1 ? atoi(argv[1]) : 8;          // waves per SIMD
  const int blocks = cus * wps;                          // a block is 4 waves, one per SIMD
  unsigned *out;
  unsigned long long *clk;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CHK(hipMalloc(&clk, (size_t)blocks * 4 * 8));
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 4096;
  printf("device %s, %d CUs, %d waves per SIMD\n", prop.gcnArchName, cus, wps);
  for (auto &en : ents) {
    hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, out, clk, 64, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per_wave_cycles = sum / h.size();                   // shader-clock cycles one wave took for its iters * 16 instructions, sharing its SIMD with wps - 1 others
    const double winstr = (double)blocks * 4 * iters * 16;
    const double per_simd = winstr / (cus * 4.0);
    const double ns_per = ms * 1e6 / per_simd;
    printf("%-22s %8.3f ms  %6.3f ns / wave-instr / SIMD   %6.2f counter ticks / wave-instr / SIMD   (%.0f ticks per us)\n", en.name, ms, ns_per,
           per_wave_cycles / ((double)iters * 16 * wps), per_wave_cycles / (ms * 1e3));
  }
  return 0;
}
