// valu_rates.hip -- instruction-rate microbenchmark for gfx950 (tooling, not part of the product).
// Each kernel runs N iterations of 16 independent instances of one VALU instruction; all CUs are filled
// with 8 waves/SIMD.  Prints wave-instructions per clock per SIMD (1/cycles-per-instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define DEF_KERNEL(NAME, ASM)                                                                  \
  __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, int iters, unsigned seed) {   \
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    unsigned b = a0 ^ 0x5bd1e995u, c = a0 + 77;                                                 \
    for (int i = 0; i < iters; i++) {                                                           \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }                                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                \
  }

#define A_MAD24(i) "v_mad_i32_i24 %" #i ", %8, %9, %" #i "\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define A_ADD(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %8, %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 8, %" #i "\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %8, %" #i ", %9\n"
#define A_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define A_DOT4U(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define A_DOT4C(i) "v_dot4c_i32_i8 %" #i ", %8, %9\n"
#define A_DOT2C(i) "v_dot2c_i32_i16 %" #i ", %8, %9\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %8, %" #i "\n"
#define A_MUL24SDWA(i) "v_mul_u32_u24_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_PKMADU16(i) "v_pk_mad_u16 %" #i ", %8, %9, %" #i "\n"
#define A_PKADDU16(i) "v_pk_add_u16 %" #i ", %8, %" #i "\n"
#define A_PKMULLO16(i) "v_pk_mul_lo_u16 %" #i ", %8, %" #i "\n"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTPK(i) "v_cvt_pk_i16_i32 %" #i ", %" #i ", %8\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %8, %" #i ", 2\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 6, %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MADU16(i) "v_mad_u16 %" #i ", %8, %9, %" #i "\n"
#define A_CVTI32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_RPI(i) "v_cvt_rpi_i32_f32 %" #i ", %" #i "\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_MQSAD(i) "v_msad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_LERP(i) "v_lerp_u8 %" #i ", %8, %" #i ", %9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define A_PKFMA(i) "v_pk_fma_f16 %" #i ", %8, %9, %" #i "\n"

DEF_KERNEL(mad_i32_i24, A_MAD24)
DEF_KERNEL(mad_u32_u24, A_MADU24)
DEF_KERNEL(fma_f32, A_FMA)
DEF_KERNEL(fmac_f32, A_FMAC)
DEF_KERNEL(add_u32, A_ADD)
DEF_KERNEL(and_b32, A_AND)
DEF_KERNEL(bfe_u32, A_BFE)
DEF_KERNEL(lshrrev, A_LSHR)
DEF_KERNEL(perm_b32, A_PERM)
DEF_KERNEL(cvt_f32_ubyte1, A_CVTUB)
DEF_KERNEL(dot4_u32_u8, A_DOT4U)
DEF_KERNEL(dot4c_i32_i8, A_DOT4C)
DEF_KERNEL(dot2c_i32_i16, A_DOT2C)
DEF_KERNEL(mul_lo_u32, A_MULLO)
DEF_KERNEL(mul_u24_sdwa, A_MUL24SDWA)
DEF_KERNEL(pk_mad_u16, A_PKMADU16)
DEF_KERNEL(pk_add_u16, A_PKADDU16)
DEF_KERNEL(pk_mul_lo_u16, A_PKMULLO16)
DEF_KERNEL(med3_i32, A_MED3)
DEF_KERNEL(cvt_pk_i16_i32, A_CVTPK)
DEF_KERNEL(alignbyte, A_ALIGNBYTE)
DEF_KERNEL(lshl_add, A_LSHLADD)
DEF_KERNEL(add3, A_ADD3)
DEF_KERNEL(mad_u16, A_MADU16)
DEF_KERNEL(cvt_i32_f32, A_CVTI32)
DEF_KERNEL(cvt_rpi_i32_f32, A_RPI)
DEF_KERNEL(sad_u8, A_SAD)
DEF_KERNEL(msad_u8, A_MQSAD)
DEF_KERNEL(lerp_u8, A_LERP)
DEF_KERNEL(bfi_b32, A_BFI)
DEF_KERNEL(pk_fma_f16, A_PKFMA)

typedef void (*kern_t)(unsigned *, int, unsigned);
struct Ent { const char *name; kern_t k; };

int main() {
  std::vector<Ent> ents = {
#define E(n) {#n, k_##n}
      E(mad_i32_i24), E(mad_u32_u24), E(fma_f32), E(fmac_f32), E(add_u32), E(and_b32), E(bfe_u32), E(lshrrev), E(perm_b32),
      E(cvt_f32_ubyte1), E(dot4_u32_u8), E(dot4c_i32_i8), E(dot2c_i32_i16), E(mul_lo_u32), E(mul_u24_sdwa), E(pk_mad_u16),
      E(pk_add_u16), E(pk_mul_lo_u16), E(med3_i32), E(cvt_pk_i16_i32), E(alignbyte), E(lshl_add), E(add3), E(mad_u16),
      E(cvt_i32_f32), E(cvt_rpi_i32_f32), E(sad_u8), E(msad_u8), E(lerp_u8), E(bfi_b32), E(pk_fma_f16)};
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;          // 8 blocks x 4 waves = 32 waves / CU = 8 waves / SIMD
  unsigned *out;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 4096;
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  for (auto &en : ents) {
    hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, out, 64, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4 * iters * 16;            // wave-instructions
    const double per_simd = winstr / (cus * 4.0);
    const double ns_per = ms * 1e6 / per_simd;
    printf("%-18s %8.3f ms  %6.3f ns / wave-instr / SIMD  (= %.2f clk @2.4GHz)   %.1f G wave-instr/s chip\n", en.name, ms, ns_per,
           ns_per * 2.4, winstr / (ms * 1e-3) / 1e9);
  }
  return 0;
}
