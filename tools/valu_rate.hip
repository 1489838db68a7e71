// tools/valu_rate.hip -- issue rate of the integer multiply-accumulate forms the scalers could use (gfx950): clocks per wave64 instruction on one SIMD, measured with
// s_memtime around 16 independent accumulator chains (no dependency stalls), 1 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/_valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP>
__global__ void k(uint32_t *out, unsigned long long *clk, int iters) {
  uint32_t a[16], x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9E3779B9u;
  for (int i = 0; i < 16; i++) a[i] = i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#define D2(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define D4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define M24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define M16(i) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define PKM(i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define ADD(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define SDW(i) asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_0" : "+v"(a[i]) : "v"(x), "v"(y));
#define PRM(i) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define D2C(i) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define D2I(i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define M64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&a[i & ~1]) : "v"(x), "v"(y) : "vcc");
    if (OP == 0) { REP16(D2) } else if (OP == 1) { REP16(D4) } else if (OP == 2) { REP16(M24) } else if (OP == 3) { REP16(M16) } else if (OP == 4) { REP16(PKM) }
    else if (OP == 5) { REP16(FMA) } else if (OP == 6) { REP16(ADD) } else if (OP == 7) { REP16(SDW) } else if (OP == 8) { REP16(PRM) } else if (OP == 9) { REP16(D2C) } else if (OP == 10) { REP16(D2I) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < 16; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int PK>
__global__ void kpk(float *out, unsigned long long *clk, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[8], x = {1.0f + threadIdx.x, 2.0f}, y = {0.5f, 0.25f};
  for (int i = 0; i < 8; i++) a[i] = f2{(float)i, 1.0f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#define PF(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 7]) : "v"(x), "v"(y));
    REP16(PF)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
  uint32_t *out; unsigned long long *clk, h;
  hipMalloc(&out, 1 << 20); hipMalloc(&clk, 8);
  const char *names[] = {"v_dot2_u32_u16", "v_dot4_u32_u8", "v_mad_u32_u24", "v_mad_u32_u16", "v_pk_mad_u16", "v_fma_f32", "v_add_u32", "v_mul_u32_u24_sdwa", "v_perm_b32", "v_dot2c_i32_i16", "v_dot2_i32_i16"};
  const int iters = 4096;
  for (int waves = 1; waves <= 8; waves *= 2) {            // block of waves * 4 wave64 on one CU: `waves` per SIMD
    printf("== %d wave(s) per SIMD: s_memtime ticks (100 MHz) per 16 instructions x %d; clocks per instruction per wave at 2.4 GHz in brackets\n", waves, iters);
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves), 0, 0, out, clk, iters); hipDeviceSynchronize(); hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); \
    printf("%-22s %8llu ticks  [%.2f clk / instr / wave, %.2f per SIMD]\n", names[OP], h, (double)h * 24.0 / (16.0 * iters), (double)h * 24.0 / (16.0 * iters * waves));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    hipLaunchKernelGGL(kpk<1>, dim3(1), dim3(256 * waves), 0, 0, (float *)out, clk, iters); hipDeviceSynchronize(); hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-22s %8llu ticks  [%.2f clk / instr / wave, %.2f per SIMD]\n", "v_pk_fma_f32", h, (double)h * 24.0 / (16.0 * iters), (double)h * 24.0 / (16.0 * iters * waves));
  }
  return 0;
}
