// lgpu_common.h -- shared device helpers / launch plumbing for liblivesgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lives_gpu.h"

namespace lgpu {

// ---- error plumbing -------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int ensure_init();   // LGPU_OK or LGPU_E_NODEVICE

#define LGPU_HIP(expr)                                                              \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      lgpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LGPU_E_HIP;                                                            \
    }                                                                               \
  } while (0)

#define LGPU_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) { lgpu::set_error("%s: %s", __func__, msg); return LGPU_E_BADARG; } \
  } while (0)

#define LGPU_CHECK_LAUNCH()                                                         \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      lgpu::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return LGPU_E_HIP;                                                            \
    }                                                                               \
  } while (0)

// 256-byte LUT passed by value as a kernel argument (lands in the kernarg segment -> scalar loads)
struct Lut8 {
  uint32_t w[64];
};
static inline Lut8 pack_lut(const uint8_t *lut8) {
  Lut8 l;
  if (lut8) __builtin_memcpy(l.w, lut8, 256);
  else for (int i = 0; i < 64; i++) { uint32_t b = 4u * i; l.w[i] = b | ((b + 1) << 8) | ((b + 2) << 16) | ((b + 3) << 24); }
  return l;
}

// The frames of one batched effect launch (lgpu_fx_batch): planes of the first input, of the second input (transitions) and of the output of every frame; the
// kernels take the frame from the grid (z index, or y where the grid is one-dimensional).  A single-frame entry point passes a table with one frame.
struct FxFrames {
  const uint8_t *in0[LGPU_FX_MAX_FRAMES][4];
  const uint8_t *in1[LGPU_FX_MAX_FRAMES][4];
  uint8_t *out[LGPU_FX_MAX_FRAMES][4];
};
// the batched forms behind the single-frame entry points and lgpu_fx_batch (argument checks included; ensure_init() is the caller's)
int softlight_n(const FxFrames &F, int nframes, const int irow[4], const int orow[4], int width, int height, int palette, int unclamped, hipStream_t st);
int blend_chroma_n(const FxFrames &X, int nframes, int irow1, int irow2, int orow, int width, int height, int psize, const int *bf, hipStream_t st);
int blend_luma_n(const FxFrames &X, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int psize, int pal_order, const int *thresh, hipStream_t st);
int blend_multi_n(const FxFrames &X, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int is_bgr, const int *bf, hipStream_t st);
int transition_n(const FxFrames &F, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int psize, const double *amounts, hipStream_t st);
int gauss5_colorkey_n(const FxFrames &F, int nframes, int irow0, int irow1, int orow, int width, int height, int psize, int is_bgr, double delta, double opac,
                      int col_r, int col_g, int col_b, hipStream_t st);
int yuv411_to_rgb_n(const FxFrames &F, int nframes, int width_mp, int height, int orow, int out_order, int out_alpha, int clamping_unclamped, hipStream_t st);

// conversion tables resident in device memory (uploaded once by lgpu_init)
struct DeviceTables {
  int32_t *yuv2rgb[4];   // [5][256] each
  int32_t *rgb2yuv[4];   // [9][256] each
  int32_t *luma;         // [3][256]: 65536-scaled unclamped BT.601 luma weights (libweed/weed-plugin-utils.c:879-895)
};
const DeviceTables *device_tables();
int device_cus();      // CUs of the current device

constexpr int kBlock = 256;   // 4 wavefronts of 64

// frames of one geometry for the batch forms of the single-plane kernels (lgpu_*_batch): the frame is the grid's z index; travels in the kernarg segment
struct FrameTab { const uint8_t *src[16]; uint8_t *dst[16]; };        // 16 = LGPU_FX_MAX_FRAMES

// ---- launch-shape / ablation switches ------------------------------------------------------------------
// Every switch the launch paths consult lives in ONE process-wide table of atomics: filled once from the environment (LGPU_<NAME>) at first use, changed
// afterwards only through lgpu_tuning_set() (tests, sweeps).  No launch path calls getenv(): the host (LiVES) calls setenv() at run time from other threads.
enum Tune {
  TUNE_PBH_ALIGNED, TUNE_PBH_TH, TUNE_PB_NO_DOUBLE, TUNE_PB_NO_HALF3, TUNE_PB_NO_PAIRS, TUNE_PB_NO_GATHER, TUNE_PB_NO_UP, TUNE_PB_UP_RB,
  TUNE_GCK_TH, TUNE_CHAIN_SPARE_WGS, TUNE_SEP2_LDS_KB, TUNE_NO_SEP2P, TUNE_NO_SEP2P_MFMA, TUNE_PLAN_DEBUG,
  TUNE_SEP2P_FORCE, TUNE_PB_CACHE_MAX, TUNE_K2_WGS, TUNE_SOFT_NO_S, TUNE_SOFT_RB, TUNE_EDGE_NO_S, TUNE_EDGE_TH, TUNE_PBH_ORDER, TUNE_PBH_OCC, TUNE_PBH_GROUP, TUNE_G5_MFMA, TUNE_RGB2YUV_NO_S, TUNE_UYVY_NO_S, TUNE_REPACK_NO_S, TUNE_DISABLE_HALF8, TUNE_NO_SEP2, TUNE_SEP2P_TH, TUNE_G5_CLASSIC,
  TUNE_GAUSS5_NO_ROWS, TUNE_PB_NO_PRE, TUNE_PB_LDS_KB, TUNE_PHASE_PROFILE, TUNE_PB_TILE_ORDER, TUNE_PB_CHAIN_GROUP, TUNE_SEAM_STAGED, TUNE_COUNT
};
int tune(Tune t);                                  // the value, or -1 when the switch is unset
static inline bool tune_on(Tune t) { return tune(t) > 0; }

static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

#if defined(__HIPCC__)
// ---- device helpers ---------------------------------------------------------------------------------
// stage a kernarg LUT into LDS (256 B) -- one dword per lane for the first wave
__device__ __forceinline__ void stage_lut(uint8_t *lds_lut, const Lut8 &lut) {
  if (threadIdx.x < 64) reinterpret_cast<uint32_t *>(lds_lut)[threadIdx.x] = lut.w[threadIdx.x];
}
__device__ __forceinline__ uint32_t lut3_rgba(const uint8_t *l, uint32_t p) {   // LUT on bytes 0..2, keep byte 3
  return (uint32_t)l[p & 0xFF] | ((uint32_t)l[(p >> 8) & 0xFF] << 8) | ((uint32_t)l[(p >> 16) & 0xFF] << 16) | (p & 0xFF000000u);
}
__device__ __forceinline__ uint32_t lut3_argb(const uint8_t *l, uint32_t p) {   // LUT on bytes 1..3, keep byte 0
  return (p & 0xFFu) | ((uint32_t)l[(p >> 8) & 0xFF] << 8) | ((uint32_t)l[(p >> 16) & 0xFF] << 16) | ((uint32_t)l[p >> 24] << 24);
}
__device__ __forceinline__ uint32_t lut4(const uint8_t *l, uint32_t p) {
  return (uint32_t)l[p & 0xFF] | ((uint32_t)l[(p >> 8) & 0xFF] << 8) | ((uint32_t)l[(p >> 16) & 0xFF] << 16) | ((uint32_t)l[p >> 24] << 24);
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
// [1 4 6 4 1] on packed 16-bit lanes, a + e + 4 (b + d) + 6 c + k, without a 32-bit multiply: the operands of the vertical pass exceed 24 bits, so `6u * c` became
// v_mul_lo_u32 (a quarter of the vector rate); ((b + c + d) << 2) + (c << 1) + (a + e + k) is two v_add3 and two v_lshl_add
__device__ __forceinline__ uint32_t gauss5_taps(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t k = 0u) { return ((b + c + d) << 2) + (c << 1) + (a + e + k); }

// --- per-lane 4-pixel gather / scatter ------------------------------------------------------------------
// 3-byte pixels: 12 contiguous bytes d0 d1 d2 -> four dwords [c0 c1 c2 x]
__device__ __forceinline__ void unpack3(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t p[4]) {
  p[0] = d0;
  p[1] = __builtin_amdgcn_perm(d1, d0, 0x0C050403u);   // d0.b3 d1.b0 d1.b1
  p[2] = __builtin_amdgcn_perm(d2, d1, 0x0C040302u);   // d1.b2 d1.b3 d2.b0
  p[3] = d2 >> 8;
}
// four dwords [o0 o1 o2 x] -> 12 contiguous bytes
__device__ __forceinline__ void pack3(const uint32_t q[4], uint32_t &w0, uint32_t &w1, uint32_t &w2) {
  w0 = __builtin_amdgcn_perm(q[1], q[0], 0x04020100u);   // q0.b0 q0.b1 q0.b2 q1.b0
  w1 = __builtin_amdgcn_perm(q[2], q[1], 0x05040201u);   // q1.b1 q1.b2 q2.b0 q2.b1
  w2 = __builtin_amdgcn_perm(q[3], q[2], 0x06050402u);   // q2.b2 q3.b0 q3.b1 q3.b2
}

#endif

}  // namespace lgpu
