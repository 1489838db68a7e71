// palette.hip -- the rest of the palette matrix (SURVEY 8a row P1 / 8f "next" 2):
//   K4  packed RGB family -> YUV888 / YUVA8888 / YUV(A)444(4)P / UYVY / YUYV / YUV420P / YUV422P
//       src/colourspace.c:5129-6440, pixel maths :2119-2192
//   K3  YUV888 / YUVA8888 / YUV(A)444(4)P / UYVY / YUYV -> packed RGB family
//       src/colourspace.c:2750-3258, :6616-7102, :7200-7498, pixel maths :2345-2459
// All of it is per-pixel table arithmetic: HBM-bound byte work, lane = one pixel pair, tables staged in LDS.
#include "lgpu_common.h"
#include <mutex>

namespace lgpu {

struct PalArgs {
  const uint8_t *src[4];
  uint8_t *dst[4];
  int irow[4], orow[4];
  int width, height;
  int order, alpha_in, fmt, alpha_out;
  int unclamped;
  const int32_t *tables;      // rgb2yuv [9][256] or yuv2rgb [5][256] of the selected (clamping, subspace)
  const uint16_t *lut16;      // k_rgb_to_yuv, UYVY / YUYV only: create_gamma_lut's 65536 entries, applied inline (rgb2uyvy_with_gamma)
};

// the frame of a batched launch (lgpu_rgb_to_yuv_batch / lgpu_yuv_to_rgb_batch): planes by the grid's z index; a single-frame call passes a table of one
__device__ __forceinline__ void pal_frame(PalArgs &a, const FxFrames &F) {
#pragma unroll
  for (int k = 0; k < 4; k++) { a.src[k] = F.in0[blockIdx.z][k]; a.dst[k] = F.out[blockIdx.z][k]; }
}

// ---- K4 -------------------------------------------------------------------------------------------------------------
struct R2Y {
  const int32_t *t;           // LDS [9][256]
  int min_y, max_y, min_uv, max_uv;
  // rgb2yuv (:2119-2127): short a = spc_rnd(sum) (>> 16 at PB_QUALITY_MED, :832-843); upper clamp first, then lower
  __device__ __forceinline__ int Y(int r, int g, int b) const {
    const int a = (short)((t[r] + t[256 + g] + t[512 + b]) >> 16);
    return a > max_y ? max_y : a < min_y ? min_y : a;
  }
  __device__ __forceinline__ int Uraw(int r, int g, int b) const { return (short)((t[768 + r] + t[1024 + g] + t[1280 + b]) >> 16); }
  __device__ __forceinline__ int Vraw(int r, int g, int b) const { return (short)((t[1536 + r] + t[1792 + g] + t[2048 + b]) >> 16); }
  __device__ __forceinline__ int cuv(int a) const { return a > max_uv ? max_uv : a < min_uv ? min_uv : a; }
  // rgb2uyvy_with_gamma (:2146-2159): the table sum's top 16 bits index the LUT, the LUT's high byte is the sample
  __device__ __forceinline__ int Yg(const uint16_t *l, int r, int g, int b) const {
    const int a = l[((uint32_t)(t[r] + t[256 + g] + t[512 + b]) >> 8) & 0xFFFF] >> 8;
    return a > max_y ? max_y : a < min_y ? min_y : a;
  }
  __device__ __forceinline__ int Ug(const uint16_t *l, int r, int g, int b) const { return cuv(l[((uint32_t)(t[768 + r] + t[1024 + g] + t[1280 + b]) >> 8) & 0xFFFF] >> 8); }
  __device__ __forceinline__ int Vg(const uint16_t *l, int r, int g, int b) const { return cuv(l[((uint32_t)(t[1536 + r] + t[1792 + g] + t[2048 + b]) >> 8) & 0xFFFF] >> 8); }
};

// init_average (:190-216): cavgu is the integer mean; cavgc mixes float and double exactly as written there:
//   fa = (float)((double)(float)(x - 128) * 255. / 244.), fb likewise, fc = (float)((double)(fa + fb) * 224. / 512. + 128.), clamped to 16 .. 240 and truncated.
// The clamped average without the double divisions: fa(x) = (float)((x - 128) * 255. / 244.) is a 256-entry table (built per workgroup with the formula above),
// and (float)((double)(fa + fb) * 224. / 512. + 128.) is ONE rounding of an exactly representable double (a 24-bit sum times 7 / 16 plus 128), i.e. fmaf(fa + fb, 0.4375f, 128.f).
__device__ __forceinline__ float cavg_fa(int x) { return (float)__ddiv_rn(__dmul_rn((double)(float)(x - 128), 255.), 244.); }
__device__ __forceinline__ int cavg_lds(int clamped, const float *s_fa, int x, int y) {
  if (!clamped) {
    const int c = (((x - 128) + (y - 128)) >> 1) + 128;
    return c > 255 ? 255 : c < 0 ? 0 : c;
  }
  const float fc = __fmaf_rn(__fadd_rn(s_fa[x], s_fa[y]), 0.4375f, 128.f);
  return (int)__builtin_amdgcn_fmed3f(fc, 16.f, 240.f);       // > 240 -> 240, < 16 -> 16, else truncated: one v_med3_f32 instead of two compares and two v_cndmask_b32_e32 (which issue at 1 / 4.6 rate)
}
// fa(x) WITHOUT the table: d = x - 128 is exact in float, 255 / 244 = KHI + KLO to 48 bits, and fmaf(d, KHI, d * KLO) rounds the exact d * KHI + fl(d * KLO) once --
// for all 256 bytes the float the double division gives (k_build_cavgc compares the two forms entry by entry; tests/test_gpu_parity.py::test_chroma_average_table).
// Four vector operations instead of an LDS gather: for kernels whose LDS pipe is full of table gathers (YUV411 -> RGB: 67 % bank-conflict cycles).
__device__ __forceinline__ float cavg_fa_arith(float d) { return __fmaf_rn(d, 0x1.0b8a7ep+0f, __fmul_rn(d, -0x1.92e2ap-28f)); }
__device__ __forceinline__ int cavg_arith(int clamped, int x, int y) {
  if (!clamped) {
    const int c = (((x - 128) + (y - 128)) >> 1) + 128;
    return c > 255 ? 255 : c < 0 ? 0 : c;
  }
  const float fc = __fmaf_rn(__fadd_rn(cavg_fa_arith((float)(x - 128)), cavg_fa_arith((float)(y - 128))), 0.4375f, 128.f);
  return (int)__builtin_amdgcn_fmed3f(fc, 16.f, 240.f);
}
__device__ unsigned int d_cavg_forms_differ = 0;
__device__ const uint8_t *d_cavgc = nullptr;
// built with the fma form (cavg_lds) the kernels use; tests/test_gpu_parity.py::test_chroma_average_table compares all 65,536 entries with the reference's table
__global__ __launch_bounds__(256) void k_build_cavgc(uint8_t *t) {
  __shared__ float s_fa[256];
  s_fa[threadIdx.x] = cavg_fa((int)threadIdx.x);
  __syncthreads();
  const int v = cavg_lds(1, s_fa, blockIdx.x, threadIdx.x);
  t[blockIdx.x * 256 + threadIdx.x] = (uint8_t)v;
  if (cavg_arith(1, blockIdx.x, threadIdx.x) != v) atomicAdd(&d_cavg_forms_differ, 1u);      // the table-free form must say the same for every pair
}
// the kernels' average: the fma form on a per-workgroup fa table in LDS.  Every kernel that calls cavg() runs cavg_init() first, before any thread returns.
// (Round 1 gathered from a 64 KB table in global memory; the table is still built, for the test that compares it with the reference's.)
__shared__ float s_cavg_fa[256];
__device__ __forceinline__ void cavg_init() {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_cavg_fa[i] = cavg_fa(i);
  __syncthreads();
}
__device__ __forceinline__ int cavg(int clamped, int x, int y) { return cavg_lds(clamped, s_cavg_fa, x & 255, y & 255); }   // x, y are bytes (the reference indexes cavg[(x << 8) + y])

__device__ __forceinline__ void load_rgb(const uint8_t *p, int order, int &r, int &g, int &b) {
  if (order == 0) { r = p[0]; g = p[1]; b = p[2]; }
  else if (order == 1) { r = p[2]; g = p[1]; b = p[0]; }
  else { r = p[1]; g = p[2]; b = p[3]; }
}

// a pair of neighbouring pixels: one 8-byte load when they are 4-byte pixels at an 8-byte aligned address (the usual RGBA32 / BGRA32 / ARGB32 row), byte loads otherwise
__device__ __forceinline__ void load_rgb_pair(const uint8_t *s, int ips, int order, int &r0, int &g0, int &b0, int &r1, int &g1, int &b1) {
  if (ips == 4 && (reinterpret_cast<uintptr_t>(s) & 7) == 0) {
    const uint2 v = *reinterpret_cast<const uint2 *>(s);
    const int sh = order == 2 ? 8 : 0;                     // ARGB: colours in bytes 1..3
    const int c0a = (v.x >> sh) & 0xFF, c1a = (v.x >> (sh + 8)) & 0xFF, c2a = (v.x >> (sh + 16)) & 0xFF;
    const int c0b = (v.y >> sh) & 0xFF, c1b = (v.y >> (sh + 8)) & 0xFF, c2b = (v.y >> (sh + 16)) & 0xFF;
    if (order == 1) { r0 = c2a; g0 = c1a; b0 = c0a; r1 = c2b; g1 = c1b; b1 = c0b; }
    else { r0 = c0a; g0 = c1a; b0 = c2a; r1 = c0b; g1 = c1b; b1 = c2b; }
    return;
  }
  load_rgb(s, order, r0, g0, b0);
  load_rgb(s + ips, order, r1, g1, b1);
}
__device__ __forceinline__ void store_y_pair(uint8_t *dy, int y0, int y1) {
  if ((reinterpret_cast<uintptr_t>(dy) & 1) == 0) *reinterpret_cast<uint16_t *>(dy) = (uint16_t)((y0 & 0xFF) | ((y1 & 0xFF) << 8));
  else { dy[0] = (uint8_t)y0; dy[1] = (uint8_t)y1; }
}

template <int ORDER, int FMT>
__global__ __launch_bounds__(kBlock) void k_rgb_to_yuv(PalArgs a, const FxFrames F) {
  pal_frame(a, F);
  if (FMT == 4) cavg_init();                              // only the 4:2:0 walk averages
  __shared__ int32_t s_t[9 * 256];
  for (int i = threadIdx.x; i < 9 * 256; i += kBlock) s_t[i] = a.tables[i];
  __syncthreads();
  R2Y c;
  c.t = s_t;
  if (a.unclamped) { c.min_y = c.min_uv = 0; c.max_y = c.max_uv = 255; }
  else { c.min_y = c.min_uv = 16; c.max_y = 235; c.max_uv = 240; }                 // set_conversion_arrays :361-370
  const int ips = (ORDER == 2 || a.alpha_in) ? 4 : 3;
  const int npairs = a.width >> 1;                                                     // an odd last pixel is never converted (:5761)
  const int px = blockIdx.x * kBlock + threadIdx.x;
  if (px >= npairs) return;
  const int x = px * 2;
  // 4:2:0 walks chroma rows (two luma rows each), everything else luma rows
  const int nrows = FMT == 4 ? a.height >> 1 : a.height;
  for (int yy = blockIdx.y; yy < nrows; yy += gridDim.y) {
    if (FMT == 4) {
      // chroma row k = avg_chroma(row 2k+2, row 2k+1); the last one is row 2k+1 alone (:6302-6315 at compact strides)
      const int k = yy;
      int cu1 = 0, cv1 = 0;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint8_t *s = a.src[0] + (size_t)(2 * k + j) * a.irow[0] + (size_t)x * ips;
        int r0, g0, b0, r1, g1, b1;
        load_rgb_pair(s, ips, ORDER, r0, g0, b0, r1, g1, b1);
        store_y_pair(a.dst[0] + (size_t)(2 * k + j) * a.orow[0] + x, c.Y(r0, g0, b0), c.Y(r1, g1, b1));
        if (j == 1) { cu1 = c.cuv(c.Uraw(r0, g0, b0)); cv1 = c.cuv(c.Vraw(r1, g1, b1)); }
      }
      if (2 * k + 2 < a.height) {
        const uint8_t *s = a.src[0] + (size_t)(2 * k + 2) * a.irow[0] + (size_t)x * ips;
        int r0, g0, b0, r1, g1, b1;
        load_rgb_pair(s, ips, ORDER, r0, g0, b0, r1, g1, b1);
        cu1 = cavg(!a.unclamped, c.cuv(c.Uraw(r0, g0, b0)), cu1);
        cv1 = cavg(!a.unclamped, c.cuv(c.Vraw(r1, g1, b1)), cv1);
      }
      a.dst[1][(size_t)k * a.orow[1] + px] = (uint8_t)cu1;
      a.dst[2][(size_t)k * a.orow[2] + px] = (uint8_t)cv1;
      continue;
    }
    const int y = yy;
    const uint8_t *s = a.src[0] + (size_t)y * a.irow[0] + (size_t)x * ips;
    int r0, g0, b0, r1, g1, b1;
    load_rgb_pair(s, ips, ORDER, r0, g0, b0, r1, g1, b1);
    if ((FMT == 2 || FMT == 3) && a.lut16) {
      // the gamma twins clamp properly in both byte orders (rgb2yuyv_with_gamma :2194-2207 has the `else` rgb2yuyv lost)
      const uint32_t gy0 = (uint32_t)c.Yg(a.lut16, r0, g0, b0), gy1 = (uint32_t)c.Yg(a.lut16, r1, g1, b1);
      const uint32_t gu = (uint32_t)c.Ug(a.lut16, r0, g0, b0), gv = (uint32_t)c.Vg(a.lut16, r1, g1, b1);
      *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + (size_t)px * 4) =
          FMT == 2 ? (gu | (gy0 << 8) | (gv << 16) | (gy1 << 24)) : (gy0 | (gu << 8) | (gy1 << 16) | (gv << 24));
      continue;
    }
    const int y0 = c.Y(r0, g0, b0), y1 = c.Y(r1, g1, b1);
    if (FMT <= 1) {
      const int al0 = ORDER == 2 ? s[0] : ips == 4 ? s[3] : 255, al1 = ORDER == 2 ? s[ips] : ips == 4 ? s[ips + 3] : 255;
      const int u0 = c.cuv(c.Uraw(r0, g0, b0)), v0 = c.cuv(c.Vraw(r0, g0, b0));
      const int u1 = c.cuv(c.Uraw(r1, g1, b1)), v1 = c.cuv(c.Vraw(r1, g1, b1));
      if (FMT == 0) {
        const int ops = a.alpha_out ? 4 : 3;
        uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + (size_t)x * ops;
        d[0] = (uint8_t)y0; d[1] = (uint8_t)u0; d[2] = (uint8_t)v0;
        if (a.alpha_out) d[3] = (uint8_t)al0;
        d[ops] = (uint8_t)y1; d[ops + 1] = (uint8_t)u1; d[ops + 2] = (uint8_t)v1;
        if (a.alpha_out) d[ops + 3] = (uint8_t)al1;
      } else {
        uint8_t *dy = a.dst[0] + (size_t)y * a.orow[0] + x, *du = a.dst[1] + (size_t)y * a.orow[1] + x, *dv = a.dst[2] + (size_t)y * a.orow[2] + x;
        dy[0] = (uint8_t)y0; dy[1] = (uint8_t)y1; du[0] = (uint8_t)u0; du[1] = (uint8_t)u1; dv[0] = (uint8_t)v0; dv[1] = (uint8_t)v1;
        if (a.alpha_out) { uint8_t *da = a.dst[3] + (size_t)y * a.orow[3] + x; da[0] = (uint8_t)al0; da[1] = (uint8_t)al1; }
      }
    } else {
      // rgb2uyvy / rgb2yuyv (:2162-2192): U of the first pixel, V of the SECOND, no averaging
      const int ur = c.Uraw(r0, g0, b0), vr = c.Vraw(r1, g1, b1);
      if (FMT == 2) {
        *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + (size_t)px * 4) =
            (uint32_t)c.cuv(ur) | ((uint32_t)y0 << 8) | ((uint32_t)c.cuv(vr) << 16) | ((uint32_t)y1 << 24);
      } else if (FMT == 3) {
        // rgb2yuyv lost its `else`: only the lower chroma clamp survives (:2183-2191)
        const uint32_t u = (uint32_t)(ur < c.min_uv ? c.min_uv : ur) & 0xFF, v = (uint32_t)(vr < c.min_uv ? c.min_uv : vr) & 0xFF;
        *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + (size_t)px * 4) = (uint32_t)y0 | (u << 8) | ((uint32_t)y1 << 16) | (v << 24);
      } else {       // 4:2:2 planar
        store_y_pair(a.dst[0] + (size_t)y * a.orow[0] + x, y0, y1);
        a.dst[1][(size_t)y * a.orow[1] + px] = (uint8_t)c.cuv(ur);
        a.dst[2][(size_t)y * a.orow[2] + px] = (uint8_t)c.cuv(vr);
      }
    }
  }
}

// RGBA32 / BGRA32 -> YUV420P for aligned frames (the sink hand-off of a 1080p frame): the arithmetic of k_rgb_to_yuv<.., 4> in the shape that paid off for K2
// (yuv.hip, k_yuv420p_to_rgb_s).  A lane owns a cell of 4 x 2 pixels -- luma rows 2u + 1 and 2u + 2, whose chroma the reference averages into chroma row u
// (:6302-6315; row 0 is luma only, the last chroma row has row 2u + 1 alone) -- so every source row is read once (16 bytes per lane), luma leaves as dwords and
// chroma as 16-bit pairs; cells are numbered linearly over the frame, the first cell's pixels are requested before the tables (9 KB + the chroma-average table) are
// staged, and a 512-thread workgroup stages them once for 2,048 cells.  k_rgb_to_yuv<.., 4>: one workgroup of 256 pixel pairs per chroma row, the second row of every
// pair of rows read twice.
// The nine rgb -> yuv tables as THREE tables of {Y, U, V} contributions per colour byte, 16 bytes an entry: three ds_read_b128 per pixel instead of six to nine
// ds_read_b32 (the cell kernels are bound by their table gathers: 4:4:4 packed went from 0.19 to 0.60 of the roofline with this layout and vector stores,
// profiles/r06/yuv444_s.txt).  Same sums, same clamps as R2Y.
typedef int pal_i4 __attribute__((ext_vector_type(4)));
struct R2Y3 {
  const pal_i4 *p;            // LDS [3][256]
  int min_y, max_y, min_uv, max_uv;
  __device__ __forceinline__ void stage(pal_i4 *s_p, const int32_t *tables, int unclamped) {      // every thread of the workgroup; the caller puts a barrier behind it
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
      const int c = i >> 8, e = i & 255;
      s_p[i] = pal_i4{tables[c * 256 + e], tables[768 + c * 256 + e], tables[1536 + c * 256 + e], 0};
    }
    p = s_p;
    if (unclamped) { min_y = min_uv = 0; max_y = max_uv = 255; } else { min_y = min_uv = 16; max_y = 235; max_uv = 240; }
  }
  __device__ __forceinline__ pal_i4 sums(int r, int g, int b) const { return p[r] + p[256 + g] + p[512 + b]; }
  __device__ __forceinline__ int Y(const pal_i4 s) const { const int a = (short)(s.x >> 16); return a > max_y ? max_y : a < min_y ? min_y : a; }
  __device__ __forceinline__ int Uraw(const pal_i4 s) const { return (short)(s.y >> 16); }
  __device__ __forceinline__ int Vraw(const pal_i4 s) const { return (short)(s.z >> 16); }
  __device__ __forceinline__ int cuv(int a) const { return a > max_uv ? max_uv : a < min_uv ? min_uv : a; }
};

template <int ORDER>
__global__ __launch_bounds__(512) void k_rgb_to_yuv420_s(PalArgs a, uint32_t gmagic, const FxFrames F) {
  pal_frame(a, F);
  __shared__ pal_i4 s_p[3 * 256];
  typedef unsigned pu4 __attribute__((ext_vector_type(4)));
  const int ngr = a.width >> 2, hc = a.height >> 1, nunits = hc + 1;      // unit 0: row 0 alone; unit u + 1: rows 2u + 1, 2u + 2 -> chroma row u
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t unit = __umulhi(idx, gmagic);                                     // floor magic: the quotient or one less
  uint32_t gx = idx - unit * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; unit++; }
  const bool valid = unit < (uint32_t)nunits;
  const int ra = unit ? 2 * (int)unit - 1 : 0, rb = ra + 1;                  // the cell's rows; rb unused for unit 0
  const bool has_b = valid && unit && rb < a.height;
  pu4 qa = {0, 0, 0, 0}, qb = {0, 0, 0, 0};
  if (valid) qa = *reinterpret_cast<const pu4 *>(a.src[0] + (size_t)ra * a.irow[0] + 16 * (size_t)gx);
  if (has_b) qb = *reinterpret_cast<const pu4 *>(a.src[0] + (size_t)rb * a.irow[0] + 16 * (size_t)gx);
  R2Y3 c;
  c.stage(s_p, a.tables, a.unclamped);
  cavg_init();                                                               // ends with the workgroup barrier
  if (!valid) return;
  auto rgb = [](uint32_t p, int &r, int &g, int &b) {
    const int c0 = p & 0xFF, c1 = (p >> 8) & 0xFF, c2 = (p >> 16) & 0xFF;
    if (ORDER == 1) { r = c2; g = c1; b = c0; } else { r = c0; g = c1; b = c2; }
  };
  // one row of the cell: four luma samples, U of pixels 0 and 2, V of pixels 1 and 3 (rgb2yuv's pairing, :6250-6322)
  auto row = [&](pu4 q, uint32_t &yy, int u[2], int v[2], bool chroma) {
    const uint32_t px[4] = {q.x, q.y, q.z, q.w};
    yy = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int r, g, b;
      rgb(px[i], r, g, b);
      const pal_i4 sm = c.sums(r, g, b);
      yy |= (uint32_t)c.Y(sm) << (8 * i);
      if (chroma) { if (i & 1) v[i >> 1] = c.cuv(c.Vraw(sm)); else u[i >> 1] = c.cuv(c.Uraw(sm)); }
    }
  };
  const int ylim = 2 * hc;                                                   // luma rows the reference writes (an odd last row is read for its chroma only)
  uint32_t ya, yb;
  int ua[2] = {0, 0}, va[2] = {0, 0}, ub[2] = {0, 0}, vb[2] = {0, 0};
  row(qa, ya, ua, va, unit != 0);
  if (ra < ylim) *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)ra * a.orow[0] + 4 * (size_t)gx) = ya;
  if (!unit) return;
  if (has_b) {
    row(qb, yb, ub, vb, true);
    if (rb < ylim) *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)rb * a.orow[0] + 4 * (size_t)gx) = yb;
#pragma unroll
    for (int i = 0; i < 2; i++) { ua[i] = cavg(!a.unclamped, ub[i], ua[i]); va[i] = cavg(!a.unclamped, vb[i], va[i]); }
  }
  const int k = (int)unit - 1;
  *reinterpret_cast<uint16_t *>(a.dst[1] + (size_t)k * a.orow[1] + 2 * (size_t)gx) = (uint16_t)((ua[0] & 0xFF) | ((ua[1] & 0xFF) << 8));
  *reinterpret_cast<uint16_t *>(a.dst[2] + (size_t)k * a.orow[2] + 2 * (size_t)gx) = (uint16_t)((va[0] & 0xFF) | ((va[1] & 0xFF) << 8));
}

// RGBA32 / BGRA32 -> UYVY / YUYV / YUV422P on aligned frames: the same cell shape, one row per cell (rgb2uyvy / rgb2yuyv :2162-2192: U of a pair's first pixel,
// V of its second, no averaging; the YUYV form keeps only the lower chroma clamp, as the reference does).  FMT as in k_rgb_to_yuv: 2 UYVY, 3 YUYV, 5 planar 4:2:2.
template <int ORDER, int FMT>
__global__ __launch_bounds__(512) void k_rgb_to_yuv422_s(PalArgs a, uint32_t gmagic, const FxFrames F) {
  pal_frame(a, F);
  __shared__ pal_i4 s_p[3 * 256];
  typedef unsigned pu4 __attribute__((ext_vector_type(4)));
  const int ngr = a.width >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);                                        // floor magic: the quotient or one less
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  const bool valid = y < (uint32_t)a.height;
  pu4 q = {0, 0, 0, 0};
  if (valid) q = *reinterpret_cast<const pu4 *>(a.src[0] + (size_t)y * a.irow[0] + 16 * (size_t)gx);
  R2Y3 c;
  c.stage(s_p, a.tables, a.unclamped);
  __syncthreads();
  if (!valid) return;
  const uint32_t px[4] = {q.x, q.y, q.z, q.w};
  uint32_t yy[4], uu[2], vv[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c0 = px[i] & 0xFF, c1 = (px[i] >> 8) & 0xFF, c2 = (px[i] >> 16) & 0xFF;
    const int r = ORDER == 1 ? c2 : c0, g = c1, b = ORDER == 1 ? c0 : c2;
    const pal_i4 sm = c.sums(r, g, b);
    yy[i] = (uint32_t)c.Y(sm);
    if (i & 1) { const int vr = c.Vraw(sm); vv[i >> 1] = FMT == 3 ? ((uint32_t)(vr < c.min_uv ? c.min_uv : vr) & 0xFF) : (uint32_t)c.cuv(vr); }
    else { const int ur = c.Uraw(sm); uu[i >> 1] = FMT == 3 ? ((uint32_t)(ur < c.min_uv ? c.min_uv : ur) & 0xFF) : (uint32_t)c.cuv(ur); }
  }
  if (FMT == 5) {
    *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + 4 * (size_t)gx) = yy[0] | (yy[1] << 8) | (yy[2] << 16) | (yy[3] << 24);
    *reinterpret_cast<uint16_t *>(a.dst[1] + (size_t)y * a.orow[1] + 2 * (size_t)gx) = (uint16_t)(uu[0] | (uu[1] << 8));
    *reinterpret_cast<uint16_t *>(a.dst[2] + (size_t)y * a.orow[2] + 2 * (size_t)gx) = (uint16_t)(vv[0] | (vv[1] << 8));
  } else {
    uint2 o;
    if (FMT == 2) { o.x = uu[0] | (yy[0] << 8) | (vv[0] << 16) | (yy[1] << 24); o.y = uu[1] | (yy[2] << 8) | (vv[1] << 16) | (yy[3] << 24); }
    else { o.x = yy[0] | (uu[0] << 8) | (yy[1] << 16) | (vv[0] << 24); o.y = yy[2] | (uu[1] << 8) | (yy[3] << 16) | (vv[1] << 24); }
    *reinterpret_cast<uint2 *>(a.dst[0] + (size_t)y * a.orow[0] + 8 * (size_t)gx) = o;
  }
}

// RGB24 / BGR24 / RGBA32 / BGRA32 -> YUV888 / YUVA8888 (packed 4:4:4) on aligned frames: a lane owns four pixels -- one 12- or 16-byte load, one 12- or 16-byte
// store (k_rgb_to_yuv<.., 0> stores six to eight single bytes per pixel pair: x16 1080p at 0.19 of the HBM roofline) -- and the nine tables sit in LDS as THREE
// tables of {Y, U, V} contributions per colour byte, 16 bytes an entry: three ds_read_b128 per pixel instead of nine ds_read_b32 (the table gathers are what
// the cell kernels above are bound by).  Same sums, same clamps as R2Y (rgb2yuv, :2119-2127).
// PLANAR: YUV444P / YUVA4444P -- the four pixels leave as one dword per plane
template <int ORDER, int IPS, int AOUT, int PLANAR = 0>
__global__ __launch_bounds__(512) void k_rgb_to_yuv444_s(PalArgs a, uint32_t gmagic, const FxFrames F) {
  pal_frame(a, F);
  typedef unsigned pu4 __attribute__((ext_vector_type(4)));
  typedef unsigned pu3 __attribute__((ext_vector_type(3)));
  typedef int pi4 __attribute__((ext_vector_type(4)));
  typedef pu3 pu3a __attribute__((aligned(4)));
  __shared__ pi4 s_p[3 * 256];
  const int ngr = a.width >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);                                        // floor magic: the quotient or one less
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  const bool valid = y < (uint32_t)a.height;
  uint32_t px[4] = {0, 0, 0, 0};
  if (valid) {
    const uint8_t *sp = a.src[0] + (size_t)y * a.irow[0] + 4 * IPS * (size_t)gx;
    if (IPS == 4) { const pu4 q = *reinterpret_cast<const pu4 *>(sp); px[0] = q.x; px[1] = q.y; px[2] = q.z; px[3] = q.w; }
    else {
      const pu3 q = *reinterpret_cast<const pu3a *>(sp);
      px[0] = q.x | 0xFF000000u; px[1] = __builtin_amdgcn_alignbyte(q.y, q.x, 3) | 0xFF000000u; px[2] = __builtin_amdgcn_alignbyte(q.z, q.y, 2) | 0xFF000000u; px[3] = (q.z >> 8) | 0xFF000000u;
    }
  }
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
    const int c = i >> 8, e = i & 255;
    s_p[i] = pi4{a.tables[c * 256 + e], a.tables[768 + c * 256 + e], a.tables[1536 + c * 256 + e], 0};
  }
  __syncthreads();
  if (!valid) return;
  const int min_y = a.unclamped ? 0 : 16, max_y = a.unclamped ? 255 : 235, min_uv = min_y, max_uv = a.unclamped ? 255 : 240;
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t c0 = px[i] & 0xFF, c1 = (px[i] >> 8) & 0xFF, c2 = (px[i] >> 16) & 0xFF;
    const pi4 er = s_p[ORDER == 1 ? c2 : c0], eg = s_p[256 + c1], eb = s_p[512 + (ORDER == 1 ? c0 : c2)];
    const int yr = (short)((er.x + eg.x + eb.x) >> 16), ur = (short)((er.y + eg.y + eb.y) >> 16), vr = (short)((er.z + eg.z + eb.z) >> 16);
    const int Y = yr > max_y ? max_y : yr < min_y ? min_y : yr, U = ur > max_uv ? max_uv : ur < min_uv ? min_uv : ur, V = vr > max_uv ? max_uv : vr < min_uv ? min_uv : vr;
    o[i] = (uint32_t)Y | ((uint32_t)U << 8) | ((uint32_t)V << 16) | (px[i] & 0xFF000000u);
  }
  if (PLANAR) {
    // byte k of every pixel -> plane k: selector bytes 0 / 4 of the pair, then of the pair of pairs
    auto plane = [&](uint32_t s01, uint32_t s23) { return __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], s23), __builtin_amdgcn_perm(o[1], o[0], s01), 0x05040100u); };
    *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + 4 * (size_t)gx) = plane(0x0C0C0400u, 0x0C0C0400u);
    *reinterpret_cast<uint32_t *>(a.dst[1] + (size_t)y * a.orow[1] + 4 * (size_t)gx) = plane(0x0C0C0501u, 0x0C0C0501u);
    *reinterpret_cast<uint32_t *>(a.dst[2] + (size_t)y * a.orow[2] + 4 * (size_t)gx) = plane(0x0C0C0602u, 0x0C0C0602u);
    if (AOUT) *reinterpret_cast<uint32_t *>(a.dst[3] + (size_t)y * a.orow[3] + 4 * (size_t)gx) = plane(0x0C0C0703u, 0x0C0C0703u);
  } else if (AOUT) *reinterpret_cast<pu4 *>(a.dst[0] + (size_t)y * a.orow[0] + 16 * (size_t)gx) = pu4{o[0], o[1], o[2], o[3]};
  else {
    const pu3 v = {(o[0] & 0xFFFFFFu) | (o[1] << 24), ((o[1] >> 8) & 0xFFFFu) | (o[2] << 16), ((o[2] >> 16) & 0xFFu) | (o[3] << 8)};
    *reinterpret_cast<pu3a *>(a.dst[0] + (size_t)y * a.orow[0] + 12 * (size_t)gx) = v;
  }
}

// ---- K3 -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_rgb(uint8_t *d, int order, int ops, const int32_t *t, int Y, int U, int V, int A) {
  // yuv2rgb_int (:2345-2349): CLAMP0255f(spc_rnd(RGB_Y[y] + R_Cr[v])) ...
  const int32_t yy = t[Y];
  const int r = clamp255((yy + t[256 + V]) >> 16), g = clamp255((yy + t[512 + U] + t[768 + V]) >> 16), b = clamp255((yy + t[1024 + U]) >> 16);
  if (ops == 4 && (reinterpret_cast<uintptr_t>(d) & 3) == 0) {           // a 4-byte pixel at a 4-byte aligned address: one store
    const uint32_t R = (uint32_t)r, G = (uint32_t)g, B = (uint32_t)b, AA = (uint32_t)A & 0xFF;
    *reinterpret_cast<uint32_t *>(d) = order == 0 ? (R | (G << 8) | (B << 16) | (AA << 24)) : order == 1 ? (B | (G << 8) | (R << 16) | (AA << 24)) : (AA | (R << 8) | (G << 16) | (B << 24));
    return;
  }
  if (order == 0) { d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)b; if (ops == 4) d[3] = (uint8_t)A; }
  else if (order == 1) { d[0] = (uint8_t)b; d[1] = (uint8_t)g; d[2] = (uint8_t)r; if (ops == 4) d[3] = (uint8_t)A; }
  else { d[0] = (uint8_t)A; d[1] = (uint8_t)r; d[2] = (uint8_t)g; d[3] = (uint8_t)b; }
}

template <int FMT, int ORDER>
__global__ __launch_bounds__(kBlock) void k_yuv_to_rgb(PalArgs a, const FxFrames F) {
  pal_frame(a, F);
  __shared__ int32_t s_t[5 * 256];
  for (int i = threadIdx.x; i < 5 * 256; i += kBlock) s_t[i] = a.tables[i];
  __syncthreads();
  const int ops = (ORDER == 2 || a.alpha_out) ? 4 : 3;
  const int npairs = (a.width + 1) >> 1;
  const int px = blockIdx.x * kBlock + threadIdx.x;
  if (px >= npairs) return;
  const int x = px * 2;
  const bool two = x + 1 < a.width;
  for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    int Y0, U0, V0, A0 = 255, Y1 = 0, U1 = 0, V1 = 0, A1 = 255;
    if (FMT == 0) {
      const int ips = a.alpha_in ? 4 : 3;
      const uint8_t *p = a.src[0] + (size_t)y * a.irow[0] + (size_t)x * ips;
      Y0 = p[0]; U0 = p[1]; V0 = p[2]; if (a.alpha_in) A0 = p[3];
      if (two) { Y1 = p[ips]; U1 = p[ips + 1]; V1 = p[ips + 2]; if (a.alpha_in) A1 = p[ips + 3]; }
    } else if (FMT == 1) {
      const uint8_t *py = a.src[0] + (size_t)y * a.irow[0] + x, *pu = a.src[1] + (size_t)y * a.irow[1] + x, *pv = a.src[2] + (size_t)y * a.irow[2] + x;
      Y0 = py[0]; U0 = pu[0]; V0 = pv[0];
      if (two) { Y1 = py[1]; U1 = pu[1]; V1 = pv[1]; }
      if (a.alpha_in) { const uint8_t *pa = a.src[3] + (size_t)y * a.irow[3] + x; A0 = pa[0]; if (two) A1 = pa[1]; }
    } else {
      const uint32_t m = *reinterpret_cast<const uint32_t *>(a.src[0] + (size_t)y * a.irow[0] + (size_t)px * 4);
      if (FMT == 2) { U0 = m & 0xFF; Y0 = (m >> 8) & 0xFF; V0 = (m >> 16) & 0xFF; Y1 = m >> 24; }      // uyvy2rgb :2410-2415
      else { Y0 = m & 0xFF; U0 = (m >> 8) & 0xFF; Y1 = (m >> 16) & 0xFF; V0 = m >> 24; }                   // yuyv2rgb :2418-2423
      U1 = U0; V1 = V0;
    }
    uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + (size_t)x * ops;
    put_rgb(d, ORDER, ops, s_t, Y0, U0, V0, A0);
    if (two) put_rgb(d + ops, ORDER, ops, s_t, Y1, U1, V1, A1);
  }
}

// UYVY / YUYV -> 4-byte RGB on aligned frames (a live-capture frame on its way into the chain): the cell shape of k_yuv420p_to_rgb_s / k_rgb_to_yuv420_s.
// A lane owns two macropixels (8 bytes in, four pixels = 16 bytes out), cells are numbered linearly over the frame, the macropixels are requested before the tables
// are staged, and the two chroma terms that share an index sit in one 8-byte LDS entry ({R_Cr, G_Cr}[v], {G_Cb, B_Cb}[u]): 1 + 2 / 2 gathers per pixel instead of 5.
// Same arithmetic as put_rgb() above: CLAMP0255f((RGB_Y[y] + ...) >> 16), alpha 255.
template <int FMT, int ORDER>
__global__ __launch_bounds__(512) void k_uyvy_to_rgb_s(PalArgs a, uint32_t gmagic, const FxFrames F) {
  pal_frame(a, F);
  __shared__ int32_t s_ty[256];
  __shared__ uint2 s_rg[256], s_gb[256];
  const int ngr = a.width >> 2;                          // cells per row
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);                    // floor magic: the quotient or one less
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  const bool valid = y < (uint32_t)a.height;
  uint2 m = make_uint2(0u, 0u);
  if (valid) m = *reinterpret_cast<const uint2 *>(a.src[0] + (size_t)y * a.irow[0] + 8 * (size_t)gx);
  for (int e = threadIdx.x; e < 256; e += blockDim.x) {
    s_ty[e] = a.tables[e];
    s_rg[e] = make_uint2((uint32_t)a.tables[256 + e], (uint32_t)a.tables[768 + e]);       // R_Cr, G_Cr (indexed by V)
    s_gb[e] = make_uint2((uint32_t)a.tables[512 + e], (uint32_t)a.tables[1024 + e]);      // G_Cb, B_Cb (indexed by U)
  }
  __syncthreads();
  if (!valid) return;
  uint32_t px[4];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const uint32_t w = h ? m.y : m.x;
    uint32_t Y0, Y1, U, V;
    if (FMT == 2) { U = w & 0xFF; Y0 = (w >> 8) & 0xFF; V = (w >> 16) & 0xFF; Y1 = w >> 24; }      // uyvy2rgb :2410-2415
    else { Y0 = w & 0xFF; U = (w >> 8) & 0xFF; Y1 = (w >> 16) & 0xFF; V = w >> 24; }                 // yuyv2rgb :2418-2423
    const uint2 rg = s_rg[V], gb = s_gb[U];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int32_t yy = s_ty[i ? Y1 : Y0];
      const uint32_t r = (uint32_t)clamp255((yy + (int32_t)rg.x) >> 16), g = (uint32_t)clamp255((yy + (int32_t)gb.x + (int32_t)rg.y) >> 16), b = (uint32_t)clamp255((yy + (int32_t)gb.y) >> 16);
      // two byte permutes per pixel (selector 0x0C = 0x00, 0x0D = 0xFF); also keeps the compiler from fusing shift + clamp + pack into v_ashr_pk_u8_i32, whose
      // upper result half it takes for zero (seen wrong on gfx950 with ROCm 7.2: the blue byte came out OR-ed with leftovers)
      px[2 * h + i] = ORDER == 0 ? __builtin_amdgcn_perm(b, __builtin_amdgcn_perm(g, r, 0x0C0C0400u), 0x0D040100u)
                    : ORDER == 1 ? __builtin_amdgcn_perm(r, __builtin_amdgcn_perm(g, b, 0x0C0C0400u), 0x0D040100u)
                                 : __builtin_amdgcn_perm(b, __builtin_amdgcn_perm(g, r, 0x0C04000Du), 0x04020100u);
    }
  }
  typedef unsigned pu4 __attribute__((ext_vector_type(4)));
  const pu4 o = {px[0], px[1], px[2], px[3]};
  *reinterpret_cast<pu4 *>(a.dst[0] + (size_t)y * a.orow[0] + 16 * (size_t)gx) = o;
}

// YUV888 / YUVA8888 -> 3- or 4-byte RGB on aligned frames: four pixels per lane (one 12- / 16-byte load, one 12- / 16-byte store; k_yuv_to_rgb<0, ..> moves single
// bytes), the table layout of k_uyvy_to_rgb_s (one gather for Y, one 8-byte gather each for the two terms indexed by V and by U).  put_rgb()'s arithmetic.
// PLANAR: YUV444P / YUVA4444P -- one dword per plane in (IPS 4: with the alpha plane)
template <int ORDER, int IPS, int OPS, int PLANAR = 0>
__global__ __launch_bounds__(512) void k_yuv444_to_rgb_s(PalArgs a, uint32_t gmagic, const FxFrames F) {
  static_assert(ORDER != 2 || OPS == 4, "ARGB32 has four bytes");
  pal_frame(a, F);
  typedef unsigned pu4 __attribute__((ext_vector_type(4)));
  typedef unsigned pu3 __attribute__((ext_vector_type(3)));
  typedef pu3 pu3a __attribute__((aligned(4)));
  __shared__ int32_t s_ty[256];
  __shared__ uint2 s_rg[256], s_gb[256];
  const int ngr = a.width >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);                    // floor magic: the quotient or one less
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  const bool valid = y < (uint32_t)a.height;
  uint32_t in[4] = {0, 0, 0, 0};
  if (valid && PLANAR) {
    const uint32_t Y4 = *reinterpret_cast<const uint32_t *>(a.src[0] + (size_t)y * a.irow[0] + 4 * (size_t)gx), U4 = *reinterpret_cast<const uint32_t *>(a.src[1] + (size_t)y * a.irow[1] + 4 * (size_t)gx);
    const uint32_t V4 = *reinterpret_cast<const uint32_t *>(a.src[2] + (size_t)y * a.irow[2] + 4 * (size_t)gx);
    const uint32_t A4 = IPS == 4 ? *reinterpret_cast<const uint32_t *>(a.src[3] + (size_t)y * a.irow[3] + 4 * (size_t)gx) : 0xFFFFFFFFu;
    const uint32_t yu01 = __builtin_amdgcn_perm(U4, Y4, 0x05010400u), yu23 = __builtin_amdgcn_perm(U4, Y4, 0x07030602u);      // Y0 U0 Y1 U1 / Y2 U2 Y3 U3
    const uint32_t va01 = __builtin_amdgcn_perm(A4, V4, 0x05010400u), va23 = __builtin_amdgcn_perm(A4, V4, 0x07030602u);      // V0 A0 V1 A1 / V2 A2 V3 A3
    in[0] = __builtin_amdgcn_perm(va01, yu01, 0x05040100u); in[1] = __builtin_amdgcn_perm(va01, yu01, 0x07060302u);
    in[2] = __builtin_amdgcn_perm(va23, yu23, 0x05040100u); in[3] = __builtin_amdgcn_perm(va23, yu23, 0x07060302u);
  } else if (valid) {
    const uint8_t *sp = a.src[0] + (size_t)y * a.irow[0] + 4 * IPS * (size_t)gx;
    if (IPS == 4) { const pu4 q = *reinterpret_cast<const pu4 *>(sp); in[0] = q.x; in[1] = q.y; in[2] = q.z; in[3] = q.w; }
    else {
      const pu3 q = *reinterpret_cast<const pu3a *>(sp);
      in[0] = q.x | 0xFF000000u; in[1] = __builtin_amdgcn_alignbyte(q.y, q.x, 3) | 0xFF000000u; in[2] = __builtin_amdgcn_alignbyte(q.z, q.y, 2) | 0xFF000000u; in[3] = (q.z >> 8) | 0xFF000000u;
    }
  }
  for (int e = threadIdx.x; e < 256; e += blockDim.x) {
    s_ty[e] = a.tables[e];
    s_rg[e] = make_uint2((uint32_t)a.tables[256 + e], (uint32_t)a.tables[768 + e]);       // R_Cr, G_Cr (indexed by V)
    s_gb[e] = make_uint2((uint32_t)a.tables[512 + e], (uint32_t)a.tables[1024 + e]);      // G_Cb, B_Cb (indexed by U)
  }
  __syncthreads();
  if (!valid) return;
  uint32_t px[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t w = in[i];
    const int32_t yy = s_ty[w & 0xFF];
    const uint2 gb = s_gb[(w >> 8) & 0xFF], rg = s_rg[(w >> 16) & 0xFF];
    const uint32_t r = (uint32_t)clamp255((yy + (int32_t)rg.x) >> 16), g = (uint32_t)clamp255((yy + (int32_t)gb.x + (int32_t)rg.y) >> 16), b = (uint32_t)clamp255((yy + (int32_t)gb.y) >> 16);
    const uint32_t al = (IPS == 4 && a.alpha_in) ? (w >> 24) : 255u;
    // (byte permutes: see k_uyvy_to_rgb_s about v_ashr_pk_u8_i32)
    const uint32_t lo = ORDER == 1 ? __builtin_amdgcn_perm(g, b, 0x0C0C0400u) : __builtin_amdgcn_perm(g, r, 0x0C0C0400u);      // c0 | c1 << 8
    const uint32_t c2 = ORDER == 1 ? r : b;
    px[i] = ORDER == 2 ? (al | (r << 8) | (g << 16) | (b << 24)) : (__builtin_amdgcn_perm(c2, lo, 0x0C040100u) | (al << 24));
  }
  if (OPS == 4) *reinterpret_cast<pu4 *>(a.dst[0] + (size_t)y * a.orow[0] + 16 * (size_t)gx) = pu4{px[0], px[1], px[2], px[3]};
  else {
    const pu3 v = {(px[0] & 0xFFFFFFu) | (px[1] << 24), ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16), ((px[2] >> 16) & 0xFFu) | (px[3] << 8)};
    *reinterpret_cast<pu3a *>(a.dst[0] + (size_t)y * a.orow[0] + 12 * (size_t)gx) = v;
  }
}

// ---- K4b: RGB -> YUV411 (src/colourspace.c:6499-6615, rgb2_411 :2322-2343) -----------------------------------------------------
// lane = four pixels -> u2 y0 y1 v2 y2 y3; chroma is the sum of the four per-pixel (>> 16) values >> 2, clamped afterwards
__global__ __launch_bounds__(kBlock) void k_rgb_to_yuv411(PalArgs a) {
  __shared__ int32_t s_t[9 * 256];
  for (int i = threadIdx.x; i < 9 * 256; i += kBlock) s_t[i] = a.tables[i];
  __syncthreads();
  const int min_y = a.unclamped ? 0 : 16, max_y = a.unclamped ? 255 : 235, min_uv = min_y, max_uv = a.unclamped ? 255 : 240;   // :361-370
  const int ips = (a.order == 2 || a.alpha_in) ? 4 : 3, wm = a.width >> 2;
  const int ro = a.order == 0 ? 0 : a.order == 1 ? 2 : 1, go = a.order == 2 ? 2 : 1, bo = a.order == 0 ? 2 : a.order == 1 ? 0 : 3;
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= wm) return;
  for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    const uint8_t *s = a.src[0] + (size_t)y * a.irow[0] + (size_t)j * 4 * ips;
    int su = 0, sv = 0, Y[4];
    uint32_t px4[4] = {0, 0, 0, 0};
    const bool quad = ips == 4 && (reinterpret_cast<uintptr_t>(s) & 15) == 0;        // four 4-byte pixels: one 16-byte load instead of twelve byte loads
    if (quad) { const uint4 v4 = *reinterpret_cast<const uint4 *>(s); px4[0] = v4.x; px4[1] = v4.y; px4[2] = v4.z; px4[3] = v4.w; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = quad ? (int)((px4[k] >> (8 * ro)) & 0xFF) : s[k * ips + ro], g = quad ? (int)((px4[k] >> (8 * go)) & 0xFF) : s[k * ips + go],
                b = quad ? (int)((px4[k] >> (8 * bo)) & 0xFF) : s[k * ips + bo];
      const int v = (s_t[r] + s_t[256 + g] + s_t[512 + b]) >> 16;
      Y[k] = v > max_y ? max_y : v < min_y ? min_y : v;
      su += (s_t[768 + r] + s_t[1024 + g] + s_t[1280 + b]) >> 16;
      sv += (s_t[1536 + r] + s_t[1792 + g] + s_t[2048 + b]) >> 16;
    }
    su >>= 2; sv >>= 2;
    su = su > max_uv ? max_uv : su < min_uv ? min_uv : su;
    sv = sv > max_uv ? max_uv : sv < min_uv ? min_uv : sv;
    uint8_t *d = a.dst[0] + ((size_t)y * wm + j) * 6;        // 2-byte aligned when the frame is
    if ((reinterpret_cast<uintptr_t>(d) & 1) == 0) {          // three 16-bit stores instead of six byte stores
      uint16_t *d2 = reinterpret_cast<uint16_t *>(d);
      d2[0] = (uint16_t)(su | (Y[0] << 8)); d2[1] = (uint16_t)(Y[1] | (sv << 8)); d2[2] = (uint16_t)(Y[2] | (Y[3] << 8));
    } else {
      d[0] = (uint8_t)su; d[1] = (uint8_t)Y[0]; d[2] = (uint8_t)Y[1]; d[3] = (uint8_t)sv; d[4] = (uint8_t)Y[2]; d[5] = (uint8_t)Y[3];
    }
  }
}

// ---- K3b: YUV411 -> RGB (src/colourspace.c:8305-8620) ---------------------------------------------------------------------
// lane = one macropixel (u2 y0 y1 v2 y2 y3 -> 4 pixels).  Its first pair blends chroma with the block on the left, its second
// pair with the block on the right (cascaded table averages, :8344-8390); the row's first and last pair use their own chroma.
// Kept as written: the pair that opens a loop iteration never gets its alpha byte (left untouched here too), and the bgr variant
// writes the row's first pixel and last pair in R,G,B order.  The averages are three deep (h -> q -> u): evaluated arithmetically here,
// a chain of dependent table gathers measured slower (30 vs 20 us per 1080p frame).
__device__ __forceinline__ void put_colour(uint8_t *d, int bgr, const int32_t *t, int Y, int U, int V) {
  const int32_t yy = t[Y];
  const int r = clamp255((yy + t[256 + V]) >> 16), g = clamp255((yy + t[512 + U] + t[768 + V]) >> 16), b = clamp255((yy + t[1024 + U]) >> 16);
  d[0] = (uint8_t)(bgr ? b : r); d[1] = (uint8_t)g; d[2] = (uint8_t)(bgr ? r : b);
}
// the same from PAIRED tables -- {R_V, G_V}[v] and {G_U, B_U}[u] as 8-byte entries: three LDS gathers per pixel instead of five (the kernel is bound by them)
__device__ __forceinline__ int clamp255m(int v) { int d; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(v), "v"(255)); return d; }
__device__ __forceinline__ uint32_t colour24p(int bgr, const int32_t *ty, const int2 *tv, const int2 *tu, int Y, int U, int V) {
  const int32_t yy = ty[Y];
  const int2 vv = tv[V], uu = tu[U];
  const uint32_t r = (uint32_t)clamp255m((yy + vv.x) >> 16), g = (uint32_t)clamp255m((yy + uu.x + vv.y) >> 16), b = (uint32_t)clamp255m((yy + uu.y) >> 16);
  return bgr ? (b | (g << 8) | (r << 16)) : (r | (g << 8) | (b << 16));
}
__device__ __forceinline__ uint32_t colour24(int bgr, const int32_t *t, int Y, int U, int V) {       // put_colour's three bytes as bits 0..23
  const int32_t yy = t[Y];
  const uint32_t r = (uint32_t)clamp255((yy + t[256 + V]) >> 16), g = (uint32_t)clamp255((yy + t[512 + U] + t[768 + V]) >> 16), b = (uint32_t)clamp255((yy + t[1024 + U]) >> 16);
  return bgr ? (b | (g << 8) | (r << 16)) : (r | (g << 8) | (b << 16));
}
// Launch shape (round 4): macropixels are numbered linearly over the frame and walked with a grid stride by at most two 256-thread workgroups per CU -- the first
// form started one workgroup per 256 macropixels of a row (2,160 for a 1080p frame, each staging 6 KB of tables for 1.5 KB of pixels) and read its ten bytes one by
// one; now a macropixel is one dword + one halfword, its neighbours' chroma one unaligned dword each (11.2 -> see profiles/r04/ops_roofline.md).
__global__ __launch_bounds__(kBlock) void k_yuv411_to_rgb(PalArgs a, uint32_t gmagic, uint32_t ncells, const FxFrames F) {
  a.src[0] = F.in0[blockIdx.y][0]; a.dst[0] = F.out[blockIdx.y][0];      // blockIdx.y: the frame of a batched launch
  __shared__ int32_t s_t[5 * 256];
  __shared__ int2 s_v2[256], s_u2[256];
  __shared__ float s_fa[256];
  for (int i = threadIdx.x; i < 5 * 256; i += kBlock) s_t[i] = a.tables[i];
  s_v2[threadIdx.x] = make_int2(a.tables[256 + threadIdx.x], a.tables[768 + threadIdx.x]);       // {R_V, G_V}
  s_u2[threadIdx.x] = make_int2(a.tables[512 + threadIdx.x], a.tables[1024 + threadIdx.x]);      // {G_U, B_U}
  s_fa[threadIdx.x] = cavg_fa((int)threadIdx.x);             // kBlock == 256
  __syncthreads();
  const int wm = a.width;                                   // macropixels per row
  const int ps = (a.order == 2 || a.alpha_out) ? 4 : 3, coff = a.order == 2 ? 1 : 0, aoff = a.order == 2 ? 0 : 3;
  const int bgr = a.order == 1, cl = !a.unclamped;
  for (uint32_t cell = blockIdx.x * kBlock + threadIdx.x; cell < ncells; cell += gridDim.x * kBlock) {
    uint32_t y = __umulhi(cell, gmagic);                     // floor magic: the quotient or one less
    uint32_t jj = cell - y * (uint32_t)wm;
    if (jj >= (uint32_t)wm) { jj -= wm; y++; }
    const int j = (int)jj;
    const uint8_t *cb = a.src[0] + ((size_t)y * wm + j) * 6;
    uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + (size_t)j * 4 * ps;
    if (ps == 4 && ((reinterpret_cast<uintptr_t>(d) & 15) == 0) && ((reinterpret_cast<uintptr_t>(cb) & 1) == 0)) {
      // four-byte pixels: the block's four pixels leave as one 16-byte store instead of 14 .. 16 byte stores; the two alpha bytes the reference never writes
      // (pixels 4j + 2, 4j + 3 of every block but the row's last, quirk K3b) are taken from what the destination holds
      const bool first = j == 0, last = j == wm - 1;
      const uint32_t w0 = *reinterpret_cast<const uint32_t *>(cb), w1 = *reinterpret_cast<const uint16_t *>(cb + 4);      // u y0 y1 v | y2 y3
      const int cu = w0 & 0xFF, cv = w0 >> 24, y0_ = (w0 >> 8) & 0xFF, y1_ = (w0 >> 16) & 0xFF, y2_ = w1 & 0xFF, y3_ = w1 >> 8;
      uint32_t c0, c1, c2, c3;
      if (first) { c0 = colour24p(0, s_t, s_v2, s_u2, y0_, cu, cv); c1 = colour24p(bgr, s_t, s_v2, s_u2, y1_, cu, cv); }
      else {
        const uint32_t pw = *reinterpret_cast<const uint32_t *>(cb - 6);          // the block on the left: u at byte 0, v at byte 3
        const int pu = pw & 0xFF, pv = pw >> 24;
        const int qu = cavg_arith(cl, cavg_arith(cl, pu, cu), cu), qv = cavg_arith(cl, cavg_arith(cl, pv, cv), cv);
        c0 = colour24p(bgr, s_t, s_v2, s_u2, y0_, cavg_arith(cl, qu, pu), cavg_arith(cl, qv, pv));
        c1 = colour24p(bgr, s_t, s_v2, s_u2, y1_, cavg_arith(cl, qu, cu), cavg_arith(cl, qv, cv));
      }
      uint32_t a2 = 0xFFu, a3 = 0xFFu;
      if (last) { c2 = colour24p(0, s_t, s_v2, s_u2, y2_, cu, cv); c3 = colour24p(0, s_t, s_v2, s_u2, y3_, cu, cv); }
      else {
        const uint32_t nw = *reinterpret_cast<const uint32_t *>(cb + 6);          // the block on the right
        const int nu = nw & 0xFF, nv = nw >> 24;
        const int qu = cavg_arith(cl, cavg_arith(cl, cu, nu), cu), qv = cavg_arith(cl, cavg_arith(cl, cv, nv), cv);
        c2 = colour24p(bgr, s_t, s_v2, s_u2, y2_, cavg_arith(cl, qu, cu), cavg_arith(cl, qv, cv));
        c3 = colour24p(bgr, s_t, s_v2, s_u2, y3_, cavg_arith(cl, qu, nu), cavg_arith(cl, qv, nv));
        const uint2 old_ = *reinterpret_cast<const uint2 *>(d + 8);
        a2 = aoff ? old_.x >> 24 : old_.x & 0xFF; a3 = aoff ? old_.y >> 24 : old_.y & 0xFF;
      }
      uint4 o;
      if (aoff) { o.x = c0 | 0xFF000000u; o.y = c1 | 0xFF000000u; o.z = c2 | (a2 << 24); o.w = c3 | (a3 << 24); }
      else { o.x = (c0 << 8) | 0xFFu; o.y = (c1 << 8) | 0xFFu; o.z = (c2 << 8) | a2; o.w = (c3 << 8) | a3; }
      *reinterpret_cast<uint4 *>(d) = o;
      continue;
    }
    const int cu = cb[0], cv = cb[3];
    if (j == 0) {                                           // row start (:8330-8337)
      put_colour(d + coff, 0, s_t, cb[1], cu, cv);
      put_colour(d + ps + coff, bgr, s_t, cb[2], cu, cv);
    } else {                                                // second half of loop iteration j (:8373-8390)
      const int pu = cb[-6], pv = cb[-3];
      const int qu = cavg_lds(cl, s_fa, cavg_lds(cl, s_fa, pu, cu), cu), qv = cavg_lds(cl, s_fa, cavg_lds(cl, s_fa, pv, cv), cv);
      put_colour(d + coff, bgr, s_t, cb[1], cavg_lds(cl, s_fa, qu, pu), cavg_lds(cl, s_fa, qv, pv));
      put_colour(d + ps + coff, bgr, s_t, cb[2], cavg_lds(cl, s_fa, qu, cu), cavg_lds(cl, s_fa, qv, cv));
    }
    if (ps == 4) d[aoff] = d[4 + aoff] = 255;
    d += 2 * ps;
    if (j == wm - 1) {                                      // row end (:8397-8406)
      put_colour(d + coff, 0, s_t, cb[4], cu, cv);
      put_colour(d + ps + coff, 0, s_t, cb[5], cu, cv);
      if (ps == 4) d[aoff] = d[4 + aoff] = 255;
    } else {                                                // first half of loop iteration j + 1 (:8344-8366): this block is "previous"
      const int nu = cb[6], nv = cb[9];
      const int qu = cavg_lds(cl, s_fa, cavg_lds(cl, s_fa, cu, nu), cu), qv = cavg_lds(cl, s_fa, cavg_lds(cl, s_fa, cv, nv), cv);
      put_colour(d + coff, bgr, s_t, cb[4], cavg_lds(cl, s_fa, qu, cu), cavg_lds(cl, s_fa, qv, cv));
      put_colour(d + ps + coff, bgr, s_t, cb[5], cavg_lds(cl, s_fa, qu, nu), cavg_lds(cl, s_fa, qv, nv));
    }
  }
}

// ---- K5: clamped <-> unclamped, in place (src/colourspace.c:10929-11090) ---------------------------------------------------
// role of a byte = position in the plane buffer modulo `period`: pattern nibbles 0 = Y table, 1 = chroma table, 2 = leave.
// lane = 16 aligned bytes (head / tail of an unaligned buffer byte-wise by the first / last lanes)
struct ClampPlanes { uint8_t *buf[3]; size_t nbytes[3]; uint32_t pattern[3]; };
__global__ __launch_bounds__(kBlock) void k_clamp_switch(ClampPlanes cp, int period, Lut8 ylut, Lut8 clut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_y[256], s_c[256];
  stage_lut(s_y, ylut);
  stage_lut(s_c, clut);
  __syncthreads();
  uint8_t *buf = cp.buf[blockIdx.y];                 // blockIdx.y = plane
  const size_t nbytes = cp.nbytes[blockIdx.y];
  const uint32_t pattern = cp.pattern[blockIdx.y];
  if (!buf) return;
  const size_t head = (16 - (reinterpret_cast<uintptr_t>(buf) & 15)) & 15;                     // bytes before the first aligned chunk
  const size_t nchunks = nbytes > head ? (nbytes - head) / 16 : 0;
  for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * kBlock) {
    const size_t off = head + c * 16;
    uint4 v = *reinterpret_cast<uint4 *>(buf + off);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int ph = (int)(off % (size_t)period);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t o = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t byte = (w[k] >> (8 * j)) & 0xFF;
        const int role = (pattern >> (4 * ph)) & 0xF;
        o |= (role == 2 ? byte : role == 0 ? (uint32_t)s_y[byte] : (uint32_t)s_c[byte]) << (8 * j);
        ph = ph + 1 == period ? 0 : ph + 1;
      }
      w[k] = o;
    }
    *reinterpret_cast<uint4 *>(buf + off) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (blockIdx.x == 0) {                                                                         // ragged ends
    const size_t tail0 = head + nchunks * 16;
    for (size_t i = threadIdx.x; i < head + (nbytes - tail0); i += kBlock) {
      const size_t off = i < head ? i : tail0 + (i - head);
      if (off >= nbytes) continue;
      const int role = (pattern >> (4 * (int)(off % (size_t)period))) & 0xF;
      if (role == 2) continue;
      const uint8_t v = buf[off];
      buf[off] = role == 0 ? s_y[v] : s_c[v];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// K5b: YUV -> YUV repacks (src/colourspace.c:7104-7198, :7500-7753, :7800-7971, :9198-9257, :10517-10639).  One thread per
// 4:2:2 macropixel (two horizontally adjacent pixels) of one row, or per 2 x 2 block when the destination is 4:2:0: every
// output byte is a gather of at most four source bytes, so the reference's serial "write, then average with what is there"
// walks reduce to closed forms.  These are byte-granular layouts of 1-3 bytes per sample; the launches are bounded by the
// launch floor at the sizes LiVES uses them (one 1080p frame), not by HBM.
// ---------------------------------------------------------------------------------------------------------------------
enum RepackKind {
  RK_COMBINE, RK_SPLIT, RK_COPY444, RK_SWAB, RK_420_TO_PK, RK_420_TO_422P, RK_444_TO_420, RK_444_TO_PK, RK_PK_TO_444, RK_PK_TO_888,
  RK_PK_TO_420, RK_888_TO_420, RK_888_TO_422, RK_PK_TO_422P
};
// ---- K5c: YUV411 <-> the other YUV palettes (src/colourspace.c:7755-7798, :7973-8033, :8272-8303, :8622-9196) ---------------------------------
// Every one of the reference functions walks its 4:1:1 side -- and, where it takes no rowstride, the other side too -- as ONE compact stream from
// the start of the plane; so do these (what that means for a layer with padded rows is the reference's behaviour, byte for byte).  Closed forms of
// the serial walks; quirks kept: the duplicated lumas of the planar 4:4:4 / packed 4:2:2 targets, all chroma rows of the 4:2:0 target landing in
// chroma row 0 whose first sample folds a whole row, planar 4:4:4 -> 4:1:1 overwriting its first macropixel, packed 4:4:4 -> 4:1:1 stopping after
// width * height BYTES.  Macropixel bytes: u2 y0 y1 v2 y2 y3.
enum { K411_TO_888 = 1, K411_TO_444, K411_TO_PK, K411_TO_422P, K411_TO_420, K411_FROM_PK, K411_FROM_888, K411_FROM_420, K411_FROM_444 };
struct R411Args {
  const uint8_t *src[4];
  uint8_t *dst[4];
  int irow;                  // source rowstride where the reference function takes one (planar / packed 4:4:4 sources)
  int width, height, kind;   // width in pixels
  int alpha;                 // 4-byte packed pixels / alpha plane on the non-4:1:1 side
  int yuyv;                  // byte order of the packed 4:2:2 side
  int is422;                 // K411_FROM_420: the source is 4:2:2 planar
  int clamped;
  int rows;                  // K411_FROM_888: rows that begin before byte width * height
};
// chroma of the two output pixels of block j that lean on the PREVIOUS block (pixels 4j, 4j+1) / on the NEXT block (4j+2, 4j+3): the
// three-deep cascade of :8650-8716 (h = the blocks' mean, q = h re-averaged with the nearer block, then with either block)
__device__ __forceinline__ void c411_fine_pair(int cl, int a, int b, int near_b, int &o0, int &o1) {   // a = block j-1 (or j), b = block j (or j+1)
  const int q = cavg(cl, cavg(cl, a, b), near_b ? b : a);
  o0 = cavg(cl, q, a); o1 = cavg(cl, q, b);
}
__global__ __launch_bounds__(kBlock) void k_yuv411_repack(R411Args a) {
  cavg_init();
  const int wm = a.width >> 2, cl = a.clamped;
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (a.kind == K411_FROM_444) {                             // one macropixel: the last one of the frame, written over the first
    if (j || blockIdx.y) return;
    const int r = a.height - 1, jj = wm - 1;
    const uint8_t *sy = a.src[0] + (size_t)r * a.width + 4 * jj, *su = a.src[1] + (size_t)r * a.irow + 4 * jj, *sv = a.src[2] + (size_t)r * a.irow + 4 * jj;
    uint8_t *m = a.dst[0];
    m[0] = (uint8_t)cavg(cl, cavg(cl, su[0], su[1]), cavg(cl, su[2], su[3])); m[1] = sy[0]; m[2] = sy[1];
    m[3] = (uint8_t)cavg(cl, cavg(cl, sv[0], sv[1]), cavg(cl, sv[2], sv[3])); m[4] = sy[2]; m[5] = sy[3];
    return;
  }
  if (j >= wm) return;
  for (int r = blockIdx.y; r < a.height; r += gridDim.y) {
    if (a.kind >= K411_FROM_PK) {                            // ---- something -> 4:1:1: lane = destination macropixel (r, j)
      uint8_t *m = a.dst[0] + ((size_t)r * wm + j) * 6;
      if (a.kind == K411_FROM_PK) {
        const uint8_t *p = a.src[0] + ((size_t)r * wm + j) * 8, *q = p + 4;
        const int uo = a.yuyv ? 1 : 0, vo = a.yuyv ? 3 : 2, ya = a.yuyv ? 0 : 1, yb = a.yuyv ? 2 : 3;
        m[0] = (uint8_t)cavg(cl, p[uo], q[uo]); m[1] = p[ya]; m[2] = p[yb];
        m[3] = (uint8_t)cavg(cl, p[vo], q[vo]); m[4] = q[ya]; m[5] = q[yb];
      } else if (a.kind == K411_FROM_888) {
        if (r >= a.rows) return;
        const int ips = a.alpha ? 4 : 3;
        const uint8_t *p = a.src[0] + (size_t)r * a.irow + (size_t)4 * j * ips;
        m[0] = (uint8_t)((p[1] + p[ips + 1] + p[2 * ips + 1] + p[3 * ips + 1]) >> 2); m[1] = p[0]; m[2] = p[ips];
        m[3] = (uint8_t)((p[2] + p[ips + 2] + p[2 * ips + 2] + p[3 * ips + 2]) >> 2); m[4] = p[2 * ips]; m[5] = p[3 * ips];
      } else {                                               // K411_FROM_420 (compact planes)
        const int hw = a.width >> 1;
        const uint8_t *sy = a.src[0] + (size_t)r * a.width + 4 * j;
        const size_t c0 = (size_t)(a.is422 ? r : r >> 1) * hw + 2 * j;
        int u = cavg(cl, a.src[1][c0], a.src[1][c0 + 1]), v = cavg(cl, a.src[2][c0], a.src[2][c0 + 1]);
        if (!a.is422 && (r & 1) && r + 1 < a.height) {       // odd rows below the last take the mean with the row that follows (:9179-9182)
          const size_t c1 = (size_t)((r + 1) >> 1) * hw + 2 * j;
          u = cavg(cl, u, cavg(cl, a.src[1][c1], a.src[1][c1 + 1])); v = cavg(cl, v, cavg(cl, a.src[2][c1], a.src[2][c1 + 1]));
        }
        m[0] = (uint8_t)u; m[1] = sy[0]; m[2] = sy[1]; m[3] = (uint8_t)v; m[4] = sy[2]; m[5] = sy[3];
      }
      continue;
    }
    // ---- 4:1:1 -> something: lane = source block j of row r, it owns output pixels 4j .. 4j+3 (macropixels 2j, 2j+1)
    const uint8_t *cb = a.src[0] + ((size_t)r * wm + j) * 6;
    const int cu = cb[0], cv = cb[3];
    const int y0 = cb[1], y1 = cb[2], y2 = cb[4], y3 = cb[5];
    const bool first = j == 0, last = j == wm - 1;
    const int pu = first ? cu : cb[-6], pv = first ? cv : cb[-3], nu = last ? cu : cb[6], nv = last ? cv : cb[9];
    if (a.kind == K411_TO_888 || a.kind == K411_TO_444) {
      int u[4], v[4];
      if (first) { u[0] = u[1] = cu; v[0] = v[1] = cv; }
      else { c411_fine_pair(cl, pu, cu, 1, u[0], u[1]); c411_fine_pair(cl, pv, cv, 1, v[0], v[1]); }
      if (last) { u[2] = u[3] = cu; v[2] = v[3] = cv; }
      else { c411_fine_pair(cl, cu, nu, 0, u[2], u[3]); c411_fine_pair(cl, cv, nv, 0, v[2], v[3]); }
      if (a.kind == K411_TO_888) {
        const int ps = a.alpha ? 4 : 3;
        uint8_t *d = a.dst[0] + ((size_t)r * a.width + 4 * j) * ps;
        const int yy[4] = {y0, y1, y2, y3};
#pragma unroll
        for (int k = 0; k < 4; k++) { d[k * ps] = (uint8_t)yy[k]; d[k * ps + 1] = (uint8_t)u[k]; d[k * ps + 2] = (uint8_t)v[k]; if (ps == 4) d[k * ps + 3] = 255; }
      } else {
        const size_t o = (size_t)r * a.width + 4 * j;
        const int yy[4] = {y0, y0, y2, last ? y3 : y2};        // :8751-8756, :8789-8818: the second luma of a pair repeats the first, except in the row's last pair
        const uint32_t yw = (uint32_t)yy[0] | ((uint32_t)yy[1] << 8) | ((uint32_t)yy[2] << 16) | ((uint32_t)yy[3] << 24);
        const uint32_t uw = (uint32_t)u[0] | ((uint32_t)u[1] << 8) | ((uint32_t)u[2] << 16) | ((uint32_t)u[3] << 24);
        const uint32_t vw = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        if (first && !last) {}                                  // (nothing special: block 0's first pair is y0 y0 as well, :8751-8756)
        uint8_t *dy = a.dst[0] + o, *du = a.dst[1] + o, *dv = a.dst[2] + o;
        for (int k = 0; k < 4; k++) { dy[k] = (uint8_t)(yw >> (8 * k)); du[k] = (uint8_t)(uw >> (8 * k)); dv[k] = (uint8_t)(vw >> (8 * k)); }
        if (a.alpha) { uint8_t *da = a.dst[3] + o; da[0] = da[1] = da[2] = da[3] = 255; }
      }
      continue;
    }
    // the two-deep chroma of the 4:2:2 / 4:2:0 targets: macropixel 2j leans on the previous block, 2j+1 on the next
    const int u0 = first ? cu : cavg(cl, cavg(cl, pu, cu), cu), v0 = first ? cv : cavg(cl, cavg(cl, pv, cv), cv);
    const int u1 = last ? cu : cavg(cl, cavg(cl, cu, nu), cu), v1 = last ? cv : cavg(cl, cavg(cl, cv, nv), cv);
    if (a.kind == K411_TO_PK) {
      uint8_t *d = a.dst[0] + ((size_t)r * wm + j) * 8;
      const int a0 = y0, b0 = first ? y1 : y0, a1 = y2, b1 = last ? y3 : y2;      // inner macropixels carry one luma twice (:8875-8878, :8893-8896)
      if (!a.yuyv) { d[0] = (uint8_t)u0; d[1] = (uint8_t)a0; d[2] = (uint8_t)v0; d[3] = (uint8_t)b0; d[4] = (uint8_t)u1; d[5] = (uint8_t)a1; d[6] = (uint8_t)v1; d[7] = (uint8_t)b1; }
      else { d[0] = (uint8_t)a0; d[1] = (uint8_t)u0; d[2] = (uint8_t)b0; d[3] = (uint8_t)v0; d[4] = (uint8_t)a1; d[5] = (uint8_t)u1; d[6] = (uint8_t)b1; d[7] = (uint8_t)v1; }
      continue;
    }
    uint8_t *dy = a.dst[0] + (size_t)r * a.width + 4 * j;
    dy[0] = (uint8_t)y0; dy[1] = (uint8_t)y1; dy[2] = (uint8_t)y2; dy[3] = (uint8_t)y3;
    if (a.kind == K411_TO_422P) {
      uint8_t *du = a.dst[1] + (size_t)r * 2 * wm + 2 * j, *dv = a.dst[2] + (size_t)r * 2 * wm + 2 * j;
      du[0] = (uint8_t)u0; du[1] = (uint8_t)u1; dv[0] = (uint8_t)v0; dv[1] = (uint8_t)v1;
    } else if (r == ((a.height - 1) & ~1)) {
      // K411_TO_420 (:9062-9141): every even row is written to chroma row 0 and the pointers step back, so only the last even row survives
      // there (its first sample is then folded with the whole last odd row by k_yuv411_420_fold)
      a.dst[1][2 * j] = (uint8_t)u0; a.dst[1][2 * j + 1] = (uint8_t)u1; a.dst[2][2 * j] = (uint8_t)v0; a.dst[2][2 * j + 1] = (uint8_t)v1;
    }
  }
}
// YUV411 -> 4:2:0, the odd row after the last even one: each of its 2 wm chroma samples is averaged INTO THE FIRST sample of chroma row 0, in order
// (the destination pointer is not advanced on odd rows, :9071-9073, :9096-9098, :9113-9115, :9134-9136): acc = cavg(acc, c[m]) for m = 0 .. 2 wm - 1, a serial
// fold.  One lane per plane walking it took 251 us at 1080p (two dependent global loads and two table gathers per step).  The fold is a composition of byte ->
// byte functions, and composition is associative: the 2 wm steps are cut into chunks, 256 lanes per chunk each fold ONE possible start value through the
// chunk (the average in its table-free form, cavg_arith) -- the chunk's function as a 256-byte table -- and a second launch walks the start value through the tables.
__global__ __launch_bounds__(512) void k_yuv411_420_fold_chunks(const uint8_t *src, int wm, int row, int cl, int per, uint8_t *tables) {
  // blockIdx.x: chunk; threads 0..255: U, 256..511: V; tables[(plane * gridDim.x + chunk) * 256 + start value]
  extern __shared__ uint8_t s_c[];                       // [2][per]: the chunk's samples c[m], both planes
  const int plane = threadIdx.x >> 8, x = threadIdx.x & 255, m0 = blockIdx.x * per, n = min(per, 2 * wm - m0);
  const uint8_t *rp = src + (size_t)row * wm * 6;
  for (int e = threadIdx.x; e < 2 * per; e += blockDim.x) {
    const int pl = e >= per, k_ = e - pl * per, m = m0 + k_, off = pl ? 3 : 0;
    int c = 0;
    if (k_ < n) {
      if (m == 0) c = rp[off];
      else if (m == 2 * wm - 1) c = rp[(size_t)(wm - 1) * 6 + off];
      else {
        const int j = (m + 1) >> 1, k = (m + 1) & 1;
        const int p = rp[(size_t)(j - 1) * 6 + off], q = rp[(size_t)j * 6 + off];
        c = cavg_arith(cl, cavg_arith(cl, p, q), k ? q : p);
      }
    }
    s_c[e] = (uint8_t)c;
  }
  __syncthreads();
  int acc = x;
  const uint8_t *c = s_c + plane * per;
  for (int k = 0; k < n; k++) acc = cavg_arith(cl, acc, c[k]);
  tables[((size_t)plane * gridDim.x + blockIdx.x) * 256 + x] = (uint8_t)acc;
}
__global__ __launch_bounds__(256) void k_yuv411_420_fold_walk(const uint8_t *tables, int nchunks, uint8_t *du, uint8_t *dv) {
  extern __shared__ uint8_t s_tab[];                     // both planes' tables: the walk is a chain of dependent lookups, through LDS instead of through L2
  const uint32_t *t4 = reinterpret_cast<const uint32_t *>(tables);
  for (int e = threadIdx.x; e < 2 * nchunks * 64; e += blockDim.x) reinterpret_cast<uint32_t *>(s_tab)[e] = t4[e];
  __syncthreads();
  if (threadIdx.x > 1) return;
  uint8_t *d = threadIdx.x ? dv : du;
  const uint8_t *t = s_tab + (size_t)threadIdx.x * nchunks * 256;
  int acc = d[0];
  for (int ch = 0; ch < nchunks; ch++) acc = t[ch * 256 + acc];
  d[0] = (uint8_t)acc;
}

// ---- K5d: 4:2:0 / 4:2:2 planar -> YUV888 / YUVA8888 (convert_quad_chroma_packed :10715-10808, convert_double_chroma_packed :10811-10873) -------------
// lane = one pixel pair of one row.  Chroma is supersampled from the neighbouring samples of the row's chroma row (JPEG / default siting: plain means;
// otherwise the 3:1 / 1:3 forms); the pair's second pixel reads sample k + 1 even for the row's last pair (the next chroma row's first sample; for the plane's
// last row the index is clamped to the plane -- the reference reads past it).  4:2:0: even rows as above; odd row i (i <= height - 3) = mean of the even rows
// around it, i.e. of two such supersampled values; the LAST odd row's chroma and the alpha of every odd row are not written.  4:2:2: the second pixel of
// a pair never gets its alpha.  Bytes the reference does not write are not written.
struct ChromaUpArgs {
  const uint8_t *src[3];
  uint8_t *dst;
  int irow[3], orow;
  int width, height;
  int is420, alpha, jpeg, clamped;
  unsigned ulast, vlast;       // last valid byte index of the chroma planes
};
__device__ __forceinline__ void chroma_up_pair(const ChromaUpArgs &a, int cr, int k, int &u_a, int &v_a, int &u_b, int &v_b) {
  const int cl = a.clamped;
  const unsigned ub = (unsigned)cr * (unsigned)a.irow[1] + (unsigned)k, vb = (unsigned)cr * (unsigned)a.irow[2] + (unsigned)k;
  const int u0 = a.src[1][min(ub, a.ulast)], v0 = a.src[2][min(vb, a.vlast)], u1 = a.src[1][min(ub + 1, a.ulast)], v1 = a.src[2][min(vb + 1, a.vlast)];
  if (k > 0) {
    const int um = a.src[1][ub - 1], vm = a.src[2][vb - 1];
    u_a = a.jpeg ? cavg(cl, um, u0) : cavg(cl, um, cavg(cl, um, u0));        // avg_chroma_3_1f
    v_a = a.jpeg ? cavg(cl, vm, v0) : cavg(cl, cavg(cl, vm, v0), v0);        // avg_chroma_1_3f
  } else { u_a = u0; v_a = v0; }
  if (a.is420) {
    u_b = a.jpeg ? cavg(cl, u0, u1) : cavg(cl, cavg(cl, u0, u1), u1);        // 1_3 for U, 3_1 for V (:10756-10757)
    v_b = a.jpeg ? cavg(cl, v0, v1) : cavg(cl, v0, cavg(cl, v0, v1));
  } else {
    u_b = a.jpeg ? cavg(cl, u0, u1) : cavg(cl, u0, cavg(cl, u0, u1));        // the first pixel's forms again (:10862-10863)
    v_b = a.jpeg ? cavg(cl, v0, v1) : cavg(cl, cavg(cl, v0, v1), v1);
  }
}
__global__ __launch_bounds__(kBlock) void k_chroma_up_packed(ChromaUpArgs a) {
  cavg_init();
  const int k = blockIdx.x * kBlock + threadIdx.x, hw = a.width >> 1, ps = a.alpha ? 4 : 3, cl = a.clamped;
  if (k >= hw) return;
  for (int i = blockIdx.y; i < a.height; i += gridDim.y) {
    uint8_t *p = a.dst + (size_t)i * a.orow + (size_t)2 * k * ps;
    const uint8_t *sy = a.src[0] + (size_t)i * a.irow[0] + 2 * k;
    p[0] = sy[0]; p[ps] = sy[1];
    if (a.is420 && (i & 1)) {
      if (i + 2 < a.height) {                                // mean of the even rows around it (:10766-10777)
        int ua, va, ub, vb, uc, vc, ud, vd;
        chroma_up_pair(a, (i - 1) >> 1, k, ua, va, ub, vb);
        chroma_up_pair(a, (i + 1) >> 1, k, uc, vc, ud, vd);
        p[1] = (uint8_t)cavg(cl, uc, ua); p[2] = (uint8_t)cavg(cl, vc, va);
        p[ps + 1] = (uint8_t)cavg(cl, ud, ub); p[ps + 2] = (uint8_t)cavg(cl, vd, vb);
      }
      continue;
    }
    int ua, va, ub, vb;
    chroma_up_pair(a, a.is420 ? i >> 1 : i, k, ua, va, ub, vb);
    p[1] = (uint8_t)ua; p[2] = (uint8_t)va; p[ps + 1] = (uint8_t)ub; p[ps + 2] = (uint8_t)vb;
    if (a.alpha) { p[3] = 255; if (a.is420) p[ps + 3] = 255; }
  }
}

struct RepackArgs {
  const uint8_t *src[4];
  uint8_t *dst[4];
  int irow[4], orow[4];
  int width, height, kind;
  int in_alpha, out_alpha;   // RK_COMBINE / RK_COPY444 / RK_PK_TO_444 / RK_PK_TO_888
  int yuyv_in, yuyv_out;     // byte order of a packed 4:2:2 source / destination
  int clamped;               // which averaging table (init_average, :190-216)
  int copy_w;                // bytes per row of the plain plane copies: the width, or the whole rowstride where the reference memcpy()s the plane
  int cshift;                // RK_420_TO_PK: chroma row of luma row y = y >> cshift (1: 4:2:0, 0: planar 4:2:2)
};

// 4:2:0 -> UYVY / YUYV on aligned frames (the playback plugin's packed 4:2:2 from a decoder's planes): a permutation, so the whole cost is the shape -- cells of four
// macropixels (8 + 4 + 4 bytes in, 16 out) numbered linearly over the frame instead of one macropixel per lane and one 256-lane workgroup per row segment
__global__ __launch_bounds__(512) void k_420_to_packed_s(RepackArgs a, uint32_t gmagic) {
  const int ngr = a.width >> 3;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);                    // floor magic: the quotient or one less
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)a.height) return;
  const uint2 y8 = *reinterpret_cast<const uint2 *>(a.src[0] + (size_t)y * a.irow[0] + 8 * (size_t)gx);
  const uint32_t u4 = *reinterpret_cast<const uint32_t *>(a.src[1] + (size_t)(y >> a.cshift) * a.irow[1] + 4 * (size_t)gx), v4 = *reinterpret_cast<const uint32_t *>(a.src[2] + (size_t)(y >> a.cshift) * a.irow[2] + 4 * (size_t)gx);
  uint32_t mpx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t yw = k < 2 ? y8.x : y8.y;
    const uint32_t y0_ = (yw >> (16 * (k & 1))) & 0xFF, y1_ = (yw >> (16 * (k & 1) + 8)) & 0xFF, u = (u4 >> (8 * k)) & 0xFF, v = (v4 >> (8 * k)) & 0xFF;
    mpx[k] = a.yuyv_out ? (y0_ | (u << 8) | (y1_ << 16) | (v << 24)) : (u | (y0_ << 8) | (v << 16) | (y1_ << 24));
  }
  typedef unsigned rk_u4 __attribute__((ext_vector_type(4)));
  const rk_u4 o = {mpx[0], mpx[1], mpx[2], mpx[3]};
  *reinterpret_cast<rk_u4 *>(a.dst[0] + (size_t)y * (size_t)((a.orow[0] / 4) * 4) + 16 * (size_t)gx) = o;
}

// planar 4:4:4 (+ alpha plane) -> packed YUV888 / YUVA8888 on aligned frames: four pixels per lane (three or four dword loads, 12 or 16 bytes out), linear cells.
// The alpha plane is walked as a compact buffer, as the reference does (:7634-7638).
__global__ __launch_bounds__(512) void k_combine_s(RepackArgs a, uint32_t gmagic) {
  const int ngr = a.width >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)a.height) return;
  const size_t si = (size_t)y * a.irow[0] + 4 * (size_t)gx;
  const uint32_t Y4 = *reinterpret_cast<const uint32_t *>(a.src[0] + si), U4 = *reinterpret_cast<const uint32_t *>(a.src[1] + si), V4 = *reinterpret_cast<const uint32_t *>(a.src[2] + si);
  if (a.out_alpha) {
    const uint32_t A4 = a.in_alpha ? *reinterpret_cast<const uint32_t *>(a.src[3] + (size_t)y * a.width + 4 * (size_t)gx) : 0xFFFFFFFFu;
    typedef unsigned rk_u4 __attribute__((ext_vector_type(4)));
    rk_u4 o;
    o.x = (Y4 & 0xFF) | ((U4 & 0xFF) << 8) | ((V4 & 0xFF) << 16) | ((A4 & 0xFF) << 24);
    o.y = ((Y4 >> 8) & 0xFF) | (U4 & 0xFF00) | ((V4 & 0xFF00) << 8) | ((A4 & 0xFF00) << 16);
    o.z = ((Y4 >> 16) & 0xFF) | ((U4 >> 8) & 0xFF00) | (V4 & 0xFF0000) | ((A4 & 0xFF0000) << 8);
    o.w = (Y4 >> 24) | ((U4 >> 16) & 0xFF00) | ((V4 >> 8) & 0xFF0000) | (A4 & 0xFF000000u);
    *reinterpret_cast<rk_u4 *>(a.dst[0] + (size_t)y * a.orow[0] + 16 * (size_t)gx) = o;
  } else {
    uint32_t *dq = reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + 12 * (size_t)gx);
    // bytes out: Y0 U0 V0 Y1 | U1 V1 Y2 U2 | V2 Y3 U3 V3
    dq[0] = (Y4 & 0xFF) | ((U4 & 0xFF) << 8) | ((V4 & 0xFF) << 16) | ((Y4 & 0xFF00) << 16);
    dq[1] = ((U4 >> 8) & 0xFF) | (V4 & 0xFF00) | (Y4 & 0xFF0000) | ((U4 & 0xFF0000) << 8);
    dq[2] = ((V4 >> 16) & 0xFF) | ((Y4 >> 16) & 0xFF00) | ((U4 >> 8) & 0xFF0000) | (V4 & 0xFF000000u);
  }
}

// packed YUV888 -> planar 4:4:4 on aligned frames: four pixels per lane (12 bytes in, one dword into each plane), linear cells
__global__ __launch_bounds__(512) void k_split_s(RepackArgs a, uint32_t gmagic) {
  const int ngr = a.width >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)a.height) return;
  const uint32_t *sp = reinterpret_cast<const uint32_t *>(a.src[0] + (size_t)y * a.irow[0] + 12 * (size_t)gx);
  const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2];              // Y0 U0 V0 Y1 | U1 V1 Y2 U2 | V2 Y3 U3 V3
  const uint32_t Y4 = (w0 & 0xFF) | ((w0 >> 16) & 0xFF00) | (w1 & 0xFF0000) | ((w2 << 16) & 0xFF000000u);
  const uint32_t U4 = ((w0 >> 8) & 0xFF) | ((w1 << 8) & 0xFF00) | ((w1 >> 8) & 0xFF0000) | ((w2 << 8) & 0xFF000000u);
  const uint32_t V4 = ((w0 >> 16) & 0xFF) | (w1 & 0xFF00) | ((w2 << 16) & 0xFF0000) | (w2 & 0xFF000000u);
  *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + 4 * (size_t)gx) = Y4;
  *reinterpret_cast<uint32_t *>(a.dst[1] + (size_t)y * a.orow[1] + 4 * (size_t)gx) = U4;
  *reinterpret_cast<uint32_t *>(a.dst[2] + (size_t)y * a.orow[2] + 4 * (size_t)gx) = V4;
}

// UYVY <-> YUYV on aligned frames: swap the bytes of every 16-bit half, four macropixels per lane, linear cells
__global__ __launch_bounds__(512) void k_swab_s(RepackArgs a, uint32_t gmagic) {
  const int ngr = a.width >> 3;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)a.height) return;
  typedef unsigned rk_u4 __attribute__((ext_vector_type(4)));
  rk_u4 v = *reinterpret_cast<const rk_u4 *>(a.src[0] + (size_t)y * a.irow[0] + 16 * (size_t)gx);
  v.x = __builtin_amdgcn_perm(0u, v.x, 0x02030001u); v.y = __builtin_amdgcn_perm(0u, v.y, 0x02030001u);
  v.z = __builtin_amdgcn_perm(0u, v.z, 0x02030001u); v.w = __builtin_amdgcn_perm(0u, v.w, 0x02030001u);
  *reinterpret_cast<rk_u4 *>(a.dst[0] + (size_t)y * a.orow[0] + 16 * (size_t)gx) = v;
}

// UYVY / YUYV -> planar 4:2:0 / planar 4:4:4 / packed 4:4:4 on aligned frames (a capture card's frame on its way to an encoder or into the effects): a lane owns four
// macropixels of a row (of two rows for 4:2:0: its chroma is the average of the pair, K5b's RK_PK_TO_420 arithmetic), one 16-byte load per row, dword / 8- / 16-byte
// stores.  k_yuv_repack moves the same bytes one at a time (9-12 us for one 1080p frame).
template <int KIND>
__global__ __launch_bounds__(512) void k_pk_to_s(RepackArgs a, uint32_t gmagic) {
  if (KIND == RK_PK_TO_420) cavg_init();
  typedef unsigned rk_u4 __attribute__((ext_vector_type(4)));
  typedef unsigned rk_u3 __attribute__((ext_vector_type(3)));
  typedef rk_u3 rk_u3a __attribute__((aligned(4)));
  const int ngr = a.width >> 3, rows = KIND == RK_PK_TO_420 ? (a.height + 1) >> 1 : a.height;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)rows) return;
  const size_t irm = (size_t)((a.irow[0] / 4) * 4);
  const int r0 = KIND == RK_PK_TO_420 ? 2 * (int)y : (int)y;
  const rk_u4 m = *reinterpret_cast<const rk_u4 *>(a.src[0] + (size_t)r0 * irm + 16 * (size_t)gx);
  const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
  // bytes of a macropixel: UYVY u y0 v y1, YUYV y0 u y1 v
  auto lum = [&](const uint32_t *w, uint32_t &lo, uint32_t &hi) {      // eight luma bytes
    const uint32_t sel = a.yuyv_in ? 0x06040200u : 0x07050301u;
    lo = __builtin_amdgcn_perm(w[1], w[0], sel); hi = __builtin_amdgcn_perm(w[3], w[2], sel);
  };
  auto chr = [&](const uint32_t *w, uint32_t &u4, uint32_t &v4) {      // four U, four V
    const uint32_t su = a.yuyv_in ? 0x0C0C0501u : 0x0C0C0400u, sv = a.yuyv_in ? 0x0C0C0703u : 0x0C0C0602u;
    u4 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(w[3], w[2], su), __builtin_amdgcn_perm(w[1], w[0], su), 0x05040100u);
    v4 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(w[3], w[2], sv), __builtin_amdgcn_perm(w[1], w[0], sv), 0x05040100u);
  };
  uint32_t ylo, yhi, u4, v4;
  lum(mw, ylo, yhi);
  chr(mw, u4, v4);
  if (KIND == RK_PK_TO_420) {
    *reinterpret_cast<uint2 *>(a.dst[0] + (size_t)r0 * a.width + 8 * (size_t)gx) = make_uint2(ylo, yhi);
    if (r0 + 1 < a.height) {
      const rk_u4 n = *reinterpret_cast<const rk_u4 *>(a.src[0] + (size_t)(r0 + 1) * irm + 16 * (size_t)gx);
      const uint32_t nw[4] = {n.x, n.y, n.z, n.w};
      uint32_t l2, h2, nu, nv;
      lum(nw, l2, h2);
      chr(nw, nu, nv);
      *reinterpret_cast<uint2 *>(a.dst[0] + (size_t)(r0 + 1) * a.width + 8 * (size_t)gx) = make_uint2(l2, h2);
      uint32_t ou = 0, ov = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        ou |= (uint32_t)cavg(a.clamped, (u4 >> (8 * k)) & 0xFF, (nu >> (8 * k)) & 0xFF) << (8 * k);
        ov |= (uint32_t)cavg(a.clamped, (v4 >> (8 * k)) & 0xFF, (nv >> (8 * k)) & 0xFF) << (8 * k);
      }
      u4 = ou; v4 = ov;
    }
    *reinterpret_cast<uint32_t *>(a.dst[1] + (size_t)y * (a.width >> 1) + 4 * (size_t)gx) = u4;
    *reinterpret_cast<uint32_t *>(a.dst[2] + (size_t)y * (a.width >> 1) + 4 * (size_t)gx) = v4;
  } else if (KIND == RK_PK_TO_444) {
    // every chroma sample serves its two pixels
    const uint32_t ul = __builtin_amdgcn_perm(0u, u4, 0x01010000u), uh = __builtin_amdgcn_perm(0u, u4, 0x03030202u);
    const uint32_t vl = __builtin_amdgcn_perm(0u, v4, 0x01010000u), vh = __builtin_amdgcn_perm(0u, v4, 0x03030202u);
    const size_t di = (size_t)y * a.orow[0] + 8 * (size_t)gx;
    *reinterpret_cast<uint2 *>(a.dst[0] + di) = make_uint2(ylo, yhi);
    *reinterpret_cast<uint2 *>(a.dst[1] + di) = make_uint2(ul, uh);
    *reinterpret_cast<uint2 *>(a.dst[2] + di) = make_uint2(vl, vh);
  } else {
    uint32_t px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t yy = ((k < 4 ? ylo : yhi) >> (8 * (k & 3))) & 0xFF, u = (u4 >> (8 * (k >> 1))) & 0xFF, v = (v4 >> (8 * (k >> 1))) & 0xFF;
      px[k] = yy | (u << 8) | (v << 16) | 0xFF000000u;
    }
    if (a.out_alpha) {
      rk_u4 *d = reinterpret_cast<rk_u4 *>(a.dst[0] + (size_t)y * a.orow[0] + 32 * (size_t)gx);
      d[0] = rk_u4{px[0], px[1], px[2], px[3]}; d[1] = rk_u4{px[4], px[5], px[6], px[7]};
    } else {
      uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + 24 * (size_t)gx;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t *o = px + 4 * h;
        *reinterpret_cast<rk_u3a *>(d + 12 * h) = rk_u3{(o[0] & 0xFFFFFFu) | (o[1] << 24), ((o[1] >> 8) & 0xFFFFu) | (o[2] << 16), ((o[2] >> 16) & 0xFFu) | (o[3] << 8)};
      }
    }
  }
}

// YUV888 / YUVA8888 -> planar 4:2:0 / planar 4:2:2 / UYVY / YUYV on aligned frames (RK_888_TO_420 / RK_888_TO_422: the chroma of a pixel pair is the table
// average of its two samples, of the 2 x 2 block the average of the two rows' averages, :8035-8270) and planar 4:2:0 -> planar 4:2:2 (RK_420_TO_422P: odd rows take
// the average of the chroma rows around them, :7163-7226): a lane owns four pixels (eight for the planar pair) of one row / row pair.
template <int KIND>
__global__ __launch_bounds__(512) void k_888_to_s(RepackArgs a, uint32_t gmagic) {
  cavg_init();
  typedef unsigned rk_u4 __attribute__((ext_vector_type(4)));
  typedef unsigned rk_u3 __attribute__((ext_vector_type(3)));
  typedef rk_u3 rk_u3a __attribute__((aligned(4)));
  const int ngr = a.width >> 2, rows = KIND == RK_888_TO_420 ? a.height >> 1 : a.height;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)rows) return;
  const int ips = a.in_alpha ? 4 : 3, hw = a.width >> 1;
  auto load4 = [&](int row, uint32_t *px) {                 // four pixels as Y | U << 8 | V << 16
    const uint8_t *sp = a.src[0] + (size_t)row * a.irow[0] + 4 * (size_t)ips * gx;
    if (ips == 4) { const rk_u4 q = *reinterpret_cast<const rk_u4 *>(sp); px[0] = q.x; px[1] = q.y; px[2] = q.z; px[3] = q.w; }
    else {
      const rk_u3 q = *reinterpret_cast<const rk_u3a *>(sp);
      px[0] = q.x; px[1] = __builtin_amdgcn_alignbyte(q.y, q.x, 3); px[2] = __builtin_amdgcn_alignbyte(q.z, q.y, 2); px[3] = q.z >> 8;
    }
  };
  auto pairavg = [&](const uint32_t *px, int p, int sh) -> int { return cavg(a.clamped, (px[2 * p] >> sh) & 0xFF, (px[2 * p + 1] >> sh) & 0xFF); };
  uint32_t p0[4];
  const int r0 = KIND == RK_888_TO_420 ? 2 * (int)y : (int)y;
  load4(r0, p0);
  const uint32_t y4 = (p0[0] & 0xFF) | ((p0[1] & 0xFF) << 8) | ((p0[2] & 0xFF) << 16) | ((p0[3] & 0xFF) << 24);
  int u[2] = {pairavg(p0, 0, 8), pairavg(p0, 1, 8)}, v[2] = {pairavg(p0, 0, 16), pairavg(p0, 1, 16)};
  if (KIND == RK_888_TO_420) {
    uint32_t p1[4];
    load4(r0 + 1, p1);
    *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)r0 * a.width + 4 * (size_t)gx) = y4;
    *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)(r0 + 1) * a.width + 4 * (size_t)gx) = (p1[0] & 0xFF) | ((p1[1] & 0xFF) << 8) | ((p1[2] & 0xFF) << 16) | ((p1[3] & 0xFF) << 24);
#pragma unroll
    for (int p = 0; p < 2; p++) { u[p] = cavg(a.clamped, u[p], pairavg(p1, p, 8)); v[p] = cavg(a.clamped, v[p], pairavg(p1, p, 16)); }
    *reinterpret_cast<uint16_t *>(a.dst[1] + (size_t)y * hw + 2 * (size_t)gx) = (uint16_t)(u[0] | (u[1] << 8));
    *reinterpret_cast<uint16_t *>(a.dst[2] + (size_t)y * hw + 2 * (size_t)gx) = (uint16_t)(v[0] | (v[1] << 8));
  } else if (a.out_alpha) {                                  // planar 4:2:2 (compact)
    *reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.width + 4 * (size_t)gx) = y4;
    *reinterpret_cast<uint16_t *>(a.dst[1] + (size_t)y * hw + 2 * (size_t)gx) = (uint16_t)(u[0] | (u[1] << 8));
    *reinterpret_cast<uint16_t *>(a.dst[2] + (size_t)y * hw + 2 * (size_t)gx) = (uint16_t)(v[0] | (v[1] << 8));
  } else {                                                   // UYVY / YUYV (compact)
    uint2 o;
    const uint32_t ya = y4 & 0xFF, yb = (y4 >> 8) & 0xFF, yc = (y4 >> 16) & 0xFF, yd = y4 >> 24;
    if (a.yuyv_out) { o.x = ya | ((uint32_t)u[0] << 8) | (yb << 16) | ((uint32_t)v[0] << 24); o.y = yc | ((uint32_t)u[1] << 8) | (yd << 16) | ((uint32_t)v[1] << 24); }
    else { o.x = (uint32_t)u[0] | (ya << 8) | ((uint32_t)v[0] << 16) | (yb << 24); o.y = (uint32_t)u[1] | (yc << 8) | ((uint32_t)v[1] << 16) | (yd << 24); }
    *reinterpret_cast<uint2 *>(a.dst[0] + (size_t)y * a.width * 2 + 8 * (size_t)gx) = o;
  }
}
__global__ __launch_bounds__(512) void k_420_to_422p_s(RepackArgs a, uint32_t gmagic) {
  cavg_init();
  const int ngr = a.width >> 3;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t y = __umulhi(idx, gmagic);
  uint32_t gx = idx - y * (uint32_t)ngr;
  if (gx >= (uint32_t)ngr) { gx -= ngr; y++; }
  if (y >= (uint32_t)a.height) return;
  *reinterpret_cast<uint2 *>(a.dst[0] + (size_t)y * a.orow[0] + 8 * (size_t)gx) = *reinterpret_cast<const uint2 *>(a.src[0] + (size_t)y * a.irow[0] + 8 * (size_t)gx);
  const int ch2 = (a.height >> 1) * 2;
  if ((int)y >= ch2) return;
#pragma unroll
  for (int p = 1; p < 3; p++) {
    uint32_t c4 = *reinterpret_cast<const uint32_t *>(a.src[p] + (size_t)(y >> 1) * a.irow[p] + 4 * (size_t)gx);
    if ((y & 1) && (int)y + 1 < ch2) {
      const uint32_t n4 = *reinterpret_cast<const uint32_t *>(a.src[p] + (size_t)((y + 1) >> 1) * a.irow[p] + 4 * (size_t)gx);
      uint32_t o = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) o |= (uint32_t)cavg(a.clamped, (c4 >> (8 * k)) & 0xFF, (n4 >> (8 * k)) & 0xFF) << (8 * k);
      c4 = o;
    }
    *reinterpret_cast<uint32_t *>(a.dst[p] + (size_t)y * a.orow[p] + 4 * (size_t)gx) = c4;
  }
}

__global__ __launch_bounds__(kBlock) void k_yuv_repack(RepackArgs a) {
  // the chroma-average table costs a workgroup a round of LDS writes and a barrier: only the kinds that average build it (kernel-uniform)
  if (a.kind > RK_420_TO_PK) cavg_init();              // RK_COMBINE .. RK_420_TO_PK are permutations
  const int mx = blockIdx.x * kBlock + threadIdx.x;         // macropixel column
  const int mw = ((a.copy_w > a.width ? a.copy_w : a.width) + 1) >> 1;
  if (mx >= mw) return;
  const int x0 = 2 * mx;
  if (x0 >= a.width && a.kind != RK_COPY444 && a.kind != RK_444_TO_420) return;
  const bool two = x0 + 1 < a.width;                         // odd widths: the last column is a single pixel where the reference allows it
  const int rows = (a.kind == RK_444_TO_420 || a.kind == RK_PK_TO_420 || a.kind == RK_888_TO_420) ? (a.height + 1) >> 1 : a.height;
  for (int y = blockIdx.y; y < rows; y += gridDim.y) {
    switch (a.kind) {
    case RK_COMBINE: {
      const int ops = a.out_alpha ? 4 : 3;
      uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + (size_t)x0 * ops;
      if (a.out_alpha && two && ((uintptr_t)d & 7) == 0) {        // two YUVA pixels = one 8-byte store
        const size_t si = (size_t)y * a.irow[0] + x0;
        const uint32_t a0 = a.in_alpha ? a.src[3][(size_t)y * a.width + x0] : 255u, a1 = a.in_alpha ? a.src[3][(size_t)y * a.width + x0 + 1] : 255u;
        const uint32_t p0 = a.src[0][si] | ((uint32_t)a.src[1][si] << 8) | ((uint32_t)a.src[2][si] << 16) | (a0 << 24);
        const uint32_t p1 = a.src[0][si + 1] | ((uint32_t)a.src[1][si + 1] << 8) | ((uint32_t)a.src[2][si + 1] << 16) | (a1 << 24);
        *reinterpret_cast<uint2 *>(d) = make_uint2(p0, p1);
        break;
      }
      if (!a.out_alpha) {
        // four pixels at a time on the even lanes (3 dword loads, 3 dword stores instead of 12 + 12 byte accesses on two lanes) where rows and pointers allow it
        const int xq = x0 & ~3;
        const bool quad = xq + 3 < a.width && (a.irow[0] & 3) == 0 && (a.orow[0] & 3) == 0 &&
                          ((((uintptr_t)a.src[0] | (uintptr_t)a.src[1] | (uintptr_t)a.src[2] | (uintptr_t)a.dst[0]) & 3) == 0);
        if (quad) {
          if (mx & 1) break;                                   // the even lane of the pair does all four
          const size_t si = (size_t)y * a.irow[0] + xq;
          const uint32_t Y4 = *reinterpret_cast<const uint32_t *>(a.src[0] + si), U4 = *reinterpret_cast<const uint32_t *>(a.src[1] + si), V4 = *reinterpret_cast<const uint32_t *>(a.src[2] + si);
          uint32_t *dq = reinterpret_cast<uint32_t *>(a.dst[0] + (size_t)y * a.orow[0] + (size_t)xq * 3);
          // bytes out: Y0 U0 V0 Y1 | U1 V1 Y2 U2 | V2 Y3 U3 V3
          dq[0] = (Y4 & 0xFF) | ((U4 & 0xFF) << 8) | ((V4 & 0xFF) << 16) | ((Y4 & 0xFF00) << 16);
          dq[1] = ((U4 >> 8) & 0xFF) | (V4 & 0xFF00) | (Y4 & 0xFF0000) | ((U4 & 0xFF0000) << 8);
          dq[2] = ((V4 >> 16) & 0xFF) | ((Y4 >> 16) & 0xFF00) | ((U4 >> 8) & 0xFF0000) | (V4 & 0xFF000000u);
          break;
        }
      }
      for (int k = 0; k < (two ? 2 : 1); k++) {
        const size_t si = (size_t)y * a.irow[0] + x0 + k;
        d[k * ops] = a.src[0][si]; d[k * ops + 1] = a.src[1][si]; d[k * ops + 2] = a.src[2][si];
        if (a.out_alpha) d[k * ops + 3] = a.in_alpha ? a.src[3][(size_t)y * a.width + x0 + k] : 255;   // alpha plane walked as compact (:7634-7638)
      }
      break;
    }
    case RK_SPLIT: {
      const uint8_t *s = a.src[0] + (size_t)y * a.irow[0] + (size_t)x0 * 3;
      for (int k = 0; k < (two ? 2 : 1); k++)
        for (int p = 0; p < 3; p++) a.dst[p][(size_t)y * a.orow[p] + x0 + k] = s[3 * k + p];
      break;
    }
    case RK_COPY444: {
      for (int k = 0; k < 2; k++)
        if (x0 + k < a.copy_w)
          for (int p = 0; p < 3; p++) a.dst[p][(size_t)y * a.orow[0] + x0 + k] = a.src[p][(size_t)y * a.irow[0] + x0 + k];
      break;
    }
    case RK_SWAB: {
      const uint8_t *s = a.src[0] + (size_t)y * a.irow[0] + 4 * (size_t)mx;
      uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + 4 * (size_t)mx;
      const uint8_t b0 = s[0], b1 = s[1], b2 = s[2], b3 = s[3];
      d[0] = b1; d[1] = b0; d[2] = b3; d[3] = b2;
      break;
    }
    case RK_420_TO_PK: {
      const uint8_t *sy = a.src[0] + (size_t)y * a.irow[0] + x0;
      const uint8_t u = a.src[1][(size_t)(y >> a.cshift) * a.irow[1] + mx], v = a.src[2][(size_t)(y >> a.cshift) * a.irow[2] + mx];
      uint8_t *d = a.dst[0] + (size_t)y * (size_t)((a.orow[0] / 4) * 4) + 4 * (size_t)mx;
      const uint32_t y0_ = sy[0], y1_ = sy[1];
      const uint32_t mp = a.yuyv_out ? (y0_ | ((uint32_t)u << 8) | (y1_ << 16) | ((uint32_t)v << 24)) : ((uint32_t)u | (y0_ << 8) | ((uint32_t)v << 16) | (y1_ << 24));
      if (((uintptr_t)d & 3) == 0) *reinterpret_cast<uint32_t *>(d) = mp;                  // one macropixel = one dword store instead of four byte stores
      else { d[0] = (uint8_t)mp; d[1] = (uint8_t)(mp >> 8); d[2] = (uint8_t)(mp >> 16); d[3] = (uint8_t)(mp >> 24); }
      break;
    }
    case RK_420_TO_422P: {
      a.dst[0][(size_t)y * a.orow[0] + x0] = a.src[0][(size_t)y * a.irow[0] + x0];
      if (two) a.dst[0][(size_t)y * a.orow[0] + x0 + 1] = a.src[0][(size_t)y * a.irow[0] + x0 + 1];
      const int ch2 = (a.height >> 1) * 2;
      if (y < ch2 && mx < (a.width >> 1)) {
        for (int p = 1; p < 3; p++) {
          int c = a.src[p][(size_t)(y >> 1) * a.irow[p] + mx];
          if ((y & 1) && y + 1 < ch2) c = cavg(a.clamped, c, a.src[p][(size_t)((y + 1) >> 1) * a.irow[p] + mx]);
          a.dst[p][(size_t)y * a.orow[p] + mx] = (uint8_t)c;
        }
      }
      break;
    }
    case RK_444_TO_420: {
      for (int r = 2 * y; r < 2 * y + 2 && r < a.height; r++)
        for (int k = 0; k < 2; k++)
          if (x0 + k < a.copy_w) a.dst[0][(size_t)r * a.orow[0] + x0 + k] = a.src[0][(size_t)r * a.irow[0] + x0 + k];
      if (mx < (a.width >> 1)) {
        for (int p = 1; p < 3; p++) {
          const uint8_t *s0 = a.src[p] + (size_t)(2 * y) * a.irow[p] + x0;
          int c = cavg(a.clamped, s0[0], s0[1]);
          if (2 * y + 1 < a.height) c = cavg(a.clamped, c, cavg(a.clamped, s0[a.irow[p]], s0[a.irow[p] + 1]));
          a.dst[p][(size_t)y * a.orow[p] + mx] = (uint8_t)c;
        }
      }
      break;
    }
    case RK_444_TO_PK: {                                       // compact source and destination (checked by the caller)
      const size_t si = (size_t)y * a.width + x0;
      uint8_t *d = a.dst[0] + (size_t)y * a.width * 2 + 4 * (size_t)mx;
      const uint8_t u = (uint8_t)cavg(a.clamped, a.src[1][si], a.src[1][si + 1]), v = (uint8_t)cavg(a.clamped, a.src[2][si], a.src[2][si + 1]);
      if (a.yuyv_out) { d[0] = a.src[0][si]; d[1] = u; d[2] = a.src[0][si + 1]; d[3] = v; }
      else { d[0] = u; d[1] = a.src[0][si]; d[2] = v; d[3] = a.src[0][si + 1]; }
      break;
    }
    case RK_888_TO_420: case RK_888_TO_422: {                   // compact destination (checked by the caller); in_alpha = 4-byte source pixels
      const int ips = a.in_alpha ? 4 : 3, hw = a.width >> 1;
      if (a.kind == RK_888_TO_420) {
        const uint8_t *s0 = a.src[0] + (size_t)(2 * y) * a.irow[0] + (size_t)x0 * ips, *s1 = s0 + a.irow[0];
        a.dst[0][(size_t)(2 * y) * a.width + x0] = s0[0]; a.dst[0][(size_t)(2 * y) * a.width + x0 + 1] = s0[ips];
        a.dst[0][(size_t)(2 * y + 1) * a.width + x0] = s1[0]; a.dst[0][(size_t)(2 * y + 1) * a.width + x0 + 1] = s1[ips];
        a.dst[1][(size_t)y * hw + mx] = (uint8_t)cavg(a.clamped, cavg(a.clamped, s0[1], s0[1 + ips]), cavg(a.clamped, s1[1], s1[1 + ips]));
        a.dst[2][(size_t)y * hw + mx] = (uint8_t)cavg(a.clamped, cavg(a.clamped, s0[2], s0[2 + ips]), cavg(a.clamped, s1[2], s1[2 + ips]));
      } else {
        const uint8_t *s0 = a.src[0] + (size_t)y * a.irow[0] + (size_t)x0 * ips;
        const uint8_t u = (uint8_t)cavg(a.clamped, s0[1], s0[1 + ips]), v = (uint8_t)cavg(a.clamped, s0[2], s0[2 + ips]);
        if (a.out_alpha) {                                        // here: planar 4:2:2 destination
          a.dst[0][(size_t)y * a.width + x0] = s0[0]; a.dst[0][(size_t)y * a.width + x0 + 1] = s0[ips];
          a.dst[1][(size_t)y * hw + mx] = u; a.dst[2][(size_t)y * hw + mx] = v;
        } else {
          uint8_t *d = a.dst[0] + (size_t)y * a.width * 2 + 4 * (size_t)mx;
          if (a.yuyv_out) { d[0] = s0[0]; d[1] = u; d[2] = s0[ips]; d[3] = v; }
          else { d[0] = u; d[1] = s0[0]; d[2] = v; d[3] = s0[ips]; }
        }
      }
      break;
    }
    case RK_PK_TO_422P: {                                         // the reference never advances its source pointer (:8102-8107): macropixel 0 everywhere
      const int yo = a.yuyv_in ? 0 : 1, uo = a.yuyv_in ? 1 : 0, vo = a.yuyv_in ? 3 : 2;
      const uint8_t *m = a.src[0];
      const size_t k = (size_t)y * (a.width >> 1) + mx;
      a.dst[0][2 * k] = m[yo]; a.dst[0][2 * k + 1] = m[yo + 2]; a.dst[1][k] = m[uo]; a.dst[2][k] = m[vo];
      break;
    }
    case RK_PK_TO_444: case RK_PK_TO_888: case RK_PK_TO_420: {
      const int yo = a.yuyv_in ? 0 : 1, uo = a.yuyv_in ? 1 : 0, vo = a.yuyv_in ? 3 : 2;
      const size_t irm = (size_t)((a.irow[0] / 4) * 4);
      if (a.kind == RK_PK_TO_444) {
        const uint8_t *m = a.src[0] + (size_t)y * irm + 4 * (size_t)mx;
        const size_t di = (size_t)y * a.orow[0] + x0;
        a.dst[0][di] = m[yo]; a.dst[0][di + 1] = m[yo + 2];
        a.dst[1][di] = a.dst[1][di + 1] = m[uo];
        a.dst[2][di] = a.dst[2][di + 1] = m[vo];
      } else if (a.kind == RK_PK_TO_888) {
        const uint8_t *m = a.src[0] + (size_t)y * irm + 4 * (size_t)mx;
        const int ops = a.out_alpha ? 4 : 3;
        uint8_t *d = a.dst[0] + (size_t)y * a.orow[0] + (size_t)x0 * ops;
        d[0] = m[yo]; d[1] = m[uo]; d[2] = m[vo];
        d[ops] = m[yo + 2]; d[ops + 1] = m[uo]; d[ops + 2] = m[vo];
        if (a.out_alpha) d[3] = d[ops + 3] = 255;
      } else {
        const uint8_t *m0 = a.src[0] + (size_t)(2 * y) * irm + 4 * (size_t)mx;
        int u = m0[uo], v = m0[vo];
        a.dst[0][(size_t)(2 * y) * a.width + x0] = m0[yo]; a.dst[0][(size_t)(2 * y) * a.width + x0 + 1] = m0[yo + 2];
        if (2 * y + 1 < a.height) {
          const uint8_t *m1 = m0 + irm;
          a.dst[0][(size_t)(2 * y + 1) * a.width + x0] = m1[yo]; a.dst[0][(size_t)(2 * y + 1) * a.width + x0 + 1] = m1[yo + 2];
          u = cavg(a.clamped, u, m1[uo]); v = cavg(a.clamped, v, m1[vo]);
        }
        a.dst[1][(size_t)y * (a.width >> 1) + mx] = (uint8_t)u; a.dst[2][(size_t)y * (a.width >> 1) + mx] = (uint8_t)v;
      }
      break;
    }
    }
  }
}
}  // namespace lgpu

using namespace lgpu;

// per-device table of the clamped chroma average (see d_cavgc above); called by every entry point whose kernels average chroma
static int ensure_cavgc() {
  static std::mutex mu;
  static uint8_t *tab[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("hipGetDevice failed"); return LGPU_E_HIP; }
  std::lock_guard<std::mutex> lk(mu);
  if (tab[dev]) return LGPU_OK;
  uint8_t *d = nullptr;
  if (hipMalloc((void **)&d, 65536) != hipSuccess) { set_error("hipMalloc of the chroma average table failed"); return LGPU_E_NOMEM; }
  hipLaunchKernelGGL(k_build_cavgc, dim3(256), dim3(256), 0, (hipStream_t)0, d);
  if (hipMemcpyToSymbol(HIP_SYMBOL(d_cavgc), &d, sizeof d) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(d);
    set_error("building the chroma average table failed");
    return LGPU_E_HIP;
  }
  // the kernels average chroma WITHOUT the table (cavg_arith: fmaf with split constants); k_build_cavgc has just compared that form with the table form for all
  // 65,536 pairs.  A compiler or target change that moved the arithmetic would otherwise give wrong 4:1:1 bytes with no error at run time: fail here instead.
  unsigned int differ = 0;
  if (hipMemcpyFromSymbol(&differ, HIP_SYMBOL(d_cavg_forms_differ), sizeof differ) != hipSuccess || differ) {
    (void)hipFree(d);
    set_error("the table-free chroma average differs from the table form for %u of 65536 pairs: this build's arithmetic is not the reference's", differ);
    return LGPU_E_HIP;
  }
  tab[dev] = d;
  return LGPU_OK;
}

// diagnostics: the device's clamped chroma-average table (init_average, src/colourspace.c:190-216) as the kernels compute it
extern "C" int lgpu_chroma_average_table(uint8_t out[65536]) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(out, "null result buffer");
  const uint8_t *d = nullptr;
  LGPU_HIP(hipMemcpyFromSymbol(&d, HIP_SYMBOL(d_cavgc), sizeof d));
  LGPU_HIP(hipMemcpy(out, d, 65536, hipMemcpyDeviceToHost));
  unsigned int differ = 0;
  LGPU_HIP(hipMemcpyFromSymbol(&differ, HIP_SYMBOL(d_cavg_forms_differ), sizeof differ));
  if (differ) { set_error("the table-free chroma average differs from the table form for %u pairs", differ); return LGPU_E_HIP; }
  return LGPU_OK;
}

// nfr frames of one geometry (srcs[f], dsts[f][plane]) as one launch: the frame is the grid's z index; the cell forms need EVERY frame aligned
static int rgb_to_yuv_impl_n(const uint8_t *const *srcs, int irow, int width, int height, int in_order, int in_alpha,
                             uint8_t *const (*dsts)[4], const int orow[4], int out_fmt, int out_alpha, int which_tables, const uint16_t *lut16_d, int nfr, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(srcs && dsts && orow && nfr >= 1 && nfr <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0, "null frame table, 1..16 frames, or empty geometry");
  const uint8_t *const src_d = srcs[0];
  uint8_t *const *const dst_d = dsts[0];
  LGPU_REQUIRE(src_d, "null frame");
  LGPU_REQUIRE(in_order >= 0 && in_order <= 2, "in_order must be 0 (RGB), 1 (BGR) or 2 (ARGB)");
  LGPU_REQUIRE(out_fmt >= 0 && out_fmt <= 5, "out_fmt must be 0 (packed 4:4:4), 1 (planar 4:4:4), 2 (UYVY), 3 (YUYV), 4 (4:2:0 planar), 5 (4:2:2 planar)");
  LGPU_REQUIRE(out_fmt < 2 || !(width & 1), "subsampled targets need an even width");
  LGPU_REQUIRE(out_fmt != 4 || !(height & 1), "4:2:0 needs an even height");
  LGPU_REQUIRE(out_fmt >= 4 || !(which_tables & 2), "only the 4:2:0 / 4:2:2 conversions take a BT.709 subspace (the reference's other entry points are YCbCr only)");
  if (out_fmt >= 4 && in_order == 2) { set_error("ARGB32 -> planar 4:2:0 / 4:2:2 is declined: the reference reads the wrong bytes there (src/colourspace.c:6353)"); return LGPU_E_UNSUPPORTED; }
  PalArgs a;
  __builtin_memset(&a, 0, sizeof a);
  const int ips = (in_order == 2 || in_alpha) ? 4 : 3;
  LGPU_REQUIRE(irow >= width * ips, "rowstride smaller than a row");
  const int nplanes = out_fmt == 0 ? 1 : out_fmt == 1 ? (out_alpha ? 4 : 3) : out_fmt <= 3 ? 1 : 3;
  for (int i = 0; i < nplanes; i++) { LGPU_REQUIRE(dst_d[i] && orow[i] > 0, "null destination plane"); a.dst[i] = dst_d[i]; a.orow[i] = orow[i]; }
  if (out_fmt == 2 || out_fmt == 3) LGPU_REQUIRE(!(orow[0] & 3) && !(reinterpret_cast<uintptr_t>(dst_d[0]) & 3), "UYVY / YUYV rows must be 4-byte aligned");
  a.src[0] = src_d; a.irow[0] = irow; a.width = width; a.height = height;
  a.order = in_order; a.alpha_in = in_alpha; a.fmt = out_fmt; a.alpha_out = (out_fmt <= 1) ? out_alpha : 0;
  a.unclamped = which_tables & 1;
  a.tables = device_tables()->rgb2yuv[which_tables & 3];
  a.lut16 = lut16_d;
  FxFrames F = {};
  uintptr_t sbits = (uintptr_t)irow, d0bits = (uintptr_t)orow[0], d12bits = 0;
  for (int f = 0; f < nfr; f++) {
    LGPU_REQUIRE(srcs[f], "null frame");
    F.in0[f][0] = srcs[f]; sbits |= (uintptr_t)srcs[f];
    for (int i = 0; i < nplanes; i++) { LGPU_REQUIRE(dsts[f][i], "null destination plane"); F.out[f][i] = dsts[f][i]; }
    d0bits |= (uintptr_t)dsts[f][0];
    if (nplanes >= 3) d12bits |= (uintptr_t)dsts[f][1] | (uintptr_t)dsts[f][2] | (uintptr_t)orow[1] | (uintptr_t)orow[2];
    if (out_fmt == 2 || out_fmt == 3) LGPU_REQUIRE(!((uintptr_t)dsts[f][0] & 3), "UYVY / YUYV rows must be 4-byte aligned");
  }
  const int npairs = width >> 1;
  if (npairs == 0) return LGPU_OK;
  // aligned 4-byte pixels -> 4:2:0: the cell form
  const bool no_s420 = tune_on(TUNE_RGB2YUV_NO_S);
  if (out_fmt == 4 && ips == 4 && in_order <= 1 && !no_s420 && (width & 3) == 0 && height >= 2 &&
      (sbits & 15) == 0 && (d0bits & 3) == 0 && (d12bits & 1) == 0) {
    const int ngr = width >> 2, nunits = (height >> 1) + 1;
    const unsigned long long cells = (unsigned long long)ngr * nunits;
    if (cells < (1ull << 31)) {
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)((cells + 511) / 512), 1, (unsigned)nfr);
      if (in_order == 0) hipLaunchKernelGGL(k_rgb_to_yuv420_s<0>, gs, dim3(512), 0, (hipStream_t)stream, a, magic, F);
      else hipLaunchKernelGGL(k_rgb_to_yuv420_s<1>, gs, dim3(512), 0, (hipStream_t)stream, a, magic, F);
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  // aligned 4-byte pixels -> packed / planar 4:2:2 without a gamma LUT: the cell form
  if ((out_fmt == 2 || out_fmt == 3 || out_fmt == 5) && ips == 4 && in_order <= 1 && !lut16_d && !no_s420 && (width & 3) == 0 &&
      (sbits & 15) == 0 &&
      (out_fmt == 5 ? ((d0bits & 3) == 0 && (d12bits & 1) == 0) : ((d0bits & 7) == 0))) {
    const int ngr = width >> 2;
    const unsigned long long cells = (unsigned long long)ngr * height;
    if (cells < (1ull << 31)) {
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)((cells + 511) / 512), 1, (unsigned)nfr);
#define K4S_CASE(O, FM) case (O) * 8 + (FM): hipLaunchKernelGGL((k_rgb_to_yuv422_s<O, FM>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); break;
      switch (in_order * 8 + out_fmt) { K4S_CASE(0, 2) K4S_CASE(0, 3) K4S_CASE(0, 5) K4S_CASE(1, 2) K4S_CASE(1, 3) K4S_CASE(1, 5) }
#undef K4S_CASE
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  // aligned rows -> packed / planar 4:4:4: the four-pixel form
  uintptr_t pbits = d0bits | d12bits;
  if (out_fmt == 1 && out_alpha) { pbits |= (uintptr_t)orow[3]; for (int f = 0; f < nfr; f++) pbits |= (uintptr_t)dsts[f][3]; }
  if (((out_fmt == 0 && (d0bits & (out_alpha ? 15 : 3)) == 0) || (out_fmt == 1 && (pbits & 3) == 0)) &&
      in_order <= 1 && !no_s420 && (width & 3) == 0 && (sbits & (ips == 4 ? 15 : 3)) == 0) {
    const int ngr = width >> 2;
    const unsigned long long cells = (unsigned long long)ngr * height;
    if (cells < (1ull << 31)) {
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)((cells + 511) / 512), 1, (unsigned)nfr);
#define K444_CASE(O, I, AO) case (O) * 4 + ((I) == 4 ? 2 : 0) + (AO): if (out_fmt == 1) hipLaunchKernelGGL((k_rgb_to_yuv444_s<O, I, AO, 1>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); \
        else hipLaunchKernelGGL((k_rgb_to_yuv444_s<O, I, AO, 0>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); break;
      switch (in_order * 4 + (ips == 4 ? 2 : 0) + (out_alpha ? 1 : 0)) {
        K444_CASE(0, 3, 0) K444_CASE(0, 3, 1) K444_CASE(0, 4, 0) K444_CASE(0, 4, 1) K444_CASE(1, 3, 0) K444_CASE(1, 3, 1) K444_CASE(1, 4, 0) K444_CASE(1, 4, 1)
      }
#undef K444_CASE
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  const int nrows = out_fmt == 4 ? height >> 1 : height;
  const dim3 grid(cdiv((unsigned)npairs, kBlock), (unsigned)(nrows < 2048 ? nrows : 2048), (unsigned)nfr);
#define K4_CASE(O, FM) case (O) * 8 + (FM): hipLaunchKernelGGL((k_rgb_to_yuv<O, FM>), grid, dim3(kBlock), 0, (hipStream_t)stream, a, F); break;
  switch (in_order * 8 + out_fmt) {
    K4_CASE(0, 0) K4_CASE(0, 1) K4_CASE(0, 2) K4_CASE(0, 3) K4_CASE(0, 4) K4_CASE(0, 5)
    K4_CASE(1, 0) K4_CASE(1, 1) K4_CASE(1, 2) K4_CASE(1, 3) K4_CASE(1, 4) K4_CASE(1, 5)
    K4_CASE(2, 0) K4_CASE(2, 1) K4_CASE(2, 2) K4_CASE(2, 3)
  }
#undef K4_CASE
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
static int rgb_to_yuv_impl(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha,
                           uint8_t *const dst_d[4], const int orow[4], int out_fmt, int out_alpha, int which_tables, const uint16_t *lut16_d, void *stream) {
  LGPU_REQUIRE(dst_d, "null destination table");
  uint8_t *const one[1][4] = {{dst_d[0], dst_d[1], dst_d[2], dst_d[3]}};
  return rgb_to_yuv_impl_n(&src_d, irow, width, height, in_order, in_alpha, one, orow, out_fmt, out_alpha, which_tables, lut16_d, 1, stream);
}
extern "C" int lgpu_rgb_to_yuv(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha,
                               uint8_t *const dst_d[4], const int orow[4], int out_fmt, int out_alpha, int which_tables, void *stream) {
  return rgb_to_yuv_impl(src_d, irow, width, height, in_order, in_alpha, dst_d, orow, out_fmt, out_alpha, which_tables, nullptr, stream);
}
// nframes frames of one geometry: dst_d[f * 4 + plane]
extern "C" int lgpu_rgb_to_yuv_batch(const uint8_t *const *src_d, int irow, int width, int height, int in_order, int in_alpha,
                                     uint8_t *const *dst_d, const int orow[4], int out_fmt, int out_alpha, int which_tables, int nframes, void *stream) {
  LGPU_REQUIRE(dst_d, "null destination table");
  return rgb_to_yuv_impl_n(src_d, irow, width, height, in_order, in_alpha, reinterpret_cast<uint8_t *const (*)[4]>(dst_d), orow, out_fmt, out_alpha, which_tables, nullptr, nframes, stream);
}
extern "C" int lgpu_rgb_to_yuv_lut16(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst_d, int orow, int out_fmt,
                                     int clamping_unclamped, const uint16_t *lut16_d, void *stream) {
  LGPU_REQUIRE(lut16_d, "null LUT");
  LGPU_REQUIRE(out_fmt == 2 || out_fmt == 3, "only the UYVY (2) and YUYV (3) entry points take a gamma LUT (rgb2uyvy_with_gamma, rgb2yuyv_with_gamma)");
  uint8_t *const dd[4] = {dst_d, nullptr, nullptr, nullptr};
  const int oo[4] = {orow, 0, 0, 0};
  return rgb_to_yuv_impl(src_d, irow, width, height, in_order, in_alpha, dd, oo, out_fmt, 0, clamping_unclamped ? 1 : 0, lut16_d, stream);
}

static int yuv_to_rgb_n(const uint8_t *const (*srcs)[4], const int irow[4], int width, int height, int in_fmt, int in_alpha,
                        uint8_t *const *dsts, int orow, int out_order, int out_alpha, int which_tables, int nfr, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(srcs && irow && dsts && nfr >= 1 && nfr <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0, "null frame table, 1..16 frames, or empty geometry");
  const uint8_t *const *const src_d = srcs[0];
  uint8_t *const dst_d = dsts[0];
  LGPU_REQUIRE(dst_d, "null frame");
  LGPU_REQUIRE(in_fmt >= 0 && in_fmt <= 3, "in_fmt must be 0 (packed 4:4:4), 1 (planar 4:4:4), 2 (UYVY) or 3 (YUYV); 4:2:0 / 4:2:2 planar: lgpu_yuv420p_to_rgb");
  LGPU_REQUIRE(out_order >= 0 && out_order <= 2, "out_order must be 0 (RGB), 1 (BGR) or 2 (ARGB)");
  LGPU_REQUIRE(in_fmt < 2 || !(width & 1), "UYVY / YUYV need an even width");
  LGPU_REQUIRE(in_fmt == 0 || !(which_tables & 2), "only the packed 4:4:4 conversions take a BT.709 subspace (the reference's other entry points are YCbCr only)");
  if (in_fmt == 1 && (out_order == 2 || (out_order == 1 && !out_alpha))) {
    set_error("planar 4:4:4 -> ARGB32 / BGR24 is declined: the reference's row arithmetic is broken there (src/colourspace.c:7475-7476, :7313)");
    return LGPU_E_UNSUPPORTED;
  }
  PalArgs a;
  __builtin_memset(&a, 0, sizeof a);
  const int nplanes = in_fmt == 1 ? (in_alpha ? 4 : 3) : 1;
  for (int i = 0; i < nplanes; i++) { LGPU_REQUIRE(src_d[i] && irow[i] > 0, "null source plane"); a.src[i] = src_d[i]; a.irow[i] = irow[i]; }
  if (in_fmt >= 2) LGPU_REQUIRE(!(irow[0] & 3) && !(reinterpret_cast<uintptr_t>(src_d[0]) & 3), "UYVY / YUYV rows must be 4-byte aligned");
  const int ops = (out_order == 2 || out_alpha) ? 4 : 3;
  LGPU_REQUIRE(orow >= width * ops, "rowstride smaller than a row");
  a.dst[0] = dst_d; a.orow[0] = orow; a.width = width; a.height = height;
  a.order = out_order; a.alpha_in = (in_fmt <= 1) ? in_alpha : 0; a.fmt = in_fmt; a.alpha_out = out_alpha;
  a.unclamped = which_tables & 1;
  a.tables = device_tables()->yuv2rgb[which_tables & 3];
  FxFrames F = {};
  uintptr_t sbits = (uintptr_t)irow[0], dbits = (uintptr_t)orow;
  for (int f = 0; f < nfr; f++) {
    LGPU_REQUIRE(dsts[f], "null frame");
    F.out[f][0] = dsts[f]; dbits |= (uintptr_t)dsts[f];
    for (int i = 0; i < nplanes; i++) { LGPU_REQUIRE(srcs[f][i], "null source plane"); F.in0[f][i] = srcs[f][i]; }
    sbits |= (uintptr_t)srcs[f][0];
    if (in_fmt >= 2) LGPU_REQUIRE(!((uintptr_t)srcs[f][0] & 3), "UYVY / YUYV rows must be 4-byte aligned");
  }
  // UYVY / YUYV -> 4-byte pixels on aligned frames: the cell form
  const bool no_s = tune_on(TUNE_UYVY_NO_S);
  if (in_fmt >= 2 && ops == 4 && !no_s && (width & 3) == 0 && (sbits & 7) == 0 && (dbits & 15) == 0) {
    const int ngr = width >> 2;
    const unsigned long long cells = (unsigned long long)ngr * height;
    if (cells < (1ull << 31)) {
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)((cells + 511) / 512), 1, (unsigned)nfr);
#define K3S_CASE(FM, O) case (FM) * 4 + (O): hipLaunchKernelGGL((k_uyvy_to_rgb_s<FM, O>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); break;
      switch (in_fmt * 4 + out_order) { K3S_CASE(2, 0) K3S_CASE(2, 1) K3S_CASE(2, 2) K3S_CASE(3, 0) K3S_CASE(3, 1) K3S_CASE(3, 2) }
#undef K3S_CASE
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  // packed 4:4:4 on aligned rows: the four-pixel form
  {
    const int ips = in_alpha ? 4 : 3;
    uintptr_t pb = 0;
    if (in_fmt == 1) for (int f = 0; f < nfr; f++) for (int i = 0; i < nplanes; i++) pb |= (uintptr_t)srcs[f][i] | (uintptr_t)irow[i];
    if (((in_fmt == 0 && (sbits & (ips == 4 ? 15 : 3)) == 0) || (in_fmt == 1 && (pb & 3) == 0)) && !no_s && (width & 3) == 0 && (dbits & (ops == 4 ? 15 : 3)) == 0) {
      const int ngr = width >> 2;
      const unsigned long long cells = (unsigned long long)ngr * height;
      if (cells < (1ull << 31)) {
        const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
        const dim3 gs((unsigned)((cells + 511) / 512), 1, (unsigned)nfr);
#define K444_CASE(O, I, OP) case (O) * 4 + ((I) == 4 ? 2 : 0) + ((OP) == 4 ? 1 : 0): if (in_fmt == 1) hipLaunchKernelGGL((k_yuv444_to_rgb_s<O, I, OP, 1>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); \
          else hipLaunchKernelGGL((k_yuv444_to_rgb_s<O, I, OP, 0>), gs, dim3(512), 0, (hipStream_t)stream, a, magic, F); break;
        switch (out_order * 4 + (ips == 4 ? 2 : 0) + (ops == 4 ? 1 : 0)) {
          K444_CASE(0, 3, 3) K444_CASE(0, 3, 4) K444_CASE(0, 4, 3) K444_CASE(0, 4, 4) K444_CASE(1, 3, 3) K444_CASE(1, 3, 4) K444_CASE(1, 4, 3) K444_CASE(1, 4, 4)
          K444_CASE(2, 3, 4) K444_CASE(2, 4, 4)
        }
#undef K444_CASE
        LGPU_CHECK_LAUNCH();
        return LGPU_OK;
      }
    }
  }
  const dim3 grid(cdiv((unsigned)((width + 1) >> 1), kBlock), (unsigned)(height < 2048 ? height : 2048), (unsigned)nfr);
#define K3_CASE(FM, O) case (FM) * 4 + (O): hipLaunchKernelGGL((k_yuv_to_rgb<FM, O>), grid, dim3(kBlock), 0, (hipStream_t)stream, a, F); break;
  switch (in_fmt * 4 + out_order) {
    K3_CASE(0, 0) K3_CASE(0, 1) K3_CASE(0, 2) K3_CASE(1, 0) K3_CASE(1, 1) K3_CASE(2, 0) K3_CASE(2, 1) K3_CASE(2, 2)
    K3_CASE(3, 0) K3_CASE(3, 1) K3_CASE(3, 2)
  }
#undef K3_CASE
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_yuv_to_rgb(const uint8_t *const src_d[4], const int irow[4], int width, int height, int in_fmt, int in_alpha,
                               uint8_t *dst_d, int orow, int out_order, int out_alpha, int which_tables, void *stream) {
  LGPU_REQUIRE(src_d, "null source table");
  const uint8_t *const one[1][4] = {{src_d[0], src_d[1], src_d[2], src_d[3]}};
  return yuv_to_rgb_n(one, irow, width, height, in_fmt, in_alpha, &dst_d, orow, out_order, out_alpha, which_tables, 1, stream);
}
// nframes frames of one geometry: src_d[f * 4 + plane]
extern "C" int lgpu_yuv_to_rgb_batch(const uint8_t *const *src_d, const int irow[4], int width, int height, int in_fmt, int in_alpha,
                                     uint8_t *const *dst_d, int orow, int out_order, int out_alpha, int which_tables, int nframes, void *stream) {
  LGPU_REQUIRE(src_d, "null source table");
  return yuv_to_rgb_n(reinterpret_cast<const uint8_t *const (*)[4]>(src_d), irow, width, height, in_fmt, in_alpha, dst_d, orow, out_order, out_alpha, which_tables, nframes, stream);
}

extern "C" int lgpu_rgb_to_yuv411(const uint8_t *src_d, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst_d,
                                  int clamping_unclamped, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(src_d && dst_d && width >= 4 && height > 0, "null frame, or fewer than 4 pixels per row");
  LGPU_REQUIRE(in_order >= 0 && in_order <= 2, "in_order must be 0 (RGB), 1 (BGR) or 2 (ARGB)");
  const int ips = (in_order == 2 || in_alpha) ? 4 : 3;
  LGPU_REQUIRE(irow >= (width >> 2) * 4 * ips, "rowstride smaller than a row");
  PalArgs a;
  __builtin_memset(&a, 0, sizeof a);
  a.src[0] = src_d; a.irow[0] = irow; a.dst[0] = dst_d; a.width = width; a.height = height;
  a.order = in_order; a.alpha_in = in_alpha; a.unclamped = clamping_unclamped ? 1 : 0;
  a.tables = device_tables()->rgb2yuv[a.unclamped];         // set_conversion_arrays(clamping, WEED_YUV_SUBSPACE_YCBCR) (:6511)
  const dim3 grid(cdiv((unsigned)(width >> 2), kBlock), (unsigned)(height < 2048 ? height : 2048));
  hipLaunchKernelGGL(k_rgb_to_yuv411, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

int lgpu::yuv411_to_rgb_n(const FxFrames &F, int nframes, int width_mp, int height, int orow, int out_order, int out_alpha, int clamping_unclamped, hipStream_t st) {
  int rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(width_mp > 0 && height > 0, "empty geometry");
  for (int f = 0; f < nframes; f++) LGPU_REQUIRE(F.in0[f][0] && F.out[f][0], "null frame");
  LGPU_REQUIRE(out_order >= 0 && out_order <= 2, "out_order must be 0 (RGB), 1 (BGR) or 2 (ARGB)");
  const int ps = (out_order == 2 || out_alpha) ? 4 : 3;
  LGPU_REQUIRE(orow >= width_mp * 4 * ps, "rowstride smaller than a row");
  PalArgs a;
  __builtin_memset(&a, 0, sizeof a);
  a.orow[0] = orow; a.width = width_mp; a.height = height;
  a.order = out_order; a.alpha_out = out_alpha; a.unclamped = clamping_unclamped ? 1 : 0;
  a.tables = device_tables()->yuv2rgb[a.unclamped];         // set_conversion_arrays(clamping, WEED_YUV_SUBSPACE_YCBCR) (:8316)
  LGPU_REQUIRE((unsigned long long)width_mp * height < (1ull << 31), "frame too large");
  const uint32_t ncells = (uint32_t)width_mp * (uint32_t)height;
  const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)width_mp - (width_mp == 1 ? 1 : 0));
  unsigned wgs = cdiv(ncells, kBlock), cap = cdiv((unsigned)device_cus() * (unsigned)(tune(TUNE_K2_WGS) > 0 ? tune(TUNE_K2_WGS) : 8), (unsigned)nframes);
  hipLaunchKernelGGL(k_yuv411_to_rgb, dim3(wgs < cap ? wgs : cap, (unsigned)nframes), dim3(kBlock), 0, st, a, magic, ncells, F);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_yuv411_to_rgb(const uint8_t *src_d, int width_mp, int height, uint8_t *dst_d, int orow, int out_order, int out_alpha,
                                  int clamping_unclamped, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  FxFrames F = {};
  F.in0[0][0] = src_d; F.out[0][0] = dst_d;
  return yuv411_to_rgb_n(F, 1, width_mp, height, orow, out_order, out_alpha, clamping_unclamped, (hipStream_t)stream);
}

// init_YUV_to_YUV_tables (src/colourspace.c:1108-1139); myround = round half away from zero (src/maths.h:118)
static void yuv_yuv_tables(uint8_t yc2u[256], uint8_t uvc2u[256], uint8_t yu2c[256], uint8_t uvu2c[256]) {
  auto rnd = [](double n) { return n >= 0. ? (int)(n + 0.5) : (int)(n - 0.5); };
  int i;
  for (i = 0; i <= 16; i++) yc2u[i] = 0;
  for (; i < 235; i++) yc2u[i] = (uint8_t)rnd((i - 16.) * 255. / (235. - 16.));
  for (; i < 256; i++) yc2u[i] = 255;
  for (i = 0; i < 16; i++) uvc2u[i] = 0;
  for (; i < 240; i++) uvc2u[i] = (uint8_t)rnd((i - 16.) * 255. / (240. - 16.));
  for (; i < 256; i++) uvc2u[i] = 255;
  for (i = 0; i < 256; i++) { yu2c[i] = (uint8_t)rnd((i / 255.) * (235. - 16.) + 16.); uvu2c[i] = (uint8_t)rnd((i / 255.) * (240. - 16.) + 16.); }
}

extern "C" int lgpu_yuv_switch_clamping(uint8_t *const planes_d[4], const int rowstrides[4], int palette, int height, int to_unclamped,
                                        void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  LGPU_REQUIRE(planes_d && rowstrides && planes_d[0] && rowstrides[0] > 0 && height > 0, "null plane or empty geometry");
  uint8_t t[4][256];
  yuv_yuv_tables(t[0], t[1], t[2], t[3]);
  const Lut8 ly = pack_lut(to_unclamped ? t[0] : t[2]), lc = pack_lut(to_unclamped ? t[1] : t[3]);
  const size_t n = (size_t)height * rowstrides[0];
  hipStream_t st = (hipStream_t)stream;
  auto launch3 = [&](uint8_t *b0, size_t n0, uint32_t p0, uint8_t *b1, size_t n1, uint32_t p1, uint8_t *b2, size_t n2, uint32_t p2, int period, int np) {
    ClampPlanes cp;
    cp.buf[0] = b0; cp.buf[1] = b1; cp.buf[2] = b2; cp.nbytes[0] = n0; cp.nbytes[1] = n1; cp.nbytes[2] = n2; cp.pattern[0] = p0; cp.pattern[1] = p1; cp.pattern[2] = p2;
    unsigned g = cdiv((unsigned)((n0 + 15) / 16), kBlock);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_clamp_switch, dim3(g, (unsigned)np), dim3(kBlock), 0, st, cp, period, ly, lc);
  };
  auto launch = [&](uint8_t *buf, size_t bytes, int period, uint32_t pattern) { launch3(buf, bytes, pattern, nullptr, 0, 0, nullptr, 0, 0, period, 1); };
  switch (palette) {
  case 588: launch(planes_d[0], n, 3, 0x110u); break;                    // Y U V over the whole buffer (:10957-10968)
  case 589: launch(planes_d[0], n, 4, 0x2110u); break;                   // Y U V A
  case 564: launch(planes_d[0], n, 4, 0x0101u); break;                   // U Y V Y
  case 565: launch(planes_d[0], n, 4, 0x1010u); break;                   // Y U Y V
  case 544: case 545: case 522: case 512: case 513: {
    LGPU_REQUIRE(planes_d[1] && planes_d[2], "null chroma plane");
    const size_t nc = palette == 522 ? n / 2 : (palette == 512 || palette == 513) ? n / 4 : n;
    launch3(planes_d[0], n, 0x0u, planes_d[1], nc, 0x1u, planes_d[2], nc, 0x1u, 1, 3);
    break;
  }
  default: set_error("lgpu_yuv_switch_clamping: palette %d is not handled", palette); return LGPU_E_UNSUPPORTED;
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}


static int unsupported(const char *why) { lgpu::set_error("lgpu_yuv_repack: %s", why); return LGPU_E_UNSUPPORTED; }

extern "C" int lgpu_yuv_repack(int in_pal, int out_pal, const uint8_t *const src_d[4], const int irow[4], uint8_t *const dst_d[4],
                               const int orow[4], int width, int height, int clamping_unclamped, int sampling, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if ((rc = ensure_cavgc())) return rc;
  enum { P_420 = 512, P_YV12 = 513, P_422 = 522, P_444 = 544, P_4444 = 545, P_UYVY = 564, P_YUYV = 565, P_888 = 588, P_8888 = 589 };
  LGPU_REQUIRE(src_d && dst_d && irow && orow && width > 0 && height > 0, "null plane tables or empty geometry");
  LGPU_REQUIRE(src_d[0] && dst_d[0], "null plane");
  const bool in444 = (in_pal == P_444 || in_pal == P_4444), in420 = (in_pal == P_420 || in_pal == P_YV12);
  const bool inpk = (in_pal == P_UYVY || in_pal == P_YUYV), outpk = (out_pal == P_UYVY || out_pal == P_YUYV);
  hipStream_t st = (hipStream_t)stream;
  // the K1 pair: convert_addpost_frame / convert_delpost_frame
  if (in_pal == P_888 && out_pal == P_8888) return lgpu_swizzle(LGPU_ADDPOST, 0, src_d[0], irow[0], dst_d[0], orow[0], width, height, nullptr, stream);
  if (in_pal == P_8888 && out_pal == P_888) return lgpu_swizzle(LGPU_DELPOST, 0, src_d[0], irow[0], dst_d[0], orow[0], width, height, nullptr, stream);
  if (in_pal == 595 || out_pal == 595) {
    // K5c: the 4:1:1 pairs (compact streams on both sides, see k_yuv411_repack)
    if (in_pal == out_pal) return unsupported("nothing to convert");
    if (width < 4 || (width & 3)) return unsupported("YUV411 pairs need a width that is a multiple of 4 pixels");
    lgpu::R411Args r = {};
    r.width = width; r.height = height; r.clamped = clamping_unclamped ? 0 : 1; r.irow = irow[0];
    const int other = in_pal == 595 ? out_pal : in_pal;
    int nin = 1, nout = 1;
    r.yuyv = (other == P_YUYV); r.alpha = (other == P_8888 || other == P_4444);
    if (in_pal == 595) {
      switch (out_pal) {
      case P_888: case P_8888: r.kind = lgpu::K411_TO_888; break;
      case P_444: case P_4444: r.kind = lgpu::K411_TO_444; nout = r.alpha ? 4 : 3; break;
      case P_UYVY: case P_YUYV: r.kind = lgpu::K411_TO_PK; break;
      case P_422: r.kind = lgpu::K411_TO_422P; nout = 3; break;
      case P_420: case P_YV12: r.kind = lgpu::K411_TO_420; nout = 3; break;
      default: return unsupported("YUV411 -> this palette does not exist in the reference either");
      }
    } else {
      switch (in_pal) {
      case P_444: case P_4444: r.kind = lgpu::K411_FROM_444; nin = 3; break;
      case P_UYVY: case P_YUYV: r.kind = lgpu::K411_FROM_PK; break;
      case P_888: case P_8888: {
        r.kind = lgpu::K411_FROM_888;
        LGPU_REQUIRE(irow[0] >= width * (r.alpha ? 4 : 3), "source rowstride smaller than a row");
        const size_t lim = (size_t)width * height;               // the end pointer is width * height BYTES past the start (:8277)
        r.rows = (int)((lim + (size_t)irow[0] - 1) / (size_t)irow[0]);
        if (r.rows > height) r.rows = height;
        break;
      }
      case P_420: case P_YV12: case P_422:
        if (in_pal != P_422 && (height & 1)) return unsupported("a 4:2:0 source has an even height");
        r.kind = lgpu::K411_FROM_420; r.is422 = (in_pal == P_422); nin = 3; break;
      default: return unsupported("this palette -> YUV411 does not exist in the reference either");
      }
    }
    for (int i = 0; i < nin; i++) { LGPU_REQUIRE(src_d[i], "null source plane"); r.src[i] = src_d[i]; }
    for (int i = 0; i < nout; i++) { LGPU_REQUIRE(dst_d[i], "null destination plane"); r.dst[i] = dst_d[i]; }
    if (out_pal == P_YV12) { uint8_t *t = r.dst[1]; r.dst[1] = r.dst[2]; r.dst[2] = t; }       // is_yvu (:9055-9061)
    const dim3 grid(cdiv((unsigned)(width >> 2), kBlock), (unsigned)(height < 2048 ? height : 2048));
    hipLaunchKernelGGL(lgpu::k_yuv411_repack, grid, dim3(kBlock), 0, st, r);
    if (r.kind == lgpu::K411_TO_420 && height >= 2 && !(height & 1)) {
      const int wm = width >> 2, per = wm >= 2048 ? 128 : 32, nch = (2 * wm + per - 1) / per;      // (the walk's tables stay within 64 KB of LDS: width < 32768)
      void *tab = nullptr;
      if ((rc = lgpu_malloc_ordered(&tab, (size_t)2 * nch * 256, st))) return rc;
      hipLaunchKernelGGL(lgpu::k_yuv411_420_fold_chunks, dim3((unsigned)nch), dim3(512), (size_t)2 * per, st, r.src[0], wm, height - 1, r.clamped, per, (uint8_t *)tab);
      hipLaunchKernelGGL(lgpu::k_yuv411_420_fold_walk, dim3(1), dim3(256), (size_t)2 * nch * 256, st, (const uint8_t *)tab, nch, r.dst[1], r.dst[2]);
      lgpu_free_ordered(tab, st);
    }
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if ((in420 || in_pal == P_422) && (out_pal == P_888 || out_pal == P_8888)) {
    // K5d: convert_quad_chroma_packed / convert_double_chroma_packed
    if (width & 1) return unsupported("a subsampled source has an even width");
    if (in420 && ((height & 1) || height < 2)) return unsupported("4:2:0 -> packed 4:4:4 needs an even height (the reference's trailing loop reads the row after the frame, colourspace.c:10798)");
    lgpu::ChromaUpArgs c = {};
    for (int i = 0; i < 3; i++) { LGPU_REQUIRE(src_d[i] && irow[i] > 0, "null source plane"); c.src[i] = src_d[i]; c.irow[i] = irow[i]; }
    c.dst = dst_d[0]; c.orow = orow[0]; c.width = width; c.height = height;
    c.is420 = in420 ? 1 : 0; c.alpha = (out_pal == P_8888); c.jpeg = (sampling == 0); c.clamped = clamping_unclamped ? 0 : 1;
    const int crows = in420 ? height >> 1 : height;
    c.ulast = (unsigned)irow[1] * (unsigned)crows - 1u; c.vlast = (unsigned)irow[2] * (unsigned)crows - 1u;
    const dim3 grid(cdiv((unsigned)(width >> 1), kBlock), (unsigned)(height < 2048 ? height : 2048));
    hipLaunchKernelGGL(lgpu::k_chroma_up_packed, grid, dim3(kBlock), 0, st, c);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  lgpu::RepackArgs a = {};
  a.width = width; a.height = height; a.clamped = clamping_unclamped ? 0 : 1;
  a.yuyv_in = (in_pal == P_YUYV); a.yuyv_out = (out_pal == P_YUYV);
  int nin = 1, nout = 1;
  if (in444 && (out_pal == P_888 || out_pal == P_8888)) {
    a.kind = lgpu::RK_COMBINE; a.in_alpha = (in_pal == P_4444); a.out_alpha = (out_pal == P_8888); nin = a.in_alpha ? 4 : 3;
  } else if (in_pal == P_888 && out_pal == P_444) {
    a.kind = lgpu::RK_SPLIT; nout = 3;
  } else if (in444 && (out_pal == P_444 || out_pal == P_4444) && in_pal != out_pal) {
    a.kind = lgpu::RK_COPY444; a.out_alpha = (out_pal == P_4444); nin = 3; nout = a.out_alpha ? 4 : 3;
    a.copy_w = (orow[0] == irow[0]) ? irow[0] : width;                     // :7658-7663 copies whole planes when the strides agree
  } else if (inpk && outpk && in_pal != out_pal) {
    if (width & 1) return unsupported("packed 4:2:2 needs an even width");
    a.kind = lgpu::RK_SWAB;
  } else if (in420 && outpk) {
    // convert_yuv420_to_uyvy_frame steps its chroma pointers back by the rowstride (:7143-7146): compact chroma planes only
    if (irow[1] != (width >> 1) || irow[2] != (width >> 1) || ((width | height) & 1)) return unsupported("4:2:0 -> packed 4:2:2 needs compact chroma planes and even dimensions (colourspace.c:7143)");
    a.kind = lgpu::RK_420_TO_PK; a.cshift = 1; nin = 3;
  } else if (in_pal == P_422 && outpk) {
    // own specification (docs/SPECS.md, "evident intent"): convert_yuv422p_to_uyvy_frame / _yuyv_frame (:6442-6497) loop `width` macropixels per row and overrun;
    // what they mean is the interleave of the 4:2:0 sibling with a chroma row per luma row
    if (width & 1) return unsupported("packed 4:2:2 needs an even width");
    a.kind = lgpu::RK_420_TO_PK; a.cshift = 0; nin = 3;
  } else if (in420 && out_pal == P_422) {
    if ((width | height) & 1) return unsupported("a 4:2:0 source has even width and height");
    a.kind = lgpu::RK_420_TO_422P; nin = 3; nout = 3;
  } else if (in444 && (out_pal == P_420 || out_pal == P_YV12)) {
    // 4:2:0 layers have even dimensions (create_empty_pixel_data :11601-11603); with an odd one the reference writes past the planes
    if ((width | height) & 1) return unsupported("a 4:2:0 destination needs even width and height");
    a.kind = lgpu::RK_444_TO_420; nin = 3; nout = 3;
    a.copy_w = (orow[0] == irow[0]) ? irow[0] : width;                     // :7711-7712
  } else if (in444 && outpk) {
    // only the compact branch of convert_yuv_planar_to_uyvy_frame stays inside its buffers (:7512-7524 vs :7526-7543)
    if (irow[0] != width || orow[0] != width * 2 || (width & 1)) return unsupported("4:4:4 planar -> packed 4:2:2 needs compact rows and an even width (colourspace.c:7526)");
    a.kind = lgpu::RK_444_TO_PK; nin = 3;
  } else if ((in_pal == P_888 || in_pal == P_8888) && (out_pal == P_420 || out_pal == P_YV12 || out_pal == P_422 || outpk)) {
    // :8035-8270: every one of these walks its destination as a compact buffer (the strided branches subtract the wrong widths)
    a.in_alpha = (in_pal == P_8888);
    if (width & 1) return unsupported("a subsampled destination needs an even width");
    if (out_pal == P_420 || out_pal == P_YV12) {
      if ((height & 1) || orow[0] != width || orow[1] != (width >> 1) || orow[2] != (width >> 1)) return unsupported("YUV888 -> 4:2:0 needs a compact destination and an even height (colourspace.c:8064-8087)");
      a.kind = lgpu::RK_888_TO_420; nout = 3;
    } else if (out_pal == P_422) {
      if (irow[0] != width * (a.in_alpha ? 4 : 3) || orow[0] != width || orow[1] != (width >> 1)) return unsupported("YUV888 -> 4:2:2 planar needs compact rows on both sides (colourspace.c:8161-8178)");
      a.kind = lgpu::RK_888_TO_422; a.out_alpha = 1; nout = 3;
    } else {
      if (orow[0] != width * 2) return unsupported("YUV888 -> packed 4:2:2 needs a compact destination (colourspace.c:8205-8222)");
      a.kind = lgpu::RK_888_TO_422; a.out_alpha = 0;
    }
  } else if (inpk && out_pal == P_422) {
    if ((width & 1) || irow[0] != width * 2 || orow[0] != width || orow[1] != (width >> 1) || orow[2] != (width >> 1)) return unsupported("packed 4:2:2 -> planar 4:2:2 needs compact rows on both sides (colourspace.c:8093-8126)");
    a.kind = lgpu::RK_PK_TO_422P; nout = 3;
  } else if (inpk && (out_pal == P_444 || out_pal == P_4444)) {
    if (width & 1) return unsupported("packed 4:2:2 needs an even width");
    if (orow[0] != orow[1] || orow[0] != orow[2]) return unsupported("packed 4:2:2 -> planar needs equal plane rowstrides (colourspace.c:7813-7816 mixes them)");
    a.kind = lgpu::RK_PK_TO_444; nout = 3;
  } else if (inpk && (out_pal == P_888 || out_pal == P_8888)) {
    if (width & 1) return unsupported("packed 4:2:2 needs an even width");
    a.kind = lgpu::RK_PK_TO_888; a.out_alpha = (out_pal == P_8888);
  } else if (inpk && (out_pal == P_420 || out_pal == P_YV12)) {
    if (((width | height) & 1) || irow[0] != width * 2 || orow[0] != width || orow[1] != (width >> 1) || orow[2] != (width >> 1))
      return unsupported("packed 4:2:2 -> 4:2:0 needs compact rows on both sides (colourspace.c:7887-7927 has no strides)");
    a.kind = lgpu::RK_PK_TO_420; nout = 3;
  } else {
    return unsupported("this YUV -> YUV pair is not taken (the reference function overruns, or depends on the destination's previous contents)");
  }
  for (int i = 0; i < nin; i++) { LGPU_REQUIRE(src_d[i], "null source plane"); a.src[i] = src_d[i]; a.irow[i] = irow[i]; }
  for (int i = 0; i < nout; i++) { LGPU_REQUIRE(dst_d[i], "null destination plane"); a.dst[i] = dst_d[i]; a.orow[i] = orow[i]; }
  if (a.kind == lgpu::RK_COPY444 && a.out_alpha) {
    LGPU_REQUIRE(dst_d[3], "null alpha plane");
    a.dst[3] = dst_d[3];
    // memset(dest[3], 255, orowstride * height) (:7686): padding included
    if ((rc = lgpu_fill(dst_d[3], 255, (size_t)orow[0] * height, stream))) return rc;
  }
  if (a.kind == lgpu::RK_PK_TO_444 && out_pal == P_4444) {
    LGPU_REQUIRE(dst_d[3], "null alpha plane");
    if ((rc = lgpu_fill(dst_d[3], 255, (size_t)orow[3] * height, stream))) return rc;                 // :7819
  }
  const bool no_s = tune_on(TUNE_REPACK_NO_S);
  if (a.kind == lgpu::RK_420_TO_PK && !no_s && (width & 7) == 0 && (((uintptr_t)src_d[0] | (uintptr_t)irow[0]) & 7) == 0 &&
      (((uintptr_t)src_d[1] | (uintptr_t)src_d[2] | (uintptr_t)irow[1] | (uintptr_t)irow[2]) & 3) == 0 &&        // dword loads on EVERY chroma row
      (((uintptr_t)dst_d[0] | (uintptr_t)((orow[0] / 4) * 4)) & 15) == 0 && (unsigned long long)(width >> 3) * height < (1ull << 31)) {
    const int ngr = width >> 3;
    const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
    const unsigned long long cells = (unsigned long long)ngr * height;
    hipLaunchKernelGGL(lgpu::k_420_to_packed_s, dim3((unsigned)((cells + 511) / 512)), dim3(512), 0, st, a, magic);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if (a.kind == lgpu::RK_COMBINE && !no_s && (width & 3) == 0 && ((irow[0] | irow[1] | irow[2]) & 3) == 0 && irow[0] == irow[1] && irow[0] == irow[2] &&
      (((uintptr_t)src_d[0] | (uintptr_t)src_d[1] | (uintptr_t)src_d[2]) & 3) == 0 && (!a.in_alpha || !a.out_alpha || (((uintptr_t)src_d[3]) & 3) == 0) &&
      (((uintptr_t)dst_d[0] | (uintptr_t)orow[0]) & (a.out_alpha ? 15 : 3)) == 0 && (unsigned long long)(width >> 2) * height < (1ull << 31)) {
    const int ngr = width >> 2;
    const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
    hipLaunchKernelGGL(lgpu::k_combine_s, dim3((unsigned)(((unsigned long long)ngr * height + 511) / 512)), dim3(512), 0, st, a, magic);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if (a.kind == lgpu::RK_SPLIT && !no_s && (width & 3) == 0 && (((uintptr_t)src_d[0] | (uintptr_t)irow[0]) & 3) == 0 &&
      (((uintptr_t)dst_d[0] | (uintptr_t)dst_d[1] | (uintptr_t)dst_d[2] | (uintptr_t)orow[0] | (uintptr_t)orow[1] | (uintptr_t)orow[2]) & 3) == 0 &&
      (unsigned long long)(width >> 2) * height < (1ull << 31)) {
    const int ngr = width >> 2;
    const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
    hipLaunchKernelGGL(lgpu::k_split_s, dim3((unsigned)(((unsigned long long)ngr * height + 511) / 512)), dim3(512), 0, st, a, magic);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if (a.kind == lgpu::RK_SWAB && !no_s && (width & 7) == 0 && (((uintptr_t)src_d[0] | (uintptr_t)irow[0] | (uintptr_t)dst_d[0] | (uintptr_t)orow[0]) & 15) == 0 &&
      (unsigned long long)(width >> 3) * height < (1ull << 31)) {
    const int ngr = width >> 3;
    const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
    hipLaunchKernelGGL(lgpu::k_swab_s, dim3((unsigned)(((unsigned long long)ngr * height + 511) / 512)), dim3(512), 0, st, a, magic);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if ((a.kind == lgpu::RK_PK_TO_420 || a.kind == lgpu::RK_PK_TO_444 || a.kind == lgpu::RK_PK_TO_888) && !no_s && (width & 7) == 0 &&
      (((uintptr_t)src_d[0] | (uintptr_t)((irow[0] / 4) * 4)) & 15) == 0 && (unsigned long long)(width >> 3) * height < (1ull << 31)) {
    bool ok;
    if (a.kind == lgpu::RK_PK_TO_420) ok = (((uintptr_t)dst_d[0]) & 7) == 0 && (((uintptr_t)dst_d[1] | (uintptr_t)dst_d[2]) & 3) == 0;      // (compact destination: width a multiple of 8)
    else if (a.kind == lgpu::RK_PK_TO_444) ok = (((uintptr_t)dst_d[0] | (uintptr_t)dst_d[1] | (uintptr_t)dst_d[2] | (uintptr_t)orow[0]) & 7) == 0;
    else ok = (((uintptr_t)dst_d[0] | (uintptr_t)orow[0]) & (a.out_alpha ? 15 : 3)) == 0;
    if (ok) {
      const int ngr = width >> 3, rows_ = a.kind == lgpu::RK_PK_TO_420 ? (height + 1) >> 1 : height;
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)(((unsigned long long)ngr * rows_ + 511) / 512));
      if (a.kind == lgpu::RK_PK_TO_420) hipLaunchKernelGGL(lgpu::k_pk_to_s<lgpu::RK_PK_TO_420>, gs, dim3(512), 0, st, a, magic);
      else if (a.kind == lgpu::RK_PK_TO_444) hipLaunchKernelGGL(lgpu::k_pk_to_s<lgpu::RK_PK_TO_444>, gs, dim3(512), 0, st, a, magic);
      else hipLaunchKernelGGL(lgpu::k_pk_to_s<lgpu::RK_PK_TO_888>, gs, dim3(512), 0, st, a, magic);
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if ((a.kind == lgpu::RK_888_TO_420 || a.kind == lgpu::RK_888_TO_422) && !no_s && (width & 3) == 0 && (((uintptr_t)src_d[0] | (uintptr_t)irow[0]) & (a.in_alpha ? 15 : 3)) == 0 &&
      (unsigned long long)(width >> 2) * height < (1ull << 31)) {
    // (the destinations of these kinds are compact: checked above)
    const bool planar = a.kind == lgpu::RK_888_TO_420 || a.out_alpha;
    const bool ok = planar ? ((((uintptr_t)dst_d[0]) & 3) == 0 && (((uintptr_t)dst_d[1] | (uintptr_t)dst_d[2]) & 1) == 0) : ((((uintptr_t)dst_d[0]) & 7) == 0);
    if (ok) {
      const int ngr = width >> 2, rows_ = a.kind == lgpu::RK_888_TO_420 ? height >> 1 : height;
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
      const dim3 gs((unsigned)(((unsigned long long)ngr * rows_ + 511) / 512));
      if (a.kind == lgpu::RK_888_TO_420) hipLaunchKernelGGL(lgpu::k_888_to_s<lgpu::RK_888_TO_420>, gs, dim3(512), 0, st, a, magic);
      else hipLaunchKernelGGL(lgpu::k_888_to_s<lgpu::RK_888_TO_422>, gs, dim3(512), 0, st, a, magic);
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if (a.kind == lgpu::RK_420_TO_422P && !no_s && (width & 7) == 0 && (unsigned long long)(width >> 3) * height < (1ull << 31) &&
      (((uintptr_t)src_d[0] | (uintptr_t)irow[0] | (uintptr_t)dst_d[0] | (uintptr_t)orow[0]) & 7) == 0 &&
      (((uintptr_t)src_d[1] | (uintptr_t)src_d[2] | (uintptr_t)irow[1] | (uintptr_t)irow[2] | (uintptr_t)dst_d[1] | (uintptr_t)dst_d[2] | (uintptr_t)orow[1] | (uintptr_t)orow[2]) & 3) == 0) {
    const int ngr = width >> 3;
    const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ngr - (ngr == 1 ? 1 : 0));
    hipLaunchKernelGGL(lgpu::k_420_to_422p_s, dim3((unsigned)(((unsigned long long)ngr * height + 511) / 512)), dim3(512), 0, st, a, magic);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  const int rows = (a.kind == lgpu::RK_444_TO_420 || a.kind == lgpu::RK_PK_TO_420 || a.kind == lgpu::RK_888_TO_420) ? (height + 1) >> 1 : height;
  const int span = a.copy_w > width ? a.copy_w : width;
  const dim3 grid(cdiv((unsigned)((span + 1) >> 1), kBlock), (unsigned)(rows < 2048 ? rows : 2048));
  hipLaunchKernelGGL(lgpu::k_yuv_repack, grid, dim3(kBlock), 0, st, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
