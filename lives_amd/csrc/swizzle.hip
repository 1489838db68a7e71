// swizzle.hip -- K1 packed-RGB swizzles, K6 gamma-LUT apply, K9 alpha (un)premultiply.
//
// All three are pure HBM streaming kernels: 1 read + 1 write of the frame, no reuse.  One lane moves
// 4 pixels (16 B of a 4-byte palette, 12 B of a 3-byte one) so every wavefront issues full-width,
// contiguous global accesses; byte shuffles are v_perm_b32; the 256-byte gamma LUT rides in the
// kernarg segment and is staged to LDS once per workgroup.
#include "lgpu_common.h"
#include <map>
#include <mutex>
#include <vector>

namespace lgpu {

// selector bytes: 0..3 = source byte of the (normalised) input pixel, 0x0D = constant 0xFF
struct SwzDesc {
  uint32_t sel;       // v_perm selector applied to {0, pixel}
  uint32_t lutmask;   // output bytes that go through the gamma LUT (colour bytes)
  int ibpp, obpp;
};

static bool swz_desc(int op, int alpha_first, SwzDesc *d) {
  // out byte k <- in byte s[k]; 0xFF = new opaque alpha; a[k] = 1 when out byte k carries alpha
  uint8_t s[4] = {0, 0, 0, 0}, a[4] = {0, 0, 0, 0};
  auto set = [&](int ib, int ob, int s0, int s1, int s2, int s3) { d->ibpp = ib; d->obpp = ob; s[0] = s0; s[1] = s1; s[2] = s2; s[3] = s3; };
  switch (op) {
  case LGPU_SWAP3: set(3, 3, 2, 1, 0, 0); break;
  case LGPU_SWAP4: set(4, 4, 3, 2, 1, 0); a[alpha_first ? 3 : 0] = 1; break;
  case LGPU_SWAP3ADDPOST: set(3, 4, 2, 1, 0, 0xFF); break;
  case LGPU_SWAP3ADDPRE: set(3, 4, 0xFF, 2, 1, 0); break;
  case LGPU_SWAP3POSTALPHA: set(4, 4, 2, 1, 0, 3); a[3] = 1; break;
  case LGPU_SWAP3PREALPHA: set(4, 4, 0, 3, 2, 1); a[0] = 1; break;
  case LGPU_ADDPOST: set(3, 4, 0, 1, 2, 0xFF); break;
  case LGPU_ADDPRE: set(3, 4, 0xFF, 0, 1, 2); break;
  case LGPU_SWAP3DELPOST: set(4, 3, 2, 1, 0, 0); break;
  case LGPU_DELPOST: set(4, 3, 0, 1, 2, 0); break;
  case LGPU_DELPRE: set(4, 3, 1, 2, 3, 0); break;
  case LGPU_SWAP3DELPRE: set(4, 3, 3, 2, 1, 0); break;
  case LGPU_SWAPPREPOST:
    if (alpha_first) { set(4, 4, 1, 2, 3, 0); a[3] = 1; } else { set(4, 4, 3, 0, 1, 2); a[0] = 1; }
    break;
  default: return false;
  }
  d->sel = 0; d->lutmask = 0;
  for (int k = 0; k < 4; k++) {
    const bool konst = (s[k] == 0xFF);
    d->sel |= (uint32_t)(konst ? 0x0D : s[k]) << (8 * k);
    if (!konst && !a[k] && k < d->obpp) d->lutmask |= 0xFFu << (8 * k);
  }
  return true;
}

template <int IB, int OB, bool LUT>
__global__ __launch_bounds__(kBlock) void k_swizzle(const FrameTab F, int irow, int orow,
                                                     int width, int height, uint32_t sel, uint32_t lutmask, Lut8 lut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  if (LUT) { stage_lut(s_lut, lut); __syncthreads(); }
  const uint8_t *__restrict__ src = F.src[blockIdx.z];          // the frame is the grid's z index (lgpu_swizzle_batch)
  uint8_t *dst = F.dst[blockIdx.z];
  const int groups = width >> 2;                     // full 4-pixel groups per row
  const int g = blockIdx.x * kBlock + threadIdx.x;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const uint8_t *ip = src + (size_t)y * irow;
    uint8_t *op = dst + (size_t)y * orow;
    if (g < groups) {
      uint32_t p[4], q[4];
      if (IB == 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ip + (size_t)g * 16);
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      } else {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(ip + (size_t)g * 12);
        unpack3(w[0], w[1], w[2], p);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t t = __builtin_amdgcn_perm(0u, p[k], sel);
        if (LUT) t = (lut4(s_lut, t) & lutmask) | (t & ~lutmask);
        q[k] = t;
      }
      if (OB == 4) {
        *reinterpret_cast<uint4 *>(op + (size_t)g * 16) = make_uint4(q[0], q[1], q[2], q[3]);
      } else {
        uint32_t w0, w1, w2;
        pack3(q, w0, w1, w2);
        uint32_t *o = reinterpret_cast<uint32_t *>(op + (size_t)g * 12);
        o[0] = w0; o[1] = w1; o[2] = w2;
      }
    } else if (g == groups) {
      // ragged tail: up to 3 pixels, byte-wise
      for (int x = groups * 4; x < width; x++) {
        uint32_t pin = 0;
        for (int b = 0; b < IB; b++) pin |= (uint32_t)ip[x * IB + b] << (8 * b);
        uint32_t t = __builtin_amdgcn_perm(0u, pin, sel);
        if (LUT) t = (lut4(s_lut, t) & lutmask) | (t & ~lutmask);
        for (int b = 0; b < OB; b++) op[x * OB + b] = (uint8_t)(t >> (8 * b));
      }
    }
  }
}

// any alignment: one pixel per lane, byte accesses
template <bool LUT>
__global__ __launch_bounds__(kBlock) void k_swizzle_bytes(const FrameTab F, int irow, int orow,
                                                           int width, int height, int ib, int ob, uint32_t sel,
                                                           uint32_t lutmask, Lut8 lut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  if (LUT) { stage_lut(s_lut, lut); __syncthreads(); }
  const uint8_t *__restrict__ src = F.src[blockIdx.z];
  uint8_t *dst = F.dst[blockIdx.z];
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= width) return;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const uint8_t *ip = src + (size_t)y * irow + (size_t)x * ib;
    uint8_t *op = dst + (size_t)y * orow + (size_t)x * ob;
    uint32_t pin = ip[0] | ((uint32_t)ip[1] << 8) | ((uint32_t)ip[2] << 16);
    if (ib == 4) pin |= (uint32_t)ip[3] << 24;
    uint32_t t = __builtin_amdgcn_perm(0u, pin, sel);
    if (LUT) t = (lut4(s_lut, t) & lutmask) | (t & ~lutmask);
    op[0] = (uint8_t)t; op[1] = (uint8_t)(t >> 8); op[2] = (uint8_t)(t >> 16);
    if (ob == 4) op[3] = (uint8_t)(t >> 24);
  }
}

// --- K6 -------------------------------------------------------------------------------------------------
// The rectangle's rows are treated as byte ranges; each lane owns one 16-byte aligned chunk.  `chanmask`
// marks the colour bytes of a dword (0x00FFFFFF RGBA/BGRA, 0xFFFFFF00 ARGB, 0xFFFFFFFF 3-byte palettes).
__global__ __launch_bounds__(kBlock) void k_gamma_apply(const FrameTab F, int rowstride, int byte0, int byte1, int height,
                                                         uint32_t chanmask, Lut8 lut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  stage_lut(s_lut, lut);
  __syncthreads();
  uint8_t *pix = F.dst[blockIdx.z];
  const int c0 = byte0 & ~15;
  const int o = c0 + (blockIdx.x * kBlock + threadIdx.x) * 16;
  if (o >= byte1) return;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    uint8_t *row = pix + (size_t)y * rowstride;
    if (o >= byte0 && o + 16 <= byte1) {
      uint4 v = *reinterpret_cast<uint4 *>(row + o);
      v.x = (lut4(s_lut, v.x) & chanmask) | (v.x & ~chanmask);
      v.y = (lut4(s_lut, v.y) & chanmask) | (v.y & ~chanmask);
      v.z = (lut4(s_lut, v.z) & chanmask) | (v.z & ~chanmask);
      v.w = (lut4(s_lut, v.w) & chanmask) | (v.w & ~chanmask);
      *reinterpret_cast<uint4 *>(row + o) = v;
    } else {
      for (int b = (o < byte0 ? byte0 : o); b < o + 16 && b < byte1; b++)
        if ((chanmask >> (8 * (b & 3))) & 0xFF) row[b] = s_lut[row[b]];
    }
  }
}
// rows that are not 16-byte aligned: one byte per lane
__global__ __launch_bounds__(kBlock) void k_gamma_apply_bytes(const FrameTab F, int rowstride, int byte0, int byte1, int height,
                                                               int psize, int alpha_first, Lut8 lut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  stage_lut(s_lut, lut);
  __syncthreads();
  uint8_t *pix = F.dst[blockIdx.z];
  const int b = byte0 + blockIdx.x * kBlock + threadIdx.x;
  if (b >= byte1) return;
  if (psize == 4 && ((b & 3) == (alpha_first ? 0 : 3))) return;   // rows start on a pixel boundary
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    uint8_t *p = pix + (size_t)y * rowstride + b;
    *p = s_lut[*p];
  }
}

// --- K9 -------------------------------------------------------------------------------------------------
// table-free: the reference tables are al[a][v] = CLAMP0255f((float)v * (255.f / a)) and
// unal[a][v] = CLAMP0255f((float)v / (255.f / a))  (src/colourspace.c:1141-1160); two IEEE float ops and a
// double-precision round reproduce every one of the 2 x 65536 entries (tests/test_premult_gpu.py sweeps them).
__device__ __forceinline__ uint32_t premult_byte(uint32_t v, float ratio, bool un) {
  const float a = un ? __fdiv_rn((float)v, ratio) : __fmul_rn((float)v, ratio);
  if (a != a) return 0;
  const double d = (double)a;
  return d >= 254.5 ? 255u : d < -0.5 ? 0u : (uint32_t)(int)(d + .5);
}
__device__ __forceinline__ uint32_t premult_pixel(uint32_t p, int alpha_first, int un, const float *s_ratio) {
  const uint32_t al = alpha_first ? (p & 0xFF) : (p >> 24);
  const float ratio = s_ratio[al];                 // 255.f / alpha, the correctly rounded quotient, from the workgroup's table instead of a division per pixel
  if (alpha_first)
    return al | (premult_byte((p >> 8) & 0xFF, ratio, un) << 8) | (premult_byte((p >> 16) & 0xFF, ratio, un) << 16) | (premult_byte(p >> 24, ratio, un) << 24);
  return premult_byte(p & 0xFF, ratio, un) | (premult_byte((p >> 8) & 0xFF, ratio, un) << 8) | (premult_byte((p >> 16) & 0xFF, ratio, un) << 16) | (al << 24);
}
// VEC: four pixels per lane, 16-byte loads and stores (rows and base 16-byte aligned, decided on the host)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_premult(const FrameTab F, int rowstride, int width, int height, int alpha_first, int un) {
  uint8_t *pix = F.dst[blockIdx.z];
  __shared__ float s_ratio[256];
  for (int i = threadIdx.x; i < 256; i += kBlock) s_ratio[i] = __fdiv_rn(255.f, (float)i);
  __syncthreads();
  const int x = (blockIdx.x * kBlock + threadIdx.x) * (VEC ? 4 : 1);
  if (x >= width) return;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    uint32_t *pp = reinterpret_cast<uint32_t *>(pix + (size_t)y * rowstride) + x;
    if (VEC && x + 4 <= width) {
      uint4 v = *reinterpret_cast<const uint4 *>(pp);
      v.x = premult_pixel(v.x, alpha_first, un, s_ratio); v.y = premult_pixel(v.y, alpha_first, un, s_ratio);
      v.z = premult_pixel(v.z, alpha_first, un, s_ratio); v.w = premult_pixel(v.w, alpha_first, un, s_ratio);
      *reinterpret_cast<uint4 *>(pp) = v;
    } else {
      for (int k = 0; k < (VEC ? 4 : 1) && x + k < width; k++) pp[k] = premult_pixel(pp[k], alpha_first, un, s_ratio);
    }
  }
}

// --- per-byte-position LUTs: negate / posterise / ccorrect (scripts/{negate,posterise,ccorrect}.script) -----------------
// Every byte position of a pixel has its own 256-entry table (identity where the effect copies the byte); lane = 4 pixels.
struct Luts4 { uint32_t w[4][64]; };
template <int PS>
__global__ __launch_bounds__(kBlock) void k_byte_luts(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, Luts4 luts) {
  __shared__ __attribute__((aligned(16))) uint8_t s_l[4 * 256];
  for (int i = threadIdx.x; i < 256; i += kBlock) reinterpret_cast<uint32_t *>(s_l)[i] = luts.w[i >> 6][i & 63];
  __syncthreads();
  const int x4 = (blockIdx.x * kBlock + threadIdx.x) * 4;
  if (x4 >= width) return;
  const int n = width - x4 < 4 ? width - x4 : 4;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const uint8_t *s = src + (size_t)y * irow + (size_t)x4 * PS;
    uint8_t *d = dst + (size_t)y * orow + (size_t)x4 * PS;
    if (PS == 4 && n == 4 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
      uint4 v = *reinterpret_cast<const uint4 *>(s);
      uint32_t *pv = &v.x;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t p = pv[k];
        pv[k] = s_l[p & 0xFF] | ((uint32_t)s_l[256 + ((p >> 8) & 0xFF)] << 8) | ((uint32_t)s_l[512 + ((p >> 16) & 0xFF)] << 16) | ((uint32_t)s_l[768 + (p >> 24)] << 24);
      }
      *reinterpret_cast<uint4 *>(d) = v;
    } else {
      for (int k = 0; k < n; k++)
#pragma unroll
        for (int c = 0; c < PS; c++) d[k * PS + c] = s_l[c * 256 + s[k * PS + c]];
    }
  }
}

// --- K9b: alpha_premult on YUVA8888 / YUVA4444P (src/colourspace.c:11995-12096) ------------------------------------------------
// unclamped: the RGB arithmetic on Y, U and V.  clamped: the four tables of init_unal (:1141-1160; 4 x 64 KB) -- evaluated, not gathered: three byte gathers per
// pixel from 64 KB tables cost a wave up to 3 x 64 cache lines (12.3 us per 1080p frame); the entries are
//   alcy / unalcy [i][j] = (int)((float)j / alpha + .5) > lim ? cap : (int)((float)(j - 16.) / alpha + 16. + .5)        alpha = 255.f / i, (lim, cap) = (224, 240) / (219, 235)
//     = floor((2 j i + 255) / 510) > lim ? cap : floor((2 (j - 16) i + 8415) / 510)   EXACTLY: 2 x i + 255 is odd and 510 even, so no quotient is nearer than 1 / 510 to the
//       rounding boundary, four orders of magnitude more than the float error -- all 2 x 65,536 entries equal the integer form (tests/test_host_cpu.py), row 0 included;
//   alcuv / unalcuv [i][j] = c255f((float)(j - off) * alpha + off), off = 128 / 16: here the real value does land on boundaries (2,154 + 2,188 entries) and 26 of them go the
//       other way in float, so the float product is kept: __fmul_rn with alpha from a 256-entry table of correctly rounded quotients (one per lane and workgroup), the
//       rest in exact double steps; alpha = inf (i = 0) gives the reference's 0 / 255 / (NaN ->) 0 through the saturating conversion.
struct PremultYuvaArgs {
  uint8_t *p[4];
  int rs[4];
  int width, height, planar, clamped, un, dword;
};
__device__ __forceinline__ uint32_t pm_div510(uint32_t n) { return __umulhi(n, 2155905153u) >> 8; }       // floor(n / 510) for every 32-bit n (510 * 2155905153 - 2^40 = 254 <= 2^8)
template <int UN>
__device__ __forceinline__ uint32_t pm_cy(uint32_t y, uint32_t al) {
  const uint32_t t = pm_div510(2u * y * al + 255u);
  const uint32_t m = pm_div510((uint32_t)(2 * ((int)y - 16) * (int)al + 8415));          // 255 .. 130,305: never negative
  return t > (UN ? 219u : 224u) ? (UN ? 235u : 240u) : m;
}
template <int UN>
__device__ __forceinline__ uint32_t pm_cuv(uint32_t v, float alpha) {
  const int off = UN ? 16 : 128;
  const float p = __fmul_rn((float)((int)v - off), alpha);
  const int r = (int)((double)p + ((double)off + .5));                    // p + off and + .5 are exact in double; NaN -> 0, +-inf saturate
  return (uint32_t)min(max(r, 0), 255);
}
template <int UN, int PLANAR>
__device__ __forceinline__ uint32_t pm_clamped_px(uint32_t px, const float *s_ratio) {        // packed Y U V A pixel
  const uint32_t al = px >> 24;
  const float alpha = s_ratio[al];
  const uint32_t ny = pm_cy<UN>(px & 255u, al);
  uint32_t nu, nv;
  if (UN || PLANAR) { nu = pm_cuv<UN>((px >> 8) & 255u, alpha); nv = pm_cuv<UN>((px >> 16) & 255u, alpha); }
  else nu = nv = pm_cuv<0>(ny, alpha);                 // the packed FORWARD loop indexes alcuv with the Y byte it has just written (:12089-12091)
  return ny | (nu << 8) | (nv << 16) | (px & 0xff000000u);
}
// 4 pixels per lane on 16-byte aligned packed rows (VEC), one otherwise; planar layers: one sample of each plane per lane
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_premult_yuva(PremultYuvaArgs a) {
  __shared__ float s_ratio[256];
  for (int i = threadIdx.x; i < 256; i += kBlock) s_ratio[i] = __fdiv_rn(255.f, (float)i);
  __syncthreads();
  const int x = (blockIdx.x * kBlock + threadIdx.x) * (VEC ? 4 : 1);
  if (x >= a.width) return;
  for (int i = blockIdx.y; i < a.height; i += gridDim.y) {
    uint8_t *py, *pu, *pv;
    uint32_t al;
    if (a.planar) {
      py = a.p[0] + (size_t)i * a.rs[0] + x; pu = a.p[1] + (size_t)i * a.rs[1] + x; pv = a.p[2] + (size_t)i * a.rs[2] + x;
      al = a.p[3][(size_t)i * a.rs[3] + x];
    } else {
      py = a.p[0] + (size_t)i * a.rs[0] + 4 * (size_t)x; pu = py + 1; pv = py + 2;
      if (VEC) {                                          // (width % 4 == 0 on this path)
        uint4 q = *reinterpret_cast<const uint4 *>(py);
        if (a.un) { q.x = pm_clamped_px<1, 0>(q.x, s_ratio); q.y = pm_clamped_px<1, 0>(q.y, s_ratio); q.z = pm_clamped_px<1, 0>(q.z, s_ratio); q.w = pm_clamped_px<1, 0>(q.w, s_ratio); }
        else { q.x = pm_clamped_px<0, 0>(q.x, s_ratio); q.y = pm_clamped_px<0, 0>(q.y, s_ratio); q.z = pm_clamped_px<0, 0>(q.z, s_ratio); q.w = pm_clamped_px<0, 0>(q.w, s_ratio); }
        *reinterpret_cast<uint4 *>(py) = q;
        continue;
      }
      if (a.dword) {                                    // 4-byte aligned rows: one dword in, one dword out
        const uint32_t px = *(const uint32_t *)py;
        uint32_t o;
        if (!a.clamped) {
          const float ratio = s_ratio[px >> 24];
          o = (premult_byte(px & 255u, ratio, a.un) & 255u) | ((premult_byte((px >> 8) & 255u, ratio, a.un) & 255u) << 8) | ((premult_byte((px >> 16) & 255u, ratio, a.un) & 255u) << 16) | (px & 0xff000000u);
        } else o = a.un ? pm_clamped_px<1, 0>(px, s_ratio) : pm_clamped_px<0, 0>(px, s_ratio);
        *(uint32_t *)py = o;
        continue;
      }
      al = py[3];
    }
    const uint32_t y = *py, u = *pu, v = *pv;
    if (!a.clamped) {
      const float ratio = s_ratio[al];
      *py = (uint8_t)premult_byte(y, ratio, a.un); *pu = (uint8_t)premult_byte(u, ratio, a.un); *pv = (uint8_t)premult_byte(v, ratio, a.un);
    } else {
      const uint32_t px = y | (u << 8) | (v << 16) | (al << 24);
      const uint32_t o = a.planar ? (a.un ? pm_clamped_px<1, 1>(px, s_ratio) : pm_clamped_px<0, 1>(px, s_ratio)) : (a.un ? pm_clamped_px<1, 0>(px, s_ratio) : pm_clamped_px<0, 0>(px, s_ratio));
      *py = (uint8_t)o; *pu = (uint8_t)(o >> 8); *pv = (uint8_t)(o >> 16);
    }
  }
}
// test hook: the four tables as the device arithmetic gives them ([unalcy, alcy, unalcuv, alcuv][256][256]), to be compared with lgpu_premult_yuv_tables byte for byte
__global__ __launch_bounds__(256) void k_premult_yuv_tables(uint8_t *out) {
  const uint32_t al = blockIdx.x, j = threadIdx.x;
  const float alpha = __fdiv_rn(255.f, (float)al);
  out[0 * 65536 + al * 256 + j] = (uint8_t)pm_cy<1>(j, al);
  out[1 * 65536 + al * 256 + j] = (uint8_t)pm_cy<0>(j, al);
  out[2 * 65536 + al * 256 + j] = (uint8_t)pm_cuv<1>(j, alpha);
  out[3 * 65536 + al * 256 + j] = (uint8_t)pm_cuv<0>(j, alpha);
}

static inline dim3 row_grid(unsigned items_per_row, int height) {
  unsigned gy = (unsigned)height;
  if (gy > 4096) gy = 4096;
  return dim3(cdiv(items_per_row, kBlock), gy, 1);
}

}  // namespace lgpu

using namespace lgpu;

// frames of one geometry as ONE launch (the frame is the grid's z index); n = 1 is the single-frame entry point.  The vector forms need every frame aligned.
static dim3 with_frames(dim3 g, int n) { g.z = (unsigned)n; return g; }
static int swizzle_n(int op, int alpha_first, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow, int width, int height, const uint8_t *lut8, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  SwzDesc d;
  LGPU_REQUIRE(swz_desc(op, alpha_first, &d), "unknown swizzle op");
  LGPU_REQUIRE(src_d && dst_d && n >= 1 && n <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0, "null frame table, 1..16 frames, or empty geometry");
  LGPU_REQUIRE(irow >= width * d.ibpp && orow >= width * d.obpp, "rowstride smaller than a row");
  FrameTab F = {};
  uintptr_t sb = (uintptr_t)irow, db = (uintptr_t)orow;
  for (int i = 0; i < n; i++) {
    LGPU_REQUIRE(src_d[i] && dst_d[i], "null frame");
    LGPU_REQUIRE(src_d[i] != dst_d[i] || d.ibpp == d.obpp, "in-place needs equal pixel sizes");
    F.src[i] = src_d[i]; F.dst[i] = dst_d[i];
    sb |= (uintptr_t)src_d[i]; db |= (uintptr_t)dst_d[i];
  }
  hipStream_t st = (hipStream_t)stream;
  const Lut8 l = pack_lut(lut8);
  const bool iv = (d.ibpp == 4) ? (sb & 15) == 0 : (sb & 3) == 0;
  const bool ov = (d.obpp == 4) ? (db & 15) == 0 : (db & 3) == 0;
  if (iv && ov) {
    const dim3 grid = with_frames(row_grid((unsigned)(width >> 2) + 1, height), n);
#define LAUNCH(IB, OB)                                                                                                   \
  do {                                                                                                                   \
    if (lut8) hipLaunchKernelGGL((k_swizzle<IB, OB, true>), grid, dim3(kBlock), 0, st, F, irow, orow, width, height, d.sel, d.lutmask, l); \
    else hipLaunchKernelGGL((k_swizzle<IB, OB, false>), grid, dim3(kBlock), 0, st, F, irow, orow, width, height, d.sel, d.lutmask, l);     \
  } while (0)
    if (d.ibpp == 3 && d.obpp == 3) LAUNCH(3, 3);
    else if (d.ibpp == 3) LAUNCH(3, 4);
    else if (d.obpp == 3) LAUNCH(4, 3);
    else LAUNCH(4, 4);
#undef LAUNCH
  } else {
    const dim3 grid = with_frames(row_grid((unsigned)width, height), n);
    if (lut8) hipLaunchKernelGGL((k_swizzle_bytes<true>), grid, dim3(kBlock), 0, st, F, irow, orow, width, height, d.ibpp, d.obpp, d.sel, d.lutmask, l);
    else hipLaunchKernelGGL((k_swizzle_bytes<false>), grid, dim3(kBlock), 0, st, F, irow, orow, width, height, d.ibpp, d.obpp, d.sel, d.lutmask, l);
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_swizzle(int op, int alpha_first, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow,
                            int width, int height, const uint8_t *lut8, void *stream) {
  return swizzle_n(op, alpha_first, &src_d, irow, &dst_d, orow, width, height, lut8, 1, stream);
}
extern "C" int lgpu_swizzle_batch(int op, int alpha_first, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow,
                                  int width, int height, const uint8_t *lut8, int nframes, void *stream) {
  return swizzle_n(op, alpha_first, src_d, irow, dst_d, orow, width, height, lut8, nframes, stream);
}

static int gamma_apply_n(uint8_t *const *pix_d, int rowstride, int x, int y, int width, int height, int psize, int alpha_first, const uint8_t *lut8, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(pix_d && n >= 1 && n <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0 && x >= 0 && y >= 0, "null frame table, 1..16 frames, or empty rectangle");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(rowstride >= (x + width) * psize, "rectangle exceeds the row");
  for (int i = 0; i < n; i++) LGPU_REQUIRE(pix_d[i], "null frame");
  if (!lut8) return LGPU_OK;   // reference: no LUT -> nothing to do (src/colourspace.c:14046)
  const Lut8 l = pack_lut(lut8);
  FrameTab F = {};
  uintptr_t bits = (uintptr_t)rowstride;
  for (int i = 0; i < n; i++) { F.dst[i] = pix_d[i] + (size_t)y * rowstride; bits |= (uintptr_t)F.dst[i]; }
  const int b0 = x * psize, b1 = (x + width) * psize;
  hipStream_t st = (hipStream_t)stream;
  if ((bits & 15) == 0) {
    const uint32_t chanmask = psize == 3 ? 0xFFFFFFFFu : alpha_first ? 0xFFFFFF00u : 0x00FFFFFFu;
    const unsigned chunks = (unsigned)((b1 - (b0 & ~15) + 15) >> 4);
    hipLaunchKernelGGL(k_gamma_apply, with_frames(row_grid(chunks, height), n), dim3(kBlock), 0, st, F, rowstride, b0, b1, height, chanmask, l);
  } else {
    hipLaunchKernelGGL(k_gamma_apply_bytes, with_frames(row_grid((unsigned)(b1 - b0), height), n), dim3(kBlock), 0, st, F, rowstride, b0, b1,
                       height, psize, alpha_first, l);
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_gamma_apply(uint8_t *pix_d, int rowstride, int x, int y, int width, int height, int psize,
                                int alpha_first, const uint8_t *lut8, void *stream) {
  return gamma_apply_n(&pix_d, rowstride, x, y, width, height, psize, alpha_first, lut8, 1, stream);
}
extern "C" int lgpu_gamma_apply_batch(uint8_t *const *pix_d, int rowstride, int x, int y, int width, int height, int psize,
                                      int alpha_first, const uint8_t *lut8, int nframes, void *stream) {
  return gamma_apply_n(pix_d, rowstride, x, y, width, height, psize, alpha_first, lut8, nframes, stream);
}

static int alpha_premult_n(uint8_t *const *pix_d, int rowstride, int width, int height, int alpha_first, int un, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(pix_d && n >= 1 && n <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0 && rowstride >= width * 4, "bad geometry or frame table");
  FrameTab F = {};
  uintptr_t bits = (uintptr_t)rowstride;
  for (int i = 0; i < n; i++) { LGPU_REQUIRE(pix_d[i], "null frame"); F.dst[i] = pix_d[i]; bits |= (uintptr_t)pix_d[i]; }
  LGPU_REQUIRE((bits & 3) == 0, "4-byte pixels must be 4-byte aligned");
  if ((bits & 15) == 0)
    hipLaunchKernelGGL(k_premult<true>, with_frames(row_grid((unsigned)((width + 3) >> 2), height), n), dim3(kBlock), 0, (hipStream_t)stream, F, rowstride, width, height, alpha_first, un);
  else
    hipLaunchKernelGGL(k_premult<false>, with_frames(row_grid((unsigned)width, height), n), dim3(kBlock), 0, (hipStream_t)stream, F, rowstride, width, height, alpha_first, un);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_alpha_premult(uint8_t *pix_d, int rowstride, int width, int height, int alpha_first, int un, void *stream) {
  return alpha_premult_n(&pix_d, rowstride, width, height, alpha_first, un, 1, stream);
}
extern "C" int lgpu_alpha_premult_batch(uint8_t *const *pix_d, int rowstride, int width, int height, int alpha_first, int un, int nframes, void *stream) {
  return alpha_premult_n(pix_d, rowstride, width, height, alpha_first, un, nframes, stream);
}

extern "C" int lgpu_byte_luts(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize, const uint8_t *luts,
                              void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && luts && width > 0 && height > 0, "null frame / tables or empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  Luts4 l = {};
  for (int c = 0; c < psize; c++)
    for (int i = 0; i < 64; i++)
      l.w[c][i] = (uint32_t)luts[c * 256 + 4 * i] | ((uint32_t)luts[c * 256 + 4 * i + 1] << 8) | ((uint32_t)luts[c * 256 + 4 * i + 2] << 16) | ((uint32_t)luts[c * 256 + 4 * i + 3] << 24);
  const dim3 grid = row_grid((unsigned)((width + 3) / 4), height);
  if (psize == 4) hipLaunchKernelGGL(k_byte_luts<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, src_d, irow, dst_d, orow, width, height, l);
  else hipLaunchKernelGGL(k_byte_luts<3>, grid, dim3(kBlock), 0, (hipStream_t)stream, src_d, irow, dst_d, orow, width, height, l);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_alpha_premult_yuva(uint8_t *const planes_d[4], const int rowstrides[4], int width, int height, int palette, int clamped, int un,
                                       void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(planes_d && rowstrides && planes_d[0] && width > 0 && height > 0, "null planes or empty geometry");
  LGPU_REQUIRE(palette == 589 || palette == 545, "palette must be YUVA8888 (589) or YUVA4444P (545)");
  PremultYuvaArgs a = {};
  a.planar = (palette == 545);
  for (int i = 0; i < (a.planar ? 4 : 1); i++) {
    LGPU_REQUIRE(planes_d[i] && rowstrides[i] >= width * (a.planar ? 1 : 4), "null plane or rowstride smaller than a row");
    a.p[i] = planes_d[i]; a.rs[i] = rowstrides[i];
  }
  a.width = width; a.height = height; a.clamped = clamped ? 1 : 0; a.un = un ? 1 : 0;
  a.dword = (!a.planar && (((uintptr_t)planes_d[0] | (uintptr_t)rowstrides[0]) & 3) == 0) ? 1 : 0;
  // clamped packed layers on 16-byte aligned rows: four pixels per lane
  if (a.clamped && !a.planar && (width & 3) == 0 && (((uintptr_t)planes_d[0] | (uintptr_t)rowstrides[0]) & 15) == 0)
    hipLaunchKernelGGL(k_premult_yuva<1>, row_grid((unsigned)width / 4, height), dim3(kBlock), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_premult_yuva<0>, row_grid((unsigned)width, height), dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// test hook: the four clamped premultiply tables as the DEVICE arithmetic of k_premult_yuva evaluates them, 4 x 65,536 bytes in the order of lgpu_premult_yuv_tables
extern "C" int lgpu_debug_premult_yuv_tables_device(uint8_t *out_host) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(out_host, "null output");
  uint8_t *d = nullptr;
  LGPU_HIP(hipMalloc((void **)&d, 4 * 65536));
  hipLaunchKernelGGL(k_premult_yuv_tables, dim3(256), dim3(256), 0, 0, d);
  const hipError_t e = hipMemcpy(out_host, d, 4 * 65536, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error("lgpu_debug_premult_yuv_tables_device: %s", hipGetErrorString(e)); return LGPU_E_HIP; }
  return LGPU_OK;
}
