// effects.hip -- the pixel loops of the built-in weed effects (F1..F5) and the letterbox blit (K8).
//
//   chroma blend / luma overlays   lives-plugins/weed-plugins/simple_blend.c:58-194
//   multiply .. burn               lives-plugins/weed-plugins/multi_blends.c:24-168
//   colour key                     lives-plugins/weed-plugins/scripts/colorkey.script <process>
//   mirror x / y / xy              lives-plugins/weed-plugins/mirrors.c:26-122
//   letterbox fill + blit          src/colourspace.c:15343-15567
//
// All of them stream 2 frames in and 1 out with no reuse -> HBM bound.  Two-input effects share one
// kernel skeleton: a lane loads 4 pixels of each input (16 B for 4-byte palettes, 12 B for 3-byte ones,
// normalised to one dword per pixel by v_perm), applies a per-pixel functor and stores 4 pixels.
#include "lgpu_common.h"

namespace lgpu {

// ---- 4-pixel load / store ------------------------------------------------------------------------------
template <int PS>
__device__ __forceinline__ void load4(const uint8_t *row, int g, bool vec, uint32_t p[4]) {
  if (PS == 4) {
    if (vec) { const uint4 v = *reinterpret_cast<const uint4 *>(row + (size_t)g * 16); p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
    else { const uint32_t *w = reinterpret_cast<const uint32_t *>(row + (size_t)g * 16); p[0] = w[0]; p[1] = w[1]; p[2] = w[2]; p[3] = w[3]; }
  } else {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(row + (size_t)g * 12);
    unpack3(w[0], w[1], w[2], p);
  }
}
template <int PS>
__device__ __forceinline__ void store4(uint8_t *row, int g, bool vec, const uint32_t q[4]) {
  if (PS == 4) {
    if (vec) *reinterpret_cast<uint4 *>(row + (size_t)g * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    else { uint32_t *w = reinterpret_cast<uint32_t *>(row + (size_t)g * 16); w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3]; }
  } else {
    uint32_t w0, w1, w2;
    pack3(q, w0, w1, w2);
    uint32_t *w = reinterpret_cast<uint32_t *>(row + (size_t)g * 12);
    w[0] = w0; w[1] = w1; w[2] = w2;
  }
}
template <int PS>
__device__ __forceinline__ uint32_t load1(const uint8_t *row, int x) {
  const uint8_t *p = row + (size_t)x * PS;
  uint32_t v = p[0];
  if (PS >= 3) v |= ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  if (PS == 4) v |= (uint32_t)p[3] << 24;
  return v;
}
template <int PS>
__device__ __forceinline__ void store1(uint8_t *row, int x, uint32_t v) {
  uint8_t *p = row + (size_t)x * PS;
  p[0] = (uint8_t)v;
  if (PS >= 3) { p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); }
  if (PS == 4) p[3] = (uint8_t)(v >> 24);
}

struct Frames2 {
  const uint8_t *s1, *s2;
  uint8_t *dst;
  int r1, r2, ro, width, height;
  int vec;       // rows are 16-byte (PS 4) / 4-byte (PS 3) aligned -> vector path
  int inplace;   // dst == s1
};

// the frames of one launch (blockIdx.z): the planes and the functor -- i.e. the parameters -- of each; geometry, rowstrides and alignment class are shared (Frames2)
template <class F>
struct Pix2Batch {
  const uint8_t *s1[LGPU_FX_MAX_FRAMES], *s2[LGPU_FX_MAX_FRAMES];
  uint8_t *dst[LGPU_FX_MAX_FRAMES];
  F fn[LGPU_FX_MAX_FRAMES];
};
// F(p1, p2, pdst) -> output pixel dword.  pdst is only loaded when F::kNeedsDst && !inplace.
template <int PS, class F>
__global__ __launch_bounds__(kBlock) void k_pixel2(Frames2 f, const Pix2Batch<F> B) {
  __shared__ int32_t s_scratch[768];
  f.s1 = B.s1[blockIdx.z]; f.s2 = B.s2[blockIdx.z]; f.dst = B.dst[blockIdx.z]; f.inplace = f.s1 == f.dst;
  F fn = B.fn[blockIdx.z];
  fn.setup(s_scratch);
  const int groups = f.width >> 2;
  const int g = blockIdx.x * kBlock + threadIdx.x;
  const bool aligned3 = (PS == 4) || f.vec;   // 3-byte rows need 4-byte alignment for dword loads
  for (int y = blockIdx.y; y < f.height; y += gridDim.y) {
    const uint8_t *a = f.s1 + (size_t)y * f.r1, *b = f.s2 + (size_t)y * f.r2;
    uint8_t *d = f.dst + (size_t)y * f.ro;
    if (g < groups && aligned3) {
      uint32_t p1[4], p2[4], pd[4], q[4];
      load4<PS>(a, g, f.vec, p1);
      load4<PS>(b, g, f.vec, p2);
      if (F::kNeedsDst && PS == 4 && !f.inplace) load4<PS>(d, g, f.vec, pd);
#pragma unroll
      for (int k = 0; k < 4; k++) q[k] = fn(p1[k], p2[k], (F::kNeedsDst && PS == 4 && !f.inplace) ? pd[k] : p1[k]);
      store4<PS>(d, g, f.vec, q);
    } else if (g <= groups) {
      const int x0 = aligned3 ? groups * 4 : g * 4, x1 = aligned3 ? f.width : (g * 4 + 4 < f.width ? g * 4 + 4 : f.width);
      if (aligned3 && g != groups) continue;
      for (int x = x0; x < x1; x++) {
        const uint32_t p1 = load1<PS>(a, x), p2 = load1<PS>(b, x);
        const uint32_t pd = (F::kNeedsDst && PS == 4 && !f.inplace) ? load1<PS>(d, x) : p1;
        store1<PS>(d, x, fn(p1, p2, pd));
      }
    }
  }
}

// ---- functors -------------------------------------------------------------------------------------------
// (bf * b + (255 - bf) * a) >> 8 on two bytes at once: bytes sit in 16-bit lanes, products < 2^16
__device__ __forceinline__ uint32_t mix_pairs(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  return ((__umul24(b, bf) + __umul24(a, nbf)) >> 8) & 0x00FF00FFu;
}
__device__ __forceinline__ uint32_t mix4(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  return mix_pairs(a & 0x00FF00FFu, b & 0x00FF00FFu, bf, nbf) | (mix_pairs((a >> 8) & 0x00FF00FFu, (b >> 8) & 0x00FF00FFu, bf, nbf) << 8);
}

// "chroma blend", 3-byte palettes and RGBA/BGRA (simple_blend.c:117-150).  For 4-byte pixels the output
// alpha byte is whatever dst already holds (the reference never writes it).
template <int PS>
struct ChromaBlend {
  static constexpr bool kNeedsDst = (PS == 4);
  uint32_t bf, nbf;
  __device__ __forceinline__ void setup(int32_t *) {}
  __device__ __forceinline__ uint32_t operator()(uint32_t p1, uint32_t p2, uint32_t pd) const {
    if (PS == 3) return mix4(p1, p2, bf, nbf);
    const uint32_t al = p2 >> 24;
    uint32_t r;
    if (al == 255) r = mix4(p1, p2, bf, nbf);
    else {
      // const float alpha = (float)a / 255., inv_alpha = 1. - alpha;  (uint8_t)((float)c * alpha)
      const float alpha = (float)((double)(float)al / 255.), inv = (float)(1. - (double)alpha);
      uint32_t s2 = 0, s1 = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s2 |= ((uint32_t)(int)__fmul_rn((float)((p2 >> (8 * c)) & 0xFF), alpha) & 0xFF) << (8 * c);
        s1 |= ((uint32_t)(int)__fmul_rn((float)((p1 >> (8 * c)) & 0xFF), inv) & 0xFF) << (8 * c);
      }
      r = mix4(s1, s2, bf, nbf);
    }
    return (r & 0x00FFFFFFu) | (pd & 0xFF000000u);
  }
};

// luma of a normalised pixel, calc_luma() (libweed/weed-plugin-utils.c:924-934); ORDER 0 RGB, 1 BGR
struct LumaTab {
  const int32_t *gl;   // device [3][256]
  int32_t *s;          // LDS copy
  __device__ __forceinline__ void stage(int32_t *lds) {
    s = lds;
    for (int i = threadIdx.x; i < 768; i += kBlock) lds[i] = gl[i];
    __syncthreads();
  }
  __device__ __forceinline__ uint32_t luma(uint32_t p, int order) const {
    const uint32_t c0 = p & 0xFF, c1 = (p >> 8) & 0xFF, c2 = (p >> 16) & 0xFF;
    const int32_t v = order ? (s[c2] + s[256 + c1] + s[512 + c0]) : (s[c0] + s[256 + c1] + s[512 + c2]);
    return (uint32_t)(v >> 16) & 0xFF;
  }
};

// luma overlay (1, 4) / underlay (2) / negative overlay (3): copies 3 bytes from layer 2 or layer 1
// (simple_blend.c:151-194; type 4 never reaches its 3x3 branch in the reference and is type 1)
struct LumaBlend {
  static constexpr bool kNeedsDst = true;   // byte 3 of 4-byte pixels is left alone
  LumaTab t;
  int type, order, ps;
  uint32_t bf, neg;
  __device__ __forceinline__ void setup(int32_t *lds) { t.stage(lds); }
  __device__ __forceinline__ uint32_t operator()(uint32_t p1, uint32_t p2, uint32_t pd) const {
    bool take2;
    if (type == 2) take2 = t.luma(p2, order) > neg;
    else if (type == 3) take2 = t.luma(p1, order) > neg;
    else take2 = t.luma(p1, order) < bf;
    const uint32_t c = take2 ? p2 : p1;
    return (c & 0x00FFFFFFu) | (pd & 0xFF000000u);
  }
};

// multi_blends.c:68-162
struct MultiBlend {
  static constexpr bool kNeedsDst = false;
  LumaTab t;
  int type, order;
  uint32_t f, b1, n1, b2, n2;
  __device__ __forceinline__ void setup(int32_t *lds) { t.stage(lds); }
  __device__ __forceinline__ uint32_t operator()(uint32_t pa, uint32_t pb, uint32_t) const {
    uint32_t px = 0;
    if (type == 2 || type == 3) {
      const uint32_t la = t.luma(pa, order), lb = t.luma(pb, order);
      px = (type == 2 ? (la <= lb) : (la >= lb)) ? pa : pb;
    } else {
      const bool screen = (type == 1) || (type == 4 && t.luma(pa, order) >= 128);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int a = (pa >> (8 * c)) & 0xFF, b = (pb >> (8 * c)) & 0xFF;
        int v;
        if (type == 5) v = (b == 255) ? 255 : ((a << 8) / (255 - b) > 255 ? 255 : (a << 8) / (255 - b));
        else if (type == 6) { if (b == 0) v = 0; else { v = 255 - (255 - (a << 8)) / b; v = v < 0 ? 0 : (v & 0xFF); } }
        else if (screen) v = (255 - (((255 - b) * (255 - a)) >> 8)) & 0xFF;
        else v = ((b * a) >> 8) & 0xFF;
        px |= (uint32_t)v << (8 * c);
      }
    }
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint32_t p = (px >> (8 * c)) & 0xFF, a = (pa >> (8 * c)) & 0xFF, b = (pb >> (8 * c)) & 0xFF;
      const uint32_t v = (f < 128) ? ((b1 * p + n1 * a) >> 8) : ((b2 * p + n2 * b) >> 8);
      out |= (v & 0xFF) << (8 * c);
    }
    return out;
  }
};

// colorkey.script <process>: RGB box test on layer 0, then a double-precision lerp, truncated
struct ColorKey {
  static constexpr bool kNeedsDst = false;
  int order;
  int rmin, rmax, gmin, gmax, bmin, bmax;
  double opac, opacx;
  __device__ __forceinline__ void setup(int32_t *) {}
  __device__ __forceinline__ uint32_t operator()(uint32_t p0, uint32_t p1, uint32_t) const {
    const int c0 = p0 & 0xFF, g = (p0 >> 8) & 0xFF, c2 = (p0 >> 16) & 0xFF;
    const int r = order ? c2 : c0, b = order ? c0 : c2;
    if (!(r >= rmin && r <= rmax && g >= gmin && g <= gmax && b >= bmin && b <= bmax)) return p0 & 0x00FFFFFFu;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double a = (double)((p0 >> (8 * c)) & 0xFF), bb = (double)((p1 >> (8 * c)) & 0xFF);
      const double v = __dadd_rn(__dmul_rn(a, opacx), __dmul_rn(bb, opac));   // no FMA contraction: x86-64 baseline has none
      out |= ((uint32_t)(int)v & 0xFF) << (8 * c);
    }
    return out;
  }
};

// "chroma blend" on ARGB32: colour bytes 1..3, alpha test on the byte FOLLOWING them = the next pixel's
// alpha (simple_blend.c:81,:128-146 with start = 1); byte 0 is never written.
__global__ __launch_bounds__(kBlock) void k_chroma_argb(Frames2 f, uint32_t bf, uint32_t nbf) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= f.width) return;
  for (int y = blockIdx.y; y < f.height; y += gridDim.y) {
    const uint32_t *a = reinterpret_cast<const uint32_t *>(f.s1 + (size_t)y * f.r1);
    const uint8_t *brow = f.s2 + (size_t)y * f.r2;
    const uint32_t *b = reinterpret_cast<const uint32_t *>(brow);
    uint32_t *d = reinterpret_cast<uint32_t *>(f.dst + (size_t)y * f.ro);
    const uint32_t p1 = a[x] >> 8, p2 = b[x] >> 8;     // colour bytes -> [c0 c1 c2 0]
    // byte (4x + 4) of layer 2: inside the row's stride, or the next row; past the last row's stride -> opaque
    uint32_t al = 255;
    if (x + 1 < f.width || 4 * (x + 1) < f.r2 || y + 1 < f.height) al = brow[4 * (x + 1)];
    uint32_t r;
    if (al == 255) r = mix4(p1, p2, bf, nbf);
    else {
      const float alpha = (float)((double)(float)al / 255.), inv = (float)(1. - (double)alpha);
      uint32_t s2 = 0, s1 = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s2 |= ((uint32_t)(int)__fmul_rn((float)((p2 >> (8 * c)) & 0xFF), alpha) & 0xFF) << (8 * c);
        s1 |= ((uint32_t)(int)__fmul_rn((float)((p1 >> (8 * c)) & 0xFF), inv) & 0xFF) << (8 * c);
      }
      r = mix4(s1, s2, bf, nbf);
    }
    const uint32_t keep = f.inplace ? a[x] : d[x];
    d[x] = (keep & 0xFFu) | (r << 8);
  }
}

// ---- mirrors ----------------------------------------------------------------------------------------------
// result[y][x] = src[ry(y)][rx(x)], rx(x) = x <= hw ? x : 2hw - x, ry(y) = y >= h - hh + 1 ? h - y : y
// (the in-place result of mirrors.c; its stray writes to pixel `width` / row `height` are not performed)
// 4-byte pixels on 16-byte aligned frames: a lane moves FOUR neighbouring pixels with one 16-byte load and one 16-byte store -- a group wholly inside the
// reflected half reads its four source pixels as one (dword-aligned) run and reverses them.  One pixel per lane (below) left the launch bound by its 138,000
// one-kilobyte workgroups: 16 x 1080p 73.6 us, 0.45 of the roofline.
__global__ __launch_bounds__(kBlock) void k_mirror_v4(const FrameTab T, int irow, int orow, int width, int height, int mx, int my) {
  const uint8_t *src = T.src[blockIdx.z];
  uint8_t *dst = T.dst[blockIdx.z];
  const int inplace = src == dst;
  const int x0 = (blockIdx.x * kBlock + threadIdx.x) * 4;
  if (x0 >= width) return;
  const int hw = width >> 1, hh = height >> 1;
  const bool whole = x0 + 4 <= width;
  const bool plain = whole && (!mx || x0 + 3 <= hw), flipped = whole && mx && x0 > hw;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const int sy = (my && y >= height - hh + 1) ? height - y : y;
    const uint32_t *srow = reinterpret_cast<const uint32_t *>(src + (size_t)sy * irow);
    uint32_t *drow = reinterpret_cast<uint32_t *>(dst + (size_t)y * orow);
    if (plain) {
      if (inplace && sy == y) continue;
      *reinterpret_cast<uint4 *>(drow + x0) = *reinterpret_cast<const uint4 *>(srow + x0);
    } else if (flipped) {
      uint32_t v[4];
      __builtin_memcpy(v, srow + (2 * hw - x0 - 3), 16);          // source pixels 2hw - x0 - 3 .. 2hw - x0: dword aligned, one load
      *reinterpret_cast<uint4 *>(drow + x0) = make_uint4(v[3], v[2], v[1], v[0]);
    } else {
      for (int x = x0; x < x0 + 4 && x < width; x++) {
        const int sx = (mx && x > hw) ? 2 * hw - x : x;
        if (inplace && sx == x && sy == y) continue;
        drow[x] = srow[sx];
      }
    }
  }
}
template <int PS>
__global__ __launch_bounds__(kBlock) void k_mirror(const FrameTab T, int irow, int orow, int width, int height,
                                                    int mx, int my) {
  const uint8_t *src = T.src[blockIdx.z];          // the frame is the grid's z index (lgpu_mirror_batch)
  uint8_t *dst = T.dst[blockIdx.z];
  const int inplace = src == dst;
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= width) return;
  const int hw = width >> 1, hh = height >> 1;
  const int sx = (mx && x > hw) ? 2 * hw - x : x;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const int sy = (my && y >= height - hh + 1) ? height - y : y;
    if (inplace && sx == x && sy == y) continue;
    store1<PS>(dst + (size_t)y * orow, x, load1<PS>(src + (size_t)sy * irow, sx));
  }
}

// ---- letterbox ------------------------------------------------------------------------------------------------
// every canvas pixel is written exactly once: inner rectangle from src, the rest opaque black
template <int PS>
__global__ __launch_bounds__(kBlock) void k_letterbox(const FrameTab T, int irow, int width, int height, int orow,
                                                       int nwidth, int nheight, int ox, int oy, uint32_t black, int vec) {
  const uint8_t *src = T.src[blockIdx.z];
  uint8_t *dst = T.dst[blockIdx.z];
  const int g = blockIdx.x * kBlock + threadIdx.x;   // group of 4 canvas pixels
  const int x0 = g * 4;
  if (x0 >= nwidth) return;
  for (int y = blockIdx.y; y < nheight; y += gridDim.y) {
    uint8_t *d = dst + (size_t)y * orow;
    const int sy = y - oy;
    const bool rowin = sy >= 0 && sy < height;
    const uint8_t *s = src + (size_t)(rowin ? sy : 0) * irow;
    if (PS == 4 && vec && x0 + 4 <= nwidth) {
      uint32_t q[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int sx = x0 + k - ox;
        q[k] = (rowin && sx >= 0 && sx < width) ? reinterpret_cast<const uint32_t *>(s)[sx] : black;
      }
      *reinterpret_cast<uint4 *>(d + (size_t)x0 * 4) = make_uint4(q[0], q[1], q[2], q[3]);
    } else {
      for (int x = x0; x < x0 + 4 && x < nwidth; x++) {
        const int sx = x - ox;
        store1<PS>(d, x, (rowin && sx >= 0 && sx < width) ? load1<PS>(s, sx) : black);
      }
    }
  }
}

// the bars alone: every canvas pixel OUTSIDE the inner rectangle gets the palette's black (letterbox_layer when the inner frame is written in place by the
// scaler, so the resized frame never exists on its own)
template <int PS>
__global__ __launch_bounds__(kBlock) void k_letterbox_bars(uint8_t *dst, int orow, int nwidth, int nheight, int ox, int oy, int width, int height, uint32_t black) {
  const int g = blockIdx.x * kBlock + threadIdx.x;   // group of 4 canvas pixels
  const int x0 = g * 4;
  if (x0 >= nwidth) return;
  for (int y = blockIdx.y; y < nheight; y += gridDim.y) {
    uint8_t *d = dst + (size_t)y * orow;
    const bool rowin = y >= oy && y < oy + height;
    if (rowin && x0 >= ox && x0 + 4 <= ox + width) continue;                        // wholly inside: the scaler's
    for (int x = x0; x < x0 + 4 && x < nwidth; x++)
      if (!(rowin && x >= ox && x < ox + width)) store1<PS>(d, x, black);
  }
}

static inline dim3 row_grid2(unsigned items_per_row, int height) {
  unsigned gy = (unsigned)height;
  if (gy > 4096) gy = 4096;
  return dim3(cdiv(items_per_row, kBlock), gy, 1);
}

static int fill_frames(Frames2 &f, const uint8_t *s1, int r1, const uint8_t *s2, int r2, uint8_t *dst, int ro, int w, int h, int ps) {
  if (!s1 || !s2 || !dst || w <= 0 || h <= 0) { set_error("null frame or empty geometry"); return LGPU_E_BADARG; }
  if (r1 < w * ps || r2 < w * ps || ro < w * ps) { set_error("rowstride smaller than a row"); return LGPU_E_BADARG; }
  f.s1 = s1; f.s2 = s2; f.dst = dst; f.r1 = r1; f.r2 = r2; f.ro = ro; f.width = w; f.height = h;
  const uintptr_t all = (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)dst | (uintptr_t)r1 | (uintptr_t)r2 | (uintptr_t)ro;
  if (ps == 4) {
    if (all & 3) { set_error("4-byte pixels must be 4-byte aligned"); return LGPU_E_BADARG; }
    f.vec = (all & 15) == 0;
  } else f.vec = (all & 3) == 0;
  f.inplace = (s1 == dst);
  return LGPU_OK;
}

}  // namespace lgpu

using namespace lgpu;

// nframes frames of one geometry through k_pixel2: the frame table of a launch, the alignment class over ALL frames
template <int PS, class F>
static int pixel2_launch(const FxFrames &X, int nframes, const F *fns, int r1, int r2, int ro, int width, int height, hipStream_t st) {
  LGPU_REQUIRE(nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames");
  Frames2 f;
  Pix2Batch<F> B = {};
  int vec = 1;
  for (int k = 0; k < nframes; k++) {
    Frames2 one;
    int rc = fill_frames(one, X.in0[k][0], r1, X.in1[k][0], r2, X.out[k][0], ro, width, height, PS);
    if (rc) return rc;
    vec &= one.vec;
    f = one;
    B.s1[k] = one.s1; B.s2[k] = one.s2; B.dst[k] = one.dst; B.fn[k] = fns[k];
  }
  f.vec = vec; f.s1 = f.s2 = nullptr; f.dst = nullptr; f.inplace = 0;        // per frame: the kernel takes them from the table
  dim3 g = row_grid2((unsigned)(width >> 2) + 1, height);
  g.z = (unsigned)nframes;
  hipLaunchKernelGGL((k_pixel2<PS, F>), g, dim3(kBlock), 0, st, f, B);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

int lgpu::blend_chroma_n(const FxFrames &X, int nframes, int irow1, int irow2, int orow, int width, int height, int psize, const int *bf, hipStream_t st) {
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(bf && nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames, a blend amount each");
  if (psize == 4) {
    ChromaBlend<4> fn[LGPU_FX_MAX_FRAMES];
    for (int k = 0; k < nframes; k++) { const uint32_t b = (uint32_t)bf[k] & 0xFF; fn[k] = ChromaBlend<4>{b, 0xFF - b}; }
    return pixel2_launch<4>(X, nframes, fn, irow1, irow2, orow, width, height, st);
  }
  ChromaBlend<3> fn[LGPU_FX_MAX_FRAMES];
  for (int k = 0; k < nframes; k++) { const uint32_t b = (uint32_t)bf[k] & 0xFF; fn[k] = ChromaBlend<3>{b, 0xFF - b}; }
  return pixel2_launch<3>(X, nframes, fn, irow1, irow2, orow, width, height, st);
}

extern "C" int lgpu_blend_chroma(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d,
                                 int orow, int width, int height, int psize, int alpha_first, int bf, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  hipStream_t st = (hipStream_t)stream;
  if (psize == 4 && alpha_first) {
    Frames2 f;
    if ((rc = fill_frames(f, src1_d, irow1, src2_d, irow2, dst_d, orow, width, height, psize))) return rc;
    const uint32_t b = (uint32_t)bf & 0xFF, nb = 0xFF - b;
    hipLaunchKernelGGL(k_chroma_argb, row_grid2((unsigned)width, height), dim3(kBlock), 0, st, f, b, nb);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  FxFrames X = {};
  X.in0[0][0] = src1_d; X.in1[0][0] = src2_d; X.out[0][0] = dst_d;
  return blend_chroma_n(X, 1, irow1, irow2, orow, width, height, psize, &bf, st);
}

int lgpu::blend_luma_n(const FxFrames &X, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int psize, int pal_order, const int *thresh, hipStream_t st) {
  LGPU_REQUIRE(type >= 1 && type <= 4, "type must be 1..4");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(pal_order == 0 || pal_order == 1, "ARGB32 luma blends: use pal_order 0/1 after a swapprepost (reference ARGB path reads across pixels)");
  LGPU_REQUIRE(thresh && nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames, a threshold each");
  LumaBlend fn[LGPU_FX_MAX_FRAMES];
  for (int k = 0; k < nframes; k++) {
    fn[k].t.gl = device_tables()->luma; fn[k].t.s = nullptr;
    fn[k].type = type; fn[k].order = pal_order; fn[k].ps = psize; fn[k].bf = (uint32_t)thresh[k] & 0xFF; fn[k].neg = 0xFF - fn[k].bf;
  }
  if (psize == 4) return pixel2_launch<4>(X, nframes, fn, irow1, irow2, orow, width, height, st);
  return pixel2_launch<3>(X, nframes, fn, irow1, irow2, orow, width, height, st);
}

extern "C" int lgpu_blend_luma(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d,
                               int orow, int width, int height, int psize, int pal_order, int thresh, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  FxFrames X = {};
  X.in0[0][0] = src1_d; X.in1[0][0] = src2_d; X.out[0][0] = dst_d;
  return blend_luma_n(X, 1, type, irow1, irow2, orow, width, height, psize, pal_order, &thresh, (hipStream_t)stream);
}

int lgpu::blend_multi_n(const FxFrames &X, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int is_bgr, const int *bf, hipStream_t st) {
  LGPU_REQUIRE(type >= 0 && type <= 6, "type must be 0..6");
  LGPU_REQUIRE(bf && nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames, a blend amount each");
  MultiBlend fn[LGPU_FX_MAX_FRAMES];
  for (int k = 0; k < nframes; k++) {
    fn[k].t.gl = device_tables()->luma; fn[k].t.s = nullptr;
    fn[k].type = type; fn[k].order = is_bgr ? 1 : 0;
    const uint8_t ff = (uint8_t)bf[k];
    fn[k].f = ff; fn[k].b1 = (uint8_t)(ff * 2); fn[k].n1 = (uint8_t)(255 - ff * 2); fn[k].b2 = (uint8_t)((255 - ff) * 2); fn[k].n2 = (uint8_t)((ff - 128) * 2);
  }
  return pixel2_launch<3>(X, nframes, fn, irow1, irow2, orow, width, height, st);
}

extern "C" int lgpu_blend_multi(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d,
                                int orow, int width, int height, int is_bgr, int bf, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  FxFrames X = {};
  X.in0[0][0] = src1_d; X.in1[0][0] = src2_d; X.out[0][0] = dst_d;
  return blend_multi_n(X, 1, type, irow1, irow2, orow, width, height, is_bgr, &bf, (hipStream_t)stream);
}

static int colorkey_n(const uint8_t *const *src0_d, int irow0, const uint8_t *const *src1_d, int irow1, uint8_t *const *dst_d, int orow,
                      int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src0_d && src1_d && dst_d && n >= 1 && n <= LGPU_FX_MAX_FRAMES, "null frame table or not 1..16 frames");
  ColorKey fn;
  // parameter preparation exactly as the script does it (host side, double)
  double xdelta = delta * 2.;
  delta /= 2.;
  fn.rmin = col_r - (int)(col_r * delta + .5);
  fn.gmin = col_g - (int)(col_g * xdelta + .5);
  fn.bmin = col_b - (int)(col_b * delta + .5);
  xdelta *= 2.;
  delta *= 2.;
  fn.rmax = col_r + (int)((255 - col_r) * delta + .5);
  fn.gmax = col_g + (int)((255 - col_g) * xdelta + .5);
  fn.bmax = col_b + (int)((255 - col_b) * delta + .5);
  fn.order = is_bgr ? 1 : 0; fn.opac = opac; fn.opacx = 1. - opac;
  FxFrames X = {};
  ColorKey fns[LGPU_FX_MAX_FRAMES];
  for (int i = 0; i < n; i++) { X.in0[i][0] = src0_d[i]; X.in1[i][0] = src1_d[i]; X.out[i][0] = dst_d[i]; fns[i] = fn; }
  return pixel2_launch<3>(X, n, fns, irow0, irow1, orow, width, height, (hipStream_t)stream);
}
extern "C" int lgpu_colorkey(const uint8_t *src0_d, int irow0, const uint8_t *src1_d, int irow1, uint8_t *dst_d, int orow,
                             int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b,
                             void *stream) {
  return colorkey_n(&src0_d, irow0, &src1_d, irow1, &dst_d, orow, width, height, is_bgr, delta, opac, col_r, col_g, col_b, 1, stream);
}
extern "C" int lgpu_colorkey_batch(const uint8_t *const *src0_d, int irow0, const uint8_t *const *src1_d, int irow1, uint8_t *const *dst_d, int orow,
                                   int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, int nframes, void *stream) {
  return colorkey_n(src0_d, irow0, src1_d, irow1, dst_d, orow, width, height, is_bgr, delta, opac, col_r, col_g, col_b, nframes, stream);
}

static int mirror_n(int mode, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow, int width, int height, int psize, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (x), 1 (y) or 2 (xy)");
  LGPU_REQUIRE(src_d && dst_d && n >= 1 && n <= LGPU_FX_MAX_FRAMES && width > 0 && height > 0, "null frame table, 1..16 frames, or empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  FrameTab T = {};
  uintptr_t bits = (uintptr_t)irow | (uintptr_t)orow;
  for (int i = 0; i < n; i++) { LGPU_REQUIRE(src_d[i] && dst_d[i], "null frame"); T.src[i] = src_d[i]; T.dst[i] = dst_d[i]; bits |= (uintptr_t)src_d[i] | (uintptr_t)dst_d[i]; }
  const int mx = (mode == 0 || mode == 2), my = (mode == 1 || mode == 2);
  if (psize == 4 && (bits & 15) == 0) {
    dim3 g4 = row_grid2((unsigned)((width + 3) >> 2), height);
    g4.z = (unsigned)n;
    hipLaunchKernelGGL(k_mirror_v4, g4, dim3(kBlock), 0, (hipStream_t)stream, T, irow, orow, width, height, mx, my);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  dim3 grid = row_grid2((unsigned)width, height);
  grid.z = (unsigned)n;
  if (psize == 4) hipLaunchKernelGGL(k_mirror<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, T, irow, orow, width, height, mx, my);
  else hipLaunchKernelGGL(k_mirror<3>, grid, dim3(kBlock), 0, (hipStream_t)stream, T, irow, orow, width, height, mx, my);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_mirror(int mode, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize,
                           void *stream) {
  return mirror_n(mode, &src_d, irow, &dst_d, orow, width, height, psize, 1, stream);
}
extern "C" int lgpu_mirror_batch(int mode, const uint8_t *const *src_d, int irow, uint8_t *const *dst_d, int orow, int width, int height, int psize,
                                 int nframes, void *stream) {
  return mirror_n(mode, src_d, irow, dst_d, orow, width, height, psize, nframes, stream);
}

extern "C" int lgpu_letterbox(const uint8_t *src_d, int irow, int width, int height, uint8_t *dst_d, int orow, int nwidth,
                              int nheight, int psize, const uint8_t black_pixel[4], void *stream) {
  // centred: offs = (outer - inner + 1) >> 1 (src/colourspace.c:15522-15523)
  return lgpu_letterbox_at(src_d, irow, width, height, dst_d, orow, nwidth, nheight, psize, black_pixel, (nwidth - width + 1) >> 1,
                           (nheight - height + 1) >> 1, stream);
}

static int letterbox_n(const uint8_t *const *srcs, int irow, int width, int height, uint8_t *const *dsts, int orow, int nwidth,
                       int nheight, int psize, const uint8_t black_pixel[4], int ox, int oy, int n, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(srcs && dsts && n >= 1 && n <= LGPU_FX_MAX_FRAMES && black_pixel && width > 0 && height > 0, "null frame table, 1..16 frames, or empty geometry");
  FrameTab T = {};
  uintptr_t sbits = (uintptr_t)irow, dbits = (uintptr_t)orow;
  for (int i = 0; i < n; i++) { LGPU_REQUIRE(srcs[i] && dsts[i], "null frame"); T.src[i] = srcs[i]; T.dst[i] = dsts[i]; sbits |= (uintptr_t)srcs[i]; dbits |= (uintptr_t)dsts[i]; }
  LGPU_REQUIRE(psize == 1 || psize == 3 || psize == 4, "psize must be 1, 3 or 4");
  LGPU_REQUIRE(nwidth >= width && nheight >= height, "canvas smaller than the inner frame");
  LGPU_REQUIRE(irow >= width * psize && orow >= nwidth * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(ox >= 0 && oy >= 0 && ox + width <= nwidth && oy + height <= nheight, "inner frame does not fit at that offset");
  uint32_t black = black_pixel[0];
  if (psize >= 3) black |= ((uint32_t)black_pixel[1] << 8) | ((uint32_t)black_pixel[2] << 16);
  if (psize == 4) black |= (uint32_t)black_pixel[3] << 24;
  dim3 grid = row_grid2((unsigned)((nwidth + 3) >> 2), nheight);
  grid.z = (unsigned)n;
  hipStream_t st = (hipStream_t)stream;
  if (psize == 4) {
    const int vec = ((dbits & 15) == 0) && ((sbits & 3) == 0);
    LGPU_REQUIRE(((sbits | dbits) & 3) == 0, "4-byte pixels must be 4-byte aligned");
    hipLaunchKernelGGL(k_letterbox<4>, grid, dim3(kBlock), 0, st, T, irow, width, height, orow, nwidth, nheight, ox, oy, black, vec);
  } else if (psize == 3) {
    hipLaunchKernelGGL(k_letterbox<3>, grid, dim3(kBlock), 0, st, T, irow, width, height, orow, nwidth, nheight, ox, oy, black, 0);
  } else {
    // single-byte planes (Y / U / V / A of the planar palettes): 4 samples per lane
    hipLaunchKernelGGL(k_letterbox<1>, grid, dim3(kBlock), 0, st, T, irow, width, height, orow, nwidth, nheight, ox, oy, black, 0);
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
extern "C" int lgpu_letterbox_at(const uint8_t *src_d, int irow, int width, int height, uint8_t *dst_d, int orow, int nwidth,
                                 int nheight, int psize, const uint8_t black_pixel[4], int ox, int oy, void *stream) {
  return letterbox_n(&src_d, irow, width, height, &dst_d, orow, nwidth, nheight, psize, black_pixel, ox, oy, 1, stream);
}
// nframes frames of one geometry into their canvases, centred as letterbox_layer centres them (src/colourspace.c:15522-15523): one launch
extern "C" int lgpu_letterbox_batch(const uint8_t *const *src_d, int irow, int width, int height, uint8_t *const *dst_d, int orow, int nwidth,
                                    int nheight, int psize, const uint8_t black_pixel[4], int nframes, void *stream) {
  return letterbox_n(src_d, irow, width, height, dst_d, orow, nwidth, nheight, psize, black_pixel, (nwidth - width + 1) >> 1, (nheight - height + 1) >> 1, nframes, stream);
}

extern "C" int lgpu_letterbox_bars(uint8_t *dst_d, int orow, int nwidth, int nheight, int psize, const uint8_t black_pixel[4], int ox, int oy, int width,
                                   int height, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(dst_d && black_pixel && width > 0 && height > 0, "null frame or empty geometry");
  LGPU_REQUIRE(psize == 1 || psize == 3 || psize == 4, "psize must be 1, 3 or 4");
  LGPU_REQUIRE(orow >= nwidth * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(ox >= 0 && oy >= 0 && ox + width <= nwidth && oy + height <= nheight, "inner frame does not fit at that offset");
  LGPU_REQUIRE(psize != 4 || (((uintptr_t)dst_d | (uintptr_t)orow) & 3) == 0, "4-byte pixels must be 4-byte aligned");
  uint32_t black = black_pixel[0];
  if (psize >= 3) black |= ((uint32_t)black_pixel[1] << 8) | ((uint32_t)black_pixel[2] << 16);
  if (psize == 4) black |= (uint32_t)black_pixel[3] << 24;
  const dim3 grid = row_grid2((unsigned)((nwidth + 3) >> 2), nheight);
  hipStream_t st = (hipStream_t)stream;
  if (psize == 4) hipLaunchKernelGGL(k_letterbox_bars<4>, grid, dim3(kBlock), 0, st, dst_d, orow, nwidth, nheight, ox, oy, width, height, black);
  else if (psize == 3) hipLaunchKernelGGL(k_letterbox_bars<3>, grid, dim3(kBlock), 0, st, dst_d, orow, nwidth, nheight, ox, oy, width, height, black);
  else hipLaunchKernelGGL(k_letterbox_bars<1>, grid, dim3(kBlock), 0, st, dst_d, orow, nwidth, nheight, ox, oy, width, height, black);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// compositor fan-in: lives-plugins/weed-plugins/gdk/compositor.c:120-125, :167-189, :288-293
// ---------------------------------------------------------------------------------------------------------------------
namespace lgpu {
struct CompArgs {
  uint8_t *dst;
  int orow, owidth, oheight, nlayers;
  uint32_t bg;                       // background pixel in destination byte order, alpha 0xFF
  lgpu_comp_layer layer[LGPU_COMP_MAX_LAYERS];   // already in paint order
};
// A lane owns one output pixel.  The walk over the layers is a dependent chain (each layer blends over the result of the one before, truncated to a byte, in double
// as paint_pixel does), but the layer PIXELS are not: every covering layer's pixel is requested before the first blend, eight layers at a time, a 4-byte pixel as one
// dword.  Measured and not kept (profiles/r04/composite_probes.txt): several rows per lane with the layer descriptors held in registers (17.6 - 21 us against 15.1),
// the running colour kept as a double between layers (15.5).
template <int PS>
__global__ __launch_bounds__(kBlock) void k_composite(CompArgs a) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= a.owidth) return;
  for (int y = blockIdx.y; y < a.oheight; y += gridDim.y) {
    int c0 = a.bg & 0xFF, c1 = (a.bg >> 8) & 0xFF, c2 = (a.bg >> 16) & 0xFF;
    for (int z0 = 0; z0 < a.nlayers; z0 += 8) {
      uint32_t px[8];
      uint32_t cov = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        px[k] = 0;
        if (z0 + k < a.nlayers) {                                          // uniform
          const lgpu_comp_layer &L = a.layer[z0 + k];
          const int lx = x - L.offs_x, ly = y - L.offs_y;
          if (ly >= 0 && ly < L.height && lx >= 0 && lx < L.width) {      // rows: uniform; columns: per lane
            const uint8_t *s = L.src_d + (size_t)ly * L.irow + (size_t)lx * PS;
            if (PS == 4 && ((L.irow | (int)(uintptr_t)L.src_d) & 3) == 0) px[k] = *reinterpret_cast<const uint32_t *>(s);
            else px[k] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
            cov |= 1u << k;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (cov & (1u << k)) {
          const double al = a.layer[z0 + k].alpha, inv = __dsub_rn(1., al);
          // paint_pixel: dst * invalpha + src * alpha in double, truncated to a byte after every layer
          c0 = (int)(uint8_t)__dadd_rn(__dmul_rn((double)c0, inv), __dmul_rn((double)(px[k] & 0xFF), al));
          c1 = (int)(uint8_t)__dadd_rn(__dmul_rn((double)c1, inv), __dmul_rn((double)((px[k] >> 8) & 0xFF), al));
          c2 = (int)(uint8_t)__dadd_rn(__dmul_rn((double)c2, inv), __dmul_rn((double)((px[k] >> 16) & 0xFF), al));
        }
      }
    }
    uint8_t *d = a.dst + (size_t)y * a.orow + (size_t)x * PS;
    if (PS == 4 && ((a.orow | (int)(uintptr_t)a.dst) & 3) == 0) *reinterpret_cast<uint32_t *>(d) = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16) | 0xFF000000u;
    else {
      d[0] = (uint8_t)c0; d[1] = (uint8_t)c1; d[2] = (uint8_t)c2;
      if (PS == 4) d[3] = 0xFF;
    }
  }
}
}  // namespace lgpu

extern "C" int lgpu_composite(uint8_t *dst_d, int orow, int owidth, int oheight, int psize, int is_bgr, const int bgcol[3],
                              const lgpu_comp_layer *layers, int nlayers, int revz, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(dst_d && bgcol && owidth > 0 && oheight > 0, "null frame or empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4 (RGB24, BGR24, RGBA32, BGRA32)");
  LGPU_REQUIRE(orow >= owidth * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(nlayers >= 0 && nlayers <= LGPU_COMP_MAX_LAYERS && (nlayers == 0 || layers), "0..LGPU_COMP_MAX_LAYERS layers");
  lgpu::CompArgs a;
  a.dst = dst_d; a.orow = orow; a.owidth = owidth; a.oheight = oheight; a.nlayers = 0;
  const int r = is_bgr ? 2 : 0, b = is_bgr ? 0 : 2;
  a.bg = (uint32_t)(bgcol[r] & 0xFF) | ((uint32_t)(bgcol[1] & 0xFF) << 8) | ((uint32_t)(bgcol[b] & 0xFF) << 16) | 0xFF000000u;
  // paint order (compositor.c:181-189): revz == 0 walks the channels from the last to the first
  for (int i = 0; i < nlayers; i++) {
    const lgpu_comp_layer &L = layers[revz ? i : nlayers - 1 - i];
    if (!L.src_d) continue;
    LGPU_REQUIRE(L.width > 0 && L.height > 0 && L.irow >= L.width * psize, "bad layer geometry");
    a.layer[a.nlayers++] = L;
  }
  const dim3 grid(cdiv((unsigned)owidth, kBlock), (unsigned)(oheight < 2048 ? oheight : 2048));
  if (psize == 4) hipLaunchKernelGGL(lgpu::k_composite<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(lgpu::k_composite<3>, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// geometric transitions: lives-plugins/weed-plugins/multi_transitions.c:86-233 (iris rectangle, iris circle, 4 way split)
// ---------------------------------------------------------------------------------------------------------------------
namespace lgpu {
struct TransArgs {
  const uint8_t *src1, *src2;      // filled in the kernel from the frame table (blockIdx.z)
  uint8_t *dst;
  int irow1, irow2, orow, width, height, wb, type;
  int xx, yy, ihwidth, ihheight;          // type 0: rectangle insets (bytes, rows); type 2: quadrant shifts (bytes of rows, bytes)
  float bf, hwidth, hheight, maxradsq;
  float uthr;                             // type 1: the largest float u with sqrt((double)(u / maxradsq)) <= bf (host, bisection over the floats): the test is u > uthr
};
struct TransAmt { int xx, yy; float bf, uthr; };      // what the transition amount decides: per frame of a batch
struct TransAmts { TransAmt v[LGPU_FX_MAX_FRAMES]; };
// where pixel (i, j / PS) comes from (multi_transitions.c:152-205), the reference's own float / double mix
template <int PS>
__device__ __forceinline__ const uint8_t *trans_from(const TransArgs &a, int i, int j) {
  if (a.type == 0)
    return (j < a.xx || j >= a.wb - a.xx || i < a.yy || i >= a.height - a.yy) ? a.src1 + (size_t)a.irow1 * i + j : a.src2 + (size_t)a.irow2 * i + j;
  if (a.type == 1) {
    // sqrt((xxf * xxf + yyf * yyf) / maxradsq) > bf: float terms, double square root (:185-187)
    const float xxf = (float)(i - a.ihheight), yyf = __fdiv_rn((float)(j - a.ihwidth), (float)PS);
    // the float division by maxradsq and the double square root are monotone in u = xxf^2 + yyf^2, so "sqrt(u / maxradsq) > bf" is "u > uthr" for the one float
    // uthr the host finds with the same IEEE operations (exact: every float u falls on the same side); a division and a square root per pixel less
    const float u = __fadd_rn(__fmul_rn(xxf, xxf), __fmul_rn(yyf, yyf));
    return (u > a.uthr) ? a.src1 + (size_t)a.irow1 * i + j : a.src2 + (size_t)a.irow2 * i + j;
  }
  const bool cross = __fdiv_rn(fabsf(__fsub_rn((float)i, a.hheight)), a.hheight) < a.bf ||
                     __fdiv_rn(fabsf(__fsub_rn((float)j, a.hwidth)), a.hwidth) < a.bf || a.bf == 1.f;
  return cross ? a.src2 + (size_t)a.irow2 * i + j
               : a.src1 + (size_t)a.irow1 * i + j + (j > a.ihwidth ? -a.yy : a.yy) + (i > a.ihheight ? -(ptrdiff_t)a.xx : (ptrdiff_t)a.xx);
}
template <int PS>
__device__ __forceinline__ void trans_copy_px(uint8_t *d, const uint8_t *from) {
  if (from == d) return;
  if (PS == 4 && (((uintptr_t)from | (uintptr_t)d) & 3) == 0) { *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(from); return; }      // a 4-byte pixel at an aligned address: one load, one store
#pragma unroll
  for (int k = 0; k < PS; k++) d[k] = from[k];
}
// A thread owns FOUR neighbouring pixels of a row (16 or 12 bytes).  The transitions are selects between two frames (the 4 way split: a shifted copy of the first): when
// the four sources are neighbours too -- everywhere but on the outline of the figure -- they move as one 16-byte (three 4-byte) load and store; the outline's groups
// go pixel by pixel.  In place (the iris classes): a group that stays what it is is not touched.
typedef uint32_t tr_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t tr_u3 __attribute__((ext_vector_type(3), aligned(4)));
template <int PS>
__global__ __launch_bounds__(kBlock) void k_transition(TransArgs a, const FxFrames F, const TransAmts A, int qpr, unsigned total, unsigned half) {
  a.src1 = F.in0[blockIdx.y][0]; a.src2 = F.in1[blockIdx.y][0]; a.dst = F.out[blockIdx.y][0];
  a.xx = A.v[blockIdx.y].xx; a.yy = A.v[blockIdx.y].yy; a.bf = A.v[blockIdx.y].bf; a.uthr = A.v[blockIdx.y].uthr;
  // two groups per thread, half a frame apart: both loads are on their way before the first store
  const unsigned t0 = blockIdx.x * kBlock + threadIdx.x;
  if (t0 >= half) return;
  const uint8_t *from[2][4];
  uint8_t *d[2];
  int npx[2];
  bool run[2];
  tr_u4 v4[2];
  tr_u3 v3[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const unsigned t = t0 + (unsigned)g * half;
    npx[g] = 0; run[g] = false; d[g] = nullptr;
    if (t >= total) continue;
    const int i = (int)(t / (unsigned)qpr), x0 = 4 * (int)(t - (unsigned)i * (unsigned)qpr);
    d[g] = a.dst + (size_t)a.orow * i + (size_t)x0 * PS;
    npx[g] = min(4, a.width - x0);
#pragma unroll
    for (int k = 0; k < 4; k++) from[g][k] = trans_from<PS>(a, i, (x0 + (k < npx[g] ? k : 0)) * PS);
    run[g] = npx[g] == 4 && from[g][1] == from[g][0] + PS && from[g][2] == from[g][0] + 2 * PS && from[g][3] == from[g][0] + 3 * PS &&
             (((uintptr_t)from[g][0] | (uintptr_t)d[g]) & 3) == 0;
    if (run[g] && from[g][0] != d[g]) {
      if (PS == 4) v4[g] = *reinterpret_cast<const tr_u4 *>(from[g][0]);
      else v3[g] = *reinterpret_cast<const tr_u3 *>(from[g][0]);
    }
  }
#pragma unroll
  for (int g = 0; g < 2; g++) {
    if (run[g]) {
      if (from[g][0] == d[g]) continue;
      if (PS == 4) *reinterpret_cast<tr_u4 *>(d[g]) = v4[g];
      else *reinterpret_cast<tr_u3 *>(d[g]) = v3[g];
      continue;
    }
    for (int k = 0; k < npx[g]; k++) trans_copy_px<PS>(d[g] + k * PS, from[g][k]);
  }
}

// slide over (slide_over.c:54-146) as a gather: every output pixel comes from one of the two sources at a shifted position
struct SlideArgs {
  const uint8_t *src1, *src2;
  uint8_t *dst;
  int irow1, irow2, orow, width, height;
  int horiz;          // 1: the bound is a column, 0: a row
  int bound;          // outputs below the bound come from `first`, the rest from the other source
  int first2;         // the part below the bound shows src2 (directions 2 / 4)
  int shift_lo, shift_hi;   // coordinate shift applied below / at-or-above the bound
};
template <int PS>
__global__ __launch_bounds__(kBlock) void k_slide_over(SlideArgs a) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= a.width) return;
  for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    const int c = a.horiz ? x : y;
    const bool lo = c < a.bound;
    const int sc = lo ? c + a.shift_lo : c - a.bound + a.shift_hi;
    const int sx = a.horiz ? sc : x, sy = a.horiz ? y : sc;
    const bool from2 = lo ? a.first2 : !a.first2;
    const uint8_t *from = from2 ? a.src2 + (size_t)a.irow2 * sy + (size_t)sx * PS : a.src1 + (size_t)a.irow1 * sy + (size_t)sx * PS;
    uint8_t *d = a.dst + (size_t)a.orow * y + (size_t)x * PS;
    if (PS == 4 && ((reinterpret_cast<uintptr_t>(from) | reinterpret_cast<uintptr_t>(d)) & 3) == 0) {      // a 4-byte pixel: one load, one store
      *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(from);
      continue;
    }
#pragma unroll
    for (int k = 0; k < PS; k++) d[k] = from[k];
  }
}

// triple split (layout_blends.c:24-113): per-pixel select between src2 (outer bands), src1 (middle band) and the border colour
struct TsplitArgs {
  const uint8_t *src1, *src2;
  uint8_t *dst;
  int irow1, irow2, orow, width, height, inplace;
  double c_lo_out, c_hi_out, c_lo_in, c_hi_in;   // width_bytes * (xstart - bw), * (xend + bw), * (xstart + bw), * (xend - bw)
  int tbs, tbe, bbs, bbe;
  uint8_t bc[4];
};
__global__ __launch_bounds__(kBlock) void k_triple_split(TsplitArgs a) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= a.width) return;
  const int j = 3 * x;
  const bool col_out = (double)j < a.c_lo_out || (double)j >= a.c_hi_out;
  const bool col_in = (double)j > a.c_lo_in && (double)j < a.c_hi_in;
  for (int r = blockIdx.y; r < a.height; r += gridDim.y) {
    uint8_t *d = a.dst + (size_t)r * a.orow + j;
    if (col_out && (r <= a.tbs || r >= a.bbe)) {
      const uint8_t *s = a.src2 + (size_t)r * a.irow2 + j;
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else if (col_in || (r > a.tbe && r < a.bbs)) {
      if (!a.inplace) { const uint8_t *s = a.src1 + (size_t)r * a.irow1 + j; d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
    } else {
      d[0] = a.bc[0]; d[1] = a.bc[1]; d[2] = a.bc[2];
    }
  }
}

// dissolve (multi_transitions.c:208-212): src2 where the per-pixel mask value is below the transition amount
template <int PS>
__global__ __launch_bounds__(kBlock) void k_dissolve(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width,
                                                     int height, const float *mask, float bf, int inplace) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= width) return;
  for (int i = blockIdx.y; i < height; i += gridDim.y) {
    const bool two = mask[(size_t)i * width + x] < bf;
    if (!two && inplace) continue;
    const uint8_t *s = two ? src2 + (size_t)i * irow2 + (size_t)x * PS : src1 + (size_t)i * irow1 + (size_t)x * PS;
    uint8_t *d = dst + (size_t)i * orow + (size_t)x * PS;
    if (PS == 4 && (((uintptr_t)s | (uintptr_t)d) & 3) == 0) { *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(s); continue; }      // a 4-byte pixel: one load, one store
#pragma unroll
    for (int k = 0; k < PS; k++) d[k] = s[k];
  }
}
}  // namespace lgpu

int lgpu::transition_n(const FxFrames &F, int nframes, int type, int irow1, int irow2, int orow, int width, int height, int psize, const double *amounts, hipStream_t st) {
  LGPU_REQUIRE(type >= 0 && type <= 2, "type must be 0 (iris rectangle), 1 (iris circle) or 2 (4 way split)");
  LGPU_REQUIRE(width > 0 && height > 0, "empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow1 >= width * psize && irow2 >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(amounts && nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames, an amount each");
  for (int f = 0; f < nframes; f++) {
    LGPU_REQUIRE(F.in0[f][0] && F.in1[f][0] && F.out[f][0], "null frame");
    LGPU_REQUIRE(type != 2 || F.in0[f][0] != F.out[f][0], "4 way split is not in place (multi_transitions.c:283)");
  }
  lgpu::TransArgs a;
  lgpu::TransAmts A = {};
  a.src1 = nullptr; a.src2 = nullptr; a.dst = nullptr; a.irow1 = irow1; a.irow2 = irow2; a.orow = orow;
  a.width = width; a.height = height; a.type = type;
  // the reference's own float / double mix (:129-150)
  float hwidth = (float)width * 0.5f;
  const float hheight = (float)height * 0.5f;
  a.maxradsq = (type == 1) ? ((hheight * hheight) + (hwidth * hwidth)) : 0.f;
  a.wb = width * psize;
  hwidth = (float)a.wb * 0.5f;
  a.hwidth = hwidth; a.hheight = hheight; a.ihwidth = a.wb >> 1; a.ihheight = height >> 1;
  a.bf = 0.f; a.uthr = 0.f; a.xx = a.yy = 0;                       // per frame: the kernel takes them from A
  for (int f = 0; f < nframes; f++) {
    lgpu::TransAmt &m = A.v[f];
    m.bf = (float)amounts[f];
    m.uthr = 0.f;
    if (type == 1) {
      auto pred = [&](uint32_t bits) { float u; __builtin_memcpy(&u, &bits, 4); const volatile float q = u / a.maxradsq; return sqrt((double)q) > (double)m.bf; };
      if (pred(0u)) m.uthr = -1.f;                                 // true for every u >= 0
      else {
        uint32_t lo = 0u, hi = 0x7F7FFFFFu;                        // pred(lo) false; the largest finite float
        if (!pred(hi)) lo = hi;
        else while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (pred(mid)) hi = mid; else lo = mid; }
        __builtin_memcpy(&m.uthr, &lo, 4);
      }
    }
    const float bfneg = 1.f - m.bf;
    m.xx = m.yy = 0;
    if (type == 0) { m.xx = (int)((int)hwidth * bfneg + .5); m.yy = (int)((int)hheight * bfneg + .5); }
    else if (type == 2) { m.xx = (int)(hheight * m.bf + .5) * irow1; m.yy = (int)(hwidth / (float)psize * m.bf + .5) * psize; }
  }
  const int qpr = (width + 3) / 4;                                  // groups of four pixels per row
  LGPU_REQUIRE((long long)qpr * height < (1ll << 31), "frame too large");
  const unsigned total = (unsigned)qpr * (unsigned)height;
  const unsigned half = (total + 1) / 2;
  const dim3 grid(cdiv(half, kBlock), (unsigned)nframes);
  if (psize == 4) hipLaunchKernelGGL(lgpu::k_transition<4>, grid, dim3(kBlock), 0, st, a, F, A, qpr, total, half);
  else hipLaunchKernelGGL(lgpu::k_transition<3>, grid, dim3(kBlock), 0, st, a, F, A, qpr, total, half);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_transition(int type, const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow,
                               int width, int height, int psize, double amount, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  FxFrames F = {};
  F.in0[0][0] = src1_d; F.in1[0][0] = src2_d; F.out[0][0] = dst_d;
  return transition_n(F, 1, type, irow1, irow2, orow, width, height, psize, &amount, (hipStream_t)stream);
}

// One launch for the instances of ONE filter on the live tracks of a tick (src/effects-weed.c:1850-2425 runs weed_apply_instance once per track): the frames share
// geometry, rowstrides and parameters; only the planes differ.  Results are those of nframes single calls.
extern "C" int lgpu_fx_batch(const lgpu_fx_params *p, const lgpu_fx_frame *frames, int nframes, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(p && frames && nframes >= 1 && nframes <= LGPU_FX_MAX_FRAMES, "1..LGPU_FX_MAX_FRAMES frames");
  FxFrames F = {};
  for (int f = 0; f < nframes; f++)
    for (int k = 0; k < 4; k++) { F.in0[f][k] = frames[f].in0[k]; F.in1[f][k] = frames[f].in1[k]; F.out[f][k] = frames[f].out[k]; }
  hipStream_t st = (hipStream_t)stream;
  LGPU_REQUIRE(!p->frame_dp0 || p->op == LGPU_FX_TRANSITION || p->op == LGPU_FX_BLEND_CHROMA || p->op == LGPU_FX_BLEND_LUMA || p->op == LGPU_FX_BLEND_MULTI,
               "frame_dp0 (a value per frame) is taken by the transitions and the blends only");
  switch (p->op) {
  case LGPU_FX_SOFTLIGHT: return softlight_n(F, nframes, p->irow0, p->orow, p->width, p->height, p->palette, p->ip[0], st);
  case LGPU_FX_TRANSITION: {
    double am[LGPU_FX_MAX_FRAMES];
    for (int f = 0; f < nframes; f++) am[f] = p->frame_dp0 ? p->frame_dp0[f] : p->dp[0];
    return transition_n(F, nframes, p->ip[0], p->irow0[0], p->irow1[0], p->orow[0], p->width, p->height, p->ip[1], am, st);
  }
  case LGPU_FX_YUV411_TO_RGB: return yuv411_to_rgb_n(F, nframes, p->width, p->height, p->orow[0], p->ip[0], p->ip[1], p->ip[2], st);
  case LGPU_FX_GAUSS5_COLORKEY: return gauss5_colorkey_n(F, nframes, p->irow0[0], p->irow1[0], p->orow[0], p->width, p->height, p->ip[0], p->ip[1], p->dp[0], p->dp[1], p->ip[2] & 0xFF,
                                                         (p->ip[2] >> 8) & 0xFF, (p->ip[2] >> 16) & 0xFF, st);
  case LGPU_FX_BLEND_CHROMA: case LGPU_FX_BLEND_LUMA: case LGPU_FX_BLEND_MULTI: {
    int v[LGPU_FX_MAX_FRAMES];
    for (int f = 0; f < nframes; f++) v[f] = (int)(p->frame_dp0 ? p->frame_dp0[f] : p->dp[0]);
    if (p->op == LGPU_FX_BLEND_CHROMA) {
      LGPU_REQUIRE(!p->ip[1], "ARGB32 (alpha first) is served by lgpu_blend_chroma frame by frame");
      return blend_chroma_n(F, nframes, p->irow0[0], p->irow1[0], p->orow[0], p->width, p->height, p->ip[0], v, st);
    }
    if (p->op == LGPU_FX_BLEND_LUMA) return blend_luma_n(F, nframes, p->ip[0], p->irow0[0], p->irow1[0], p->orow[0], p->width, p->height, p->ip[1], p->ip[2], v, st);
    return blend_multi_n(F, nframes, p->ip[0], p->irow0[0], p->irow1[0], p->orow[0], p->width, p->height, p->ip[1], v, st);
  }
  default: set_error("lgpu_fx_batch: unknown op %d", p->op); return LGPU_E_BADARG;
  }
}

extern "C" int lgpu_slide_over(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow,
                               int width, int height, int psize, int amount, int direction, int slide_lower, int slide_upper, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(direction >= 1 && direction <= 4, "direction must be 1..4 (slide_over.c:40-51; the host resolves 0 = random)");
  LGPU_REQUIRE(src1_d && src2_d && dst_d && width > 0 && height > 0, "null frame or empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow1 >= width * psize && irow2 >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(amount >= 0 && amount <= 255, "amount is 0..255");
  LGPU_REQUIRE(dst_d != src1_d && dst_d != src2_d, "slide over is not in place");
  const int mvl = slide_lower ? 1 : 0, mvu = slide_upper ? 1 : 0;
  lgpu::SlideArgs a;
  a.src1 = src1_d; a.src2 = src2_d; a.dst = dst_d; a.irow1 = irow1; a.irow2 = irow2; a.orow = orow; a.width = width; a.height = height;
  // the reference's float / double mix for the dividing line (:93, :109, :125, :136)
  switch (direction) {
  case 3: a.horiz = 0; a.bound = (int)((float)height * (1. - amount / 255.)); a.first2 = 0;
    a.shift_lo = mvu ? height - a.bound : 0; a.shift_hi = mvl ? 0 : a.bound; break;
  case 4: a.horiz = 0; a.bound = (int)((float)height * (amount / 255.)); a.first2 = 1;
    a.shift_lo = mvl ? height - a.bound : 0; a.shift_hi = mvu ? 0 : a.bound; break;
  case 1: a.horiz = 1; a.bound = (int)((float)width * (1. - amount / 255.)); a.first2 = 0;
    a.shift_lo = mvu ? width - a.bound : 0; a.shift_hi = mvl ? 0 : a.bound; break;
  default: a.horiz = 1; a.bound = (int)((float)width * (amount / 255.)); a.first2 = 1;
    a.shift_lo = mvl ? width - a.bound : 0; a.shift_hi = mvu ? 0 : a.bound; break;
  }
  const dim3 grid(cdiv((unsigned)width, kBlock), (unsigned)(height < 2048 ? height : 2048));
  if (psize == 4) hipLaunchKernelGGL(lgpu::k_slide_over<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(lgpu::k_slide_over<3>, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_triple_split(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow, int width, int height,
                                 int is_bgr, double start, int symmetrical, double end, int split_rows, double border_width, const int *border_rgb,
                                 void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src1_d && src2_d && dst_d && border_rgb && width > 0 && height > 0, "null frame or empty geometry");
  LGPU_REQUIRE(irow1 >= width * 3 && irow2 >= width * 3 && orow >= width * 3, "rowstride smaller than a row");
  LGPU_REQUIRE(dst_d != src2_d, "only the first input may be the output (layout_blends.c:36)");
  lgpu::TsplitArgs a;
  a.src1 = src1_d; a.src2 = src2_d; a.dst = dst_d; a.irow1 = irow1; a.irow2 = irow2; a.orow = orow; a.width = width; a.height = height;
  a.inplace = (src1_d == dst_d);
  // the reference's own double arithmetic (:56-83); the byte column j is compared as an int promoted to double
  double xstart = start, xend = end;
  const double bw = border_width;
  if (symmetrical) { xstart /= 2.; xend = 1. - xstart; }
  if (xstart > xend) { const double t = xend; xend = xstart; xstart = t; }
  a.bc[0] = (uint8_t)border_rgb[is_bgr ? 2 : 0]; a.bc[1] = (uint8_t)border_rgb[1]; a.bc[2] = (uint8_t)border_rgb[is_bgr ? 0 : 2]; a.bc[3] = 0;
  a.tbs = a.tbe = a.bbs = a.bbe = height;
  if (split_rows) {
    a.tbs = (int)(height * (xstart - bw) + .5); a.tbe = (int)(height * (xstart + bw) + .5);
    a.bbs = (int)(height * (xend - bw) + .5); a.bbe = (int)(height * (xend + bw) + .5);
    xstart = xend = -bw;
  }
  const int wb = width * 3;
  a.c_lo_out = wb * (xstart - bw); a.c_hi_out = wb * (xend + bw); a.c_lo_in = wb * (xstart + bw); a.c_hi_in = wb * (xend - bw);
  const dim3 grid(cdiv((unsigned)width, kBlock), (unsigned)(height < 2048 ? height : 2048));
  hipLaunchKernelGGL(lgpu::k_triple_split, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_dissolve(const uint8_t *src1_d, int irow1, const uint8_t *src2_d, int irow2, uint8_t *dst_d, int orow, int width, int height,
                             int psize, const float *mask_d, double amount, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src1_d && src2_d && dst_d && mask_d && width > 0 && height > 0, "null frame / mask or empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow1 >= width * psize && irow2 >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  const dim3 grid(cdiv((unsigned)width, kBlock), (unsigned)(height < 2048 ? height : 2048));
  const int inplace = (src1_d == dst_d);
  if (psize == 4) hipLaunchKernelGGL(lgpu::k_dissolve<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, src1_d, irow1, src2_d, irow2, dst_d, orow, width, height, mask_d, (float)amount, inplace);
  else hipLaunchKernelGGL(lgpu::k_dissolve<3>, grid, dim3(kBlock), 0, (hipStream_t)stream, src1_d, irow1, src2_d, irow2, dst_d, orow, width, height, mask_d, (float)amount, inplace);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
