// pixbuf.hip -- the "pixbuf" resize arithmetic: what the reference's non-swscale resize body computes.
//
//   resize_layer_full -> layer_to_pixbuf -> lives_pixbuf_scale_simple(pixbuf, width, height, interp)     src/colourspace.c:15262-15322 (call :15295)
//   compositor: gdk_pixbuf_scale_simple(in_pixbuf, owidth, oheight, up_interp / down_interp)            lives-plugins/weed-plugins/gdk/compositor.c:263-265
//   LIVES_INTERP_BEST / NORMAL / FAST = GDK_INTERP_HYPER / BILINEAR / NEAREST                           src/widget-helper-gtk.h:1136-1138
//
// The scaler itself is gdk-pixbuf's (third party, not in the LiVES tree).  Its arithmetic, as pinned byte for byte against the runtime library
// 2.42.8 by tests/test_pixbuf_scale.py and tests/golden/pixbuf_scale.npz:
//   * source position of destination pixel j: x = j * x_step + floor(offset * 65536) in 16.16, x_step = (int)(65536 / scale); first tap at x >> 16,
//     phase (x >> 12) & 15; the same vertically
//   * one two-dimensional integer weight table per (y phase, x phase), n_y x n_x taps summing to exactly 65536:
//       BILINEAR enlarging: 2 taps, centre aligned; BILINEAR reducing: box over the source span, n = ceil(1 + 1 / scale);
//       HYPER: the bilinear kernel integrated over the destination pixel, n = ceil(1 / scale + 3), offset -1
//   * no alpha (3 bytes / pixel): c = (sum w * q + 0xffff) >> 16 (0x8000 when the filter is 2 x 2); pixels whose taps leave the row take the
//     library's per-pixel path, c = (sum 255 * w * q + 0xffffff) >> 24
//   * alpha (4 bytes / pixel): ta = alpha * w; c = (uint8_t)((double)(sum ta * q) * (1.0 / (double)(sum ta))), alpha' = (sum ta) >> 16, all 0 when sum ta == 0
//   * NEAREST: source pixel ((j * step + step / 2) >> 16), clamped
//   * reductions so strong that n_x * n_y > 1000 go through the library's two-step scaler: not covered, LGPU_E_UNSUPPORTED
//
// Kernels: k_pb_window (a 64 x tile_h output tile per workgroup; the clamped source window staged in LDS one dword per pixel, weights by scalar
// loads when the x phase is the same for every pixel -- integer ratios -- and per-lane loads otherwise), k_pb_direct (any ratio, source through
// the caches), k_pb_nearest.  HBM-bound work with an integer MAC per tap and channel; no matrix-core formulation: the per-phase tables are
// two-dimensional (rounded and corrected per phase), not separable.
#include "lgpu_common.h"
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace lgpu {

struct PbArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, sw, sh, dw, dh;
  int x_step, y_step, xoff, yoff;
  int n_x, n_y;
  const int *table;     // device: [16][16][n_y][n_x]
  unsigned rnd;         // 3-byte interior rounding term
  int tile_h, win_w, win_h;
};

__device__ __forceinline__ int pb_clamp(int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; }

template <int CH>
__device__ __forceinline__ uint32_t pb_load_px(const uint8_t *row, int x) {
  if (CH == 4) return *reinterpret_cast<const uint32_t *>(row + 4 * (size_t)x);
  const uint8_t *p = row + 3 * (size_t)x;
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
}

// the accumulators of one destination pixel -> its bytes
template <int CH>
__device__ __forceinline__ void pb_finish(uint8_t *d, unsigned r, unsigned g, unsigned b, unsigned a, bool edge, unsigned rnd) {
  if (CH == 4) {
    uint32_t o = 0;
    if (a) {
      const double ia = 1.0 / (double)a;
      o = (uint32_t)(uint8_t)((double)r * ia) | ((uint32_t)(uint8_t)((double)g * ia) << 8) | ((uint32_t)(uint8_t)((double)b * ia) << 16) | ((a >> 16) << 24);
    }
    *reinterpret_cast<uint32_t *>(d) = o;
  } else if (edge) {
    d[0] = (uint8_t)((r * 255u + 0xffffffu) >> 24); d[1] = (uint8_t)((g * 255u + 0xffffffu) >> 24); d[2] = (uint8_t)((b * 255u + 0xffffffu) >> 24);
  } else {
    d[0] = (uint8_t)((r + rnd) >> 16); d[1] = (uint8_t)((g + rnd) >> 16); d[2] = (uint8_t)((b + rnd) >> 16);
  }
}

template <int CH>
__device__ __forceinline__ void pb_tap(uint32_t q, unsigned w, unsigned &r, unsigned &g, unsigned &b, unsigned &a) {
  if (CH == 4) {
    const unsigned ta = __umul24(q >> 24, w);            // alpha * weight < 2^24
    r = __umul24(ta, q & 0xFF) + r; g = __umul24(ta, (q >> 8) & 0xFF) + g; b = __umul24(ta, (q >> 16) & 0xFF) + b; a += ta;
  } else {
    r = __umul24(w, q & 0xFF) + r; g = __umul24(w, (q >> 8) & 0xFF) + g; b = __umul24(w, (q >> 16) & 0xFF) + b;
  }
}

// 64 x tile_h output pixels per workgroup (4 waves, a wave per output row), clamped source window in LDS
template <int CH, int UNIFORM_X>
__global__ __launch_bounds__(256) void k_pb_window(const PbArgs A) {
  extern __shared__ uint32_t win[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * A.tile_h;
  const int x0 = (int)(((long long)j0 * A.x_step + A.xoff) >> 16), y0 = (int)(((long long)i0 * A.y_step + A.yoff) >> 16);
  for (int wy = wave; wy < A.win_h; wy += 4) {
    const uint8_t *row = A.src + (size_t)pb_clamp(y0 + wy, A.sh - 1) * A.irow;
    uint32_t *wr = win + wy * A.win_w;
    for (int wx = lane; wx < A.win_w; wx += 64) wr[wx] = pb_load_px<CH>(row, pb_clamp(x0 + wx, A.sw - 1));
  }
  __syncthreads();
  const int j = j0 + lane;
  const long long x = (long long)j * A.x_step + A.xoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15;
  const bool edge = xs < 0 || xs + A.n_x > A.sw;
  const int nn = A.n_x * A.n_y;
  for (int r_ = wave; r_ < A.tile_h; r_ += 4) {
    const int i = i0 + r_;
    if (i >= A.dh) break;
    const long long y = (long long)i * A.y_step + A.yoff;
    const int ys = (int)(y >> 16), yph = (int)(y >> 12) & 15;
    const uint32_t *wp = win + (ys - y0) * A.win_w + (xs - x0);
    unsigned r = 0, g = 0, b = 0, a = 0;
    if (j < A.dw) {
      if (UNIFORM_X) {
        const int *wt = A.table + (size_t)__builtin_amdgcn_readfirstlane((yph * 16 + xph) * nn);
        for (int ty = 0; ty < A.n_y; ty++, wp += A.win_w, wt += A.n_x)
          for (int tx = 0; tx < A.n_x; tx++) pb_tap<CH>(wp[tx], (unsigned)wt[tx], r, g, b, a);
      } else {
        const int *wt = A.table + (size_t)(yph * 16 + xph) * nn;
        for (int ty = 0; ty < A.n_y; ty++, wp += A.win_w, wt += A.n_x)
          for (int tx = 0; tx < A.n_x; tx++) pb_tap<CH>(wp[tx], (unsigned)wt[tx], r, g, b, a);
      }
      pb_finish<CH>(A.dst + (size_t)i * A.orow + (size_t)j * CH, r, g, b, a, edge, A.rnd);
    }
  }
}

// any ratio: a thread per destination pixel, taps straight from memory
template <int CH>
__global__ __launch_bounds__(256) void k_pb_direct(const PbArgs A) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= A.dw || i >= A.dh) return;
  const long long x = (long long)j * A.x_step + A.xoff, y = (long long)i * A.y_step + A.yoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15, ys = (int)(y >> 16), yph = (int)(y >> 12) & 15;
  const bool edge = xs < 0 || xs + A.n_x > A.sw;
  const int *wt = A.table + (size_t)(yph * 16 + xph) * A.n_x * A.n_y;
  unsigned r = 0, g = 0, b = 0, a = 0;
  for (int ty = 0; ty < A.n_y; ty++, wt += A.n_x) {
    const uint8_t *row = A.src + (size_t)pb_clamp(ys + ty, A.sh - 1) * A.irow;
    for (int tx = 0; tx < A.n_x; tx++) pb_tap<CH>(pb_load_px<CH>(row, pb_clamp(xs + tx, A.sw - 1)), (unsigned)wt[tx], r, g, b, a);
  }
  pb_finish<CH>(A.dst + (size_t)i * A.orow + (size_t)j * CH, r, g, b, a, edge, A.rnd);
}

template <int CH>
__global__ __launch_bounds__(256) void k_pb_nearest(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int x_step, int y_step) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= dw || i >= dh) return;
  const int ys = pb_clamp((int)(((long long)i * y_step + y_step / 2) >> 16), sh - 1);
  const int xs = pb_clamp((int)(((long long)j * x_step + x_step / 2) >> 16), sw - 1);
  const uint8_t *s = src + (size_t)ys * irow + (size_t)xs * CH;
  uint8_t *d = dst + (size_t)i * orow + (size_t)j * CH;
  if (CH == 4) *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(s);
  else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
}

// ---- host: the per-phase weight tables ----------------------------------------------------------------------------------------------------
struct PbDim { int n; double offset; std::vector<double> w; };   // w[phase * n + tap]

static double ramp_integral(double lo, double hi) {   // integral of t over [lo, hi] intersected with [0, 1]
  if (lo > 0.) { if (lo >= 1.) return 0.; } else { if (hi <= 0.) return 0.; lo = 0.; }
  const double top = hi < 1. ? hi : 1.;
  return 0.5 * (top * top - lo * lo);
}

static PbDim pb_dimension(int interp, double scale) {
  PbDim d;
  const bool hyper = interp == 3, grow = scale > 1.0;
  d.n = hyper ? (int)ceil(1 / scale + 3.0) : grow ? 2 : (int)ceil(1.0 + 1.0 / scale);
  d.offset = hyper ? -1.0 : grow ? 0.5 * (1 / scale - 1) : 0.0;
  d.w.resize((size_t)16 * d.n);
  for (int ph = 0; ph < 16; ph++) {
    const double x = (double)ph / 16, a = x + 1 / scale;
    for (int i = 0; i < d.n; i++) {
      double w;
      if (hyper) w = (ramp_integral(0.5 + i - a, 0.5 + i - x) + ramp_integral(1.5 + x - i, 1.5 + a - i)) * scale;
      else if (grow) w = (((i == 0) ? (1 - x) : x) / scale) * scale;
      else if (i < x) w = (i + 1 > x) ? (fmin(i + 1, a) - x) * scale : 0.;
      else w = (a > i) ? (fmin(i + 1, a) - i) * scale : 0.;
      d.w[(size_t)ph * d.n + i] = w;
    }
  }
  return d;
}

// rounding residue of one phase spread from the last tap backwards until the table sums to 65536
static void pb_fix_sum(int *w, int count, int total) {
  const int correction = 65536 - total;
  int remaining = correction;
  for (int d = 1, c = correction; c != 0 && remaining != 0; d++, c = correction / d)
    for (int i = count - 1; i >= 0 && c != 0 && remaining != 0; i--)
      if (w[i] + c >= 0) {
        w[i] += c;
        remaining -= c;
        if ((0 < remaining && remaining < c) || (0 > remaining && remaining > c)) c = remaining;
      }
}

struct PbTable { int n_x, n_y, xoff, yoff, uniform_x; int *table_d; std::vector<int> host; };
static std::mutex g_pb_mu;
static std::map<std::tuple<int, int, int, int, int, int>, PbTable *> g_pb_tables;   // (device, interp, sw, sh, dw, dh); entries live as long as the library

static int pb_build(int interp, int sw, int sh, int dw, int dh, PbTable *t, bool upload) {
  const PbDim fx = pb_dimension(interp, (double)dw / sw), fy = pb_dimension(interp, (double)dh / sh);
  t->n_x = fx.n; t->n_y = fy.n;
  t->xoff = (int)floor(fx.offset * 65536); t->yoff = (int)floor(fy.offset * 65536);
  t->table_d = nullptr;
  if ((long long)fx.n * fy.n > 1000) return LGPU_E_UNSUPPORTED;       // the library's two-step scaler takes over there
  const int nn = fx.n * fy.n;
  t->host.assign((size_t)256 * nn, 0);
  for (int yp = 0; yp < 16; yp++)
    for (int xp = 0; xp < 16; xp++) {
      int *pw = t->host.data() + (size_t)(yp * 16 + xp) * nn, total = 0;
      for (int i = 0; i < fy.n; i++)
        for (int j = 0; j < fx.n; j++) {
          const double weight = fx.w[(size_t)xp * fx.n + j] * fy.w[(size_t)yp * fy.n + i] * 1.0 * 65536 + 0.5;
          pw[i * fx.n + j] = (int)weight;
          total += (int)weight;
        }
      pb_fix_sum(pw, nn, total);
    }
  if (upload) {
    LGPU_HIP(hipMalloc((void **)&t->table_d, t->host.size() * sizeof(int)));
    LGPU_HIP(hipMemcpy(t->table_d, t->host.data(), t->host.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return LGPU_OK;
}

static int pb_table(int interp, int sw, int sh, int dw, int dh, const PbTable **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_pb_mu);
  const auto key = std::make_tuple(dev, interp, sw, sh, dw, dh);
  auto it = g_pb_tables.find(key);
  if (it == g_pb_tables.end()) {
    PbTable *t = new PbTable();
    const int rc = pb_build(interp, sw, sh, dw, dh, t, true);
    if (rc != LGPU_OK && rc != LGPU_E_UNSUPPORTED) { delete t; return rc; }
    it = g_pb_tables.emplace(key, t).first;
  }
  *out = it->second;
  return it->second->table_d ? LGPU_OK : LGPU_E_UNSUPPORTED;
}

}  // namespace lgpu

using namespace lgpu;

extern "C" int lgpu_pixbuf_weights(int interp, int sw, int sh, int dw, int dh, int *n_x, int *n_y, int *xoff, int *yoff, int32_t *table, size_t table_ints) {
  if ((interp != 2 && interp != 3) || sw < 1 || sh < 1 || dw < 1 || dh < 1 || !n_x || !n_y || !xoff || !yoff) return LGPU_E_BADARG;
  PbTable t;
  const int rc = pb_build(interp, sw, sh, dw, dh, &t, false);
  *n_x = t.n_x; *n_y = t.n_y; *xoff = t.xoff; *yoff = t.yoff;
  if (rc) return rc;
  if (table) {
    if (table_ints < t.host.size()) return LGPU_E_BADARG;
    __builtin_memcpy(table, t.host.data(), t.host.size() * sizeof(int));
  }
  return LGPU_OK;
}

extern "C" int lgpu_pixbuf_scale(const uint8_t *src_d, int irow, int sw, int sh, uint8_t *dst_d, int orow, int dw, int dh, int channels, int interp,
                                 void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && sw > 0 && sh > 0 && dw > 0 && dh > 0, "null frame or empty geometry");
  LGPU_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (no alpha) or 4 (alpha)");
  LGPU_REQUIRE(interp == 0 || interp == 2 || interp == 3, "interp must be 0 (NEAREST), 2 (BILINEAR) or 3 (HYPER)");
  LGPU_REQUIRE(irow >= sw * channels && orow >= dw * channels, "rowstride smaller than a row");
  LGPU_REQUIRE(src_d != dst_d, "scaling cannot run in place");
  LGPU_REQUIRE(sw < 32768 && sh < 32768 && dw < 32768 && dh < 32768, "frame sides must stay below 32768 (16.16 positions)");
  if (channels == 4) LGPU_REQUIRE((((uintptr_t)src_d | (uintptr_t)dst_d | (unsigned)irow | (unsigned)orow) & 3) == 0, "4-byte pixels must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dw == sw && dh == sh) return lgpu_copy_rows(dst_d, orow, src_d, irow, sw * channels, sh, stream);   // gdk_pixbuf_scale_simple: a plain copy
  const double scale_x = (double)dw / sw, scale_y = (double)dh / sh;
  const int x_step = (int)(65536 / scale_x), y_step = (int)(65536 / scale_y);
  if (x_step == 0 || y_step == 0) { set_error("lgpu_pixbuf_scale: enlargement beyond 65536x"); return LGPU_E_UNSUPPORTED; }
  const dim3 grid(cdiv((unsigned)dw, 64), cdiv((unsigned)dh, 4)), block(256);
  if (interp == 0) {
    if (channels == 4) hipLaunchKernelGGL(k_pb_nearest<4>, grid, block, 0, st, src_d, irow, sw, sh, dst_d, orow, dw, dh, x_step, y_step);
    else hipLaunchKernelGGL(k_pb_nearest<3>, grid, block, 0, st, src_d, irow, sw, sh, dst_d, orow, dw, dh, x_step, y_step);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  const PbTable *t;
  if ((rc = pb_table(interp, sw, sh, dw, dh, &t))) {
    if (rc == LGPU_E_UNSUPPORTED) set_error("lgpu_pixbuf_scale: %dx%d -> %dx%d needs %d x %d taps; the library's two-step scaler is not covered", sw, sh, dw, dh, t->n_x, t->n_y);
    return rc;
  }
  PbArgs a;
  a.src = src_d; a.dst = dst_d; a.irow = irow; a.orow = orow; a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
  a.x_step = x_step; a.y_step = y_step; a.xoff = t->xoff; a.yoff = t->yoff; a.n_x = t->n_x; a.n_y = t->n_y; a.table = t->table_d;
  a.rnd = (t->n_x == 2 && t->n_y == 2 && channels == 3) ? 0x8000u : 0xffffu;
  a.win_w = (int)((63LL * x_step + 65535) >> 16) + t->n_x;
  a.tile_h = 0;
  for (int th = 16; th >= 1; th >>= 1) {
    const int wh = (int)(((long long)(th - 1) * y_step + 65535) >> 16) + t->n_y;
    if ((size_t)a.win_w * wh * 4 <= 48 * 1024) { a.tile_h = th; a.win_h = wh; break; }
  }
  const bool uniform = (x_step & 0xFFFF) == 0;
  if (a.tile_h) {
    const dim3 g(cdiv((unsigned)dw, 64), cdiv((unsigned)dh, (unsigned)a.tile_h));
    const size_t lds = (size_t)a.win_w * a.win_h * 4;
    if (channels == 4) {
      if (uniform) hipLaunchKernelGGL((k_pb_window<4, 1>), g, block, lds, st, a);
      else hipLaunchKernelGGL((k_pb_window<4, 0>), g, block, lds, st, a);
    } else {
      if (uniform) hipLaunchKernelGGL((k_pb_window<3, 1>), g, block, lds, st, a);
      else hipLaunchKernelGGL((k_pb_window<3, 0>), g, block, lds, st, a);
    }
  } else {
    if (channels == 4) hipLaunchKernelGGL(k_pb_direct<4>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_pb_direct<3>, grid, block, 0, st, a);
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
