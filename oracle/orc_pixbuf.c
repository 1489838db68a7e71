/* orc_pixbuf.c -- CPU restatement of gdk_pixbuf_scale_simple() for the three interpolation types LiVES uses.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product.
 *
 * Why it exists: the reference's non-swscale resize body is
 *     resize_layer_full -> layer_to_pixbuf -> lives_pixbuf_scale_simple(pixbuf, width, height, interp)
 *     (/root/reference/src/colourspace.c:15262-15322, call :15295; the wrapper src/widget-helper.c:3796 is gdk_pixbuf_scale_simple),
 * with LIVES_INTERP_BEST / NORMAL / FAST = GDK_INTERP_HYPER / BILINEAR / NEAREST (src/widget-helper-gtk.h:1136-1138); the
 * compositor scales its layers with the same call (lives-plugins/weed-plugins/gdk/compositor.c:263-265).  The pixbuf is 3 channels
 * without alpha for RGB24 / BGR24 / YUV888 and 4 channels WITH alpha for RGBA32 / BGRA32 / YUVA8888
 * (lives_pixbuf_new_from_data_wrapper, colourspace.c:14219-14225: has_alpha = weed_palette_has_alpha(pal)).
 *
 * The arithmetic lives in a third-party dependency that is NOT under /root/reference: gdk-pixbuf (configure.ac pkg-check, version
 * unpinned by LiVES; the build image carries the runtime library 2.42.8, no headers, no source).  This file restates the published
 * algorithm of gdk-pixbuf's pixops scaler (pixops/pixops.c: 16.16 source positions, 16 x 16 sub-sample phases, per-phase two-dimensional
 * integer weight tables that sum to 65536, alpha-weighted colour accumulation when the source has alpha).  It is PINNED, not trusted:
 * tests/test_pixbuf_scale.py compares it byte for byte with the live libgdk_pixbuf-2.0.so.0 wherever that library loads and with the
 * committed fixtures tests/golden/pixbuf_scale.npz (made by oracle/ref/gen_golden_pixbuf.py from the same library).
 */
#include "lives_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PB_SCALE_SHIFT 16
#define PB_SUB_BITS 4
#define PB_SUB 16
#define PB_SUB_MASK 15

/* one filter dimension: n taps, PB_SUB phases, first tap at floor(pos + offset) */
typedef struct { int n; double offset; double *w; } pb_dim;

/* GDK_INTERP_BILINEAR, one dimension: enlarging -> 2-tap linear, centre-aligned; reducing -> box ("tile") filter over the source span */
static int pb_dim_bilinear(pb_dim *d, double scale) {
  int n;
  if (scale > 1.0) { n = 2; d->offset = 0.5 * (1 / scale - 1); }
  else { n = (int)ceil(1.0 + 1.0 / scale); d->offset = 0.0; }
  d->n = n;
  d->w = (double *)malloc(sizeof(double) * PB_SUB * n);
  if (!d->w) return -1;
  double *pw = d->w;
  for (int off = 0; off < PB_SUB; off++) {
    const double x = (double)off / PB_SUB;
    if (scale > 1.0) {
      for (int i = 0; i < n; i++) *pw++ = (((i == 0) ? (1 - x) : x) / scale) * scale;
    } else {
      const double a = x + 1 / scale;
      for (int i = 0; i < n; i++) {
        if (i < x) {
          if (i + 1 > x) *pw++ = (fmin(i + 1, a) - x) * scale;
          else *pw++ = 0;
        } else {
          if (a > i) *pw++ = (fmin(i + 1, a) - i) * scale;
          else *pw++ = 0;
        }
      }
    }
  }
  return 0;
}

/* integral of the linear ramp on [0,1] restricted to [b0,b1] */
static double pb_box_half(double b0, double b1) {
  const double a0 = 0., a1 = 1.;
  double x0, x1;
  if (a0 < b0) {
    if (a1 > b0) { x0 = b0; x1 = fmin(a1, b1); }
    else return 0;
  } else {
    if (b1 > a0) { x0 = a0; x1 = fmin(a1, b1); }
    else return 0;
  }
  return 0.5 * (x1 * x1 - x0 * x0);
}

/* GDK_INTERP_HYPER, one dimension: bilinear reconstruction integrated over the destination pixel's box */
static int pb_dim_hyper(pb_dim *d, double scale) {
  const int n = (int)ceil(1 / scale + 3.0);
  d->n = n;
  d->offset = -1.0;
  d->w = (double *)malloc(sizeof(double) * PB_SUB * n);
  if (!d->w) return -1;
  double *pw = d->w;
  for (int off = 0; off < PB_SUB; off++) {
    const double x = (double)off / PB_SUB;
    const double a = x + 1 / scale;
    for (int i = 0; i < n; i++) {
      double w = pb_box_half(0.5 + i - a, 0.5 + i - x);
      w += pb_box_half(1.5 + x - i, 1.5 + a - i);
      *pw++ = w * scale;
    }
  }
  return 0;
}

/* spread the rounding error of one phase's table so that it sums to exactly 65536 */
static void pb_correct_total(int *w, int n_x, int n_y, int total) {
  const int correction = 65536 - total;
  if (correction == 0) return;
  int remaining = correction;
  for (int d = 1, c = correction; c != 0 && remaining != 0; d++, c = correction / d)
    for (int i = n_x * n_y - 1; i >= 0 && c != 0 && remaining != 0; i--)
      if (w[i] + c >= 0) {
        w[i] += c;
        remaining -= c;
        if ((0 < remaining && remaining < c) || (0 > remaining && remaining > c)) c = remaining;
      }
}

/* the 16 x 16 per-phase two-dimensional tables: [ yphase ][ xphase ][ n_y ][ n_x ] */
static int *pb_filter_table(const pb_dim *fx, const pb_dim *fy) {
  const int n_x = fx->n, n_y = fy->n;
  int *weights = (int *)malloc(sizeof(int) * PB_SUB * PB_SUB * n_x * n_y);
  if (!weights) return NULL;
  for (int io = 0; io < PB_SUB; io++)
    for (int jo = 0; jo < PB_SUB; jo++) {
      int *pw = weights + ((io * PB_SUB) + jo) * n_x * n_y;
      int total = 0;
      for (int i = 0; i < n_y; i++)
        for (int j = 0; j < n_x; j++) {
          const double weight = fx->w[jo * n_x + j] * fy->w[io * n_y + i] * 1.0 * 65536 + 0.5;
          total += (int)weight;
          pw[n_x * i + j] = (int)weight;
        }
      pb_correct_total(pw, n_x, n_y, total);
    }
  return weights;
}

/* exported for the tests and for recovering the device tables: returns n_x, n_y, offsets and the 16*16*n_y*n_x table (caller frees with orc_pixbuf_free) */
int *orc_pixbuf_weights(int interp, int sw, int sh, int dw, int dh, int *n_x, int *n_y, int *xoff, int *yoff) {
  const double scale_x = (double)dw / sw, scale_y = (double)dh / sh;
  pb_dim fx = {0, 0, NULL}, fy = {0, 0, NULL};
  int rc;
  if (interp == 2) rc = pb_dim_bilinear(&fx, scale_x) | pb_dim_bilinear(&fy, scale_y);
  else if (interp == 3) rc = pb_dim_hyper(&fx, scale_x) | pb_dim_hyper(&fy, scale_y);
  else return NULL;
  int *t = rc ? NULL : pb_filter_table(&fx, &fy);
  *n_x = fx.n; *n_y = fy.n;
  *xoff = (int)floor(fx.offset * (1 << PB_SCALE_SHIFT));
  *yoff = (int)floor(fy.offset * (1 << PB_SCALE_SHIFT));
  free(fx.w); free(fy.w);
  return t;
}
void orc_pixbuf_free(void *p) { free(p); }

static inline int pb_clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* GDK_INTERP_NEAREST: sample at the 16.16 position of the destination pixel centre, truncated */
static void pb_nearest(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int ch,
                       double scale_x, double scale_y, int y0, int y1) {
  const int x_step = (int)((1 << PB_SCALE_SHIFT) / scale_x);
  const int y_step = (int)((1 << PB_SCALE_SHIFT) / scale_y);
  (void)dh;
  for (int i = y0; i < y1; i++) {
    int y_pos = (int)(((int64_t)i * y_step + y_step / 2) >> PB_SCALE_SHIFT);
    y_pos = pb_clampi(y_pos, 0, sh - 1);
    const uint8_t *s = src + (size_t)y_pos * irow;
    uint8_t *d = dst + (size_t)(i - y0) * orow;
    int64_t x = x_step / 2;
    for (int j = 0; j < dw; j++, x += x_step) {
      int xp = (int)(x >> PB_SCALE_SHIFT);
      xp = pb_clampi(xp, 0, sw - 1);
      memcpy(d + (size_t)j * ch, s + (size_t)xp * ch, ch);
    }
  }
}

/* would the library pre-shrink first (its two-step scaler for very small ratios)?  Such ratios are declined: rc -2. */
static int pb_two_step(int n_x, int n_y) { return (int64_t)n_x * n_y > 1000; }

/* src: 3 channels (no alpha) or 4 channels (has alpha); dst the same channel count.  interp: 0 NEAREST, 2 BILINEAR, 3 HYPER.
   returns 0, -1 on bad arguments / allocation, -2 for ratios this restatement does not cover. */
int orc_pixbuf_scale(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int channels, int interp) {
  return orc_pixbuf_scale_rows(src, irow, sw, sh, dst, orow, dw, dh, channels, interp, 0, dh);
}

/* destination rows [y0, y1) only; dst points at row y0 (the row-sliced CPU baseline runner uses this) */
int orc_pixbuf_scale_rows(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh, int channels, int interp, int y0, int y1) {
  if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || (channels != 3 && channels != 4) || y0 < 0 || y1 > dh || y0 > y1) return -1;
  if (dw == sw && dh == sh) {            /* gdk_pixbuf_scale_simple returns a plain copy */
    for (int y = y0; y < y1; y++) memcpy(dst + (size_t)(y - y0) * orow, src + (size_t)y * irow, (size_t)sw * channels);
    return 0;
  }
  const double scale_x = (double)dw / sw, scale_y = (double)dh / sh;
  if (interp == 0) { pb_nearest(src, irow, sw, sh, dst, orow, dw, dh, channels, scale_x, scale_y, y0, y1); return 0; }
  if (interp != 2 && interp != 3) return -1;
  int n_x, n_y, xoff, yoff;
  int *table = orc_pixbuf_weights(interp, sw, sh, dw, dh, &n_x, &n_y, &xoff, &yoff);
  if (!table) return -1;
  if (pb_two_step(n_x, n_y)) { free(table); return -2; }
  const int x_step = (int)((1 << PB_SCALE_SHIFT) / scale_x);
  const int y_step = (int)((1 << PB_SCALE_SHIFT) / scale_y);
  if (x_step == 0 || y_step == 0) { free(table); return -2; }
  /* the library's 2 x 2, 3 -> 3 channel line function rounds to nearest; every other case rounds up */
  const unsigned rnd = (n_x == 2 && n_y == 2 && channels == 3) ? 0x8000u : 0xffffu;
  int64_t y = yoff + (int64_t)y0 * y_step;
  for (int i = y0; i < y1; i++, y += y_step) {
    const int y_start = (int)(y >> PB_SCALE_SHIFT);
    const int *run = table + (size_t)((y >> (PB_SCALE_SHIFT - PB_SUB_BITS)) & PB_SUB_MASK) * n_x * n_y * PB_SUB;
    uint8_t *d = dst + (size_t)(i - y0) * orow;
    int64_t x = xoff;
    for (int j = 0; j < dw; j++, x += x_step, d += channels) {
      const int x_start = (int)(x >> PB_SCALE_SHIFT);
      const int *pw = run + (size_t)((x >> (PB_SCALE_SHIFT - PB_SUB_BITS)) & PB_SUB_MASK) * n_x * n_y;
      /* edge pixels go through the library's per-pixel function, which rounds differently from its line function when there is no alpha */
      const int edge = x_start < 0 || x_start + n_x > sw;
      unsigned r = 0, g = 0, b = 0, a = 0;
      for (int ty = 0; ty < n_y; ty++) {
        const uint8_t *line = src + (size_t)pb_clampi(y_start + ty, 0, sh - 1) * irow;
        const int *lw = pw + n_x * ty;
        for (int tx = 0; tx < n_x; tx++) {
          const uint8_t *q = line + (size_t)pb_clampi(x_start + tx, 0, sw - 1) * channels;
          if (channels == 4) {
            const unsigned ta = (unsigned)q[3] * (unsigned)lw[tx];
            r += ta * q[0]; g += ta * q[1]; b += ta * q[2]; a += ta;
          } else if (edge) {
            const unsigned ta = 0xffu * (unsigned)lw[tx];
            r += ta * q[0]; g += ta * q[1]; b += ta * q[2];
          } else {
            r += q[0] * (unsigned)lw[tx]; g += q[1] * (unsigned)lw[tx]; b += q[2] * (unsigned)lw[tx];
          }
        }
      }
      if (channels == 4) {
        /* measured on 2.42.8: the colour is r * (1.0 / a) in double, truncated -- NOT the integer quotient (an exact quotient q comes out q - 1
           whenever 1.0 / a rounds down) */
        if (a) { const double ia = 1.0 / (double)a; d[0] = (uint8_t)((double)r * ia); d[1] = (uint8_t)((double)g * ia); d[2] = (uint8_t)((double)b * ia); d[3] = (uint8_t)(a >> 16); }
        else d[0] = d[1] = d[2] = d[3] = 0;
      } else if (edge) {
        d[0] = (uint8_t)((r + 0xffffffu) >> 24); d[1] = (uint8_t)((g + 0xffffffu) >> 24); d[2] = (uint8_t)((b + 0xffffffu) >> 24);
      } else {
        d[0] = (uint8_t)((r + rnd) >> 16); d[1] = (uint8_t)((g + rnd) >> 16); d[2] = (uint8_t)((b + rnd) >> 16);
      }
    }
  }
  free(table);
  return 0;
}
