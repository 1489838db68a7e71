/* lives_oracle.c -- CPU restatement of the LiVES per-frame hot path (see lives_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product.
 *
 * Written from the behaviour of the reference (file:line cited per function), not copied from it.
 * Integer / byte work is meant to be bit-identical to the reference; the places where the reference
 * invokes undefined behaviour are listed in DESIGN.md ("Reference quirks") with the decision taken.
 */
#include "lives_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * conversion tables                                      reference: src/colourspace.c:851-1105
 * constants                                              reference: src/colourspace.h:50-63,86-131
 * ---------------------------------------------------------------------------------------------- */
#define SCALE 65793.            /* (2^24 - 1) / (2^8 - 1) */
#define Y_LO 16.
#define Y_HI 235.
#define UV_HI 240.
#define BIAS 128.

static inline int rnd_half_away(double n) { return n >= 0. ? (int)(n + 0.5) : (int)(n - 0.5); }   /* src/maths.h:118 */

static int32_t T_r2y[4][9][256], T_y2r[4][5][256];
static int tables_ready = 0;

static void build_tables(void) {
  /* clamp factors are parenthesised constants in the reference (colourspace.h:117-118) */
  const double cfy = (Y_HI - Y_LO) / (255. - 0.), cfuv = (UV_HI - Y_LO) / (255. - 0.);
  for (int which = 0; which < 4; which++) {
    const int unclamped = which & 1, hd = which & 2;
    const double kr = hd ? 0.2126 : 0.299, kb = hd ? 0.0722 : 0.114;
    int32_t (*f)[256] = T_r2y[which], (*b)[256] = T_y2r[which];
    for (int i = 0; i < 256; i++) {
      const double x = (double)i;
      double fac;
      if (!unclamped) {
        f[0][i] = rnd_half_away(kr * x * cfy * SCALE);
        f[1][i] = rnd_half_away((1. - kr - kb) * x * cfy * SCALE);
        f[2][i] = rnd_half_away((kb * x * cfy + Y_LO) * SCALE);
        fac = .5 / (1. - kb);
        f[3][i] = rnd_half_away(-fac * kr * x * cfuv * SCALE);
        f[4][i] = rnd_half_away(-fac * (1. - kb - kr) * x * cfuv * SCALE);
        f[5][i] = rnd_half_away((0.5 * x * cfuv + BIAS) * SCALE);
        fac = .5 / (1. - kr);
        f[6][i] = rnd_half_away((0.5 * x * cfuv + BIAS) * SCALE);
        f[7][i] = rnd_half_away(-fac * (1. - kb - kr) * x * cfuv * SCALE);
        f[8][i] = rnd_half_away(-fac * kb * x * cfuv * SCALE);
      } else {
        f[0][i] = rnd_half_away(kr * x * SCALE);
        f[1][i] = rnd_half_away((1. - kr - kb) * x * SCALE);
        f[2][i] = rnd_half_away(kb * x * SCALE);
        fac = .5 / (1. - kb);
        f[3][i] = rnd_half_away(-fac * kr * x * SCALE);
        f[4][i] = rnd_half_away(-fac * (1. - kb - kr) * x * SCALE);
        f[5][i] = rnd_half_away((0.5 * x + BIAS) * SCALE);
        fac = .5 / (1. - kr);
        f[6][i] = rnd_half_away((0.5 * x + BIAS) * SCALE);
        f[7][i] = rnd_half_away(-fac * (1. - kb - kr) * x * SCALE);
        f[8][i] = rnd_half_away(-fac * kb * x * SCALE);
      }
    }
    /* YUV -> RGB.  The G_Cb divisor is (1 + Kb + Kr) for YCbCr but (1 + Kb + Kb) for BT.709
       (src/colourspace.c:1017 vs :1061) -- replicated. */
    const double gcb_div = hd ? (1. + kb + kb) : (1. + kb + kr);
    for (int i = 0; i < 256; i++) {
      const double x = (double)i;
      if (unclamped) {
        b[0][i] = (int32_t)(i * SCALE);
        b[1][i] = rnd_half_away(2. * (1. - kr) * (x - BIAS) * SCALE);
        b[2][i] = rnd_half_away(-.5 / gcb_div * (x - BIAS) * SCALE);
        b[3][i] = rnd_half_away(-.5 / (1. - kr) * (x - BIAS) * SCALE);
        b[4][i] = rnd_half_away(2. * (1. - kb) * (x - BIAS) * SCALE);
        continue;
      }
      /* luma: 0 up to 16, ramp to 234, saturate from 235 */
      if (i <= 16) b[0][i] = 0;
      else if (i < 235) b[0][i] = rnd_half_away((x - Y_LO) / (Y_HI - Y_LO) * 255. * SCALE);
      else b[0][i] = (int32_t)(255 * SCALE);
      if (i <= 16) b[1][i] = b[2][i] = b[3][i] = b[4][i] = 0;
      else {
        double c;
        if (i < 240) c = ((x - Y_LO) / (UV_HI - Y_LO) * 255.) - BIAS;
        else c = hd ? (255. - BIAS) : (((UV_HI - Y_LO) / (UV_HI - Y_LO) * 255.) - BIAS);
        b[1][i] = rnd_half_away(2. * (1. - kr) * c * SCALE);
        b[2][i] = rnd_half_away(-.5 / gcb_div * c * SCALE);
        b[3][i] = rnd_half_away(-.5 / (1. - kr) * c * SCALE);
        b[4][i] = rnd_half_away(2. * (1. - kb) * c * SCALE);
      }
    }
  }
  tables_ready = 1;
}

void orc_tables(int which, int32_t *rgb2yuv, int32_t *yuv2rgb) {
  if (!tables_ready) build_tables();
  if (rgb2yuv) memcpy(rgb2yuv, T_r2y[which & 3], sizeof(T_r2y[0]));
  if (yuv2rgb) memcpy(yuv2rgb, T_y2r[which & 3], sizeof(T_y2r[0]));
}

/* ------------------------------------------------------------------------------------------------
 * gamma LUT builder                                      reference: src/colourspace.c:655-736
 * gamma constants                                        reference: src/colourspace.h:152-185
 * Deterministic quirks kept on purpose: X->LINEAR is the identity; the source gamma is
 * forgotten after the first table entry; the encode branch tops out at 246.
 * ---------------------------------------------------------------------------------------------- */
/* ids: libweed/weed-palettes.h:180-183, src/colourspace.h:27 */
enum { G_UNKNOWN = 0, G_LINEAR = -1, G_SRGB = 1, G_BT709 = 2, G_MONITOR = 1024 };
typedef struct { float offs, lin, thresh, pf; } gconst_t;

static gconst_t gconst_for(int gtype) {
  gconst_t g;
  if (gtype == G_BT709) { g.lin = 4.5; g.thresh = 0.018; g.pf = 1. / .45; }
  else { g.lin = 12.92; g.thresh = 0.04045; g.pf = 2.4; }     /* sRGB; also used for unknown ids (index 0) */
  g.offs = (powf((g.thresh / g.lin), (1. / g.pf)) - g.thresh) / (1. - (powf((g.thresh / g.lin), (1. / g.pf))));
  return g;
}

static inline uint8_t clamp_int_0_255(int n) { return n < 0 ? 0 : n > 255 ? 255 : (uint8_t)n; }

int orc_gamma_lut8(double fileg, int gamma_from, int gamma_to, double screen_gamma, uint8_t *lut) {
  float inv_gamma = 0., a, x = 0.;
  if (fileg == 1.0 && (gamma_to == gamma_from || gamma_to == G_UNKNOWN || gamma_from == G_UNKNOWN)) return 0;
  if (gamma_to == G_MONITOR) inv_gamma = 1. / (float)screen_gamma;
  lut[0] = 0;
  for (int i = 1; i < 256; ++i) {
    x = a = (float)i / 255.;
    if (fileg != 1.0) x = powf(a, fileg);
    if (gamma_from == G_MONITOR) { x = powf(a, screen_gamma); gamma_from = G_SRGB; }
    if (gamma_from != G_LINEAR && !(gamma_from == G_SRGB && gamma_to == G_MONITOR)) {
      const gconst_t g = gconst_for(gamma_from);
      a = (a < g.thresh) ? a / g.lin : powf((a + g.offs) / (1. + g.offs), g.pf);
      gamma_from = G_LINEAR;          /* sticks for every later entry */
    }
    if (gamma_to != G_LINEAR) {
      const gconst_t g = gconst_for(gamma_to == G_MONITOR ? G_SRGB : gamma_to);
      x = (a < (g.thresh) / g.lin) ? a * g.lin : powf((1. + g.offs) * a, 1. / g.pf) - g.offs;
    }
    if (gamma_to == G_MONITOR) x = powf(a, inv_gamma);
    lut[i] = clamp_int_0_255((int)(x * 255.));
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * alpha pre-multiply tables                              reference: src/colourspace.c:1141-1160
 * CLAMP0255f                                             reference: src/maths.h:88
 * ---------------------------------------------------------------------------------------------- */
static inline int clamp_round_f(float a) {
  if (a != a) return 0;                     /* 0 * inf at alpha == 0: x86-64 gcc yields 0 (fixture-pinned) */
  return a >= 254.5 ? 255 : a < -0.5 ? 0 : (uint8_t)(a + .5);
}
int orc_unal(int alpha, int v) { float al = (float)255. / (float)alpha; return clamp_round_f((float)v / al); }
int orc_al(int alpha, int v) { float al = (float)255. / (float)alpha; return clamp_round_f((float)v * al); }

void orc_alpha_premult(uint8_t *pix, int rowstride, int width, int height, int alpha_first, int un) {
  /* src/colourspace.c:12063-12082: RGBA/BGRA colour bytes 0..2, alpha 3; ARGB colour bytes 1..3, alpha 0 */
  const int aoffs = alpha_first ? 0 : 3, c0 = alpha_first ? 1 : 0;
  for (int i = 0; i < height; i++) {
    uint8_t *p = pix + (size_t)i * rowstride;
    for (int j = 0; j < width * 4; j += 4) {
      const int alpha = p[j + aoffs];
      for (int c = c0; c < c0 + 3; c++) p[j + c] = (uint8_t)(un ? orc_unal(alpha, p[j + c]) : orc_al(alpha, p[j + c]));
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * K1: packed RGB swizzles                                reference: src/colourspace.c:9259-10577
 * Each op is a per-pixel byte selection; 0xFF in the selector means "constant 255" (new alpha).
 * The gamma LUT touches colour bytes only.  swapprepost / swap4 follow the evident intent
 * (RGBA<->ARGB rotate, BGRA<->ARGB reverse): several of the reference bodies corrupt memory or
 * mis-order bytes (DESIGN.md quirk list).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int ibpp, obpp; uint8_t sel[4]; uint8_t is_alpha[4]; } swz_t;

static swz_t swz_for(int op, int alpha_first) {
  swz_t s; memset(&s, 0, sizeof s);
#define SET(ib, ob, a, b, c, d) do { s.ibpp = ib; s.obpp = ob; s.sel[0] = a; s.sel[1] = b; s.sel[2] = c; s.sel[3] = d; } while (0)
  switch (op) {
  case ORC_SWAP3: SET(3, 3, 2, 1, 0, 0); break;
  case ORC_SWAP4:
    SET(4, 4, 3, 2, 1, 0);
    if (alpha_first) s.is_alpha[3] = 1; else s.is_alpha[0] = 1;     /* in ARGB -> out BGRA ; in BGRA -> out ARGB */
    break;
  case ORC_SWAP3ADDPOST: SET(3, 4, 2, 1, 0, 0xFF); break;
  case ORC_SWAP3ADDPRE: SET(3, 4, 0xFF, 2, 1, 0); break;
  case ORC_SWAP3POSTALPHA: SET(4, 4, 2, 1, 0, 3); s.is_alpha[3] = 1; break;
  case ORC_SWAP3PREALPHA: SET(4, 4, 0, 3, 2, 1); s.is_alpha[0] = 1; break;
  case ORC_ADDPOST: SET(3, 4, 0, 1, 2, 0xFF); break;
  case ORC_ADDPRE: SET(3, 4, 0xFF, 0, 1, 2); break;
  case ORC_SWAP3DELPOST: SET(4, 3, 2, 1, 0, 0); break;
  case ORC_DELPOST: SET(4, 3, 0, 1, 2, 0); break;
  case ORC_DELPRE: SET(4, 3, 1, 2, 3, 0); break;
  case ORC_SWAP3DELPRE: SET(4, 3, 3, 2, 1, 0); break;
  case ORC_SWAPPREPOST:
    if (alpha_first) { SET(4, 4, 1, 2, 3, 0); s.is_alpha[3] = 1; }   /* ARGB -> RGBA */
    else { SET(4, 4, 3, 0, 1, 2); s.is_alpha[0] = 1; }               /* RGBA -> ARGB */
    break;
  default: s.ibpp = 0;
  }
#undef SET
  return s;
}

int orc_swizzle(int op, int alpha_first, const uint8_t *src, int irow, uint8_t *dst, int orow,
                int width, int height, const uint8_t *lut8) {
  const swz_t s = swz_for(op, alpha_first);
  if (!s.ibpp) return -1;
  for (int y = 0; y < height; y++) {
    const uint8_t *ip = src + (size_t)y * irow;
    uint8_t *op_ = dst + (size_t)y * orow;
    for (int x = 0; x < width; x++, ip += s.ibpp, op_ += s.obpp) {
      uint8_t in[4] = {ip[0], ip[1], ip[2], s.ibpp == 4 ? ip[3] : 0}, out[4];
      for (int k = 0; k < s.obpp; k++) {
        if (s.sel[k] == 0xFF) out[k] = 255;
        else if (s.is_alpha[k] || !lut8) out[k] = in[s.sel[k]];
        else out[k] = lut8[in[s.sel[k]]];
      }
      memcpy(op_, out, s.obpp);      /* after reading: in-place safe for equal bpp */
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K6: gamma apply                                        reference: src/colourspace.c:14034-14060
 * ---------------------------------------------------------------------------------------------- */
void orc_gamma_apply(uint8_t *pix, int rowstride, int width, int height, int psize, int alpha_first,
                     const uint8_t *lut8) {
  const int ncol = psize < 3 ? psize : 3, start = alpha_first ? 1 : 0;
  if (!lut8) return;
  for (int y = 0; y < height; y++) {
    uint8_t *p = pix + (size_t)y * rowstride + start;
    for (int x = 0; x < width; x++, p += psize)
      for (int c = 0; c < ncol; c++) p[c] = lut8[p[c]];
  }
}

/* ------------------------------------------------------------------------------------------------
 * K2: YUV420P / YUV422P -> RGB                           reference: src/colourspace.c:3260-3904
 * per-pixel maths                                        reference: src/colourspace.c:2351-2356, :832-843
 * chroma clamp macros                                    reference: src/colourspace.h:19-23
 * ---------------------------------------------------------------------------------------------- */
static inline int clamp_uv_c(int n) {          /* CLAMP16_240 */
  if (n < 0) return 16;
  if (n > 255 || (n & 0xF0) == 0xF0) return 240;
  return (n & 0xF0) ? n : 16;
}
static inline int clamp_uv_u(int n) { return n < 0 ? 0 : n > 255 ? 255 : n; }   /* CLAMP0_255 */

static inline int fix_shift(int32_t v, int quality) {         /* _spc_rnd */
  if (quality != 3) return v >> 16;
  return (int32_t)((float)v / 65536.);
}

typedef struct {
  const int32_t *ty, *rcr, *gcb, *gcr, *bcb;
  const uint8_t *lut8;
  const uint16_t *lut16;       /* xyuv2rgb_with_gamma (:2386-2390): 16-bit indexed LUT, takes precedence over lut8 */
  int quality, opsize, order, clamped;
} yuvctx_t;

static inline void put_px(const yuvctx_t *c, uint8_t *d, int y, int u, int v) {
  const int32_t yy = c->ty[y];
  uint8_t r = clamp_int_0_255(fix_shift(yy + c->rcr[v], c->quality));
  uint8_t g = clamp_int_0_255(fix_shift(yy + c->gcb[u] + c->gcr[v], c->quality));
  uint8_t b = clamp_int_0_255(fix_shift(yy + c->bcb[u], c->quality));
  if (c->lut16) {                                   /* lut[CLAMP16biti(sum >> 8)] >> 8 */
#define C16(n) ((n) > 65535 ? 65535 : (n) < 0 ? 0 : (n))
    r = (uint8_t)(c->lut16[C16((yy + c->rcr[v]) >> 8)] >> 8);
    g = (uint8_t)(c->lut16[C16((yy + c->gcb[u] + c->gcr[v]) >> 8)] >> 8);
    b = (uint8_t)(c->lut16[C16((yy + c->bcb[u]) >> 8)] >> 8);
#undef C16
  } else if (c->lut8) { r = c->lut8[r]; g = c->lut8[g]; b = c->lut8[b]; }
  switch (c->order) {
  case 0: d[0] = r; d[1] = g; d[2] = b; if (c->opsize == 4) d[3] = 255; break;
  case 1: d[0] = b; d[1] = g; d[2] = r; if (c->opsize == 4) d[3] = 255; break;
  default: d[0] = 255; d[1] = r; d[2] = g; d[3] = b; break;
  }
}

static inline int cuv(const yuvctx_t *c, int n) { return c->clamped ? clamp_uv_c(n) : clamp_uv_u(n); }

/* (2a + b) / 3 and (a + 2b) / 3 on doubled sums: (int)(s / 3. + .5) == (s + 1) / 3 for s >= 0 */
static inline void vblend(const yuvctx_t *c, int s1, int s2, int *top, int *bot) {
  if (c->quality != 1) {
    *top = cuv(c, (s1 + (s2 >> 1) + 1) / 3);
    *bot = cuv(c, ((s1 >> 1) + s2 + 1) / 3);
  } else { *top = cuv(c, s1 >> 1); *bot = cuv(c, s2 >> 1); }
}

static int k2_impl(const uint8_t *y, const uint8_t *u, const uint8_t *v, const int istrides[3],
                   long u_size, long v_size, uint8_t *dst, int orow, int width, int height,
                   int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                   const uint8_t *lut8, const uint16_t *lut16, int fix_edges) {
  yuvctx_t c;
  const int ys = istrides[0], us = istrides[1], vs = istrides[2], hw = width >> 1;
  if (!tables_ready) build_tables();
  if ((width & 1) || width < 2 || height < 1) return -1;
  c.ty = T_y2r[which_tables & 3][0]; c.rcr = T_y2r[which_tables & 3][1]; c.gcb = T_y2r[which_tables & 3][2];
  c.gcr = T_y2r[which_tables & 3][3]; c.bcb = T_y2r[which_tables & 3][4];
  c.lut8 = lut8; c.lut16 = lut16; c.quality = pb_quality; c.opsize = (out_order == 2) ? 4 : opsize; c.order = out_order;
  c.clamped = !(which_tables & 1);
  const int ops = c.opsize;
  /* plane fetch with the index held inside the plane (the reference reads one sample past the row
     end for the right-hand pixel of the last pair: next row's first sample, or past the plane) */
#define PU(r, k) (u[((long)(r) * us + (k)) < u_size ? ((long)(r) * us + (k)) : u_size - 1])
#define PV(r, k) (v[((long)(r) * vs + (k)) < v_size ? ((long)(r) * vs + (k)) : v_size - 1])

  if (is_422) {
    /* :3593-3640 (clamped) / :3858-3901 (unclamped): row i, "last/this" seeded from chroma row i>>1 */
    for (int i = 0; i < height; i++) {
      int lu = PU(i >> 1, 0), lv = PV(i >> 1, 0), tu = lu, tv = lv;
      uint8_t *d = dst + (size_t)i * orow;
      for (int k = 0; k < hw; k++) {
        const int nu = PU(i, k + 1), nv = PV(i, k + 1);
        put_px(&c, d + (2 * k) * ops, y[(size_t)i * ys + 2 * k], cuv(&c, (tu + lu) >> 1), cuv(&c, (tv + lv) >> 1));
        put_px(&c, d + (2 * k + 1) * ops, y[(size_t)i * ys + 2 * k + 1], cuv(&c, (tu + nu) >> 1), cuv(&c, (tv + nv) >> 1));
        lu = tu; lv = tv; tu = nu; tv = nv;
      }
    }
    return 0;
  }

  /* row 0 (:3399-3443): left pixel of each pair averages this and previous chroma sample.  The right
     pixel indexes the tables with an un-halved sum in the reference (undefined) -> evident intent. */
  for (int k = 0; k < hw; k++) {
    const int kp = k ? k - 1 : 0, kn = (k + 1 < hw) ? k + 1 : hw - 1;
    put_px(&c, dst + (2 * k) * ops, y[2 * k], cuv(&c, (PU(0, k) + PU(0, kp)) >> 1), cuv(&c, (PV(0, k) + PV(0, kp)) >> 1));
    put_px(&c, dst + (2 * k + 1) * ops, y[2 * k + 1], cuv(&c, (PU(0, k) + PU(0, kn)) >> 1), cuv(&c, (PV(0, k) + PV(0, kn)) >> 1));
  }

  /* rows 1..h-2 in pairs (i, i+1) between chroma rows r and r+1 (:3445-3554) */
  int i;
  for (i = 1; i < height - 1; i += 2) {
    const int r = i >> 1;
    uint8_t *d0 = dst + (size_t)i * orow, *d1 = d0 + orow;
    const uint8_t *y0 = y + (size_t)i * ys, *y1 = y0 + ys;
    for (int k = 0; k < hw; k++) {
      int s1, s2, t, b, tv_, bv_;
      /* left pixel: the reference builds the second-row U sum from the FIRST row (:3461), pairs V of
         row r with the previous V of row r+1 (:3544 stores this_v2 into last_v1) and never advances
         last_v2 from column 0 -- all deterministic, all replicated */
      const int lu1 = k ? PU(r, k - 1) : PU(r, 0);
      const int lv1 = k ? PV(r + 1, k - 1) : PV(r, 0);
      const int lv2 = PV(r + 1, 0);
      s1 = PU(r, k) + lu1; s2 = s1;
      vblend(&c, s1, s2, &t, &b);
      s1 = PV(r, k) + lv1; s2 = PV(r + 1, k) + lv2;
      vblend(&c, s1, s2, &tv_, &bv_);
      put_px(&c, d0 + (2 * k) * ops, y0[2 * k], t, tv_);
      put_px(&c, d1 + (2 * k) * ops, y1[2 * k], b, bv_);
      /* right pixel: this + next on both chroma rows */
      s1 = PU(r, k) + PU(r, k + 1); s2 = PU(r + 1, k) + PU(r + 1, k + 1);
      vblend(&c, s1, s2, &t, &b);
      s1 = PV(r, k) + PV(r, k + 1); s2 = PV(r + 1, k) + PV(r + 1, k + 1);
      vblend(&c, s1, s2, &tv_, &bv_);
      put_px(&c, d0 + (2 * k + 1) * ops, y0[2 * k + 1], t, tv_);
      put_px(&c, d1 + (2 * k + 1) * ops, y1[2 * k + 1], b, bv_);
    }
  }

  /* trailing row (:3556-3592).  Reference, 1 thread: luma is taken from ROW 0, "next" chroma from
     chroma row 0, the right-hand pixels land in row 0 (undefined values) and the last row's odd
     pixels are never written.  fix_edges == 0 keeps the deterministic part of that (even x). */
  if (i < height) {
    const int r = i >> 1;
    uint8_t *d = dst + (size_t)i * orow;
    if (!fix_edges) {
      int lu = PU(r, 0), lv = PV(r, 0), tu = lu, tv = lv;
      for (int k = 0; k < hw; k++) {
        const int kn = (k + 1 < hw) ? k + 1 : hw - 1;
        put_px(&c, d + (2 * k) * ops, y[2 * k], cuv(&c, (tu + lu) >> 1), cuv(&c, (tv + lv) >> 1));
        /* odd x: unwritten by the reference -> evident intent */
        put_px(&c, d + (2 * k + 1) * ops, y[(size_t)i * ys + 2 * k + 1],
               cuv(&c, (PU(r, k) + PU(r, kn)) >> 1), cuv(&c, (PV(r, k) + PV(r, kn)) >> 1));
        lu = tu; lv = tv; tu = PU(0, k + 1); tv = PV(0, k + 1);
      }
    } else {
      for (int k = 0; k < hw; k++) {
        const int kp = k ? k - 1 : 0, kn = (k + 1 < hw) ? k + 1 : hw - 1;
        put_px(&c, d + (2 * k) * ops, y[(size_t)i * ys + 2 * k],
               cuv(&c, (PU(r, k) + PU(r, kp)) >> 1), cuv(&c, (PV(r, k) + PV(r, kp)) >> 1));
        put_px(&c, d + (2 * k + 1) * ops, y[(size_t)i * ys + 2 * k + 1],
               cuv(&c, (PU(r, k) + PU(r, kn)) >> 1), cuv(&c, (PV(r, k) + PV(r, kn)) >> 1));
      }
    }
  }
#undef PU
#undef PV
  return 0;
}

int orc_yuv420p_to_rgb(const uint8_t *y, const uint8_t *u, const uint8_t *v, const int istrides[3],
                       long u_size, long v_size, uint8_t *dst, int orow, int width, int height,
                       int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                       const uint8_t *lut8, int fix_edges) {
  return k2_impl(y, u, v, istrides, u_size, v_size, dst, orow, width, height, opsize, out_order, is_422, which_tables, pb_quality, lut8, NULL,
                 fix_edges);
}
int orc_yuv420p_to_rgb_lut16(const uint8_t *y, const uint8_t *u, const uint8_t *v, const int istrides[3],
                             long u_size, long v_size, uint8_t *dst, int orow, int width, int height,
                             int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                             const uint16_t *lut16, int fix_edges) {
  return k2_impl(y, u, v, istrides, u_size, v_size, dst, orow, width, height, opsize, out_order, is_422, which_tables, pb_quality, NULL, lut16,
                 fix_edges);
}

/* create_gamma_lut (src/colourspace.c:738-808): the 65536-entry sibling of create_gamma_lut8, same mutation of
   gamma_from inside the loop; CLAMP16bit (src/colourspace.h:16) */
int orc_gamma_lut16(double fileg, int gamma_from, int gamma_to, double screen_gamma, uint16_t *lut) {
  float inv_gamma = 0., a, x = 0.;
  if (fileg == 1.0 && (gamma_to == gamma_from || gamma_to == G_UNKNOWN || gamma_from == G_UNKNOWN)) return 0;
  if (gamma_to == G_MONITOR) inv_gamma = 1. / (float)screen_gamma;
  lut[0] = 0;
  for (int i = 1; i < 65536; ++i) {
    x = a = (float)i / 65536.;
    if (fileg != 1.0) x = powf(a, fileg);
    if (gamma_from == G_MONITOR) { x = powf(a, screen_gamma); gamma_from = G_SRGB; }
    if (gamma_from != G_LINEAR && !(gamma_from == G_SRGB && gamma_to == G_MONITOR)) {
      const gconst_t g = gconst_for(gamma_from);
      a = (a < g.thresh) ? a / g.lin : powf((a + g.offs) / (1. + g.offs), g.pf);
      gamma_from = G_LINEAR;
    }
    if (gamma_to != G_LINEAR) {
      const gconst_t g = gconst_for(gamma_to == G_MONITOR ? G_SRGB : gamma_to);
      x = (a < (g.thresh) / g.lin) ? a * g.lin : powf((1. + g.offs) * a, 1. / g.pf) - g.offs;
    }
    if (gamma_to == G_MONITOR) x = powf(a, inv_gamma);
    lut[i] = (x) >= 0.99999 ? 65535 : x < 0.00001 ? 0 : (uint16_t)(x * 65535.9999);
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * K8: letterbox                                          reference: src/colourspace.c:15343-15567
 * black fill of `nwidth` pixels per row                  reference: src/colourspace.c:11109-11119
 * ---------------------------------------------------------------------------------------------- */
void orc_letterbox(const uint8_t *src, int irow, int width, int height, uint8_t *dst, int orow,
                   int nwidth, int nheight, int psize, const uint8_t *black_pixel) {
  const int ox = ((nwidth - width + 1) >> 1) * psize, oy = (nheight - height + 1) >> 1;
  for (int y = 0; y < nheight; y++) {
    uint8_t *d = dst + (size_t)y * orow;
    for (int x = 0; x < nwidth; x++) memcpy(d + x * psize, black_pixel, psize);
  }
  for (int y = 0; y < height; y++)
    memcpy(dst + (size_t)(y + oy) * orow + ox, src + (size_t)y * irow, (size_t)width * psize);
}

/* ------------------------------------------------------------------------------------------------
 * F1: chroma blend                    reference: lives-plugins/weed-plugins/simple_blend.c:29-33, :117-150
 * ---------------------------------------------------------------------------------------------- */
static inline uint8_t mix8(int bf, int a2, int a1) { return (uint8_t)((bf * a2 + (255 - bf) * a1) >> 8); }

void orc_blend_chroma(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow,
                      int width, int height, int psize, int alpha_first, int bf) {
  bf &= 0xFF;
  for (int y = 0; y < height; y++) {
    const uint8_t *a = src1 + (size_t)y * irow1, *b = src2 + (size_t)y * irow2;
    uint8_t *d = dst + (size_t)y * orow;
    if (psize == 3) {
      for (int j = 0; j < width * 3; j++) d[j] = mix8(bf, b[j], a[j]);
      continue;
    }
    /* 4-byte palettes: colour bytes j..j+2, "alpha" = byte j+3 of layer 2 (for ARGB, start = 1, that is the
       NEXT pixel's alpha -- reference behaviour, kept).  dst alpha is never written. */
    for (int j = alpha_first ? 1 : 0; j < width * 4; j += 4) {
      const int al = b[j + 3];
      if (al == 255) {
        for (int c = 0; c < 3; c++) d[j + c] = mix8(bf, b[j + c], a[j + c]);
      } else {
        const float alpha = (float)al / 255., inv_alpha = 1. - alpha;
        for (int c = 0; c < 3; c++)
          d[j + c] = mix8(bf, (uint8_t)((float)b[j + c] * alpha), (uint8_t)((float)a[j + c] * inv_alpha));
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * luma of a pixel, 65536-scaled unclamped BT.601 tables  reference: libweed/weed-plugin-utils.c:879-895, :924-934
 * ---------------------------------------------------------------------------------------------- */
static int32_t L_r[256], L_g[256], L_b[256];
static int luma_ready = 0;
static void build_luma(void) {
  for (int i = 0; i < 256; i++) {
    L_r[i] = rnd_half_away(0.299 * (double)i * 65536.);
    L_g[i] = rnd_half_away((1. - 0.299 - 0.114) * (double)i * 65536.);
    L_b[i] = rnd_half_away(0.114 * (double)i * 65536.);
  }
  luma_ready = 1;
}
static inline uint8_t luma_of(const uint8_t *p, int order) {
  switch (order) {
  case 0: return (uint8_t)((L_r[p[0]] + L_g[p[1]] + L_b[p[2]]) >> 16);
  case 1: return (uint8_t)((L_r[p[2]] + L_g[p[1]] + L_b[p[0]]) >> 16);
  default: return (uint8_t)((L_r[p[1]] + L_g[p[2]] + L_b[p[3]]) >> 16);
  }
}

/* F2: luma overlay family              reference: lives-plugins/weed-plugins/simple_blend.c:151-194
   type 4 ("averaged luma overlay") never enters its 3x3 branch in the reference (`row` stays 0) and falls
   through to type 1. */
void orc_blend_luma(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst,
                    int orow, int width, int height, int psize, int pal_order, int thresh, int inplace) {
  const int bf = thresh & 0xFF, neg = 0xFF - bf;
  if (!luma_ready) build_luma();
  for (int y = 0; y < height; y++) {
    const uint8_t *a = src1 + (size_t)y * irow1, *b = src2 + (size_t)y * irow2;
    uint8_t *d = dst + (size_t)y * orow;
    for (int j = (pal_order == 2) ? 1 : 0; j < width * psize; j += psize) {
      int take2;
      switch (type) {
      case 2: take2 = luma_of(b + j, pal_order) > neg; break;
      case 3: take2 = luma_of(a + j, pal_order) > neg; break;
      default: take2 = luma_of(a + j, pal_order) < bf; break;
      }
      if (take2) memcpy(d + j, b + j, 3);
      else if (!inplace) memcpy(d + j, a + j, 3);
    }
  }
}

/* F3: multiply / screen / darken / lighten / overlay / dodge / burn
                                        reference: lives-plugins/weed-plugins/multi_blends.c:24-168 */
void orc_blend_multi(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst,
                     int orow, int width, int height, int is_bgr, int bf) {
  const uint8_t f = (uint8_t)bf;
  const uint8_t b1 = (uint8_t)(f * 2), n1 = (uint8_t)(255 - f * 2), b2 = (uint8_t)((255 - f) * 2), n2 = (uint8_t)((f - 128) * 2);
  if (!luma_ready) build_luma();
  for (int y = 0; y < height; y++) {
    const uint8_t *a = src1 + (size_t)y * irow1, *b = src2 + (size_t)y * irow2;
    uint8_t *d = dst + (size_t)y * orow;
    for (int j = 0; j < width * 3; j += 3) {
      uint8_t px[3];
      int la, lb, v;
      switch (type) {
      case 0: for (int c = 0; c < 3; c++) px[c] = (uint8_t)((b[j + c] * a[j + c]) >> 8); break;
      case 1: for (int c = 0; c < 3; c++) px[c] = (uint8_t)(255 - (((255 - b[j + c]) * (255 - a[j + c])) >> 8)); break;
      case 2: la = luma_of(a + j, is_bgr); lb = luma_of(b + j, is_bgr); memcpy(px, (la <= lb) ? a + j : b + j, 3); break;
      case 3: la = luma_of(a + j, is_bgr); lb = luma_of(b + j, is_bgr); memcpy(px, (la >= lb) ? a + j : b + j, 3); break;
      case 4:
        la = luma_of(a + j, is_bgr);
        for (int c = 0; c < 3; c++)
          px[c] = (la < 128) ? (uint8_t)((b[j + c] * a[j + c]) >> 8)
                  : (uint8_t)(255 - (((255 - b[j + c]) * (255 - a[j + c])) >> 8));
        break;
      case 5:
        for (int c = 0; c < 3; c++) {
          if (b[j + c] == 255) px[c] = 255;
          else { v = ((int)a[j + c] << 8) / (255 - b[j + c]); px[c] = v > 255 ? 255 : (uint8_t)v; }
        }
        break;
      default:
        for (int c = 0; c < 3; c++) {
          if (b[j + c] == 0) px[c] = 0;
          else { v = 255 - (255 - ((int)a[j + c] << 8)) / (int)b[j + c]; px[c] = v < 0 ? 0 : (uint8_t)v; }
        }
        break;
      }
      if (f < 128) for (int c = 0; c < 3; c++) d[j + c] = (uint8_t)((b1 * px[c] + n1 * a[j + c]) >> 8);
      else for (int c = 0; c < 3; c++) d[j + c] = (uint8_t)((b2 * px[c] + n2 * b[j + c]) >> 8);
    }
  }
}

/* F4: colour key                       reference: lives-plugins/weed-plugins/scripts/colorkey.script <process> */
void orc_colorkey(const uint8_t *src0, int irow0, const uint8_t *src1, int irow1, uint8_t *dst, int orow,
                  int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b,
                  int inplace) {
  double xdelta = delta * 2., opacx = 1. - opac;
  int rmin, gmin, bmin, rmax, gmax, bmax;
  delta /= 2.;
  rmin = col_r - (int)(col_r * delta + .5);
  gmin = col_g - (int)(col_g * xdelta + .5);
  bmin = col_b - (int)(col_b * delta + .5);
  xdelta *= 2.; delta *= 2.;
  rmax = col_r + (int)((255 - col_r) * delta + .5);
  gmax = col_g + (int)((255 - col_g) * xdelta + .5);
  bmax = col_b + (int)((255 - col_b) * delta + .5);
  for (int y = 0; y < height; y++) {
    const uint8_t *a = src0 + (size_t)y * irow0, *b = src1 + (size_t)y * irow1;
    uint8_t *d = dst + (size_t)y * orow;
    for (int j = 0; j < width * 3; j += 3) {
      const int r = is_bgr ? a[j + 2] : a[j], g = a[j + 1], bl = is_bgr ? a[j] : a[j + 2];
      if (r >= rmin && r <= rmax && g >= gmin && g <= gmax && bl >= bmin && bl <= bmax) {
        for (int c = 0; c < 3; c++) d[j + c] = (uint8_t)(a[j + c] * opacx + b[j + c] * opac);
      } else if (!inplace) memcpy(d + j, a + j, 3);
    }
  }
}

/* the same key on 4-byte pixels (RGBA32 / BGRA32) -- an EXTENSION (the reference filter takes RGB24 / BGR24 only): the three colour bytes as above, the alpha
   byte is the first frame's.  Used for BASELINE config 4's "RGBA32 extension for the headline" (SURVEY 8d); own spec, unpinned. */
void orc_colorkey4(const uint8_t *src0, int irow0, const uint8_t *src1, int irow1, uint8_t *dst, int orow,
                   int width, int height, int is_bgr, double delta, double opac, int col_r, int col_g, int col_b) {
  double xdelta = delta * 2., opacx = 1. - opac;
  int rmin, gmin, bmin, rmax, gmax, bmax;
  delta /= 2.;
  rmin = col_r - (int)(col_r * delta + .5);
  gmin = col_g - (int)(col_g * xdelta + .5);
  bmin = col_b - (int)(col_b * delta + .5);
  xdelta *= 2.; delta *= 2.;
  rmax = col_r + (int)((255 - col_r) * delta + .5);
  gmax = col_g + (int)((255 - col_g) * xdelta + .5);
  bmax = col_b + (int)((255 - col_b) * delta + .5);
  for (int y = 0; y < height; y++) {
    const uint8_t *a = src0 + (size_t)y * irow0, *b = src1 + (size_t)y * irow1;
    uint8_t *d = dst + (size_t)y * orow;
    for (int j = 0; j < width * 4; j += 4) {
      const int r = is_bgr ? a[j + 2] : a[j], g = a[j + 1], bl = is_bgr ? a[j] : a[j + 2];
      memcpy(d + j, a + j, 4);
      if (r >= rmin && r <= rmax && g >= gmin && g <= gmax && bl >= bmin && bl <= bmax)
        for (int c = 0; c < 3; c++) d[j + c] = (uint8_t)(a[j + c] * opacx + b[j + c] * opac);
    }
  }
}

/* F5: mirrors                          reference: lives-plugins/weed-plugins/mirrors.c:26-122
   The reference's stray writes (pixel `width` of each row for even widths, row `height`) are not
   performed; rows / pixels it leaves unwritten in non-inplace mode get the in-place result. */
static void mirror_x_rows(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize) {
  const int hw = width >> 1;
  for (int y = 0; y < height; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    uint8_t *d = dst + (size_t)y * orow;
    if (s != d) memcpy(d, s, (size_t)hw * psize);
    for (int k = 0; k <= hw; k++) {
      const int t = 2 * hw - k;
      if (t < width) memmove(d + (size_t)t * psize, s + (size_t)k * psize, psize);
    }
  }
}
static void mirror_y_rows(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize) {
  const int hh = height >> 1;
  const size_t rb = (size_t)width * psize;
  if (src != dst) for (int y = 0; y < height; y++) memcpy(dst + (size_t)y * orow, src + (size_t)y * irow, rb);
  for (int i = 1; i < hh; i++) memcpy(dst + (size_t)(height - i) * orow, src + (size_t)i * irow, rb);
}
void orc_mirror(int mode, const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize) {
  if (mode == 0) mirror_x_rows(src, irow, dst, orow, width, height, psize);
  else if (mode == 1) mirror_y_rows(src, irow, dst, orow, width, height, psize);
  else { mirror_y_rows(src, irow, dst, orow, width, height, psize); mirror_x_rows(dst, orow, dst, orow, width, height, psize); }
}

/* ------------------------------------------------------------------------------------------------
 * K4 / K3: the rest of the palette matrix              reference: src/colourspace.c (ranges in lives_oracle.h)
 * ---------------------------------------------------------------------------------------------- */
int orc_cavg(int clamped, int x, int y) {              /* init_average, :190-216 (the non-MULT_AVG branch) */
  if (clamped) {
    const float fa = (float)(x - 128.) * 255. / 244., fb = (float)(y - 128.) * 255. / 244.;
    const float fc = (fa + fb) * 224. / 512. + 128.;
    return (uint8_t)(fc > 240. ? 240 : fc < 16. ? 16 : fc);
  } else {
    const short sa = (short)(x - 128), sb = (short)(y - 128);
    const short c = (short)(((sa + sb) >> 1) + 128);
    return (uint8_t)(c > 255 ? 255 : c < 0 ? 0 : c);
  }
}
typedef struct { const int32_t (*t)[256]; int min_y, max_y, min_uv, max_uv; } r2y_t;
static r2y_t r2y_for(int which) {
  r2y_t c;
  if (!tables_ready) build_tables();
  c.t = T_r2y[which & 3];
  if (which & 1) { c.min_y = c.min_uv = 0; c.max_y = c.max_uv = 255; }
  else { c.min_y = c.min_uv = 16; c.max_y = 235; c.max_uv = 240; }      /* set_conversion_arrays :361-370 */
  return c;
}
/* rgb2yuv (:2119-2127): short a = spc_rnd(sum); upper clamp first, then lower */
static inline uint8_t r2y_Y(const r2y_t *c, int r, int g, int b) {
  const short a = (short)((c->t[0][r] + c->t[1][g] + c->t[2][b]) >> 16);
  return (uint8_t)(a > c->max_y ? c->max_y : a < c->min_y ? c->min_y : a);
}
static inline short r2y_Uraw(const r2y_t *c, int r, int g, int b) { return (short)((c->t[3][r] + c->t[4][g] + c->t[5][b]) >> 16); }
static inline short r2y_Vraw(const r2y_t *c, int r, int g, int b) { return (short)((c->t[6][r] + c->t[7][g] + c->t[8][b]) >> 16); }
static inline uint8_t r2y_clampuv(const r2y_t *c, short a) { return (uint8_t)(a > c->max_uv ? c->max_uv : a < c->min_uv ? c->min_uv : a); }
static inline void px_rgb(const uint8_t *p, int order, int *r, int *g, int *b) {
  switch (order) {
  case 0: *r = p[0]; *g = p[1]; *b = p[2]; break;
  case 1: *r = p[2]; *g = p[1]; *b = p[0]; break;
  default: *r = p[1]; *g = p[2]; *b = p[3]; break;
  }
}
int orc_rgb_to_yuv(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha,
                   uint8_t *const dst[4], const int orow[4], int out_fmt, int out_alpha, int which_tables) {
  const r2y_t c = r2y_for(which_tables);
  const int ips = (in_order == 2 || in_alpha) ? 4 : 3;
  int r, g, b;
  if (in_order < 0 || in_order > 2 || out_fmt < 0 || out_fmt > 5 || width < 1 || height < 1) return -1;
  if (out_fmt >= 2 && (width & 1)) return -1;
  if (out_fmt <= 3 && (which_tables & 2)) return -1;      /* only the 4:2:0 / 4:2:2 entry points take a subspace */
  if (out_fmt >= 4 && in_order == 2) return -1;
  if (out_fmt == 4 && (height & 1)) return -1;
  if (out_fmt <= 1) {
    const int w = (width >> 1) << 1;                                    /* :5761 */
    for (int y = 0; y < height; y++) {
      const uint8_t *s = src + (size_t)y * irow;
      for (int x = 0; x < w; x++) {
        const uint8_t *p = s + x * ips;
        const uint8_t alpha = in_order == 2 ? p[0] : ips == 4 ? p[3] : 255;
        px_rgb(p, in_order, &r, &g, &b);
        const uint8_t Y = r2y_Y(&c, r, g, b), U = r2y_clampuv(&c, r2y_Uraw(&c, r, g, b)), V = r2y_clampuv(&c, r2y_Vraw(&c, r, g, b));
        if (out_fmt == 0) {
          uint8_t *d = dst[0] + (size_t)y * orow[0] + x * (out_alpha ? 4 : 3);
          d[0] = Y; d[1] = U; d[2] = V;
          if (out_alpha) d[3] = alpha;
        } else {
          dst[0][(size_t)y * orow[0] + x] = Y; dst[1][(size_t)y * orow[1] + x] = U; dst[2][(size_t)y * orow[2] + x] = V;
          if (out_alpha) dst[3][(size_t)y * orow[3] + x] = alpha;
        }
      }
    }
    return 0;
  }
  /* pair formats: rgb2uyvy / rgb2yuyv (:2162-2192): U from the first pixel, V from the second */
  for (int y = 0; y < height; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    for (int x = 0; x < width; x += 2) {
      int r1, g1, b1;
      px_rgb(s + x * ips, in_order, &r, &g, &b);
      px_rgb(s + (x + 1) * ips, in_order, &r1, &g1, &b1);
      const uint8_t y0 = r2y_Y(&c, r, g, b), y1 = r2y_Y(&c, r1, g1, b1);
      const short ur = r2y_Uraw(&c, r, g, b), vr = r2y_Vraw(&c, r1, g1, b1);
      if (out_fmt == 2) {
        uint8_t *d = dst[0] + (size_t)y * orow[0] + x * 2;
        d[0] = r2y_clampuv(&c, ur); d[1] = y0; d[2] = r2y_clampuv(&c, vr); d[3] = y1;
      } else if (out_fmt == 3) {
        uint8_t *d = dst[0] + (size_t)y * orow[0] + x * 2;                /* rgb2yuyv: the `else` is missing, the upper clamp is lost */
        d[0] = y0; d[1] = (uint8_t)(ur < c.min_uv ? c.min_uv : ur); d[2] = y1; d[3] = (uint8_t)(vr < c.min_uv ? c.min_uv : vr);
      } else {
        dst[0][(size_t)y * orow[0] + x] = y0; dst[0][(size_t)y * orow[0] + x + 1] = y1;
        const uint8_t cu = r2y_clampuv(&c, ur), cv = r2y_clampuv(&c, vr);
        if (out_fmt == 5) { dst[1][(size_t)y * orow[1] + (x >> 1)] = cu; dst[2][(size_t)y * orow[2] + (x >> 1)] = cv; }
        else {
          /* :6302-6315 at compact strides: an even row k2 = 2k (k > 0) is averaged INTO chroma row k-1 (which holds luma row
             2k-1), then overwritten by luma row 2k+1 */
          const int k = y >> 1;
          uint8_t *pu = dst[1] + (size_t)k * orow[1] + (x >> 1), *pv = dst[2] + (size_t)k * orow[2] + (x >> 1);
          if (!(y & 1) && y > 0) {
            uint8_t *qu = pu - orow[1], *qv = pv - orow[2];
            *qu = (uint8_t)orc_cavg(!(which_tables & 1), cu, *qu);
            *qv = (uint8_t)orc_cavg(!(which_tables & 1), cv, *qv);
          }
          *pu = cu; *pv = cv;
        }
      }
    }
  }
  return 0;
}

int orc_yuv_to_rgb(const uint8_t *const src[4], const int irow[4], int width, int height, int in_fmt, int in_alpha,
                   uint8_t *dst, int orow, int out_order, int out_alpha, int which_tables) {
  yuvctx_t c;
  if (!tables_ready) build_tables();
  if (in_fmt < 0 || in_fmt > 3 || out_order < 0 || out_order > 2 || width < 1 || height < 1) return -1;
  if (in_fmt >= 2 && (width & 1)) return -1;
  if (in_fmt >= 1 && (which_tables & 2)) return -1;
  if (in_fmt == 1 && out_order == 2) return -1;                 /* :7475-7476 subtracts the output stride twice */
  if (in_fmt == 1 && out_order == 1 && !out_alpha) return -1;   /* :7313 steps 4 bytes per pixel into a BGR24 row */
  c.ty = T_y2r[which_tables & 3][0]; c.rcr = T_y2r[which_tables & 3][1]; c.gcb = T_y2r[which_tables & 3][2];
  c.gcr = T_y2r[which_tables & 3][3]; c.bcb = T_y2r[which_tables & 3][4];
  c.lut8 = NULL; c.lut16 = NULL; c.quality = 2; c.order = out_order; c.clamped = !(which_tables & 1);
  c.opsize = (out_order == 2 || out_alpha) ? 4 : 3;
  for (int y = 0; y < height; y++) {
    uint8_t *d = dst + (size_t)y * orow;
    for (int x = 0; x < width; x++) {
      int Y, U, V, A = 255;
      if (in_fmt == 0) {
        const uint8_t *p = src[0] + (size_t)y * irow[0] + x * (in_alpha ? 4 : 3);
        Y = p[0]; U = p[1]; V = p[2]; if (in_alpha) A = p[3];
      } else if (in_fmt == 1) {
        Y = src[0][(size_t)y * irow[0] + x]; U = src[1][(size_t)y * irow[1] + x]; V = src[2][(size_t)y * irow[2] + x];
        if (in_alpha) A = src[3][(size_t)y * irow[3] + x];
      } else {
        const uint8_t *p = src[0] + (size_t)y * irow[0] + (x >> 1) * 4;
        if (in_fmt == 2) { U = p[0]; Y = p[1 + 2 * (x & 1)]; V = p[2]; }
        else { Y = p[2 * (x & 1)]; U = p[1]; V = p[3]; }
      }
      put_px(&c, d + x * c.opsize, Y, U, V);
      if (c.opsize == 4) d[x * 4 + (out_order == 2 ? 0 : 3)] = (uint8_t)A;
    }
  }
  return 0;
}

/* K4b: RGB24 / RGBA32 / BGR24 / BGRA32 / ARGB32 -> YUV411     reference: src/colourspace.c:6499-6615, rgb2_411 :2322-2343;
 * dispatcher :12627-12632 etc.  Four pixels -> u2 y0 y1 v2 y2 y3; chroma = (sum of the four per-pixel >> FP_BITS values) >> 2, clamped
 * afterwards; width % 4 pixels on the right are dropped; the destination is compact rows of (width >> 2) macropixels. */
/* K4 with the 16-bit gamma LUT inline: rgb2uyvy_with_gamma / rgb2yuyv_with_gamma (src/colourspace.c:2146-2159, :2194-2207), what the UYVY / YUYV entry points run
 * when convert_layer_palette_full changes the gamma on the way (:12565-12580, :12646-12660 ...: create_gamma_lut(1.0, gamma_type, new_gamma_type)).  Each table sum
 * indexes the LUT by its top 16 bits (>> 8), the LUT's high byte is the sample; both forms clamp properly here (the missing `else` of rgb2yuyv is not in
 * its gamma twin).  out_fmt 2 UYVY, 3 YUYV; U from the first pixel, V from the second. */
int orc_rgb_to_yuv_lut16(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst, int orow, int out_fmt,
                         int unclamped, const uint16_t *lut16) {
  const r2y_t c = r2y_for(unclamped ? 1 : 0);
  const int ips = (in_order == 2 || in_alpha) ? 4 : 3;
  if (!src || !dst || !lut16 || in_order < 0 || in_order > 2 || (out_fmt != 2 && out_fmt != 3) || width < 2 || (width & 1) || height < 1) return -1;
  for (int y = 0; y < height; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    uint8_t *d = dst + (size_t)y * orow;          /* the reference's row step (:5206) only works for compact rows; any rowstride is honoured here, as in orc_rgb_to_yuv */
    for (int x = 0; x < width; x += 2, d += 4) {
      int r0, g0, b0, r1, g1, b1;
      px_rgb(s + (size_t)x * ips, in_order, &r0, &g0, &b0);
      px_rgb(s + (size_t)(x + 1) * ips, in_order, &r1, &g1, &b1);
      const int y0 = lut16[((uint32_t)(c.t[0][r0] + c.t[1][g0] + c.t[2][b0]) >> 8) & 0xFFFF] >> 8, y1 = lut16[((uint32_t)(c.t[0][r1] + c.t[1][g1] + c.t[2][b1]) >> 8) & 0xFFFF] >> 8;
      const int u = lut16[((uint32_t)(c.t[3][r0] + c.t[4][g0] + c.t[5][b0]) >> 8) & 0xFFFF] >> 8, v = lut16[((uint32_t)(c.t[6][r1] + c.t[7][g1] + c.t[8][b1]) >> 8) & 0xFFFF] >> 8;
      const uint8_t Y0 = (uint8_t)(y0 > c.max_y ? c.max_y : y0 < c.min_y ? c.min_y : y0), Y1 = (uint8_t)(y1 > c.max_y ? c.max_y : y1 < c.min_y ? c.min_y : y1);
      const uint8_t U = (uint8_t)(u > c.max_uv ? c.max_uv : u < c.min_uv ? c.min_uv : u), V = (uint8_t)(v > c.max_uv ? c.max_uv : v < c.min_uv ? c.min_uv : v);
      if (out_fmt == 2) { d[0] = U; d[1] = Y0; d[2] = V; d[3] = Y1; }
      else { d[0] = Y0; d[1] = U; d[2] = Y1; d[3] = V; }
    }
  }
  return 0;
}

int orc_rgb_to_yuv411(const uint8_t *src, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst, int unclamped) {
  const r2y_t c = r2y_for(unclamped ? 1 : 0);
  const int ips = (in_order == 2 || in_alpha) ? 4 : 3, wm = width >> 2;
  if (!src || !dst || in_order < 0 || in_order > 2 || wm < 1 || height < 1 || irow < wm * 4 * ips) return -1;
  for (int y = 0; y < height; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    uint8_t *d = dst + (size_t)y * wm * 6;
    for (int j = 0; j < wm; j++, d += 6) {
      int su = 0, sv = 0, Y[4];
      for (int k = 0; k < 4; k++) {
        int r, g, b;
        px_rgb(s + (size_t)(4 * j + k) * ips, in_order, &r, &g, &b);
        const int a = (c.t[0][r] + c.t[1][g] + c.t[2][b]) >> 16;
        Y[k] = a > c.max_y ? c.max_y : a < c.min_y ? c.min_y : a;
        su += (c.t[3][r] + c.t[4][g] + c.t[5][b]) >> 16;
        sv += (c.t[6][r] + c.t[7][g] + c.t[8][b]) >> 16;
      }
      su >>= 2; sv >>= 2;
      d[0] = (uint8_t)(su > c.max_uv ? c.max_uv : su < c.min_uv ? c.min_uv : su);
      d[1] = (uint8_t)Y[0]; d[2] = (uint8_t)Y[1];
      d[3] = (uint8_t)(sv > c.max_uv ? c.max_uv : sv < c.min_uv ? c.min_uv : sv);
      d[4] = (uint8_t)Y[2]; d[5] = (uint8_t)Y[3];
    }
  }
  return 0;
}

/* K3b: YUV411 (u2 y0 y1 v2 y2 y3, 4 pixels in 6 bytes) -> RGB24 / RGBA32 / BGR24 / BGRA32 / ARGB32
 * reference: src/colourspace.c:8305-8411 (rgb), :8413-8520 (bgr), :8522-8620 (argb); dispatcher :13755-13795.
 * The source is walked as compact rows of `width_mp` macropixels (no input rowstride in the reference).  Kept as written:
 *  - chroma of the inner pixels is a cascade of table averages (avg_chromaf = cavg of the clamping, :2099-2101) between neighbouring blocks;
 *  - the first pair the loop writes per block boundary never gets its alpha byte (RGBA / BGRA / ARGB): those bytes keep what the
 *    destination held;
 *  - the bgr variant writes the row's first pixel and its last two pixels in R,G,B order (uyvy2rgb with swapped pointers only for
 *    the second pixel of the first pair, :8445, :8515). */
int orc_yuv411_to_rgb(const uint8_t *src, int width_mp, int height, uint8_t *dst, int orow, int out_order, int out_alpha, int unclamped) {
  yuvctx_t c;
  if (!tables_ready) build_tables();
  if (!src || !dst || width_mp < 1 || height < 1 || out_order < 0 || out_order > 2) return -1;
  const int w = unclamped ? 1 : 0, cl = !unclamped;
  c.ty = T_y2r[w][0]; c.rcr = T_y2r[w][1]; c.gcb = T_y2r[w][2]; c.gcr = T_y2r[w][3]; c.bcb = T_y2r[w][4];
  c.lut8 = NULL; c.lut16 = NULL; c.quality = 2; c.clamped = cl;
  const int ps = (out_order == 2 || out_alpha) ? 4 : 3;
  c.opsize = 3;                                                 /* colour bytes only: alpha is written (or not) below */
  const int coff = out_order == 2 ? 1 : 0, aoff = out_order == 2 ? 0 : 3;
  if (orow < width_mp * 4 * ps) return -1;
  for (int i = 0; i < height; i++) {
    const uint8_t *row = src + (size_t)i * width_mp * 6;
    uint8_t *d = dst + (size_t)i * orow;
    /* row start (:8330-8337): block 0's y0 y1 with its own chroma */
    c.order = out_order == 1 ? 0 : (out_order == 2 ? 0 : 0);
    if (ps == 4) d[aoff] = d[4 + aoff] = 255;
    c.order = 0; put_px(&c, d + coff, row[1], row[0], row[3]);                       /* R,G,B order even in the bgr variant */
    c.order = out_order == 1 ? 1 : 0; put_px(&c, d + ps + coff, row[2], row[0], row[3]);
    d += 2 * ps;
    int j;
    for (j = 1; j < width_mp; j++) {
      const uint8_t *pb = row + (size_t)(j - 1) * 6, *cb = row + (size_t)j * 6;
      const int pu = pb[0], pv = pb[3], cu = cb[0], cv = cb[3];
      const int hu = orc_cavg(cl, pu, cu), hv = orc_cavg(cl, pv, cv);
      int qu = orc_cavg(cl, hu, pu), qv = orc_cavg(cl, hv, pv);
      c.order = out_order == 1 ? 1 : 0;
      put_px(&c, d + coff, pb[4], orc_cavg(cl, qu, pu), orc_cavg(cl, qv, pv));        /* y2 of the previous block; no alpha write */
      put_px(&c, d + ps + coff, pb[5], orc_cavg(cl, qu, cu), orc_cavg(cl, qv, cv));
      d += 2 * ps;
      qu = orc_cavg(cl, hu, cu); qv = orc_cavg(cl, hv, cv);
      put_px(&c, d + coff, cb[1], orc_cavg(cl, qu, pu), orc_cavg(cl, qv, pv));
      put_px(&c, d + ps + coff, cb[2], orc_cavg(cl, qu, cu), orc_cavg(cl, qv, cv));
      if (ps == 4) d[aoff] = d[4 + aoff] = 255;
      d += 2 * ps;
    }
    /* row end (:8397-8406): the last block's y2 y3 with its own chroma, R,G,B order in the bgr variant too */
    const uint8_t *lb = row + (size_t)(j - 1) * 6;
    if (ps == 4) d[aoff] = d[4 + aoff] = 255;
    c.order = 0;
    put_px(&c, d + coff, lb[4], lb[0], lb[3]);
    put_px(&c, d + ps + coff, lb[5], lb[0], lb[3]);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K5: clamping switch                   reference: src/colourspace.c:1108-1139, :1163-1230, :10929-11090
 * ---------------------------------------------------------------------------------------------- */
void orc_yuv_yuv_tables(uint8_t *yc2u, uint8_t *uvc2u, uint8_t *yu2c, uint8_t *uvu2c) {
  int i;
  for (i = 0; i <= 16; i++) yc2u[i] = 0;
  for (; i < 235; i++) yc2u[i] = (uint8_t)rnd_half_away((i - 16.) * 255. / (235. - 16.));
  for (; i < 256; i++) yc2u[i] = 255;
  for (i = 0; i < 16; i++) uvc2u[i] = 0;
  for (; i < 240; i++) uvc2u[i] = (uint8_t)rnd_half_away((i - 16.) * 255. / (240. - 16.));
  for (; i < 256; i++) uvc2u[i] = 255;
  for (i = 0; i < 256; i++) {
    yu2c[i] = (uint8_t)rnd_half_away((i / 255.) * (235. - 16.) + 16.);
    uvu2c[i] = (uint8_t)rnd_half_away((i / 255.) * (240. - 16.) + 16.);
  }
}
int orc_switch_yuv_clamping(uint8_t *const planes[4], const int rowstrides[4], int palette, int height, int to_unclamped) {
  uint8_t t[4][256];
  orc_yuv_yuv_tables(t[0], t[1], t[2], t[3]);
  const uint8_t *Y = to_unclamped ? t[0] : t[2], *C = to_unclamped ? t[1] : t[3];
  const size_t n = (size_t)height * rowstrides[0];
  uint8_t *p = planes[0];
  switch (palette) {
  /* packed 4:4:4: the reference walks the WHOLE buffer as one Y,U,V,... stream (:10946-10968), so with a rowstride that is
     not a multiple of 3 the byte roles drift from row to row -- kept: role = offset in the buffer mod 3 */
  case 588: for (size_t i = 0; i < n; i++) p[i] = (i % 3 == 0) ? Y[p[i]] : C[p[i]]; break;
  case 589: for (size_t i = 0; i < n; i++) if ((i & 3) != 3) p[i] = ((i & 3) == 0) ? Y[p[i]] : C[p[i]]; break;
  case 564: for (size_t i = 0; i < n; i += 4) { p[i] = C[p[i]]; p[i + 1] = Y[p[i + 1]]; p[i + 2] = C[p[i + 2]]; p[i + 3] = Y[p[i + 3]]; } break;
  case 565: for (size_t i = 0; i < n; i += 4) { p[i] = Y[p[i]]; p[i + 1] = C[p[i + 1]]; p[i + 2] = Y[p[i + 2]]; p[i + 3] = C[p[i + 3]]; } break;
  case 544: case 545: case 522: case 512: case 513: {
    const size_t nc = palette == 522 ? n / 2 : (palette == 512 || palette == 513) ? n / 4 : n;
    for (size_t i = 0; i < n; i++) p[i] = Y[p[i]];
    for (size_t i = 0; i < nc; i++) { planes[1][i] = C[planes[1][i]]; planes[2][i] = C[planes[2][i]]; }
    break;
  }
  default: return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * F8: slide over                          reference: lives-plugins/weed-plugins/slide_over.c:54-146
 * dirn 1 "left to right" .. 4 "bottom to top" (the value sover_init stores in "plugin_direction", :40-51; 0 = random is
 * resolved by the host side); bound = the dividing line, computed in the reference's float / double mix.
 * ---------------------------------------------------------------------------------------------- */
void orc_slide_over(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height,
                    int psize, int transval, int dirn, int mvlower, int mvupper) {
  int bound, j;
  mvlower = !!mvlower; mvupper = !!mvupper;
  switch (dirn) {
  case 3:
    bound = (float)height * (1. - transval / 255.);                                         /* :93 */
    if (mvupper) src1 += (size_t)irow1 * (height - bound);
    for (j = 0; j < bound; j++) {
      memcpy(dst, src1, (size_t)width * psize);
      src1 += irow1;
      if (!mvlower) src2 += irow2;
      dst += orow;
    }
    for (j = bound; j < height; j++) { memcpy(dst, src2, (size_t)width * psize); src2 += irow2; dst += orow; }
    break;
  case 4:
    bound = (float)height * (transval / 255.);                                              /* :109 */
    if (mvlower) src2 += (size_t)irow2 * (height - bound);
    if (!mvupper) src1 += (size_t)irow1 * bound;
    for (j = 0; j < bound; j++) { memcpy(dst, src2, (size_t)width * psize); src2 += irow2; dst += orow; }
    for (j = bound; j < height; j++) { memcpy(dst, src1, (size_t)width * psize); src1 += irow1; dst += orow; }
    break;
  case 1:
    bound = (float)width * (1. - transval / 255.);                                          /* :125 */
    for (j = 0; j < height; j++) {
      memcpy(dst, src1 + (size_t)(width - bound) * psize * mvupper, (size_t)bound * psize);
      memcpy(dst + (size_t)bound * psize, src2 + (size_t)bound * psize * !mvlower, (size_t)(width - bound) * psize);
      src1 += irow1; src2 += irow2; dst += orow;
    }
    break;
  case 2:
    bound = (float)width * (transval / 255.);                                               /* :136 */
    for (j = 0; j < height; j++) {
      memcpy(dst, src2 + (size_t)(width - bound) * psize * mvlower, (size_t)bound * psize);
      memcpy(dst + (size_t)bound * psize, src1 + (size_t)!mvupper * bound * psize, (size_t)(width - bound) * psize);
      src1 += irow1; src2 += irow2; dst += orow;
    }
    break;
  default: break;
  }
}

/* ------------------------------------------------------------------------------------------------
 * F7: geometric transitions             reference: lives-plugins/weed-plugins/multi_transitions.c:86-233
 * ---------------------------------------------------------------------------------------------- */
void orc_transition(int type, const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow,
                    int width, int height, int psize, double amount) {
  const int ihheight = height >> 1;
  float hwidth = (float)width * 0.5f;
  const float hheight = (float)height * 0.5f;
  float maxradsq = 0.f;
  if (type == 1) maxradsq = ((hheight * hheight) + (hwidth * hwidth));                   /* :137, in pixels */
  const int wb = width * psize;                                                            /* :139: from here on width is in bytes */
  hwidth = (float)wb * 0.5f;
  const int ihwidth = wb >> 1;
  const float bf = (float)amount, bfneg = 1.f - bf;
  int xx = 0, yy = 0;
  if (type == 2) {
    xx = (int)(hheight * bf + .5) * irow1;
    yy = (int)(hwidth / (float)psize * bf + .5) * psize;
  }
  for (int i = 0; i < height; i++)
    for (int j = 0; j < wb; j += psize) {
      const uint8_t *from;
      if (type == 0) {
        xx = (int)hwidth * bfneg + .5;
        yy = (int)hheight * bfneg + .5;
        from = (j < xx || j >= (wb - xx) || i < yy || i >= (height - yy)) ? src1 + (size_t)irow1 * i + j : src2 + (size_t)irow2 * i + j;
      } else if (type == 1) {
        const float xxf = (float)(i - ihheight), yyf = (float)(j - ihwidth) / (float)psize;
        from = (sqrt((xxf * xxf + yyf * yyf) / maxradsq) > bf) ? src1 + (size_t)irow1 * i + j : src2 + (size_t)irow2 * i + j;
      } else {
        if (fabsf(i - hheight) / hheight < bf || fabsf(j - hwidth) / hwidth < bf || bf == 1.f) from = src2 + (size_t)irow2 * i + j;
        else from = src1 + (size_t)irow1 * i + j + (j > ihwidth ? -yy : yy) + (i > ihheight ? -xx : xx);
      }
      memmove(dst + (size_t)orow * i + j, from, (size_t)psize);
    }
}

/* ------------------------------------------------------------------------------------------------
 * F6a: softlight                        reference: lives-plugins/weed-plugins/softlight.c:34-47 (sqrti), :62-141
 * Per interior luma sample (the reference's own operand choice, including the two terms that differ from a
 * textbook Sobel: row0 ends with (S[+1][+1] - S[+1][-1]) and row1 ends with the SUM S[+1][+1] + S[+1][-1]):
 *   row0 = (S[+1][-1] - S[-1][-1]) + 2 (S[+1][0] - S[-1][0]) + (S[+1][+1] - S[+1][-1])
 *   row1 = (S[-1][+1] - S[-1][-1]) + 2 (S[0][+1] - S[0][-1]) + (S[+1][+1] + S[+1][-1])
 *   sum  = clamp(((3 * isqrt(row0^2 + row1^2) / 2) * 384) >> 8);  out = clamp((64 * sum + 192 * S[0][0]) >> 8)
 * ---------------------------------------------------------------------------------------------- */
static uint32_t isqrt_u32(uint32_t n) {            /* softlight.c:34-47, digit-by-digit floor square root */
  uint32_t root = 0, rem = n, place = 0x40000000u, tmp;
  while (place > rem) place >>= 2;
  while (place) {
    if (rem >= (tmp = root + place)) { rem -= tmp; root += place << 1; }
    root >>= 1;
    place >>= 2;
  }
  return root;
}
void orc_softlight_y(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int unclamped) {
  const int ymin = unclamped ? 0 : 16, ymax = unclamped ? 255 : 235, scale = 384, mix = 192;
  memcpy(dst, src, (size_t)width);                                                  /* :83 top scanline */
  for (int y = 1; y < height - 1; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    uint8_t *d = dst + (size_t)y * orow;
    d[0] = s[0];                                                                      /* :110 */
    for (int x = 1; x < width - 1; x++) {
      const uint8_t *c = s + x;
      /* `* 2`: the reference writes `<< 1` on differences that can be negative (softlight.c:119-123); same value with gcc, but not UB */
      const int row0 = (c[irow - 1] - c[-irow - 1]) + ((c[irow] - c[-irow]) * 2) + (c[irow + 1] - c[irow - 1]);
      const int row1 = (c[-irow + 1] - c[-irow - 1]) + ((c[1] - c[-1]) * 2) + (c[irow + 1] + c[irow - 1]);
      int sum = (int)(((3 * isqrt_u32((uint32_t)(row0 * row0 + row1 * row1)) / 2) * scale) >> 8);
      sum = sum < ymin ? ymin : sum > ymax ? ymax : sum;
      sum = ((256 - mix) * sum + mix * c[0]) >> 8;
      d[x] = (uint8_t)(sum < ymin ? ymin : sum > ymax ? ymax : sum);
    }
    d[width - 1] = s[width - 1];                                                      /* :131 */
  }
  memcpy(dst + (size_t)(height - 1) * orow, src + (size_t)(height - 1) * irow, (size_t)width);   /* :141 bottom row */
}

/* ------------------------------------------------------------------------------------------------
 * F6b: edge detect                      reference: lives-plugins/weed-plugins/edge.c:93-125 (copywalpha), :129-248
 * Per pass (1 pass for modes 0 / 1; luma then the three colour bytes for mode 2):
 *   l    = luma (calc_luma) or byte (pass - 1) of the pixel           -- for ARGB32 that byte index starts at A
 *   gh   = l[x+1] - l[x-1], gv = l[y+1] - l[y-1]                       (1 <= x < w-1, 1 <= y < h-1)
 *   v0   = gh[y-1] + gh[y] + gh[y+1], v1 = gv[x-1] + gv[x] + gv[x+1]   (2 <= x < w-2, 2 <= y < h-2)
 *   map  = (uint16_t)(sqrtf((float)(v0*v0) + (float)(v1*v1)) * 0.94f); border cells keep their calloc'd 0
 *   Otsu over the 1017-bin histogram in doubles; the running sums bh/nbh/bl/nbl and threshmax/difmax are
 *   function-scope in the reference, i.e. they are NOT reset between the passes of mode 2 -- kept.
 *   pixels with map >= thresh are painted (white / source / one channel to 255), others black on pass 0.
 * ---------------------------------------------------------------------------------------------- */
static void edge_paint(uint8_t *dest, size_t doffs, const uint8_t *src, size_t offs, int red, int green, int blue,
                       int aoffs, int inplace) {   /* copywalpha, edge.c:93-125 */
  if (aoffs == 1) {
    if (!inplace) dest[doffs] = src[offs];
    offs++; doffs++;
  }
  if (red != 3) dest[doffs] = red == 1 ? src[offs] : red == 0 ? 0 : 255;
  if (green != 3) dest[doffs + 1] = green == 1 ? src[offs + 1] : green == 0 ? 0 : 255;
  if (blue != 3) dest[doffs + 2] = blue == 1 ? src[offs + 2] : blue == 0 ? 0 : 255;
  if (aoffs != 0 || inplace) return;
  dest[doffs + 3] = src[offs + 3];
}
void orc_edge(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int pal, int mode,
              int16_t *map16, int inplace) {
  const int psize = (pal == 1 || pal == 2) ? 3 : 4;
  const int order = pal == 1 || pal == 3 ? 0 : pal == 2 || pal == 4 ? 1 : 2;
  const int offs = pal == 5 ? 1 : (pal == 1 || pal == 2) ? -1 : 0;
  const int TMAX = 1017;
  const size_t n = (size_t)width * height;
  uint8_t *mapl = (uint8_t *)calloc(n, 1);
  int16_t *maph = (int16_t *)calloc(n, 2), *mapv = (int16_t *)calloc(n, 2);
  int64_t *pr = (int64_t *)malloc(1024 * sizeof(int64_t));
  uint16_t thresh, threshmax = 0;
  uint64_t bh = 0, bl = 0, nbh = 0, nbl = 0, nn;
  double abl, abh, dif, difmax = 0.;
  if (!luma_ready) build_luma();
  for (int pass = 0; pass < 4; pass++) {
    memset(pr, 0, 8192);
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++)
        mapl[(size_t)y * width + x] = pass == 0 ? luma_of(src + (size_t)y * irow + x * psize, order)
                                                : src[(size_t)y * irow + x * psize + pass - 1];
    for (int y = 1; y < height - 1; y++)
      for (int x = 1; x < width - 1; x++) {
        const size_t i = (size_t)y * width + x;
        maph[i] = (int16_t)(-mapl[i - 1] + mapl[i + 1]);
        mapv[i] = (int16_t)(-mapl[i - width] + mapl[i + width]);
      }
    for (int y = 2; y < height - 2; y++)
      for (int x = 2; x < width - 2; x++) {
        const size_t i = (size_t)y * width + x;
        const int16_t v0 = (int16_t)(maph[i - width] + maph[i] + maph[i + width]);
        const int16_t v1 = (int16_t)(mapv[i - 1] + mapv[i] + mapv[i + 1]);
        const uint16_t val = (uint16_t)(sqrtf((float)(v0 * v0) + (float)(v1 * v1)) * 0.94f);
        map16[i] = (int16_t)val;
        pr[val]++;
        bh += val;
        nbh++;
      }
    for (thresh = 0; thresh < TMAX; thresh++) {                                      /* :186-204 */
      nn = (uint64_t)(pr[thresh] * thresh);
      bl += nn; nbl += (uint64_t)pr[thresh];
      bh -= nn; nbh -= (uint64_t)pr[thresh];
      abh = (double)bh / (double)nbh;
      abl = (double)bl / (double)nbl;
      dif = (double)(nbl * nbh) * (abh - abl) * (abh - abl);
      if (thresh > 0 && dif > difmax) { difmax = dif; threshmax = thresh; }
    }
    thresh = threshmax;
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) {
        const size_t d = (size_t)y * orow + x * psize, s = (size_t)y * irow + x * psize;
        if ((uint16_t)map16[(size_t)y * width + x] >= thresh) {
          if (pass == 0) { if (mode == 1) edge_paint(dst, d, src, s, 2, 2, 2, offs, inplace); else edge_paint(dst, d, src, s, 1, 1, 1, offs, inplace); }
          else if (pass == 1) edge_paint(dst, d, src, s, 2, 3, 3, offs, inplace);
          else if (pass == 2) edge_paint(dst, d, src, s, 3, 2, 3, offs, inplace);
          else edge_paint(dst, d, src, s, 3, 3, 2, offs, inplace);
        } else if (pass == 0) edge_paint(dst, d, src, s, 0, 0, 0, offs, inplace);
      }
    if (mode < 2) break;
  }
  free(mapl); free(maph); free(mapv); free(pr);
}

/* ------------------------------------------------------------------------------------------------
 * F6c: blurzoom                         reference: lives-plugins/weed-plugins/blurzoom.c (ranges in lives_oracle.h)
 * ---------------------------------------------------------------------------------------------- */
struct orc_blurzoom {
  int vw, vh, bw, bh, blocks, ml, mr, threshold, snap_time, snap_interval;
  uint8_t *buf;            /* 2 * bw * bh: feedback plane, then the blur result */
  uint32_t *zx;            /* one 32-bit step mask per block of 32 columns */
  int *zy;                 /* per-row pointer delta */
  int16_t *bg;
  uint8_t *diff;
  uint32_t *snap;
  uint32_t pal[256];
};
orc_blurzoom *orc_blurzoom_new(int width, int height, int palette) {
  orc_blurzoom *z = (orc_blurzoom *)calloc(1, sizeof *z);
  const double RATIO = 0.95;
  z->vw = width; z->vh = height;
  z->blocks = width / 32; z->bw = z->blocks * 32; z->bh = height;                    /* :260-267 */
  z->ml = (width - z->bw) / 2; z->mr = width - z->bw - z->ml;
  z->buf = (uint8_t *)calloc((size_t)z->bw * z->bh * 2, 1);
  z->zx = (uint32_t *)calloc((size_t)z->bw + 1, sizeof(uint32_t));
  z->zy = (int *)calloc((size_t)z->bh + 1, sizeof(int));
  z->bg = (int16_t *)calloc((size_t)width * height, sizeof(int16_t));
  z->diff = (uint8_t *)calloc((size_t)width * height, 4);
  z->snap = (uint32_t *)calloc((size_t)width * height, 4);
  z->threshold = 40 * 7; z->snap_time = 0; z->snap_interval = 3;                     /* :288, :318-319 */
  {                                                                                    /* setTable :106-145 */
    const int HW = z->bw / 2, HH = z->bh / 2;
    int prevptr = (int)(0.5 + RATIO * (-HW) + HW), ptr, tx, ty, xx;
    for (int b = 0; b < z->blocks; b++) {
      uint32_t bits = 0;
      for (int x = 0; x < 32; x++) {
        ptr = (int)(0.5 + RATIO * (b * 32 + x - HW) + HW);
        bits >>= 1;
        if (ptr != prevptr) bits |= 0x80000000u;
        prevptr = ptr;
      }
      z->zx[b] = bits;
    }
    ty = (int)(0.5 + RATIO * (-HH) + HH);
    tx = (int)(0.5 + RATIO * (-HW) + HW);
    xx = (int)(0.5 + RATIO * (z->bw - 1 - HW) + HW);
    z->zy[0] = ty * z->bw + tx;
    prevptr = ty * z->bw + xx;
    for (int y = 1; y < z->bh; y++) {
      ty = (int)(0.5 + RATIO * (y - HH) + HH);
      z->zy[y] = ty * z->bw + tx - prevptr;
      prevptr = ty * z->bw + xx;
    }
  }
  {                                                                                    /* makePalette :201-237 */
    const int COLORS = 32, DELTA = 255 / (32 / 2 - 1);
    uint32_t *P = z->pal;
    for (int i = 0; i < COLORS / 2; i++) {
      if (palette == 3) { P[i] = (uint32_t)(i * DELTA) << 16; P[COLORS * 2 + i] = (uint32_t)(i * DELTA); }
      else { P[i] = (uint32_t)(i * DELTA); P[COLORS * 2 + i] = (uint32_t)(i * DELTA) << 16; }
      P[COLORS + i] = (uint32_t)(i * DELTA) << 8;
    }
    for (int i = 0; i < COLORS / 2; i++) {
      const uint32_t d = (uint32_t)(i * DELTA);
      if (palette == 3) { P[i + COLORS / 2] = (255u << 16) | d << 8 | d; P[COLORS * 2 + i + COLORS / 2] = 255u | d << 16 | d << 8; }
      else { P[i + COLORS / 2] = 255u | d << 16 | d << 8; P[COLORS * 2 + i + COLORS / 2] = (255u << 16) | d << 8 | d; }
      P[COLORS + i + COLORS / 2] = (255u << 8) | d << 16 | d;
    }
    for (int i = 0; i < COLORS; i++) P[COLORS * 3 + i] = (uint32_t)(255 * i / COLORS) * 0x10101u;
    for (int i = 0; i < COLORS * 4; i++) P[i] &= 0xfefeffu;
  }
  return z;
}
void orc_blurzoom_free(orc_blurzoom *z) {
  if (!z) return;
  free(z->buf); free(z->zx); free(z->zy); free(z->bg); free(z->diff); free(z->snap); free(z);
}
int orc_blurzoom_process(orc_blurzoom *z, const uint8_t *src8, int irow_b, uint8_t *dst8, int orow_b, int mode, int pattern) {
  const int vw = z->vw, vh = z->vh, bw = z->bw, bh = z->bh;
  const size_t area = (size_t)bw * bh;
  if ((irow_b & 3) || (orow_b & 3) || z->blocks < 1) return -1;
  if ((mode == 1 || mode == 2) && irow_b != vw * 4) return -1;
  const uint32_t *src = (const uint32_t *)src8;
  uint32_t *dst = (uint32_t *)dst8;
  int irow = irow_b / 4;
  const int orow = orow_b / 4;
  if (mode != 2 || z->snap_time <= 0) {                                               /* :368 */
    for (int y = 0; y < vh; y++)                                                       /* image_bgsubtract_update_y :74-101 */
      for (int x = 0; x < vw; x++) {
        const uint32_t p = src[(size_t)y * irow + x];
        const int R = (int)((p & 0xff0000) >> (16 - 1)), G = (int)((p & 0xff00) >> (8 - 2)), B = (int)(p & 0xff);
        const int v = (R + G + B) - (int)z->bg[(size_t)y * vw + x];
        z->bg[(size_t)y * vw + x] = (int16_t)(R + G + B);
        z->diff[(size_t)y * vw + x] = (uint8_t)(((v + z->threshold) >> 24) | ((z->threshold - v) >> 24));
      }
    if (mode == 0 || z->snap_time <= 0) {
      for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) z->buf[(size_t)y * bw + x] |= z->diff[(size_t)y * vw + z->ml + x] >> 3;
      if (mode == 1 || mode == 2)
        for (int y = 0; y < vh; y++) memcpy(z->snap + (size_t)y * vw, src + (size_t)y * irow, (size_t)vw * 4);
    }
  }
  {                                                                                    /* blur :149-168 */
    const uint8_t *p = z->buf + bw + 1;
    uint8_t *q = z->buf + area + bw + 1;
    for (int y = bh - 2; y > 0; y--) {
      for (int x = bw - 2; x > 0; x--) {
        uint8_t v;
        if ((v = (uint8_t)((p[-bw] + p[-1] + p[1] + p[bw]) / 4 - 1)) == 255) v = 0;
        *(q++) = v;
        p++;
      }
      p += 2; q += 2;
    }
  }
  {                                                                                    /* zoom :171-192 */
    const uint8_t *p = z->buf + area;
    uint8_t *q = z->buf;
    for (int y = 0; y < bh; y++) {
      p += z->zy[y];
      for (int b = 0; b < z->blocks; b++) {
        uint32_t dx = z->zx[b];
        for (int x = 0; x < 32; x++) { p += (dx & 1); *q++ = *p; dx >>= 1; }
      }
    }
  }
  {
    const uint32_t *s = (mode == 1 || mode == 2) ? z->snap : src;                     /* :391-396: keeps the source's row padding */
    const uint8_t *p = z->buf;
    const int ipad = irow - vw, opad = orow - vw;
    uint32_t *d = dst;
    for (int y = 0; y < vh; y++) {
      for (int x = 0; x < z->ml; x++) *d++ = *s++;
      for (int x = 0; x < bw; x++) {
        uint32_t a = *s & 0xfefeff, b = z->pal[32 * pattern + *p++];
        a += b;
        b = a & 0x10101;
        *d++ = (*s++ & 0xff000000u) | ((a | (b - (b >> 8))) & 0xffffffu);
      }
      for (int x = 0; x < z->mr; x++) *d++ = *s++;
      s += ipad; d += opad;
    }
  }
  if (mode == 1 || mode == 2) { z->snap_time--; if (z->snap_time < 0) z->snap_time = z->snap_interval; }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * C1: compositor                        reference: lives-plugins/weed-plugins/gdk/compositor.c:120-125, :167-189, :288-293
 * ---------------------------------------------------------------------------------------------- */
void orc_composite(uint8_t *dst, int orow, int owidth, int oheight, int psize, int is_bgr, const int bgcol[3],
                   const orc_comp_layer *layers, int nlayers, int revz) {
  const int r = is_bgr ? 2 : 0, b = is_bgr ? 0 : 2;
  for (int y = 0; y < oheight; y++)                                         /* :171-178 */
    for (int x = 0; x < owidth; x++) {
      uint8_t *d = dst + (size_t)y * orow + x * psize;
      d[0] = (uint8_t)bgcol[r]; d[1] = (uint8_t)bgcol[1]; d[2] = (uint8_t)bgcol[b];
      if (psize == 4) d[3] = 0xFF;
    }
  const int starti = revz ? 0 : nlayers - 1, endi = revz ? nlayers : -1, stepi = revz ? 1 : -1;   /* :181-189 */
  for (int z = starti; z != endi; z += stepi) {
    const orc_comp_layer *L = &layers[z];
    if (!L->src) continue;
    const double alpha = L->alpha, invalpha = 1. - alpha;
    for (int y = L->offs_y; y < oheight && y < L->offs_y + L->height; y++)
      for (int x = L->offs_x; x < owidth && x < L->offs_x + L->width; x++) {
        if (x < 0 || y < 0) continue;
        uint8_t *d = dst + (size_t)y * orow + x * psize;
        const uint8_t *sp = L->src + (size_t)(y - L->offs_y) * L->irow + (x - L->offs_x) * psize;
        d[0] = (uint8_t)(d[0] * invalpha + sp[0] * alpha);                  /* paint_pixel :120-125 */
        d[1] = (uint8_t)(d[1] * invalpha + sp[1] * alpha);
        d[2] = (uint8_t)(d[2] * invalpha + sp[2] * alpha);
      }
  }
}

/* ------------------------------------------------------------------------------------------------
 * R1: resize -- UNPINNED.  The reference hands this to FFmpeg libswscale (src/colourspace.c:14711,
 * flags :14991-14997), which is neither vendored nor version-pinned.  Spec "lgpu-polyphase-v1"
 * (DESIGN.md): separable polyphase FIR, horizontal then vertical, Q14 coefficients, 15-bit
 * intermediate with 7 fractional bits, edge replicate, support widened by the downscale ratio.
 * ---------------------------------------------------------------------------------------------- */
static double kern_eval(int kernel, double x) {
  x = fabs(x);
  if (kernel == 0) {                     /* triangle */
    return x < 1. ? 1. - x : 0.;
  } else if (kernel == 1) {              /* cubic, B = 0, C = 0.6 */
    const double B = 0., C = 0.6;
    if (x < 1.) return ((12. - 9. * B - 6. * C) * x * x * x + (-18. + 12. * B + 6. * C) * x * x + (6. - 2. * B)) / 6.;
    if (x < 2.) return ((-B - 6. * C) * x * x * x + (6. * B + 30. * C) * x * x + (-12. * B - 48. * C) * x + (8. * B + 24. * C)) / 6.;
    return 0.;
  } else {                               /* lanczos, a = 3 */
    if (x < 1e-12) return 1.;
    if (x >= 3.) return 0.;
    const double px = M_PI * x;
    return 3. * sin(px) * sin(px / 3.) / (px * px);
  }
}
static double kern_radius(int kernel) { return kernel == 0 ? 1. : kernel == 1 ? 2. : 3.; }

/* kernel: 0 triangle, 1 cubic(0, 0.6), 2 lanczos3.  pos[dstn], coef[dstn * ntaps].  returns 0 / -1 */
int orc_make_filter(int srcn, int dstn, int kernel, int *ntaps_out, int32_t *pos, int16_t *coef, int maxtaps) {
  const double ratio = (double)srcn / (double)dstn, scale = ratio > 1. ? ratio : 1.;
  const double support = kern_radius(kernel) * scale;
  const int ntaps = (int)ceil(2. * support);
  double w[256];
  if (ntaps > maxtaps || ntaps > 256) return -1;
  *ntaps_out = ntaps;
  for (int i = 0; i < dstn; i++) {
    const double centre = ((double)i + 0.5) * ratio - 0.5;
    const int left = (int)floor(centre - support) + 1;
    double sum = 0.;
    int q[256], qs = 0, big = 0;
    for (int j = 0; j < ntaps; j++) { w[j] = kern_eval(kernel, ((double)(left + j) - centre) / scale); sum += w[j]; }
    for (int j = 0; j < ntaps; j++) {
      q[j] = (int)floor(w[j] / sum * 16384. + 0.5);
      qs += q[j];
      if (q[j] > q[big]) big = j;
    }
    q[big] += 16384 - qs;                /* rows sum to exactly 1.0 in Q14 */
    pos[i] = left;
    for (int j = 0; j < ntaps; j++) coef[(size_t)i * ntaps + j] = (int16_t)q[j];
  }
  return 0;
}

static int kernel_for(int interp, int upscale) {
  /* LIVES_INTERP_BEST -> bicubic when shrinking, lanczos when enlarging; NORMAL/FAST -> bilinear
     (src/colourspace.c:14991-14997; GdkInterpType values src/widget-helper-gtk.h:1136-1138) */
  if (interp == ORC_INTERP_HYPER) return upscale ? 2 : 1;
  return 0;
}

int orc_resize(const uint8_t *src, int irow, int sw, int sh, uint8_t *dst, int orow, int dw, int dh,
               int psize, int interp) {
  const int kernel = kernel_for(interp, dw > sw || dh > sh);
  int nth = 0, ntv = 0, rc = -1;
  int32_t *hpos = malloc(sizeof(int32_t) * dw), *vpos = malloc(sizeof(int32_t) * dh);
  int16_t *hco = malloc(sizeof(int16_t) * (size_t)dw * 256), *vco = malloc(sizeof(int16_t) * (size_t)dh * 256);
  int16_t *tmp = malloc(sizeof(int16_t) * (size_t)sh * dw * psize);
  if (!hpos || !vpos || !hco || !vco || !tmp) goto out;
  if (orc_make_filter(sw, dw, kernel, &nth, hpos, hco, 256)) goto out;
  if (orc_make_filter(sh, dh, kernel, &ntv, vpos, vco, 256)) goto out;
  for (int y = 0; y < sh; y++) {                      /* horizontal pass */
    const uint8_t *s = src + (size_t)y * irow;
    int16_t *t = tmp + (size_t)y * dw * psize;
    for (int x = 0; x < dw; x++)
      for (int c = 0; c < psize; c++) {
        int32_t acc = 0;
        for (int j = 0; j < nth; j++) {
          int sx = hpos[x] + j;
          sx = sx < 0 ? 0 : sx >= sw ? sw - 1 : sx;
          acc += (int32_t)hco[(size_t)x * nth + j] * s[sx * psize + c];
        }
        acc = (acc + 64) >> 7;
        t[x * psize + c] = (int16_t)(acc < -32768 ? -32768 : acc > 32767 ? 32767 : acc);
      }
  }
  for (int y = 0; y < dh; y++) {                      /* vertical pass */
    uint8_t *d = dst + (size_t)y * orow;
    for (int x = 0; x < dw * psize; x++) {
      int32_t acc = 0;
      for (int j = 0; j < ntv; j++) {
        int sy = vpos[y] + j;
        sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
        acc += (int32_t)vco[(size_t)y * ntv + j] * tmp[(size_t)sy * dw * psize + x];
      }
      acc = (acc + (1 << 20)) >> 21;
      d[x] = clamp_int_0_255(acc);
    }
  }
  rc = 0;
out:
  free(hpos); free(vpos); free(hco); free(vco); free(tmp);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * B1: 5x5 gaussian -- UNPINNED (no reference loop).  Build-defined: [1 4 6 4 1] per axis,
 * edge replicate, exact 12-bit row sums, one rounding: (sum + 128) >> 8.  All channels incl. alpha.
 * ---------------------------------------------------------------------------------------------- */
void orc_gauss5(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize) {
  static const int kw[5] = {1, 4, 6, 4, 1};
  const int rowlen = width * psize;
  uint16_t *tmp = malloc(sizeof(uint16_t) * (size_t)height * rowlen);
  for (int y = 0; y < height; y++) {
    const uint8_t *s = src + (size_t)y * irow;
    for (int x = 0; x < width; x++)
      for (int c = 0; c < psize; c++) {
        int acc = 0;
        for (int j = -2; j <= 2; j++) {
          int sx = x + j; sx = sx < 0 ? 0 : sx >= width ? width - 1 : sx;
          acc += kw[j + 2] * s[sx * psize + c];
        }
        tmp[(size_t)y * rowlen + x * psize + c] = (uint16_t)acc;
      }
  }
  for (int y = 0; y < height; y++) {
    uint8_t *d = dst + (size_t)y * orow;
    for (int x = 0; x < rowlen; x++) {
      int acc = 0;
      for (int j = -2; j <= 2; j++) {
        int sy = y + j; sy = sy < 0 ? 0 : sy >= height ? height - 1 : sy;
        acc += kw[j + 2] * tmp[(size_t)sy * rowlen + x];
      }
      d[x] = (uint8_t)((acc + 128) >> 8);
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * the headline chain = composition of the single ops (BASELINE.json config 5 / north_star chain)
 * ---------------------------------------------------------------------------------------------- */
int orc_chain(const uint8_t *src, int irow, int sw, int sh, const uint8_t *layer2, int irow2,
              uint8_t *dst, int orow, int dw, int dh, int swap_rb, int interp, int do_blur, int bf,
              const uint8_t *lut8) {
  uint8_t *conv = malloc((size_t)sw * 4 * sh), *rs = malloc((size_t)dw * 4 * dh), *bl = NULL;
  int rc = -1;
  if (!conv || !rs) goto out;
  if (swap_rb) orc_swizzle(ORC_SWAP3POSTALPHA, 0, src, irow, conv, sw * 4, sw, sh, NULL);
  else for (int y = 0; y < sh; y++) memcpy(conv + (size_t)y * sw * 4, src + (size_t)y * irow, (size_t)sw * 4);
  if (interp & 0x100) {      /* LGPU_INTERP_PIXBUF: the resize stage is the reference's gdk-pixbuf body (orc_pixbuf.c), 4 channels with alpha */
    if (orc_pixbuf_scale(conv, sw * 4, sw, sh, rs, dw * 4, dw, dh, 4, interp & 0xFF)) goto out;
  } else if (orc_resize(conv, sw * 4, sw, sh, rs, dw * 4, dw, dh, 4, interp)) goto out;
  if (do_blur) {
    bl = malloc((size_t)dw * 4 * dh);
    if (!bl) goto out;
    orc_gauss5(rs, dw * 4, bl, dw * 4, dw, dh, 4);
  }
  /* blend in place on the track (host "inplace" channel): dst alpha = track alpha */
  { uint8_t *trk = bl ? bl : rs;
    orc_blend_chroma(trk, dw * 4, layer2, irow2, trk, dw * 4, dw, dh, 4, 0, bf);
    if (lut8) orc_gamma_apply(trk, dw * 4, dw, dh, 4, 0, lut8);
    for (int y = 0; y < dh; y++) memcpy(dst + (size_t)y * orow, trk + (size_t)y * dw * 4, (size_t)dw * 4); }
  rc = 0;
out:
  free(conv); free(rs); free(bl);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * K5b: YUV -> YUV repacks                 reference: src/colourspace.c, the functions cited per case
 * Palettes are the WEED_PALETTE_* numbers; width is in PIXELS (the reference's dispatcher passes pixels, and macropixels for a
 * UYVY / YUYV source, :13130-13330).  Returns 0, or -1 for a pair / layout this restatement does not take: the reference
 * function for it overruns its buffers, mixes up strides or leaves the result depending on the destination's previous
 * contents (list in DESIGN.md "YUV -> YUV").  Bytes the reference does not write are not written here either.
 * ---------------------------------------------------------------------------------------------- */
enum { P_420 = 512, P_YV12 = 513, P_422 = 522, P_444 = 544, P_4444 = 545, P_UYVY = 564, P_YUYV = 565, P_888 = 588, P_8888 = 589 };

static void copy_rows(const uint8_t *s, int irow, uint8_t *d, int orow, int nbytes, int rows) {
  if (s == d) return;
  for (int y = 0; y < rows; y++) memcpy(d + (size_t)y * orow, s + (size_t)y * irow, (size_t)nbytes);
}

/* ---- YUV411 <-> the other YUV palettes (src/colourspace.c:7755-7796, :7973-8033, :8272-8303, :8622-9146, :9148-9196; dispatcher :13024-13029,
 * :13223-13228, :13323-13328, :13416-13421, :13508-13513, :13636-13641, :13743-13748, :13793-13846).  `width` is in PIXELS (a multiple of 4).
 * Every one of these walks its 4:1:1 side -- and, where it has no rowstride argument, the other side too -- as one compact stream from the start of
 * the plane, whatever rowstride the layer has: restated the same way (a strided destination "drifts" exactly like the reference's).  Bytes the
 * reference does not write are not written here.  Macropixel bytes: u2 y0 y1 v2 y2 y3. */
enum { P_411 = 595 };
/* chroma of output pixel p (0 .. 4 wm - 1) of one row: the three-deep average cascade between neighbouring blocks (:8650-8716) */
static int c411_fine(int cl, const uint8_t *row, int wm, int p, int off) {
  if (p < 2) return row[off];
  if (p >= 4 * wm - 2) return row[(size_t)(wm - 1) * 6 + off];
  const int j = (p + 2) >> 2, k = (p + 2) & 3;
  const int pu = row[(size_t)(j - 1) * 6 + off], cu = row[(size_t)j * 6 + off];
  const int h = orc_cavg(cl, pu, cu), q = orc_cavg(cl, h, k < 2 ? pu : cu);
  return orc_cavg(cl, q, (k & 1) ? cu : pu);
}
/* chroma of output macropixel m (0 .. 2 wm - 1): the two-deep form of the 4:2:2 targets (:8860-8895, :9001-9024) */
static int c411_half(int cl, const uint8_t *row, int wm, int m, int off) {
  if (m == 0) return row[off];
  if (m == 2 * wm - 1) return row[(size_t)(wm - 1) * 6 + off];
  const int j = (m + 1) >> 1, k = (m + 1) & 1;
  const int pu = row[(size_t)(j - 1) * 6 + off], cu = row[(size_t)j * 6 + off];
  return orc_cavg(cl, orc_cavg(cl, pu, cu), k ? cu : pu);
}
static int y411(const uint8_t *row, int p) { static const int yo[4] = {1, 2, 4, 5}; return row[(size_t)(p >> 2) * 6 + yo[p & 3]]; }

static int orc_yuv411_repack(int in_pal, int out_pal, const uint8_t *const src[4], const int irow[4], uint8_t *const dst[4], int width, int height, int cl) {
  if (width < 4 || (width & 3) || height < 1) return -1;
  const int wm = width >> 2;
  if (in_pal == P_411) {
    const uint8_t *s = src[0];
    if (out_pal == P_888 || out_pal == P_8888) {                 /* convert_yuv411_to_yuv888_frame :8622-8736 */
      const int ps = out_pal == P_8888 ? 4 : 3;
      for (int r = 0; r < height; r++) {
        const uint8_t *row = s + (size_t)r * wm * 6;
        uint8_t *d = dst[0] + (size_t)r * width * ps;
        for (int p = 0; p < width; p++) {
          d[p * ps] = (uint8_t)y411(row, p); d[p * ps + 1] = (uint8_t)c411_fine(cl, row, wm, p, 0); d[p * ps + 2] = (uint8_t)c411_fine(cl, row, wm, p, 3);
          if (ps == 4) d[p * ps + 3] = 255;
        }
      }
      return 0;
    }
    if (out_pal == P_444 || out_pal == P_4444) {                 /* convert_yuv411_to_yuvp_frame :8739-8838: the second luma of every pair but the last is the first one again */
      for (int r = 0; r < height; r++) {
        const uint8_t *row = s + (size_t)r * wm * 6;
        const size_t o = (size_t)r * width;
        for (int p = 0; p < width; p++) {
          dst[0][o + p] = (uint8_t)y411(row, p >= width - 2 ? p : (p & ~1));
          dst[1][o + p] = (uint8_t)c411_fine(cl, row, wm, p, 0); dst[2][o + p] = (uint8_t)c411_fine(cl, row, wm, p, 3);
          if (out_pal == P_4444) dst[3][o + p] = 255;
        }
      }
      return 0;
    }
    if (out_pal == P_UYVY || out_pal == P_YUYV) {                /* :8842-8979: inner macropixels carry one luma twice */
      for (int r = 0; r < height; r++) {
        const uint8_t *row = s + (size_t)r * wm * 6;
        uint8_t *d = dst[0] + (size_t)r * width * 2;
        for (int m = 0; m < 2 * wm; m++) {
          const int inner = m > 0 && m < 2 * wm - 1;
          const int ya = y411(row, 2 * m), yb = inner ? ya : y411(row, 2 * m + 1);
          const int u = c411_half(cl, row, wm, m, 0), v = c411_half(cl, row, wm, m, 3);
          if (out_pal == P_UYVY) { d[4 * m] = (uint8_t)u; d[4 * m + 1] = (uint8_t)ya; d[4 * m + 2] = (uint8_t)v; d[4 * m + 3] = (uint8_t)yb; }
          else { d[4 * m] = (uint8_t)ya; d[4 * m + 1] = (uint8_t)u; d[4 * m + 2] = (uint8_t)yb; d[4 * m + 3] = (uint8_t)v; }
        }
      }
      return 0;
    }
    if (out_pal == P_422) {                                      /* convert_yuv411_to_yuv422_frame :8982-9036 */
      for (int r = 0; r < height; r++) {
        const uint8_t *row = s + (size_t)r * wm * 6;
        for (int p = 0; p < width; p++) dst[0][(size_t)r * width + p] = (uint8_t)y411(row, p);
        for (int m = 0; m < 2 * wm; m++) {
          dst[1][(size_t)r * 2 * wm + m] = (uint8_t)c411_half(cl, row, wm, m, 0);
          dst[2][(size_t)r * 2 * wm + m] = (uint8_t)c411_half(cl, row, wm, m, 3);
        }
      }
      return 0;
    }
    if (out_pal == P_420 || out_pal == P_YV12) {
      /* convert_yuv411_to_yuv420_frame :9039-9146: after an even row the chroma pointers go back to where they were, and on the odd row every
         sample is averaged into *d_u WITHOUT advancing it: all rows land in chroma row 0, whose first sample collects the whole odd row */
      uint8_t *du = dst[out_pal == P_YV12 ? 2 : 1], *dv = dst[out_pal == P_YV12 ? 1 : 2];
      for (int r = 0; r < height; r++) {
        const uint8_t *row = s + (size_t)r * wm * 6;
        for (int p = 0; p < width; p++) dst[0][(size_t)r * width + p] = (uint8_t)y411(row, p);
        for (int m = 0; m < 2 * wm; m++) {
          const int u = c411_half(cl, row, wm, m, 0), v = c411_half(cl, row, wm, m, 3);
          if (!(r & 1)) { du[m] = (uint8_t)u; dv[m] = (uint8_t)v; }
          else { du[0] = (uint8_t)orc_cavg(cl, du[0], u); dv[0] = (uint8_t)orc_cavg(cl, dv[0], v); }
        }
      }
      return 0;
    }
    return -1;
  }
  if (out_pal != P_411) return -1;
  uint8_t *d = dst[0];
  if (in_pal == P_444 || in_pal == P_4444) {
    /* convert_yuvp_to_yuv411_frame :7755-7796: the destination pointer is never advanced, so every macropixel is written over the FIRST one and the
       frame ends up holding only the last one there; the Y pointer never skips the row padding either */
    const int r = height - 1, j = wm - 1;
    const uint8_t *sy = src[0] + (size_t)r * width + 4 * j, *su = src[1] + (size_t)r * irow[0] + 4 * j, *sv = src[2] + (size_t)r * irow[0] + 4 * j;
    d[0] = (uint8_t)orc_cavg(cl, orc_cavg(cl, su[0], su[1]), orc_cavg(cl, su[2], su[3]));
    d[1] = sy[0]; d[2] = sy[1];
    d[3] = (uint8_t)orc_cavg(cl, orc_cavg(cl, sv[0], sv[1]), orc_cavg(cl, sv[2], sv[3]));
    d[4] = sy[2]; d[5] = sy[3];
    return 0;
  }
  if (in_pal == P_UYVY || in_pal == P_YUYV) {                    /* :7973-8033: no rowstride argument, the source is one stream */
    const int uo = in_pal == P_UYVY ? 0 : 1, vo = in_pal == P_UYVY ? 2 : 3, ya = in_pal == P_UYVY ? 1 : 0, yb = in_pal == P_UYVY ? 3 : 2;
    for (size_t k = 0; k < (size_t)wm * height; k++) {
      const uint8_t *a = src[0] + k * 8, *b = a + 4;
      uint8_t *m = d + k * 6;
      m[0] = (uint8_t)orc_cavg(cl, a[uo], b[uo]); m[1] = a[ya]; m[2] = a[yb];
      m[3] = (uint8_t)orc_cavg(cl, a[vo], b[vo]); m[4] = b[ya]; m[5] = b[yb];
    }
    return 0;
  }
  if (in_pal == P_888 || in_pal == P_8888) {
    /* convert_yuv888_to_yuv411_frame :8272-8303: the end pointer is width * height BYTES past the start, tested once per row: only the rows that
       begin before it are converted; chroma is the plain mean of the four samples (no table) */
    const int ips = in_pal == P_8888 ? 4 : 3;
    for (int r = 0; (size_t)r * irow[0] < (size_t)width * height && r < height; r++)
      for (int j = 0; j < wm; j++) {
        const uint8_t *p = src[0] + (size_t)r * irow[0] + (size_t)4 * j * ips;
        uint8_t *m = d + ((size_t)r * wm + j) * 6;
        m[0] = (uint8_t)((p[1] + p[ips + 1] + p[2 * ips + 1] + p[3 * ips + 1]) >> 2);
        m[1] = p[0]; m[2] = p[ips];
        m[3] = (uint8_t)((p[2] + p[ips + 2] + p[2 * ips + 2] + p[3 * ips + 2]) >> 2);
        m[4] = p[2 * ips]; m[5] = p[3 * ips];
      }
    return 0;
  }
  if (in_pal == P_420 || in_pal == P_YV12 || in_pal == P_422) {
    /* convert_yuv420_to_yuv411_frame :9148-9196 (compact planes): chroma = average of two neighbouring samples of the row's chroma row; for 4:2:0 every
       odd row below the last is then averaged with the row that follows it (:9179-9182) */
    const int is422 = in_pal == P_422, hw = width >> 1;
    const uint8_t *pu = src[1], *pv = src[2];              /* a YVU420P layer arrives with its chroma pointers swapped by the caller, like every 4:2:0 source */
    for (int r = 0; r < height; r++)
      for (int j = 0; j < wm; j++) {
        const uint8_t *sy = src[0] + (size_t)r * width + 4 * j;
        const size_t c0 = (size_t)(is422 ? r : r >> 1) * hw + 2 * j;
        int u = orc_cavg(cl, pu[c0], pu[c0 + 1]), v = orc_cavg(cl, pv[c0], pv[c0 + 1]);
        if (!is422 && (r & 1) && r + 1 < height) {
          const size_t c1 = (size_t)((r + 1) >> 1) * hw + 2 * j;
          u = orc_cavg(cl, u, orc_cavg(cl, pu[c1], pu[c1 + 1])); v = orc_cavg(cl, v, orc_cavg(cl, pv[c1], pv[c1 + 1]));
        }
        uint8_t *m = d + ((size_t)r * wm + j) * 6;
        m[0] = (uint8_t)u; m[1] = sy[0]; m[2] = sy[1]; m[3] = (uint8_t)v; m[4] = sy[2]; m[5] = sy[3];
      }
    return 0;
  }
  return -1;
}

int orc_yuv_repack(int in_pal, int out_pal, const uint8_t *const src[4], const int irow[4], uint8_t *const dst[4], const int orow[4],
                   int width, int height, int clamping_unclamped, int sampling) {
  const int cl = !clamping_unclamped;               /* set_conversion_arrays(clamping, YCBCR): cavg = cavgc when clamped */
  if (in_pal == P_411 || out_pal == P_411) return orc_yuv411_repack(in_pal, out_pal, src, irow, dst, width, height, cl);
  const int in444 = (in_pal == P_444 || in_pal == P_4444), in420 = (in_pal == P_420 || in_pal == P_YV12);
  const int inpk422 = (in_pal == P_UYVY || in_pal == P_YUYV);
  if (width < 1 || height < 1) return -1;

  /* OWN SPECIFICATION (docs/SPECS.md, "evident intent"), not a restatement: convert_yuv422p_to_uyvy_frame / _yuyv_frame (:6442-6497) walk `width` macropixels per
     row -- twice the row -- and overrun every buffer, so there is no reference behaviour to follow.  What the function's own comments and its 4:2:0 sibling
     (:7100-7160) say it means: macropixel k of row y = (U[y][k], Y[y][2k], V[y][k], Y[y][2k + 1]) (UYVY; YUYV: Y U Y V), every plane with its own rowstride. */
  if (in_pal == P_422 && (out_pal == P_UYVY || out_pal == P_YUYV)) {
    if (width & 1) return -1;
    for (int y = 0; y < height; y++) {
      uint8_t *d = dst[0] + (size_t)y * (size_t)((orow[0] / 4) * 4);
      for (int k = 0; k < width / 2; k++) {
        const uint8_t y0 = src[0][(size_t)y * irow[0] + 2 * k], y1 = src[0][(size_t)y * irow[0] + 2 * k + 1], u = src[1][(size_t)y * irow[1] + k], v = src[2][(size_t)y * irow[2] + k];
        if (out_pal == P_UYVY) { d[4 * k] = u; d[4 * k + 1] = y0; d[4 * k + 2] = v; d[4 * k + 3] = y1; }
        else { d[4 * k] = y0; d[4 * k + 1] = u; d[4 * k + 2] = y1; d[4 * k + 3] = v; }
      }
    }
    return 0;
  }
  /* convert_combineplanes_frame :7593-7641 -- both of its branches write the same bytes */
  if (in444 && (out_pal == P_888 || out_pal == P_8888)) {
    const int ops = out_pal == P_8888 ? 4 : 3;
    for (int y = 0; y < height; y++) {
      uint8_t *d = dst[0] + (size_t)y * orow[0];
      for (int x = 0; x < width; x++) {
        d[x * ops + 0] = src[0][(size_t)y * irow[0] + x];
        d[x * ops + 1] = src[1][(size_t)y * irow[0] + x];            /* one irowstride for all planes (:7624-7640) */
        d[x * ops + 2] = src[2][(size_t)y * irow[0] + x];
        /* the alpha pointer is never moved past the row padding (:7634-7638): the plane is walked as if it were compact */
        if (ops == 4) d[x * ops + 3] = in_pal == P_4444 ? src[3][(size_t)y * width + x] : 255;
      }
    }
    return 0;
  }
  /* convert_splitplanes_frame :9198-9257 -- only 888 -> 444P: with a source or destination alpha the strided branch
     walks the alpha plane / the source with the wrong step (:9229-9233, :9236-9243) */
  if (in_pal == P_888 && out_pal == P_444) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++)
        for (int p = 0; p < 3; p++) dst[p][(size_t)y * orow[p] + x] = src[0][(size_t)y * irow[0] + x * 3 + p];
    return 0;
  }
  /* convert_yuvap_to_yuvp_frame / convert_yuvp_to_yuvap_frame :7643-7688 (one rowstride in, one out) */
  if (in444 && (out_pal == P_444 || out_pal == P_4444) && in_pal != out_pal) {
    for (int p = 0; p < 3; p++) copy_rows(src[p], irow[0], dst[p], orow[0], orow[0] == irow[0] ? irow[0] : width, height);
    if (out_pal == P_4444) memset(dst[3], 255, (size_t)orow[0] * height);
    return 0;
  }
  /* convert_addpost_frame / convert_delpost_frame (K1 family, :9709-9837 / :10026-10113) without a LUT */
  if (in_pal == P_888 && out_pal == P_8888) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) {
        memcpy(dst[0] + (size_t)y * orow[0] + x * 4, src[0] + (size_t)y * irow[0] + x * 3, 3);
        dst[0][(size_t)y * orow[0] + x * 4 + 3] = 255;
      }
    return 0;
  }
  if (in_pal == P_8888 && out_pal == P_888) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) memcpy(dst[0] + (size_t)y * orow[0] + x * 3, src[0] + (size_t)y * irow[0] + x * 4, 3);
    return 0;
  }
  /* convert_swab_frame :10517-10575: in place in the reference (dst may equal src) */
  if (inpk422 && (out_pal == P_UYVY || out_pal == P_YUYV) && in_pal != out_pal) {
    if (width & 1) return -1;
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width * 2; x += 2) {
        const uint8_t a = src[0][(size_t)y * irow[0] + x], b = src[0][(size_t)y * irow[0] + x + 1];
        dst[0][(size_t)y * orow[0] + x] = b; dst[0][(size_t)y * orow[0] + x + 1] = a;
      }
    return 0;
  }
  /* convert_quad_chroma_packed :10715-10808 (4:2:0 planar -> YUV888 / YUVA8888) and convert_double_chroma_packed :10811-10873 (4:2:2 planar -> the same).
     Chroma is supersampled horizontally from the neighbouring samples (JPEG / default siting: plain averages; other sitings: the 3:1 / 1:3 forms);
     the second pixel of a row's last pair reads the sample one past the chroma row (the next row's first; for the plane's last row one past the plane:
     clamped to the plane here, the tests mask that pixel when the planes are compact).
     4:2:0: even rows are computed, on every odd row i >= 3 the chroma of row i - 2 becomes the mean of its even neighbours; the LAST odd row's chroma and
     the alpha of all odd rows are never written (the caller's zeroed frame shows through).  Even heights only: with an odd one the trailing loop (:10798-10807)
     reads the row after the frame.  4:2:2: the second pixel of every pair never gets its alpha byte (:10857-10870). */
  if ((in420 || in_pal == P_422) && (out_pal == P_888 || out_pal == P_8888)) {
    const int ps = out_pal == P_8888 ? 4 : 3, jpeg = sampling == 0, is420 = in420;
    const int hw = width >> 1, crows = is420 ? height >> 1 : height;
    if ((width & 1) || (is420 && ((height & 1) || height < 2))) return -1;
    const size_t ulast = (size_t)irow[1] * crows - 1, vlast = (size_t)irow[2] * crows - 1;
    #define CH_U(cr, k) src[1][((size_t)(cr) * irow[1] + (k)) > ulast ? ulast : ((size_t)(cr) * irow[1] + (k))]
    #define CH_V(cr, k) src[2][((size_t)(cr) * irow[2] + (k)) > vlast ? vlast : ((size_t)(cr) * irow[2] + (k))]
    for (int i = 0; i < height; i++) {
      uint8_t *d = dst[0] + (size_t)i * orow[0];
      const uint8_t *sy = src[0] + (size_t)i * irow[0];
      if (is420 && (i & 1)) {
        for (int x = 0; x < width; x++) d[x * ps] = sy[x];
        continue;
      }
      const int cr = is420 ? i >> 1 : i;
      for (int k = 0; k < hw; k++) {
        const int u0 = CH_U(cr, k), v0 = CH_V(cr, k), u1 = CH_U(cr, k + 1), v1 = CH_V(cr, k + 1);
        uint8_t *p = d + (size_t)2 * k * ps;
        p[0] = sy[2 * k];
        if (k > 0) {
          const int um = CH_U(cr, k - 1), vm = CH_V(cr, k - 1);
          p[1] = (uint8_t)(jpeg ? orc_cavg(cl, um, u0) : orc_cavg(cl, um, orc_cavg(cl, um, u0)));          /* avg_chroma_3_1f */
          p[2] = (uint8_t)(jpeg ? orc_cavg(cl, vm, v0) : orc_cavg(cl, orc_cavg(cl, vm, v0), v0));          /* avg_chroma_1_3f */
        } else { p[1] = (uint8_t)u0; p[2] = (uint8_t)v0; }
        if (ps == 4) p[3] = 255;
        p += ps;
        p[0] = sy[2 * k + 1];
        if (is420) {
          p[1] = (uint8_t)(jpeg ? orc_cavg(cl, u0, u1) : orc_cavg(cl, orc_cavg(cl, u0, u1), u1));          /* 1_3 for U, 3_1 for V (:10756-10757) */
          p[2] = (uint8_t)(jpeg ? orc_cavg(cl, v0, v1) : orc_cavg(cl, v0, orc_cavg(cl, v0, v1)));
          if (ps == 4) p[3] = 255;
        } else {
          p[1] = (uint8_t)(jpeg ? orc_cavg(cl, u0, u1) : orc_cavg(cl, u0, orc_cavg(cl, u0, u1)));          /* the same 3_1 / 1_3 as the first pixel (:10862-10863) */
          p[2] = (uint8_t)(jpeg ? orc_cavg(cl, v0, v1) : orc_cavg(cl, orc_cavg(cl, v0, v1), v1));
        }
      }
    }
    if (is420)
      for (int i = 1; i + 2 < height; i += 2) {                   /* row i = mean of rows i - 1 and i + 1, written while row i + 2 is walked (:10766-10777) */
        uint8_t *d = dst[0] + (size_t)i * orow[0];
        for (int x = 0; x < hw * 2; x++) {
          d[x * ps + 1] = (uint8_t)orc_cavg(cl, d[x * ps + 1 + orow[0]], d[x * ps + 1 - orow[0]]);
          d[x * ps + 2] = (uint8_t)orc_cavg(cl, d[x * ps + 2 + orow[0]], d[x * ps + 2 - orow[0]]);
        }
      }
    #undef CH_U
    #undef CH_V
    return 0;
  }
  /* convert_yuv420_to_uyvy_frame / _yuyv_frame :7104-7198: `i` is never advanced, so the "average with the row above"
     step never runs; after an even row the chroma pointers go back by the ROWSTRIDE (not by the half width), which only
     lands on the start of the same chroma row when the chroma planes are compact */
  if (in420 && (out_pal == P_UYVY || out_pal == P_YUYV)) {
    const int hw = width >> 1;
    if (irow[1] != hw || irow[2] != hw || ((width | height) & 1)) return -1;
    for (int y = 0; y < height; y++) {
      uint8_t *d = dst[0] + (size_t)y * (orow[0] / 4) * 4;                                   /* orow / 4 macropixels (:7120) */
      const uint8_t *sy = src[0] + (size_t)y * irow[0], *su = src[1] + (size_t)(y >> 1) * hw, *sv = src[2] + (size_t)(y >> 1) * hw;
      for (int j = 0; j < hw; j++) {
        if (out_pal == P_UYVY) { d[4 * j] = su[j]; d[4 * j + 1] = sy[2 * j]; d[4 * j + 2] = sv[j]; d[4 * j + 3] = sy[2 * j + 1]; }
        else { d[4 * j] = sy[2 * j]; d[4 * j + 1] = su[j]; d[4 * j + 2] = sy[2 * j + 1]; d[4 * j + 3] = sv[j]; }
      }
    }
    return 0;
  }
  /* 420P -> 422P: Y plane copied (weed_layer_copy_single_plane, :13593), convert_double_chroma :10612-10639 on
     (width >> 1, height >> 1): every chroma row twice, then each odd row averaged with the even row below it */
  if (in420 && out_pal == P_422) {
    const int cw = width >> 1, ch = height >> 1;
    if ((width | height) & 1) return -1;
    copy_rows(src[0], irow[0], dst[0], orow[0], width, height);
    for (int p = 1; p < 3; p++) {
      int i2 = 0, chroma = 0;
      for (int i = 0; i < ch * 2; i++) {
        memcpy(dst[p] + (size_t)orow[p] * i, src[p] + (size_t)irow[p] * i2, (size_t)cw);
        if (!chroma && i > 0)
          for (int j = 0; j < cw; j++) {
            uint8_t *q = dst[p] + (size_t)orow[p] * (i - 1) + j;
            *q = (uint8_t)orc_cavg(cl, *q, dst[p][(size_t)orow[p] * i + j]);
          }
        if (chroma) i2++;
        chroma = !chroma;
      }
    }
    return 0;
  }
  /* convert_yuvp_to_yuv420_frame :7690-7753 (444P / 4444P -> 420P): horizontal pairs averaged, then the two rows */
  if (in444 && (out_pal == P_420 || out_pal == P_YV12)) {
    const int hw = width >> 1;
    if ((width | height) & 1) return -1;             /* 4:2:0 layers are even (:11601-11603); the reference overruns otherwise */
    if (dst[0] != src[0]) {
      if (irow[0] == orow[0]) memcpy(dst[0], src[0], (size_t)irow[0] * height);
      else copy_rows(src[0], irow[0], dst[0], orow[0], width, height);
    }
    for (int p = 1; p < 3; p++) {
      uint8_t *d = dst[p];
      const uint8_t *s = src[p];
      int chroma = 0;
      for (int i = 0; i < height; i++) {
        for (int j = 0; j < hw; j++) {
          const int x = orc_cavg(cl, s[j * 2], s[j * 2 + 1]);
          d[j] = (uint8_t)(!chroma ? x : orc_cavg(cl, d[j], x));
        }
        if (chroma) d += orow[p];
        chroma = !chroma;
        s += irow[p];
      }
    }
    return 0;
  }
  /* convert_yuv_planar_to_uyvy_frame / _yuyv_frame :7500-7591: only the compact branch stays inside its buffers (the strided
     one runs `width` macropixels per row) */
  if (in444 && (out_pal == P_UYVY || out_pal == P_YUYV)) {
    if (irow[0] != width || orow[0] != width * 2 || (width & 1)) return -1;
    const size_t n = ((size_t)width * height) >> 1;
    for (size_t k = 0; k < n; k++) {
      uint8_t *d = dst[0] + 4 * k;
      const int u = orc_cavg(cl, src[1][2 * k], src[1][2 * k + 1]), v = orc_cavg(cl, src[2][2 * k], src[2][2 * k + 1]);
      if (out_pal == P_UYVY) { d[0] = (uint8_t)u; d[1] = src[0][2 * k]; d[2] = (uint8_t)v; d[3] = src[0][2 * k + 1]; }
      else { d[0] = src[0][2 * k]; d[1] = (uint8_t)u; d[2] = src[0][2 * k + 1]; d[3] = (uint8_t)v; }
    }
    return 0;
  }
  /* YUV888 / YUVA8888 -> subsampled (:8035-8270): horizontal pairs averaged; every one of these walks its DESTINATION as a
     compact buffer (the strided branches of :8161-8178 / :8205-8222 / :8250-8267 subtract the wrong widths or step macropixel
     pointers by byte counts), the packed source may have row padding */
  if ((in_pal == P_888 || in_pal == P_8888) && (out_pal == P_420 || out_pal == P_YV12 || out_pal == P_422 || out_pal == P_UYVY || out_pal == P_YUYV)) {
    const int ips = in_pal == P_8888 ? 4 : 3, hw = width >> 1;
    if (width & 1) return -1;
    if (out_pal == P_420 || out_pal == P_YV12) {                 /* convert_yuv888_to_yuv420_frame :8035-8090 */
      if ((height & 1) || orow[0] != width || orow[1] != hw || orow[2] != hw) return -1;
      for (int y = 0; y < height; y++) {
        const uint8_t *s = src[0] + (size_t)y * irow[0];
        for (int j = 0; j < hw; j++) {
          const int xu = orc_cavg(cl, s[2 * j * ips + 1], s[2 * j * ips + 1 + ips]), xv = orc_cavg(cl, s[2 * j * ips + 2], s[2 * j * ips + 2 + ips]);
          uint8_t *pu = dst[1] + (size_t)(y >> 1) * hw + j, *pv = dst[2] + (size_t)(y >> 1) * hw + j;
          dst[0][(size_t)y * width + 2 * j] = s[2 * j * ips]; dst[0][(size_t)y * width + 2 * j + 1] = s[2 * j * ips + ips];
          if (!(y & 1)) { *pu = (uint8_t)xu; *pv = (uint8_t)xv; }
          else { *pu = (uint8_t)orc_cavg(cl, *pu, xu); *pv = (uint8_t)orc_cavg(cl, *pv, xv); }
        }
      }
      return 0;
    }
    if (out_pal == P_422) {                                      /* convert_yuv888_to_yuv422_frame :8129-8181, compact branch */
      if (irow[0] != width * ips || orow[0] != width || orow[1] != hw) return -1;
    } else if (orow[0] != width * 2) return -1;                  /* convert_yuv888_to_uyvy_frame / _yuyv_frame :8184-8270 */
    for (int y = 0; y < height; y++) {
      const uint8_t *s = src[0] + (size_t)y * irow[0];
      for (int j = 0; j < hw; j++) {
        const uint8_t y0 = s[2 * j * ips], y1 = s[2 * j * ips + ips];
        const uint8_t u = (uint8_t)orc_cavg(cl, s[2 * j * ips + 1], s[2 * j * ips + 1 + ips]), v = (uint8_t)orc_cavg(cl, s[2 * j * ips + 2], s[2 * j * ips + 2 + ips]);
        if (out_pal == P_422) {
          dst[0][(size_t)y * width + 2 * j] = y0; dst[0][(size_t)y * width + 2 * j + 1] = y1;
          dst[1][(size_t)y * hw + j] = u; dst[2][(size_t)y * hw + j] = v;
        } else {
          uint8_t *d = dst[0] + (size_t)y * width * 2 + 4 * j;
          if (out_pal == P_UYVY) { d[0] = u; d[1] = y0; d[2] = v; d[3] = y1; }
          else { d[0] = y0; d[1] = u; d[2] = y1; d[3] = v; }
        }
      }
    }
    return 0;
  }
  /* convert_uyvy_to_yuv422_frame / convert_yuyv_to_yuv422_frame :8093-8126: flat walks, compact on both sides */
  if (inpk422 && out_pal == P_422) {
    const int mw = width >> 1, yo = in_pal == P_UYVY ? 1 : 0, uo = in_pal == P_UYVY ? 0 : 1, vo = in_pal == P_UYVY ? 2 : 3;
    if ((width & 1) || irow[0] != width * 2 || orow[0] != width || orow[1] != mw || orow[2] != mw) return -1;
    for (size_t k = 0; k < (size_t)mw * height; k++) {
      const uint8_t *m = src[0];                                  /* the source pointer is never advanced (:8102-8107, :8120-8125) */
      dst[0][2 * k] = m[yo]; dst[0][2 * k + 1] = m[yo + 2]; dst[1][k] = m[uo]; dst[2][k] = m[vo];
    }
    return 0;
  }
  /* UYVY / YUYV sources (:7800-7971): chroma replicated, no interpolation */
  if (inpk422) {
    const int mw = width >> 1, yo = in_pal == P_UYVY ? 1 : 0, uo = in_pal == P_UYVY ? 0 : 1, vo = in_pal == P_UYVY ? 2 : 3;
    const int irm = (irow[0] / 4) * 4;                                                        /* irow /= 4 macropixels */
    if (width & 1) return -1;
    if (out_pal == P_444 || out_pal == P_4444) {          /* convert_uyvy_to_yuvp_frame: plane strides are mixed up -> equal strides only */
      if (orow[0] != orow[1] || orow[0] != orow[2]) return -1;
      for (int k = 0; k < height; k++)
        for (int x = 0; x < mw; x++) {
          const uint8_t *m = src[0] + (size_t)k * irm + 4 * x;
          dst[0][(size_t)k * orow[0] + 2 * x] = m[yo]; dst[0][(size_t)k * orow[0] + 2 * x + 1] = m[yo + 2];
          dst[1][(size_t)k * orow[0] + 2 * x] = dst[1][(size_t)k * orow[0] + 2 * x + 1] = m[uo];
          dst[2][(size_t)k * orow[0] + 2 * x] = dst[2][(size_t)k * orow[0] + 2 * x + 1] = m[vo];
        }
      if (out_pal == P_4444) memset(dst[3], 255, (size_t)orow[3] * height);
      return 0;
    }
    if (out_pal == P_888 || out_pal == P_8888) {          /* convert_uyvy_to_yuv888_frame / convert_yuyv_to_yuv888_frame */
      const int ops = out_pal == P_8888 ? 4 : 3;
      for (int y = 0; y < height; y++)
        for (int x = 0; x < mw; x++) {
          const uint8_t *m = src[0] + (size_t)y * irm + 4 * x;
          uint8_t *d = dst[0] + (size_t)y * orow[0] + (size_t)2 * x * ops;
          d[0] = m[yo]; d[1] = m[uo]; d[2] = m[vo];
          if (ops == 4) d[3] = 255;
          d[ops] = m[yo + 2]; d[ops + 1] = m[uo]; d[ops + 2] = m[vo];
          if (ops == 4) d[ops + 3] = 255;
        }
      return 0;
    }
    if (out_pal == P_420 || out_pal == P_YV12) {          /* convert_uyvy_to_yuv420_frame: no strides at all -> compact only */
      if ((height & 1) || irow[0] != width * 2 || orow[0] != width || orow[1] != mw || orow[2] != mw) return -1;
      for (int y = 0; y < height; y++)
        for (int x = 0; x < mw; x++) {
          const uint8_t *m = src[0] + (size_t)y * irm + 4 * x;
          uint8_t *pu = dst[1] + (size_t)(y >> 1) * mw + x, *pv = dst[2] + (size_t)(y >> 1) * mw + x;
          dst[0][(size_t)y * width + 2 * x] = m[yo]; dst[0][(size_t)y * width + 2 * x + 1] = m[yo + 2];
          if (!(y & 1)) { *pu = m[uo]; *pv = m[vo]; }
          else { *pu = (uint8_t)orc_cavg(cl, *pu, m[uo]); *pv = (uint8_t)orc_cavg(cl, *pv, m[vo]); }
        }
      return 0;
    }
    return -1;
  }
  return -1;
}

/* ------------------------------------------------------------------------------------------------
 * F9: deinterlace                         reference: lives-plugins/weed-plugins/deinterlace.c:45-308
 * Packed palettes only (WEED_PALETTE_* numbers 1..5 RGB family, 588 / 589, 564 / 565): for a planar frame the reference's
 * pixel_size() is 0 and its loop does nothing (:91, :112), and YUV444P dereferences a pointer it never sets (:146).
 * Works on pixel triples of every odd row r < height - 2: row r - 1 <- row r; row r <- the mean of rows r and r + 2 where the
 * alternate rows differ less than the consecutive ones, else row r + 1.  Bytes the reference does not write are left alone
 * (alpha of 4-byte pixels is copied for row r only, :271-277).  -1: ARGB32 out of place (the `x++` at :119 makes the triple
 * walk drift by one byte per triple), or rows too tight for the last partial triple (the reference then writes into the next row).
 * ---------------------------------------------------------------------------------------------- */
int orc_deinterlace(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int palette) {
  const int inplace = (src == dst);
  int psize, pcpy, green = 0, packed422 = 0;
  switch (palette) {
  case 1: case 2: psize = 3; pcpy = 3; green = 1; break;
  case 588: psize = 3; pcpy = 3; break;
  case 3: case 4: psize = 4; pcpy = 3; green = 1; break;
  case 589: psize = 4; pcpy = 3; break;
  case 5: psize = 4; pcpy = 3; green = 2; if (!inplace) return -1; break;
  case 564: psize = 4; pcpy = 4; packed422 = 1; break;
  case 565: psize = 4; pcpy = 4; packed422 = 2; break;
  default: return -1;
  }
  const int widthx = width * psize, ntrip = (width + 2) / 3;
  if (ntrip * 3 * psize > irow || ntrip * 3 * psize > orow) return -1;
  for (int r = 1; r < height - 2; r += 2) {
    const uint8_t *r0 = src + (size_t)(r - 1) * irow, *r1 = src + (size_t)r * irow, *r2 = src + (size_t)(r + 1) * irow, *r3 = src + (size_t)(r + 2) * irow;
    uint8_t *o0 = dst + (size_t)(r - 1) * orow, *o1 = dst + (size_t)r * orow;
    for (int x = 0; x < widthx; x += 3 * psize) {
      const int xc = x + 2 * psize;
      int m1, m2, m3, m4;
      if (packed422) {
        const int yo = packed422 == 1 ? 1 : 0;
        m1 = (r0[x + yo] + r0[x + yo + 2] + r0[xc + yo] + r0[xc + yo + 2]) >> 2;
        m2 = (r2[x + yo] + r2[x + yo + 2] + r2[xc + yo] + r2[xc + yo + 2]) >> 2;
        m3 = (r1[x + yo] + r1[x + yo + 2] + r1[xc + yo] + r1[xc + yo + 2]) >> 2;
        m4 = (r3[x + yo] + r3[x + yo + 2] + r3[xc + yo] + r3[xc + yo + 2]) >> 2;
      } else {
        m1 = (r0[x + green] + r0[xc + green]) >> 1; m2 = (r2[x + green] + r2[xc + green]) >> 1;
        m3 = (r1[x + green] + r1[xc + green]) >> 1; m4 = (r3[x + green] + r3[xc + green]) >> 1;
      }
      const int d1 = abs(m1 - m2) + abs(m3 - m4), d2 = abs(m1 - m4) + abs(m3 - m2);
      uint8_t top[12], bot[12];
      for (int p = 0; p < 3; p++)
        for (int k = 0; k < pcpy; k++) {
          const int i = x + p * psize + k;
          top[p * 4 + k] = r1[i];
          bot[p * 4 + k] = (d1 < d2) ? (uint8_t)((r1[i] + r3[i]) >> 1) : r2[i];
        }
      for (int p = 0; p < 3; p++)
        for (int k = 0; k < pcpy; k++) { o0[x + p * psize + k] = top[p * 4 + k]; o1[x + p * psize + k] = bot[p * 4 + k]; }
      if (!inplace && (palette == 3 || palette == 4 || palette == 589)) { o1[x + 3] = r1[x + 3]; o1[x + 7] = r1[x + 7]; o1[x + 11] = r1[x + 11]; }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * F10: RGBdelay / YUVdelay                reference: lives-plugins/weed-plugins/RGBdelay.c:36-431
 * Stateful: a ring of up to 50 compact frames; per cached frame j the channels switched on are scaled through a LUT
 * (strength normalised per channel over all enabled frames) and ADDED into the output with 8-bit wrap-around.
 * on[3 * j + c] = the RED_ON / GREEN_ON / BLUE_ON switches, strength[j] = STRENGTH(j), for j = 0..50 (:129-132).
 * The host-ease path (:180-183, :407-412) is not taken (no "ease_out" leaf): the cache always ramps up.
 * ---------------------------------------------------------------------------------------------- */
struct orc_rgbdelay { int ccache, tcache; uint8_t *cache[51]; int is_bgr[51]; };

orc_rgbdelay *orc_rgbdelay_new(void) { return (orc_rgbdelay *)calloc(1, sizeof(orc_rgbdelay)); }
void orc_rgbdelay_free(orc_rgbdelay *s) {
  if (!s) return;
  for (int i = 0; i < s->tcache; i++) free(s->cache[i]);
  free(s);
}
static void rgbd_make_lut(uint8_t *lut, double val, int min) {            /* make_lut :36-52 */
  int mina = min, minb = 0;
  double rnd = 0.5;
  if (min < 0) { mina = 0; minb = -min; rnd += (double)minb; }
  for (int i = 0; i < 256; i++) {
    double rval = (double)(i - mina) * val + rnd;
    if (rval < 0.) rval = 0.;
    if (rval > 255.) rval = 255.;
    lut[i] = (uint8_t)rval;
  }
}
int orc_rgbdelay_process(orc_rgbdelay *s, const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int palette,
                         int yuv_clamped, int maxcache, const int *on, const double *strength) {
  const int wb = width * 3, inplace = (src == dst);
  const int is_bgr = (palette == 2), is_yuv = (palette == 588);
  double tstr[3] = {0., 0., 0.}, yscale = 1., uvscale = 1.;
  int yuvmin = 0, uvmin = 0, maxneeded = 0;
  uint8_t lut[3][256], *tmpcache = NULL;
  if (palette != 1 && palette != 2 && palette != 588) return -1;
  if (maxcache < 0) maxcache = 0; else if (maxcache > 50) maxcache = 50;
  for (int i = 1; i < maxcache; i++) if (on[3 * i] || on[3 * i + 1] || on[3 * i + 2]) maxneeded = i + 1;   /* :186-193 */
  if (maxneeded != s->tcache) {                                                                          /* realloc_cache :54-91 */
    for (int i = s->tcache; i > maxneeded; i--) { free(s->cache[i - 1]); s->cache[i - 1] = NULL; }
    for (int i = s->tcache; i < maxneeded; i++) s->cache[i] = (uint8_t *)malloc((size_t)wb * height);
    s->tcache = maxneeded;
    if (s->ccache > s->tcache) s->ccache = s->tcache;
  }
  if (s->tcache > 1) tmpcache = s->cache[s->tcache - 1];
  for (int i = s->tcache - 1; i >= 0; i--) {                                                             /* :208-228 */
    if (i > 0) { s->cache[i] = s->cache[i - 1]; s->is_bgr[i] = s->is_bgr[i - 1]; }
    for (int c = 0; c < 3; c++) if (on[3 * i + c]) tstr[c] += strength[i];
  }
  s->is_bgr[0] = is_bgr;
  if (s->tcache > 0) {
    for (int y = 0; y < height; y++) memcpy(tmpcache + (size_t)y * wb, src + (size_t)y * irow, (size_t)wb);
    s->cache[0] = tmpcache;
  }
  for (int c = 0; c < 3; c++) if (tstr[c] < 1.) tstr[c] = 1.;
  if (is_yuv && yuv_clamped) { yuvmin = 16; uvmin = 16; yscale = 255. / 219.; uvscale = 255. / 224.; }
  if (s->tcache == 0) {                                                                                  /* :254-309 */
    int b[3] = {on[0] != 0, on[1] != 0, on[2] != 0}, red = 0, blue = 2;
    const double cstr = strength[0];
    if (is_bgr) { const int t = b[0]; b[0] = b[2]; b[2] = t; red = 2; blue = 0; }
    rgbd_make_lut(lut[red], cstr / tstr[0] * yscale, yuvmin);
    rgbd_make_lut(lut[1], cstr / tstr[1] * uvscale, yuvmin);
    rgbd_make_lut(lut[blue], cstr / tstr[2] * uvscale, yuvmin);
    for (int y = 0; y < height; y++) {
      const uint8_t *sp = src + (size_t)y * irow;
      uint8_t *dp = dst + (size_t)y * orow;
      for (int i = 0; i < wb; i += 3) {
        if (b[0]) dp[i] = lut[0][sp[i]]; else if (inplace) dp[i] = (uint8_t)yuvmin;
        if (b[1]) dp[i + 1] = lut[1][sp[i + 1]]; else if (inplace) dp[i + 1] = (uint8_t)uvmin;
        if (b[2]) dp[i + 2] = lut[2][sp[i + 2]]; else if (inplace) dp[i + 2] = (uint8_t)uvmin;
      }
    }
  } else {
    memset(dst, 0, (size_t)orow * height);                                                               /* :311 */
    for (int j = 0; j < s->tcache; j++) {
      const int k = (j <= s->ccache) ? j : s->ccache;
      int b[3] = {on[3 * j] != 0, on[3 * j + 1] != 0, on[3 * j + 2] != 0}, red, blue;
      if (!b[0] && !b[1] && !b[2] && j > 0) continue;
      const int cross = ((!is_bgr && s->is_bgr[j]) || (is_bgr && !s->is_bgr[j])) ? 2 : 0;
      const double cstr = strength[j];
      if (s->is_bgr[j]) { const int t = b[0]; b[0] = b[2]; b[2] = t; red = 2; blue = 0; } else { red = 0; blue = 2; }
      rgbd_make_lut(lut[red], cstr / tstr[0] * yscale, yuvmin);
      rgbd_make_lut(lut[1], cstr / tstr[1] * uvscale, yuvmin);
      rgbd_make_lut(lut[blue], cstr / tstr[2] * uvscale, yuvmin);
      for (int y = 0; y < height; y++) {
        const uint8_t *cp = s->cache[k] + (size_t)y * wb;
        uint8_t *dp = dst + (size_t)y * orow;
        for (int i = 0; i < wb; i += 3) {
          if (b[0]) dp[i] = (uint8_t)(dp[i] + lut[0][cp[i + cross]]);
          if (b[1]) dp[i + 1] = (uint8_t)(dp[i + 1] + lut[1][cp[i + 1]]);
          if (b[2]) dp[i + 2] = (uint8_t)(dp[i + 2] + lut[2][cp[i + 2 - cross]]);
        }
      }
    }
  }
  if (is_yuv && yuvmin == 16) {                                                                          /* :393-403 */
    rgbd_make_lut(lut[0], 1. / yscale, -yuvmin);
    rgbd_make_lut(lut[1], 1. / uvscale, -yuvmin);
    for (int y = 0; y < height; y++) {
      uint8_t *dp = dst + (size_t)y * orow;
      for (int i = 0; i < wb; i += 3) { dp[i] = lut[0][dp[i]]; dp[i + 1] = lut[1][dp[i + 1]]; dp[i + 2] = lut[1][dp[i + 2]]; }
    }
  }
  if (s->ccache < s->tcache) s->ccache++;                                                                /* :413-416 */
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * F11: negate / posterise / ccorrect       reference: lives-plugins/weed-plugins/scripts/{negate,posterise,ccorrect}.script
 * All three apply a table per byte position of the pixel.  kind 0 negate (<process>: colour bytes ^ 0xFF, alpha copied),
 * 1 posterise (levmask built as in <process>; bytes 0..2 masked, byte 3 of a 4-byte pixel copied whatever the palette; no
 * ARGB32), 2 ccorrect (<static> make_table + <process>: r / g / b tables at the palette's colour positions, alpha copied out
 * of place / left in place).  Returns psize, 0 for a palette the effect does not list.
 * ---------------------------------------------------------------------------------------------- */
int orc_fx_luts(int kind, int palette, double p0, double p1, double p2, uint8_t *luts) {
  if (palette < 1 || palette > 5) return 0;
  const int psize = palette <= 2 ? 3 : 4, alpha = palette == 5 ? 0 : (psize == 4 ? 3 : -1);
  for (int c = 0; c < psize; c++)
    for (int i = 0; i < 256; i++) luts[c * 256 + i] = (uint8_t)i;
  if (kind == 0) {
    for (int c = 0; c < psize; c++)
      if (c != alpha) for (int i = 0; i < 256; i++) luts[c * 256 + i] = (uint8_t)(0xFF ^ i);
    return psize;
  }
  if (kind == 1) {
    unsigned char levmask = 128;
    if (palette == 5) return 0;
    for (int i = 1; i < (int)p0; i++) levmask += 128 >> i;
    for (int c = 0; c < 3; c++) for (int i = 0; i < 256; i++) luts[c * 256 + i] = (uint8_t)(i & levmask);
    return psize;
  }
  if (kind == 2) {
    const double val[3] = {p0, p1, p2};
    const int bgr = (palette == 2 || palette == 4), offs = palette == 5 ? 1 : 0;
    for (int k = 0; k < 3; k++) {
      uint8_t *t = luts + (size_t)(offs + (bgr ? 2 - k : k)) * 256;
      for (int i = 0; i < 256; i++) { const int ival = (val[k] * i + .5); t[i] = ival > 255 ? (uint8_t)255 : (uint8_t)ival; }
    }
    return psize;
  }
  return 0;
}
void orc_byte_luts(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height, int psize, const uint8_t *luts) {
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++)
      for (int c = 0; c < psize; c++) dst[(size_t)y * orow + x * psize + c] = luts[c * 256 + src[(size_t)y * irow + x * psize + c]];
}

/* ------------------------------------------------------------------------------------------------
 * F12: triple split                       reference: lives-plugins/weed-plugins/layout_blends.c:24-113
 * RGB24 / BGR24.  The middle band shows src1, the outer bands src2, a border of colour bc between them; vert = the bands are rows.
 * The comparisons are kept in the reference's types: byte column j (int) against width-in-bytes * double, rows as row indices
 * (the reference compares row pointers; without `vert` all four row bounds are the end pointer).
 * ---------------------------------------------------------------------------------------------- */
void orc_triple_split(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height, int is_bgr,
                      double xstart, int sym, double xend, int vert, double bw, const int *bc_in) {
  const int wb = width * 3, inplace = (src1 == dst);
  int bc[3] = {bc_in[0], bc_in[1], bc_in[2]};
  int tbs = height, tbe = height, bbs = height, bbe = height;
  if (sym) { xstart /= 2.; xend = 1. - xstart; }
  if (xstart > xend) { const double t = xend; xend = xstart; xstart = t; }
  if (is_bgr) { const int t = bc[2]; bc[2] = bc[0]; bc[0] = t; }
  if (vert) {
    tbs = (int)(height * (xstart - bw) + .5); tbe = (int)(height * (xstart + bw) + .5);
    bbs = (int)(height * (xend - bw) + .5); bbe = (int)(height * (xend + bw) + .5);
    xstart = xend = -bw;
  }
  for (int r = 0; r < height; r++) {
    const uint8_t *s1 = src1 + (size_t)r * irow1, *s2 = src2 + (size_t)r * irow2;
    uint8_t *d = dst + (size_t)r * orow;
    for (int j = 0; j < wb; j += 3) {
      if ((j < wb * (xstart - bw) || j >= wb * (xend + bw)) && (r <= tbs || r >= bbe)) { memcpy(d + j, s2 + j, 3); continue; }
      if ((j > wb * (xstart + bw) && j < wb * (xend - bw)) || (r > tbe && r < bbs)) { if (!inplace) memcpy(d + j, s1 + j, 3); continue; }
      d[j] = (uint8_t)bc[0]; d[j + 1] = (uint8_t)bc[1]; d[j + 2] = (uint8_t)bc[2];
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * F7b: dissolve                           reference: multi_transitions.c:41-69 (mask), :208-212 (select);
 *                                         RNG libweed/weed-plugin-utils.c:666, :686-704 (xorshift64 on the instance's seed leaf)
 * mask[i] = (float)(xorshift64 chain value / 0xFFFFFFFF / 0xFFFFFFFF), seeded with the instance's "random_seed"; a pixel shows
 * src2 where mask < (float)amount.  ("rand replace" draws from the time-seeded global generator: not reproducible, not taken.)
 * ---------------------------------------------------------------------------------------------- */
void orc_dissolve_mask(uint64_t seed, int width, int height, float *mask) {
  static const double divd = (double)(0xFFFFFFFF);
  uint64_t x = seed;
  for (size_t i = 0; i < (size_t)width * height; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const double val = (double)x / divd;
    mask[i] = (float)(val / divd * 1.);
  }
}
void orc_dissolve(const uint8_t *src1, int irow1, const uint8_t *src2, int irow2, uint8_t *dst, int orow, int width, int height, int psize,
                  const float *mask, double amount) {
  const float bf = (float)amount;
  const int inplace = (src1 == dst);
  for (int i = 0; i < height; i++)
    for (int x = 0; x < width; x++) {
      const size_t j = (size_t)x * psize;
      if (mask[(size_t)i * width + x] < bf) memcpy(dst + (size_t)i * orow + j, src2 + (size_t)i * irow2 + j, (size_t)psize);
      else if (!inplace) memcpy(dst + (size_t)i * orow + j, src1 + (size_t)i * irow1 + j, (size_t)psize);
    }
}

/* ------------------------------------------------------------------------------------------------
 * K9b: alpha_premult for YUVA8888 / YUVA4444P            reference: src/colourspace.c:11995-12096, tables init_unal :1141-1160
 * Unclamped layers use the RGB tables al / unal on Y, U and V alike.  Clamped layers use alcy / alcuv (FORWARD) and unalcy / unalcuv
 * (REVERSE): note alcy tests against the UV range and both "clamped Y" tables DIVIDE by 255 / alpha in either direction; the packed
 * FORWARD loop (:12086-12092) indexes alcuv with the Y byte it has just written, for U and V alike.  All of it is kept.
 * ---------------------------------------------------------------------------------------------- */
static int k9b_c255f(double a) { return a >= 254.5 ? 255 : a < -0.5 ? 0 : (uint8_t)(a + .5); }      /* CLAMP0255f, src/maths.h:88 */
void orc_premult_yuv_tables(uint8_t *unalcy, uint8_t *alcy, uint8_t *unalcuv, uint8_t *alcuv) {   /* each [256][256], stored as bytes the way the loops store them */
  for (int i = 0; i < 256; i++) {
    const float alpha = (float)255. / (float)i;
    for (int j = 0; j < 256; j++) {
      const int t_unalcuv = k9b_c255f((float)(j - 16.) * alpha + 16.);
      const int t_alcuv = k9b_c255f((float)(j - 128.) * alpha + 128.);
      const int t_unalcy = (int)((float)j / alpha + .5) > (235. - 16.) ? (int)235. : (int)((float)(j - 16.) / alpha + 16. + .5);
      const int t_alcy = (int)((float)j / alpha + .5) > (240. - 16.) ? (int)240. : (int)((float)(j - 16.) / alpha + 16. + .5);
      unalcuv[i * 256 + j] = (uint8_t)t_unalcuv; alcuv[i * 256 + j] = (uint8_t)t_alcuv;
      unalcy[i * 256 + j] = (uint8_t)t_unalcy; alcy[i * 256 + j] = (uint8_t)t_alcy;
    }
  }
}
/* palette 589 YUVA8888 (planes[0] packed) or 545 YUVA4444P (four planes); clamped = the layer's YUV_clamping is CLAMPED; un = REVERSE */
int orc_alpha_premult_yuva(uint8_t *const planes[4], const int rows[4], int width, int height, int palette, int clamped, int un) {
  static uint8_t t[4][65536];
  static int ready = 0;
  if (!ready) { orc_premult_yuv_tables(t[0], t[1], t[2], t[3]); ready = 1; }
  const uint8_t *cy = un ? t[0] : t[1], *cuv = un ? t[2] : t[3];
  if (palette == 589) {
    for (int i = 0; i < height; i++) {
      uint8_t *p = planes[0] + (size_t)i * rows[0];
      for (int j = 0; j < width * 4; j += 4) {
        const int a = p[j + 3];
        if (!clamped) { for (int c = 0; c < 3; c++) p[j + c] = (uint8_t)(un ? orc_unal(a, p[j + c]) : orc_al(a, p[j + c])); }
        else if (un) { p[j] = cy[a * 256 + p[j]]; p[j + 1] = cuv[a * 256 + p[j + 1]]; p[j + 2] = cuv[a * 256 + p[j + 2]]; }
        else { p[j] = cy[a * 256 + p[j]]; p[j + 1] = cuv[a * 256 + p[j]]; p[j + 2] = cuv[a * 256 + p[j]]; }     /* :12089-12091 */
      }
    }
    return 0;
  }
  if (palette == 545) {
    for (int i = 0; i < height; i++)
      for (int j = 0; j < width; j++) {
        const int a = planes[3][(size_t)i * rows[3] + j];
        uint8_t *y = planes[0] + (size_t)i * rows[0] + j, *u = planes[1] + (size_t)i * rows[1] + j, *v = planes[2] + (size_t)i * rows[2] + j;
        if (!clamped) { *y = (uint8_t)(un ? orc_unal(a, *y) : orc_al(a, *y)); *u = (uint8_t)(un ? orc_unal(a, *u) : orc_al(a, *u)); *v = (uint8_t)(un ? orc_unal(a, *v) : orc_al(a, *v)); }
        else { *y = cy[a * 256 + *y]; *u = cuv[a * 256 + *u]; *v = cuv[a * 256 + *v]; }
      }
    return 0;
  }
  return -1;
}
