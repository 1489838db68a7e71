/* orc_resizable.c -- TEST INFRASTRUCTURE (the CPU oracle; nothing under lives_amd/ or include/ uses it).
 *
 * A restatement of the palette resolution resize_layer_full runs in front of its body, for a build WITHOUT swscale:
 *   weed_palette_conv_resizable / _is_resizable   src/colourspace.c:2596-2654   (the #else branch)
 *   get_masq_pal                                  src/colourspace.c:14500-14513
 *   get_inter_pal                                 src/colourspace.c:14516-14575
 *   get_resizable                                 src/colourspace.c:14577-14669
 *   get_tgt_gamma                                 src/colourspace.c:14736-14740
 *   can_inline_gamma / pconv_can_inplace          src/colourspace.c:12128-12157
 * Pinned by tests/golden/resizable.npz: the outputs of the reference's own lines (oracle/ref/build_resizable_slice.py ->
 * gen_golden_resizable.py) for every pair of the 15 integer palettes plus the hints NONE / ANY, both scale directions.
 * Palettes are described by a property table (class, planar, alpha) as the reference's advp[] describes them (:1535-1645). */
#include "lives_oracle.h"

enum { P_RGB = 1, P_YUV = 2, P_PLANAR = 4, P_ALPHA = 8, P_SCALES = 16 };   /* P_SCALES: a case of the non-swscale switch (:2621-2643) */
static int props(int pal) {
  switch (pal) {
  case 1: case 2: return P_RGB | P_SCALES;                     /* RGB24, BGR24 */
  case 3: case 4: return P_RGB | P_ALPHA | P_SCALES;           /* RGBA32, BGRA32 */
  case 5: return P_RGB | P_ALPHA;                              /* ARGB32 */
  case 512: case 513: case 522: case 544: return P_YUV | P_PLANAR;
  case 545: return P_YUV | P_PLANAR | P_ALPHA;                 /* YUVA4444P */
  case 564: case 565: case 595: return P_YUV;                  /* UYVY, YUYV, YUV411 */
  case 588: return P_YUV | P_SCALES;                           /* YUV888 */
  case 589: return P_YUV | P_ALPHA | P_SCALES;                 /* YUVA8888 */
  default: return 0;                                           /* NONE, ANY, anything unknown: no channels at all */
  }
}
static int masq(int pal) {
  if (pal == 3 || pal == 4 || pal == 589) return 3;
  if (pal == 1 || pal == 2 || pal == 588) return 1;
  if (pal == 513) return 512;
  return 0;
}
static int inter(int in, int out, int upscale) {
  const int pi = props(in), po = props(out);
  const int alpha = (pi & P_ALPHA) && (po & P_ALPHA), planar = (pi | po) & P_PLANAR;
  int to_yuv;                                                  /* which family the intermediate palette belongs to */
  if ((pi & P_RGB) && (po & P_RGB)) to_yuv = 0;
  else if ((pi & P_YUV) && (po & P_YUV)) to_yuv = 1;
  else if (pi & P_RGB) to_yuv = upscale;                       /* rgb -> yuv: convert first when the frame grows */
  else to_yuv = !upscale;                                      /* yuv (or unknown) -> rgb */
  if (!to_yuv) return alpha ? 3 : 1;
  if (planar) return alpha ? 545 : 544;
  return alpha ? 589 : 588;
}

/* io[0..3] in: palette, opal_hint, oclamp_hint, upscale; out: io[0] resolved, io[1] xpalette, io[2] oclamp_hint, io[3] opal_hint, io[4] xopal_hint.
   1 = LIVES_RESULT_SUCCESS, 0 = LIVES_RESULT_FAIL, -1 = the reference's LIVES_FATAL (io untouched) */
int orc_get_resizable(int *io) {
  const int pal = io[0], upscale = io[3];
  int opal = io[1], res = 0, xpal = pal, xopal = opal;
  const int in_ok = props(pal) & P_SCALES, out_ok = props(opal) & P_SCALES;
  if (in_ok) {
    if (opal != -1) {
      if (out_ok) res = pal;
      else if (upscale && masq(opal)) { res = opal; xpal = xopal = masq(opal); }
    }
    if (!res) res = xopal = opal = xpal = pal;
  } else if (out_ok) {
    if ((!upscale || opal == -1) && masq(pal)) { res = opal = pal; xpal = xopal = masq(pal); }
    if (!res) res = xpal = xopal = opal;
  } else {
    int m = res = inter(pal, opal, upscale);
    if (!(props(res) & P_SCALES)) {
      m = masq(res);
      if (!m) return -1;
    }
    opal = res;
    xpal = xopal = m;
  }
  if (!res) return 0;
  io[0] = res; io[1] = xpal; io[3] = opal; io[4] = xopal;
  if ((props(res) & P_YUV) && (props(xpal) & P_RGB)) io[2] = 1;      /* WEED_YUV_CLAMPING_UNCLAMPED */
  return 1;
}
int orc_get_tgt_gamma(int ipal, int opal) { return ((props(ipal) & P_RGB) && (props(opal) & P_YUV)) ? 1 : 0; }
int orc_can_inline_gamma(int in, int out) {
  if ((props(in) & P_RGB) && (props(out) & P_RGB)) return 1;
  if ((in == 512 || in == 513 || in == 522 || in == 544) && (props(out) & P_RGB)) return 1;
  if (out >= 1 && out <= 5) return 1;
  if (out == 564 || out == 565) return in == 1 || in == 3 || in == 564 || in == 565 || in == 2 || in == 4 || in == 5;
  return 0;
}
int orc_pconv_can_inplace(int in, int out) {
  if ((props(in) & P_RGB) && (props(out) & P_RGB)) return ((in <= 2) == (out <= 2));      /* equal pixel size */
  return (in == 512 && out == 513) || (in == 513 && out == 512);
}
