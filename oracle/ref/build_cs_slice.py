#!/usr/bin/env python3
"""Build oracle/_ref/libcsref.so from line-range slices of the REFERENCE's src/colourspace.c.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference).

src/colourspace.c cannot be compiled as a whole here (#include "main.h" drags in GTK / glib /
libswscale headers that this image lacks -- SURVEY.md section 8c).  The pixel loops themselves are
plain C, so this script concatenates the *reference's own text*, by line range, straight from
/root/reference/src/{maths.h,colourspace.h,colourspace.c} into a scratch translation unit under
oracle/_ref/ (git-ignored; never committed), in front of which sits a prelude that only supplies
typedefs / attribute macros / a `prefs` struct / weed constants (no algorithmic code: rounding,
clamping and table maths all come from the reference's lines), and after which sit thin extern
wrappers so that ctypes can reach the `static` functions.

What is exported (all call the reference's functions unmodified):
  csref_tables(...)           conversion tables   src/colourspace.c:851-1105
  csref_gamma_lut8(...)       create_gamma_lut8   src/colourspace.c:655-736
  csref_unal(...)             al / unal tables    src/colourspace.c:1141-1160
  csref_cavg(...)             cavg tables         src/colourspace.c:190-217
  csref_yuv420p_to_rgb(...)   convert_yuv420p_to_rgb_frame   :3260-3904
  csref_k1(...)               the 13 RGB<->RGB swizzles      :9259-10577
  csref_gamma_apply(...)      gamma_convert_layer_thread     :14034-14060
  csref_k4(...) / csref_k3(...)  the RGB -> YUV and YUV -> RGB conversions :2750-3258, :5129-6440, :6616-7498
"""
import os
import subprocess
import sys

REF = os.environ.get("LIVES_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.normpath(os.path.join(HERE, "..", "_ref"))
# scratch translation units and the include-path symlinks stay outside the repo (build_ref.sh sets LIVES_REF_WORK); only the .so lands in oracle/_ref
WORK = os.environ.get("LIVES_REF_WORK", os.path.join(os.environ.get("TMPDIR", "/tmp"), "lives_ref_work"))


def lines(path, a, b):
    with open(os.path.join(REF, path), "r", errors="replace") as f:
        all_lines = f.readlines()
    return "/* ---- %s:%d-%d ---- */\n" % (path, a, b) + "".join(all_lines[a - 1:b]) + "\n"


PRELUDE = r'''
/* prelude: typedefs / attributes / constants only (see build_cs_slice.py docstring) */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <unistd.h>
typedef int boolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
typedef void weed_layer_t;
#define WEED_MAXPPLANES 4
#define LIVES_RESTRICT __restrict__
#define LIVES_INLINE static inline
#define LIVES_LOCAL_INLINE static inline
#define LIVES_GLOBAL_INLINE
#define LIVES_CONST
#define LIVES_HOT
#define LIVES_FLATTEN
#define LIVES_UNLIKELY(a) (a)
#define LIVES_LIKELY(a) (a)
#define LIVES_THRDATTR_PRIORITY 0
#define USE_THREADS 1
#define lives_calloc calloc
#define lives_malloc malloc
#define lives_free free
#define lives_memcpy memcpy
#define lives_memset memset
#define MIN(a, b) ((a) < (b) ? (a) : (b))
/* weed palette / clamping / gamma constants: the reference's own header */
#include <weed/weed-palettes.h>
#define PB_QUALITY_LOW 1
#define PB_QUALITY_MED 2
#define PB_QUALITY_HIGH 3
typedef struct { short pb_quality; int nfx_threads; double screen_gamma; } _prefs;
static _prefs _the_prefs = {PB_QUALITY_MED, 1, 1.4};
static _prefs *prefs = &_the_prefs;
/* threads: run the slice workers synchronously on the calling thread */
typedef int lives_thread_t;
typedef void *(*lives_thread_func_t)(void *);
#define lives_thread_create(thr, attr, func, arg) (*(thr) = NULL, (void)(func)(arg), 0)
#define lives_thread_join(thr, ret) do {} while (0)
'''

THREADVAR = r'''
static struct _conv_array _tv_conv_arrays;
#define THREADVAR(x) _tv_##x
'''

# forward declarations for the *_frame_thread functions referenced before definition
FWD = r'''
static void *convert_yuv420p_to_rgb_frame_thread(void *);
static void *convert_swap3_frame_thread(void *);
static void *convert_swap4_frame_thread(void *);
static void *convert_swap3addpost_frame_thread(void *);
static void *convert_swap3addpre_frame_thread(void *);
static void *convert_swap3postalpha_frame_thread(void *);
static void *convert_swap3prealpha_frame_thread(void *);
static void *convert_addpost_frame_thread(void *);
static void *convert_addpre_frame_thread(void *);
static void *convert_swap3delpost_frame_thread(void *);
static void *convert_delpost_frame_thread(void *);
static void *convert_delpre_frame_thread(void *);
static void *convert_swap3delpre_frame_thread(void *);
static void *convert_swapprepost_frame_thread(void *);
static void *convert_swab_frame_thread(void *);
'''

WRAPPERS = r'''
/* ---- extern wrappers (this repo's code; they only forward to the reference's functions) ---- */
void csref_set_prefs(int pb_quality, int nfx_threads, double screen_gamma) {
  prefs->pb_quality = (short)pb_quality; prefs->nfx_threads = nfx_threads; prefs->screen_gamma = screen_gamma;
}
static void ensure_tables(void) {
  if (!conv_RY_inited) init_RGB_to_YUV_tables();
  if (!conv_YR_inited) init_YUV_to_RGB_tables();
  if (!conv_YY_inited) init_YUV_to_YUV_tables();
  if (!avg_inited) init_average();
  if (!unal_inited) init_unal();
}
/* which: 0 = YCbCr clamped, 1 = YCbCr unclamped, 2 = BT709 clamped, 3 = BT709 unclamped
   rgb2yuv[9][256]: Y_R Y_G Y_B Cb_R Cb_G Cb_B Cr_R Cr_G Cr_B ; yuv2rgb[5][256]: RGB_Y R_Cr G_Cb G_Cr B_Cb */
void csref_tables(int which, int *rgb2yuv, int *yuv2rgb) {
  int clamping = (which & 1) ? WEED_YUV_CLAMPING_UNCLAMPED : WEED_YUV_CLAMPING_CLAMPED;
  int subspace = (which & 2) ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR;
  ensure_tables();
  set_conversion_arrays(clamping, subspace);
  memcpy(rgb2yuv + 0 * 256, Y_R, 1024); memcpy(rgb2yuv + 1 * 256, Y_G, 1024); memcpy(rgb2yuv + 2 * 256, Y_B, 1024);
  memcpy(rgb2yuv + 3 * 256, Cb_R, 1024); memcpy(rgb2yuv + 4 * 256, Cb_G, 1024); memcpy(rgb2yuv + 5 * 256, Cb_B, 1024);
  memcpy(rgb2yuv + 6 * 256, Cr_R, 1024); memcpy(rgb2yuv + 7 * 256, Cr_G, 1024); memcpy(rgb2yuv + 8 * 256, Cr_B, 1024);
  memcpy(yuv2rgb + 0 * 256, RGB_Y, 1024); memcpy(yuv2rgb + 1 * 256, R_Cr, 1024); memcpy(yuv2rgb + 2 * 256, G_Cb, 1024);
  memcpy(yuv2rgb + 3 * 256, G_Cr, 1024); memcpy(yuv2rgb + 4 * 256, B_Cb, 1024);
}
void csref_yuv_yuv_tables(uint8_t *yc2u, uint8_t *uvc2u, uint8_t *yu2c, uint8_t *uvu2c) {
  ensure_tables();
  memcpy(yc2u, Yclamped_to_Yunclamped, 256); memcpy(uvc2u, UVclamped_to_UVunclamped, 256);
  memcpy(yu2c, Yunclamped_to_Yclamped, 256); memcpy(uvu2c, UVunclamped_to_UVclamped, 256);
}
void csref_cavg(uint8_t *c, uint8_t *u) { ensure_tables(); memcpy(c, cavgc, 65536); memcpy(u, cavgu, 65536); }
void csref_unal(int *t_unal, int *t_al) { ensure_tables(); memcpy(t_unal, unal, 65536 * 4); memcpy(t_al, al, 65536 * 4); }
void csref_unal_yuv(int *t_unalcy, int *t_alcy, int *t_unalcuv, int *t_alcuv) {
  ensure_tables();
  memcpy(t_unalcy, unalcy, 65536 * 4); memcpy(t_alcy, alcy, 65536 * 4);
  memcpy(t_unalcuv, unalcuv, 65536 * 4); memcpy(t_alcuv, alcuv, 65536 * 4);
}
/* NB: call once per fresh process for a clean LUT (cache-poisoning quirk, SURVEY appendix A2) */
int csref_gamma_lut8(double fileg, int from, int to, uint8_t *out) {
  uint8_t *lut;
  init_gamma_tx();
  lut = create_gamma_lut8(fileg, from, to);
  if (!lut) return 0;
  memcpy(out, lut, 256);
  return 1;
}
int csref_gamma_lut16(double fileg, int from, int to, uint16_t *out) {
  uint16_t *lut;
  init_gamma_tx();
  lut = create_gamma_lut(fileg, from, to);
  if (!lut) return 0;
  memcpy(out, lut, 65536 * 2);
  return 1;
}
void csref_gamma_consts(float *out8) {
  init_gamma_tx();
  for (int i = 0; i < N_GAMMA_TYPES; i++) {
    out8[i * 4 + 0] = gamma_tx[i].offs; out8[i * 4 + 1] = gamma_tx[i].lin;
    out8[i * 4 + 2] = gamma_tx[i].thresh; out8[i * 4 + 3] = gamma_tx[i].pf;
  }
}
void csref_yuv420p_to_rgb(uint8_t *y, uint8_t *u, uint8_t *v, int width, int height, int *istrides,
                          int orowstride, uint8_t *dest, int add_alpha, int is_422, int clamping, int subspace,
                          uint16_t *lut16) {
  uint8_t *src[3] = {y, u, v};
  ensure_tables();
  /* tgt_gamma = 0: an explicit LUT16 (may be NULL) is passed straight through */
  convert_yuv420p_to_rgb_frame(src, width, height, 0, istrides, orowstride, dest, add_alpha, is_422,
                               WEED_YUV_SAMPLING_DEFAULT, clamping, subspace, 0, 0, lut16, -1);
}
/* op ids follow the order of the definitions in src/colourspace.c:9259-10577 */
int csref_k1(int op, uint8_t *src, int width, int height, int irow, int orow, uint8_t *dest, uint8_t *lut8,
             int alpha_first) {
  /* some threaded bodies free() the LUT they were handed (lives_gamma_lut8_free): give them a heap copy */
  if (lut8) { uint8_t *cp = (uint8_t *)malloc(256); memcpy(cp, lut8, 256); lut8 = cp; }
  switch (op) {
  case 0: convert_swap3_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 1: convert_swap4_frame(src, width, height, irow, orow, dest, lut8, alpha_first, -1); break;
  case 2: convert_swap3addpost_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 3: convert_swap3addpre_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 4: convert_swap3postalpha_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 5: convert_swap3prealpha_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 6: convert_addpost_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 7: convert_addpre_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 8: convert_swap3delpost_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 9: convert_delpost_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 10: convert_delpre_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 11: convert_swap3delpre_frame(src, width, height, irow, orow, dest, lut8, -1); break;
  case 12: convert_swapprepost_frame(src, width, height, irow, orow, dest, lut8, alpha_first, -1); break;
  default: return -1;
  }
  return 0;
}
/* K4: in_order 0 RGB 1 BGR 2 ARGB; out_fmt 0 packed 1 planar 2 UYVY 3 YUYV 4 420P 5 422P (see oracle/lives_oracle.h) */
int csref_k4(int in_order, int in_alpha, int out_fmt, int out_alpha, uint8_t *src, int irow, int width, int height,
             uint8_t **dst, int *orows, int clamping, int bt709) {
  const int subspace = bt709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR;
  ensure_tables();
  avg_chromaf = avg_chromaf_fast;
  switch (out_fmt) {
  case 0:
    if (in_order == 0) convert_rgb_to_yuv_frame(src, width, height, irow, orows[0], dst[0], in_alpha, out_alpha, clamping, -1);
    else if (in_order == 1) convert_bgr_to_yuv_frame(src, width, height, irow, orows[0], dst[0], in_alpha, out_alpha, clamping, -1);
    else convert_argb_to_yuv_frame(src, width, height, irow, orows[0], dst[0], out_alpha, clamping, -1);
    return 0;
  case 1:
    if (in_order == 0) convert_rgb_to_yuvp_frame(src, width, height, irow, orows[0], dst, in_alpha, out_alpha, clamping, -1);
    else if (in_order == 1) convert_bgr_to_yuvp_frame(src, width, height, irow, orows[0], dst, in_alpha, out_alpha, clamping, -1);
    else convert_argb_to_yuvp_frame(src, width, height, irow, orows[0], dst, out_alpha, clamping, -1);
    return 0;
  case 2:
    if (in_order == 0) convert_rgb_to_uyvy_frame(src, width, height, irow, orows[0], (uyvy_macropixel *)dst[0], in_alpha, clamping, NULL, -1);
    else if (in_order == 1) convert_bgr_to_uyvy_frame(src, width, height, irow, orows[0], (uyvy_macropixel *)dst[0], in_alpha, clamping, NULL, -1);
    else convert_argb_to_uyvy_frame(src, width, height, irow, orows[0], (uyvy_macropixel *)dst[0], clamping, NULL, -1);
    return 0;
  case 3:
    if (in_order == 0) convert_rgb_to_yuyv_frame(src, width, height, irow, orows[0], (yuyv_macropixel *)dst[0], in_alpha, clamping, NULL, -1);
    else if (in_order == 1) convert_bgr_to_yuyv_frame(src, width, height, irow, orows[0], (yuyv_macropixel *)dst[0], in_alpha, clamping, NULL, -1);
    else convert_argb_to_yuyv_frame(src, width, height, irow, orows[0], (yuyv_macropixel *)dst[0], clamping, NULL, -1);
    return 0;
  case 4: case 5:
    if (in_order == 0) convert_rgb_to_yuv420_frame(src, width, height, irow, orows, dst, out_fmt == 5, in_alpha, subspace, clamping);
    else if (in_order == 1) convert_bgr_to_yuv420_frame(src, width, height, irow, orows, dst, out_fmt == 5, in_alpha, subspace, clamping);
    else return -1;
    return 0;
  }
  return -1;
}
/* K4 with a 16-bit gamma LUT (rgb2uyvy_with_gamma / rgb2yuyv_with_gamma): out_fmt 2 UYVY, 3 YUYV.  The frame functions take ownership of nothing here:
   with one thread they never free the LUT */
int csref_k4_lut16(int in_order, int in_alpha, uint8_t *src, int width, int height, int irow, int out_fmt, uint8_t *dst, int orow, int clamping, uint16_t *lut) {
  ensure_tables();
  if (out_fmt == 2) {
    if (in_order == 0) convert_rgb_to_uyvy_frame(src, width, height, irow, orow, (uyvy_macropixel *)dst, in_alpha, clamping, lut, -1);
    else if (in_order == 1) convert_bgr_to_uyvy_frame(src, width, height, irow, orow, (uyvy_macropixel *)dst, in_alpha, clamping, lut, -1);
    else convert_argb_to_uyvy_frame(src, width, height, irow, orow, (uyvy_macropixel *)dst, clamping, lut, -1);
    return 0;
  }
  if (out_fmt == 3) {
    if (in_order == 0) convert_rgb_to_yuyv_frame(src, width, height, irow, orow, (yuyv_macropixel *)dst, in_alpha, clamping, lut, -1);
    else if (in_order == 1) convert_bgr_to_yuyv_frame(src, width, height, irow, orow, (yuyv_macropixel *)dst, in_alpha, clamping, lut, -1);
    else convert_argb_to_yuyv_frame(src, width, height, irow, orow, (yuyv_macropixel *)dst, clamping, lut, -1);
    return 0;
  }
  return -1;
}
/* K5b: YUV -> YUV repacks with the arguments the dispatcher passes (:12937-13750); WEED_PALETTE_* numbers, width in pixels */
int csref_yuv_repack(int in_pal, int out_pal, uint8_t **src, int *irows_in, uint8_t **dst, int *orows_in, int width, int height,
                     int clamping, int sampling) {
  int irows[4], orows[4];
  const int in444 = (in_pal == 544 || in_pal == 545), in420 = (in_pal == 512 || in_pal == 513), inpk = (in_pal == 564 || in_pal == 565);
  memcpy(irows, irows_in, sizeof(irows)); memcpy(orows, orows_in, sizeof(orows));
  ensure_tables();
  avg_chromaf = avg_chromaf_fast;
  /* K5c: the 4:1:1 pairs; width in pixels, the dispatcher hands macropixels (:13793-13846) resp. pixels (:13024-13029 ...) */
  if (in_pal == 595) {
    yuv411_macropixel *s = (yuv411_macropixel *)src[0];
    const int wm = width >> 2;
    switch (out_pal) {
    case 588: convert_yuv411_to_yuv888_frame(s, wm, height, dst[0], FALSE, clamping); return 0;
    case 589: convert_yuv411_to_yuv888_frame(s, wm, height, dst[0], TRUE, clamping); return 0;
    case 544: convert_yuv411_to_yuvp_frame(s, wm, height, dst, FALSE, clamping); return 0;
    case 545: convert_yuv411_to_yuvp_frame(s, wm, height, dst, TRUE, clamping); return 0;
    case 564: convert_yuv411_to_uyvy_frame(s, wm, height, (uyvy_macropixel *)dst[0], clamping); return 0;
    case 565: convert_yuv411_to_yuyv_frame(s, wm, height, (yuyv_macropixel *)dst[0], clamping); return 0;
    case 522: convert_yuv411_to_yuv422_frame(s, wm, height, dst, clamping); return 0;
    case 512: convert_yuv411_to_yuv420_frame(s, wm, height, dst, FALSE, clamping); return 0;
    case 513: convert_yuv411_to_yuv420_frame(s, wm, height, dst, TRUE, clamping); return 0;
    }
    return -1;
  }
  if (out_pal == 595) {
    yuv411_macropixel *d = (yuv411_macropixel *)dst[0];
    if (in444) { convert_yuvp_to_yuv411_frame(src, width, height, irows[0], d, clamping); return 0; }
    if (in_pal == 564) { convert_uyvy_to_yuv411_frame((uyvy_macropixel *)src[0], width >> 1, height, d, clamping); return 0; }
    if (in_pal == 565) { convert_yuyv_to_yuv411_frame((yuyv_macropixel *)src[0], width >> 1, height, d, clamping); return 0; }
    if (in_pal == 588 || in_pal == 589) {
      set_conversion_arrays(clamping, WEED_YUV_SUBSPACE_YCBCR);
      convert_yuv888_to_yuv411_frame(src[0], width, height, irows[0], d, in_pal == 589);
      return 0;
    }
    if (in420 || in_pal == 522) { convert_yuv420_to_yuv411_frame(src, width, height, d, in_pal == 522, clamping); return 0; }
    return -1;
  }
  if (in444 && (out_pal == 588 || out_pal == 589)) { convert_combineplanes_frame(src, width, height, irows[0], orows[0], dst[0], in_pal == 545, out_pal == 589); return 0; }
  if (in_pal == 588 && out_pal == 544) { convert_splitplanes_frame(src[0], width, height, irows[0], orows, dst, FALSE, FALSE); return 0; }
  if (in_pal == 545 && out_pal == 544) { convert_yuvap_to_yuvp_frame(src, width, height, irows[0], orows[0], dst); return 0; }
  if (in_pal == 544 && out_pal == 545) { convert_yuvp_to_yuvap_frame(src, width, height, irows[0], orows[0], dst); return 0; }
  if (in_pal == 588 && out_pal == 589) { convert_addpost_frame(src[0], width, height, irows[0], orows[0], dst[0], NULL, -1); return 0; }
  if (in_pal == 589 && out_pal == 588) { convert_delpost_frame(src[0], width, height, irows[0], orows[0], dst[0], NULL, -1); return 0; }
  if (inpk && (out_pal == 564 || out_pal == 565) && in_pal != out_pal) {      /* in place (:13139) */
    for (int y = 0; y < height; y++) memcpy(dst[0] + (size_t)y * orows[0], src[0] + (size_t)y * irows[0], (size_t)width * 2);
    convert_swab_frame(dst[0], width >> 1, height, orows[0], -1);
    return 0;
  }
  if (in420 && (out_pal == 588 || out_pal == 589)) { convert_quad_chroma_packed(src, width, height, irows, orows[0], dst[0], out_pal == 589, sampling, clamping); return 0; }
  if (in_pal == 522 && (out_pal == 588 || out_pal == 589)) { convert_double_chroma_packed(src, width, height, irows, orows[0], dst[0], out_pal == 589, sampling, clamping); return 0; }
  if (in420 && out_pal == 564) { convert_yuv420_to_uyvy_frame(src, width, height, irows, orows[0], (uyvy_macropixel *)dst[0], clamping); return 0; }
  if (in420 && out_pal == 565) { convert_yuv420_to_yuyv_frame(src, width, height, irows, orows[0], (yuyv_macropixel *)dst[0], clamping); return 0; }
  if (in420 && out_pal == 522) {
    for (int y = 0; y < height; y++) memcpy(dst[0] + (size_t)y * orows[0], src[0] + (size_t)y * irows[0], (size_t)width);   /* weed_layer_copy_single_plane */
    convert_double_chroma(src, width >> 1, height >> 1, irows, orows, dst, clamping);
    return 0;
  }
  if (in444 && (out_pal == 512 || out_pal == 513)) { convert_yuvp_to_yuv420_frame(src, width, height, irows, orows, dst, clamping); return 0; }
  if (in444 && out_pal == 564) { convert_yuv_planar_to_uyvy_frame(src, width, height, irows[0], orows[0], (uyvy_macropixel *)dst[0], clamping); return 0; }
  if (in444 && out_pal == 565) { convert_yuv_planar_to_yuyv_frame(src, width, height, irows[0], orows[0], (yuyv_macropixel *)dst[0], clamping); return 0; }
  if ((in_pal == 588 || in_pal == 589) && (out_pal == 512 || out_pal == 513)) { convert_yuv888_to_yuv420_frame(src[0], width, height, irows[0], orows, dst, in_pal == 589, clamping); return 0; }
  if ((in_pal == 588 || in_pal == 589) && out_pal == 522) { convert_yuv888_to_yuv422_frame(src[0], width, height, irows[0], orows, dst, in_pal == 589, clamping); return 0; }
  if ((in_pal == 588 || in_pal == 589) && out_pal == 564) { convert_yuv888_to_uyvy_frame(src[0], width, height, irows[0], orows[0], (uyvy_macropixel *)dst[0], in_pal == 589, clamping); return 0; }
  if ((in_pal == 588 || in_pal == 589) && out_pal == 565) { convert_yuv888_to_yuyv_frame(src[0], width, height, irows[0], orows[0], (yuyv_macropixel *)dst[0], in_pal == 589, clamping); return 0; }
  if (inpk && out_pal == 522) {
    if (in_pal == 564) convert_uyvy_to_yuv422_frame((uyvy_macropixel *)src[0], width >> 1, height, dst);
    else convert_yuyv_to_yuv422_frame((yuyv_macropixel *)src[0], width >> 1, height, dst);
    return 0;
  }
  if (inpk) {
    const int mw = width >> 1;
    if (out_pal == 544 || out_pal == 545) {
      if (in_pal == 564) convert_uyvy_to_yuvp_frame((uyvy_macropixel *)src[0], mw, height, irows[0], orows, dst, out_pal == 545);
      else convert_yuyv_to_yuvp_frame((yuyv_macropixel *)src[0], mw, height, irows[0], orows, dst, out_pal == 545);
      return 0;
    }
    if (out_pal == 588 || out_pal == 589) {
      if (in_pal == 564) convert_uyvy_to_yuv888_frame((uyvy_macropixel *)src[0], mw, height, irows[0], orows[0], dst[0], out_pal == 589);
      else convert_yuyv_to_yuv888_frame((yuyv_macropixel *)src[0], mw, height, irows[0], orows[0], dst[0], out_pal == 589);
      return 0;
    }
    if (out_pal == 512 || out_pal == 513) {
      if (in_pal == 564) convert_uyvy_to_yuv420_frame((uyvy_macropixel *)src[0], mw, height, dst, clamping);
      else convert_yuyv_to_yuv420_frame((yuyv_macropixel *)src[0], mw, height, dst, clamping);
      return 0;
    }
  }
  return -1;
}
/* K3: in_fmt 0 packed 1 planar 2 UYVY 3 YUYV; out_order 0 RGB 1 BGR 2 ARGB; width in pixels */
int csref_k3(int in_fmt, int in_alpha, int out_order, int out_alpha, uint8_t **src, int *irows, int width, int height,
             uint8_t *dst, int orow, int clamping, int bt709) {
  const int subspace = bt709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR;
  ensure_tables();
  switch (in_fmt) {
  case 0:
    if (!in_alpha) {
      if (out_order == 0) convert_yuv888_to_rgb_frame(src[0], width, height, irows[0], orow, dst, out_alpha, clamping, subspace, -1);
      else if (out_order == 1) convert_yuv888_to_bgr_frame(src[0], width, height, irows[0], orow, dst, out_alpha, clamping, subspace, -1);
      else convert_yuv888_to_argb_frame(src[0], width, height, irows[0], orow, dst, clamping, subspace, -1);
    } else {
      if (out_order == 0) convert_yuva8888_to_rgba_frame(src[0], width, height, irows[0], orow, dst, !out_alpha, clamping, subspace, -1);
      else if (out_order == 1) convert_yuva8888_to_bgra_frame(src[0], width, height, irows[0], orow, dst, !out_alpha, clamping, subspace, -1);
      else convert_yuva8888_to_argb_frame(src[0], width, height, irows[0], orow, dst, clamping, subspace, -1);
    }
    return 0;
  case 1:
    if (out_order == 0) convert_yuv_planar_to_rgb_frame(src, width, height, irows[0], orow, dst, in_alpha, out_alpha, clamping, -1);
    else if (out_order == 1) convert_yuv_planar_to_bgr_frame(src, width, height, irows[0], orow, dst, in_alpha, out_alpha, clamping, -1);
    else return -1;
    return 0;
  case 2:
    if (out_order == 0) convert_uyvy_to_rgb_frame((uyvy_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, out_alpha, clamping, subspace, -1);
    else if (out_order == 1) convert_uyvy_to_bgr_frame((uyvy_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, out_alpha, clamping, -1);
    else convert_uyvy_to_argb_frame((uyvy_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, clamping, -1);
    return 0;
  case 3:
    if (out_order == 0) convert_yuyv_to_rgb_frame((yuyv_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, out_alpha, clamping, -1);
    else if (out_order == 1) convert_yuyv_to_bgr_frame((yuyv_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, out_alpha, clamping, -1);
    else convert_yuyv_to_argb_frame((yuyv_macropixel *)src[0], width >> 1, height, irows[0], orow, dst, clamping, -1);
    return 0;
  }
  return -1;
}
/* K4b: RGB -> YUV411; width in pixels (:12627-12632) */
int csref_rgb_to_yuv411(uint8_t *src, int irow, int width, int height, int in_order, int in_alpha, uint8_t *dst, int clamping) {
  ensure_tables();
  if (in_order == 0) convert_rgb_to_yuv411_frame(src, width, height, irow, (yuv411_macropixel *)dst, in_alpha, clamping);
  else if (in_order == 1) convert_bgr_to_yuv411_frame(src, width, height, irow, (yuv411_macropixel *)dst, in_alpha, clamping);
  else convert_argb_to_yuv411_frame(src, width, height, irow, (yuv411_macropixel *)dst, clamping);
  return 0;
}
/* K3b: YUV411 -> RGB; width in macropixels (what the dispatcher passes, :13755-13795) */
int csref_yuv411_to_rgb(uint8_t *src, int width_mp, int height, uint8_t *dst, int orow, int out_order, int out_alpha, int clamping) {
  ensure_tables();
  avg_chromaf = avg_chromaf_fast;
  if (out_order == 0) convert_yuv411_to_rgb_frame((yuv411_macropixel *)src, width_mp, height, orow, dst, out_alpha, clamping);
  else if (out_order == 1) convert_yuv411_to_bgr_frame((yuv411_macropixel *)src, width_mp, height, orow, dst, out_alpha, clamping);
  else convert_yuv411_to_argb_frame((yuv411_macropixel *)src, width_mp, height, orow, dst, clamping);
  return 0;
}
void csref_gamma_apply(uint8_t *pixels, int width, int height, int rowstride, int psize, int alpha_first,
                       int xoffset_px, uint8_t *lut8) {
  lives_cc_params cc;
  memset(&cc, 0, sizeof(cc));
  cc.src = pixels; cc.hsize = width; cc.vsize = height; cc.psize = psize; cc.orowstrides[0] = rowstride;
  cc.alpha_first = alpha_first; cc.xoffset = (size_t)xoffset_px * psize; cc.lut8 = lut8;
  gamma_convert_layer_thread(&cc);
}
'''


def find_line(path, needle, start=1):
    with open(os.path.join(REF, path), "r", errors="replace") as f:
        for i, l in enumerate(f, 1):
            if i >= start and needle in l:
                return i
    raise SystemExit("needle not found: %s in %s" % (needle, path))


def main():
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(WORK, exist_ok=True)
    cs = "src/colourspace.c"
    ch = "src/colourspace.h"
    parts = [PRELUDE]
    parts.append(lines("src/memory.c", 1481, 1484))   # union split4
    parts.append(lines("src/memory.c", 1491, 1498))   # swab2 (over libc's swab)
    parts.append(lines("src/memory.c", 1501, 1522))   # swab4: the byte reversal the K1 swap4 family calls -- the prelude carries no pixel code
    parts.append(lines("src/maths.h", 88, 88))      # CLAMP0255f
    parts.append(lines("src/maths.h", 101, 104))    # CEIL, ALIGN_CEIL
    parts.append(lines("src/maths.h", 118, 118))    # myround
    parts.append(lines(ch, 12, 14))                 # #define USE_EXTEND (unconditional in the reference)
    parts.append(lines(ch, 16, 29))                 # CLAMP16bit ... CLAMP0_255i, gamma ids
    parts.append(lines(ch, 45, 47))                 # RS_ALIGN
    parts.append(lines(ch, 50, 63))                 # FP_BITS, SCALE_FACTOR
    parts.append(lines(ch, 65, 131))                # struct _conv_array, K*, clamp consts
    parts.append(lines(ch, 140, 143))
    parts.append(lines(ch, 152, 185))               # gamma_const_t, INIT_GAMMA, init_gamma_tx
    parts.append(lines(ch, 189, 258))               # macropixels, lives_cc_params
    parts.append(THREADVAR)
    parts.append(lines(cs, 54, 419))                # tables, init_average, set_conversion_arrays, accessors
    parts.append(lines(cs, 575, 575))               # unal_inited
    parts.append(lines(cs, 592, 829))               # clamp0255f, YY tables, gamma statics, LUT builders
    parts.append(lines(cs, 832, 843))               # spc_rnd
    parts.append(lines(cs, 851, 1160))              # table inits, init_unal
    parts.append(lines(cs, 2345, 2365))             # yuv2rgb_int, xyuv2rgb, SETVARS
    parts.append(lines(cs, 2386, 2392))             # xyuv2rgb_with_gamma
    parts.append(FWD)
    parts.append(lines(cs, 1985, 2019))             # forward declarations of the K3 / K4 thread bodies
    parts.append(lines(cs, 2040, 2070))             # forward declarations of the pixel helpers
    parts.append(lines(cs, 2079, 2083))             # avg_chroma macros, avg_chromaf pointer
    parts.append(lines(cs, 2097, 2101))             # avg_chromaf_fast
    parts.append(lines(cs, 2116, 2117))             # avg_chroma_3_1f, avg_chroma_1_3f
    parts.append(lines(cs, 2119, 2264))             # rgb2yuv, *_with_gamma, rgb2uyvy, rgb2yuyv, rgb16_2uyvy
    parts.append(lines(cs, 2366, 2385))             # yuv2rgb_float, yuv2rgb / yuv2bgr macros, yuv2rgb_with_gamma
    parts.append(lines(cs, 2408, 2459))             # uyvy2rgb, yuyv2rgb, yuv888_2_rgb ...
    parts.append(lines(cs, 2750, 3258))             # K3: yuv888 / yuva8888 -> rgb / bgr / argb
    parts.append(lines(cs, 3260, 3925))             # convert_yuv420p_to_rgb_frame (+ thread)
    parts.append(lines(cs, 5129, 5698))             # K4: rgb / bgr / argb -> uyvy / yuyv
    parts.append(lines(cs, 5700, 6248))             # K4: rgb / bgr / argb -> yuv888 / yuva8888 / yuv(a)444p
    parts.append(lines(cs, 6250, 6440))             # K4: rgb / argb / bgr -> yuv420p / yuv422p
    parts.append(lines(cs, 6616, 7102))             # K3: uyvy / yuyv -> rgb / bgr / argb
    parts.append(lines(cs, 7200, 7498))             # K3: yuv(a)444p -> rgb / bgr / argb
    parts.append(lines(cs, 7104, 7198))             # K5b: yuv420p -> uyvy / yuyv
    parts.append(lines(cs, 7500, 7753))             # K5b: yuv(a)444p -> uyvy / yuyv / yuv(a)888(8) / yuv(a)444p / yuv420p
    parts.append(lines(cs, 7800, 7971))             # K5b: uyvy / yuyv -> yuv(a)444p / yuv(a)888(8) / yuv420p
    parts.append(lines(cs, 2461, 2474))             # K5b: uyvy_2_yuv422, yuyv_2_yuv422
    parts.append(lines(cs, 8035, 8270))             # K5b: yuv(a)888(8) -> yuv420p / yuv422p / uyvy / yuyv, uyvy / yuyv -> yuv422p
    parts.append(lines(cs, 2322, 2343))             # K4b: rgb2_411
    parts.append(lines(cs, 6499, 6615))             # K4b: rgb / bgr / argb -> yuv411
    parts.append(lines(cs, 8305, 8620))             # K3b: yuv411 -> rgb / bgr / argb
    parts.append(lines(cs, 7755, 7798))             # K5c: yuv(a)444p -> yuv411
    parts.append(lines(cs, 7973, 8033))             # K5c: uyvy / yuyv -> yuv411
    parts.append(lines(cs, 8272, 8303))             # K5c: yuv(a)888(8) -> yuv411
    parts.append(lines(cs, 8622, 9196))             # K5c: yuv411 -> yuv(a)888(8) / yuv(a)444p / uyvy / yuyv / yuv422p / yuv420p, yuv420p / yuv422p -> yuv411
    parts.append(lines(cs, 9198, 9257))             # K5b: convert_splitplanes_frame
    parts.append(lines(cs, 10578, 10639))           # K5b: convert_halve_chroma, convert_double_chroma
    parts.append(lines(cs, 10715, 10873))           # K5d: convert_quad_chroma_packed, convert_double_chroma_packed
    parts.append(lines(cs, 9259, 10577))            # K1 swizzle family
    parts.append(lines(cs, 14034, 14060))           # gamma_convert_layer_thread
    parts.append(WRAPPERS)
    src = os.path.join(WORK, "cs_slice.c")
    with open(src, "w") as f:
        f.write("/* GENERATED SCRATCH FILE -- contains reference text; never commit (oracle/_ref is git-ignored) */\n")
        f.write("".join(parts))
    so = os.path.join(OUT, "libcsref.so")
    cmd = ["gcc", "-shared", "-fPIC", "-O1", "-w", "-fno-strict-aliasing", "-I", os.path.join(WORK, "inc"), "-o", so, src, "-lm"]
    print(" ".join(cmd))
    r = subprocess.run(cmd)
    if r.returncode:
        sys.exit(r.returncode)
    print("built", so)
    # compositor: only paint_pixel (lives-plugins/weed-plugins/gdk/compositor.c:120-125) is sliceable -- the rest of that
    # plugin needs gdk-pixbuf headers.  The wrapper walks the paint loop of :288-293 over one layer.
    comp = os.path.join(WORK, "comp_slice.c")
    with open(comp, "w") as f:
        f.write("/* GENERATED SCRATCH FILE -- contains reference text; never commit */\n#include <stdint.h>\n")
        f.write(lines("lives-plugins/weed-plugins/gdk/compositor.c", 120, 125))
        f.write('''
void compref_paint_layer(unsigned char *dst, int orowstride, int owidth, int oheight, int psize, unsigned char *src, int irowstride,
                         int out_width, int out_height, int myoffsx, int myoffsy, double myalpha) {
  int x, y;
  for (y = myoffsy; y < oheight && y < myoffsy + out_height; y++)
    for (x = myoffsx; x < owidth && x < myoffsx + out_width; x++)
      paint_pixel(dst, y * orowstride + x * psize, src, (y - myoffsy) * irowstride + (x - myoffsx) * psize, myalpha);
}
''')
    so2 = os.path.join(OUT, "libcompref.so")
    r = subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-w", "-o", so2, comp])
    if r.returncode:
        sys.exit(r.returncode)
    print("built", so2)


if __name__ == "__main__":
    main()
