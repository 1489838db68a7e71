// host_tables.cpp -- host-side (CPU, one-off) builders the kernels consume: colour-conversion tables,
// gamma LUTs, polyphase filter banks, rowstride rule.  These are the product's own implementations of
//   init_RGB_to_YUV_tables / init_YUV_to_RGB_tables   src/colourspace.c:851-1105
//   create_gamma_lut8                                  src/colourspace.c:655-736
//   calc_rowstrides                                    src/colourspace.c:11252-11366
// and are pinned against reference-generated fixtures in tests/ (they are NOT shared with oracle/).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/lives_gpu.h"
#include "../../include/lives_gpu_weed_abi.h"

namespace {

// reference fixed-point scale: (2^24 - 1) / (2^8 - 1)   (src/colourspace.h:50-63, USE_EXTEND)
constexpr double kScale = 65793.;

inline int32_t nearest(double v) { return v >= 0. ? (int32_t)(v + 0.5) : (int32_t)(v - 0.5); }   // src/maths.h:118

struct Primaries { double kr, kb; };
inline Primaries primaries(bool bt709) { return bt709 ? Primaries{0.2126, 0.0722} : Primaries{0.299, 0.114}; }

// One row of the forward (RGB -> YUV) table set for input level `lvl`.
// The multiplication order follows the reference expressions so every double rounds identically.
void forward_entry(const Primaries &p, bool full_range, int lvl, int32_t out[9]) {
  const double v = (double)lvl;
  const double ys = full_range ? 1. : (235. - 16.) / (255. - 0.);
  const double cs = full_range ? 1. : (240. - 16.) / (255. - 0.);
  const double kg = 1. - p.kr - p.kb;          // luma green weight:   (1 - Kr) - Kb
  const double kg2 = 1. - p.kb - p.kr;         // chroma green weight: (1 - Kb) - Kr
  const double fb = .5 / (1. - p.kb), fr = .5 / (1. - p.kr);
  if (full_range) {
    out[0] = nearest(p.kr * v * kScale);
    out[1] = nearest(kg * v * kScale);
    out[2] = nearest(p.kb * v * kScale);
    out[3] = nearest(-fb * p.kr * v * kScale);
    out[4] = nearest(-fb * kg2 * v * kScale);
    out[5] = nearest((0.5 * v + 128.) * kScale);
    out[6] = nearest((0.5 * v + 128.) * kScale);
    out[7] = nearest(-fr * kg2 * v * kScale);
    out[8] = nearest(-fr * p.kb * v * kScale);
  } else {
    out[0] = nearest(p.kr * v * ys * kScale);
    out[1] = nearest(kg * v * ys * kScale);
    out[2] = nearest((p.kb * v * ys + 16.) * kScale);
    out[3] = nearest(-fb * p.kr * v * cs * kScale);
    out[4] = nearest(-fb * kg2 * v * cs * kScale);
    out[5] = nearest((0.5 * v * cs + 128.) * kScale);
    out[6] = nearest((0.5 * v * cs + 128.) * kScale);
    out[7] = nearest(-fr * kg2 * v * cs * kScale);
    out[8] = nearest(-fr * p.kb * v * cs * kScale);
  }
}

// One row of the inverse (YUV -> RGB) table set: {RGB_Y, R_Cr, G_Cb, G_Cr, B_Cb}.
void inverse_entry(const Primaries &p, bool bt709, bool full_range, int lvl, int32_t out[5]) {
  const double v = (double)lvl;
  // green-from-Cb divisor: (1 + Kb + Kr) for BT.601 tables, (1 + Kb + Kb) for the BT.709 ones -- that is
  // what the reference computes (src/colourspace.c:1017 vs :1061) and therefore what parity requires
  const double gdiv = bt709 ? (1. + p.kb + p.kb) : (1. + p.kb + p.kr);
  double chroma;   // centred chroma level, full-range units
  if (full_range) {
    out[0] = (int32_t)(lvl * kScale);
    chroma = v - 128.;
  } else {
    out[0] = lvl <= 16 ? 0 : lvl < 235 ? nearest((v - 16.) / (235. - 16.) * 255. * kScale) : (int32_t)(255 * kScale);
    if (lvl <= 16) { out[1] = out[2] = out[3] = out[4] = 0; return; }
    if (lvl < 240) chroma = ((v - 16.) / (240. - 16.) * 255.) - 128.;
    else chroma = bt709 ? (255. - 128.) : (((240. - 16.) / (240. - 16.) * 255.) - 128.);
  }
  out[1] = nearest(2. * (1. - p.kr) * chroma * kScale);
  out[2] = nearest(-.5 / gdiv * chroma * kScale);
  out[3] = nearest(-.5 / (1. - p.kr) * chroma * kScale);
  out[4] = nearest(2. * (1. - p.kb) * chroma * kScale);
}

// ---- gamma ----------------------------------------------------------------------------------------------
struct Transfer { float offs, lin, thresh, pf; };

Transfer transfer_for(int gamma_type) {   // src/colourspace.h:152-185 (INIT_GAMMA)
  Transfer t;
  if (gamma_type == WEED_GAMMA_BT709) { t.lin = 4.5; t.thresh = 0.018; t.pf = 1. / .45; }
  else { t.lin = 12.92; t.thresh = 0.04045; t.pf = 2.4; }
  const float knee = powf((t.thresh / t.lin), (1. / t.pf));
  t.offs = (knee - t.thresh) / (1. - (powf((t.thresh / t.lin), (1. / t.pf))));
  return t;
}

inline uint8_t sat8(int n) { return n < 0 ? 0 : n > 255 ? 255 : (uint8_t)n; }

// ---- polyphase filter kernels ------------------------------------------------------------------------------
double tap_weight(int kernel, double d) {
  d = fabs(d);
  switch (kernel) {
  case 0: return d < 1. ? 1. - d : 0.;
  case 1: {
    const double B = 0., C = 0.6;   // swscale's default bicubic parameters
    if (d < 1.) return ((12. - 9. * B - 6. * C) * d * d * d + (-18. + 12. * B + 6. * C) * d * d + (6. - 2. * B)) / 6.;
    if (d < 2.) return ((-B - 6. * C) * d * d * d + (6. * B + 30. * C) * d * d + (-12. * B - 48. * C) * d + (8. * B + 24. * C)) / 6.;
    return 0.;
  }
  default: {
    if (d < 1e-12) return 1.;
    if (d >= 3.) return 0.;
    const double pd = M_PI * d;
    return 3. * sin(pd) * sin(pd / 3.) / (pd * pd);
  }
  }
}

}  // namespace

extern "C" {

int lgpu_conversion_tables(int which, int32_t *rgb2yuv, int32_t *yuv2rgb) {
  const bool full_range = which & 1, bt709 = which & 2;
  const Primaries p = primaries(bt709);
  for (int lvl = 0; lvl < 256; lvl++) {
    int32_t f[9], b[5];
    forward_entry(p, full_range, lvl, f);
    inverse_entry(p, bt709, full_range, lvl, b);
    if (rgb2yuv) for (int t = 0; t < 9; t++) rgb2yuv[t * 256 + lvl] = f[t];
    if (yuv2rgb) for (int t = 0; t < 5; t++) yuv2rgb[t * 256 + lvl] = b[t];
  }
  return LGPU_OK;
}

// The chroma blend's scaling of translucent pixels (lives-plugins/weed-plugins/simple_blend.c:137-145):
//   alpha = (float)a / 255., inv_alpha = 1. - alpha;  s2 = (uint8_t)((float)c2 * alpha);  s1 = (uint8_t)((float)c1 * inv_alpha)
// as integer arithmetic: for every alpha a there is a constant K with (c * K) >> 16 == (uint8_t)((float)c * factor) for all
// c in 0..255 (the float product truncated).  The builder evaluates the reference's expression for all 2 x 65,536 operand
// pairs and intersects the admissible K intervals, so a table it returns is proven; LGPU_E_UNSUPPORTED if some alpha had
// none (never on IEEE hosts).  k2[a] scales layer 2 by alpha, k1[a] scales the track by 1 - alpha; a = 255 is the opaque
// branch of the reference (:128-131, no scaling at all): both constants are 65536, the identity.
int lgpu_alpha_scalers(uint32_t k2[256], uint32_t k1[256]) {
  for (int a = 0; a < 256; a++) {
    const float alpha = (float)a / 255., inv_alpha = 1. - alpha;
    for (int which = 0; which < 2; which++) {
      const float f = which ? inv_alpha : alpha;
      uint32_t *out = which ? k1 : k2;
      if (a == 255) { out[a] = 65536u; continue; }
      int64_t lo = 0, hi = (1 << 24) - 1;                  // 24-bit operand of v_mul_u32_u24
      if ((uint8_t)((float)0 * f) != 0) return LGPU_E_UNSUPPORTED;
      for (int c = 1; c < 256; c++) {
        const int64_t t = (uint8_t)((float)c * f);
        const int64_t l = ((t << 16) + c - 1) / c, h = (((t + 1) << 16) - 1) / c;     // t * 2^16 <= c * K < (t + 1) * 2^16
        if (l > lo) lo = l;
        if (h < hi) hi = h;
      }
      if (lo > hi) return LGPU_E_UNSUPPORTED;
      out[a] = (uint32_t)lo;
    }
  }
  return LGPU_OK;
}

int lgpu_gamma_lut8(double file_gamma, int gamma_from, int gamma_to, double screen_gamma, uint8_t lut[256]) {
  if (file_gamma == 1.0 &&
      (gamma_to == gamma_from || gamma_to == WEED_GAMMA_UNKNOWN || gamma_from == WEED_GAMMA_UNKNOWN)) return 0;
  const float inv_screen = (gamma_to == LIVES_GAMMA_MONITOR) ? 1. / (float)screen_gamma : 0.f;
  // The reference mutates its `gamma_from` argument while filling the table (src/colourspace.c:694,:701):
  // entry 1 sees the caller's source gamma, entries 2..255 see LINEAR (or SRGB then LINEAR for MONITOR).
  // `src` below is that evolving state; keeping it is what makes the LUT bytes identical.
  int src = gamma_from;
  lut[0] = 0;
  for (int i = 1; i < 256; ++i) {
    float lin_v, enc_v;
    lin_v = enc_v = (float)i / 255.;
    if (file_gamma != 1.0) enc_v = powf(lin_v, file_gamma);
    if (src == LIVES_GAMMA_MONITOR) { enc_v = powf(lin_v, screen_gamma); src = WEED_GAMMA_SRGB; }
    if (src != WEED_GAMMA_LINEAR && !(src == WEED_GAMMA_SRGB && gamma_to == LIVES_GAMMA_MONITOR)) {
      const Transfer t = transfer_for(src);
      lin_v = (lin_v < t.thresh) ? lin_v / t.lin : powf((lin_v + t.offs) / (1. + t.offs), t.pf);
      src = WEED_GAMMA_LINEAR;
    }
    if (gamma_to != WEED_GAMMA_LINEAR) {
      const Transfer t = transfer_for(gamma_to == LIVES_GAMMA_MONITOR ? WEED_GAMMA_SRGB : gamma_to);
      enc_v = (lin_v < (t.thresh) / t.lin) ? lin_v * t.lin : powf((1. + t.offs) * lin_v, 1. / t.pf) - t.offs;
    }
    if (gamma_to == LIVES_GAMMA_MONITOR) enc_v = powf(lin_v, inv_screen);
    lut[i] = sat8((int)(enc_v * 255.));
  }
  return 1;
}

// create_gamma_lut (src/colourspace.c:738-808): the 65536-entry table the reference fuses into YUV -> RGB when it is given a
// target gamma; same evolving source-gamma state as the 8-bit builder above, CLAMP16bit (src/colourspace.h:16) at the end
int lgpu_gamma_lut16(double file_gamma, int gamma_from, int gamma_to, double screen_gamma, uint16_t *lut) {
  if (!lut) return 0;
  if (file_gamma == 1.0 &&
      (gamma_to == gamma_from || gamma_to == WEED_GAMMA_UNKNOWN || gamma_from == WEED_GAMMA_UNKNOWN)) return 0;
  const float inv_screen = (gamma_to == LIVES_GAMMA_MONITOR) ? 1. / (float)screen_gamma : 0.f;
  int src = gamma_from;
  lut[0] = 0;
  for (int i = 1; i < 65536; ++i) {
    float lin_v, enc_v;
    lin_v = enc_v = (float)i / 65536.;
    if (file_gamma != 1.0) enc_v = powf(lin_v, file_gamma);
    if (src == LIVES_GAMMA_MONITOR) { enc_v = powf(lin_v, screen_gamma); src = WEED_GAMMA_SRGB; }
    if (src != WEED_GAMMA_LINEAR && !(src == WEED_GAMMA_SRGB && gamma_to == LIVES_GAMMA_MONITOR)) {
      const Transfer t = transfer_for(src);
      lin_v = (lin_v < t.thresh) ? lin_v / t.lin : powf((lin_v + t.offs) / (1. + t.offs), t.pf);
      src = WEED_GAMMA_LINEAR;
    }
    if (gamma_to != WEED_GAMMA_LINEAR) {
      const Transfer t = transfer_for(gamma_to == LIVES_GAMMA_MONITOR ? WEED_GAMMA_SRGB : gamma_to);
      enc_v = (lin_v < (t.thresh) / t.lin) ? lin_v * t.lin : powf((1. + t.offs) * lin_v, 1. / t.pf) - t.offs;
    }
    if (gamma_to == LIVES_GAMMA_MONITOR) enc_v = powf(lin_v, inv_screen);
    lut[i] = enc_v >= 0.99999 ? 65535 : enc_v < 0.00001 ? 0 : (uint16_t)(enc_v * 65535.9999);
  }
  return 1;
}

int lgpu_calc_rowstrides(int width, int palette, int alignment, int rs[4]) {
  int nplanes = 1, bytes;
  rs[0] = rs[1] = rs[2] = rs[3] = 0;
  switch (palette) {
  case WEED_PALETTE_RGBA32: case WEED_PALETTE_BGRA32: case WEED_PALETTE_ARGB32: case WEED_PALETTE_YUVA8888:
  case WEED_PALETTE_UYVY: case WEED_PALETTE_YUYV: bytes = width * 4; break;
  case WEED_PALETTE_RGB24: case WEED_PALETTE_BGR24: case WEED_PALETTE_YUV888: bytes = width * 3; break;
  case WEED_PALETTE_YUV420P: case WEED_PALETTE_YVU420P: case WEED_PALETTE_YUV422P: case WEED_PALETTE_YUV444P:
    bytes = width; nplanes = 3; break;
  case WEED_PALETTE_YUVA4444P: bytes = width; nplanes = 4; break;
  case WEED_PALETTE_YUV411: bytes = width * 6; break;
  case WEED_PALETTE_A8: bytes = width; break;
  case WEED_PALETTE_A1: bytes = (width + 7) >> 3; break;
  default: return 0;
  }
  if (alignment != -1) {
    if (alignment < 4 || (alignment & 3)) alignment = 32;   // RA_MIN / RS_ALIGN_DEF
    if (alignment > 128) alignment = 128;                   // RA_MAX
    bytes = (bytes + alignment - 1) / alignment * alignment;
  }
  rs[0] = bytes;
  switch (palette) {
  case WEED_PALETTE_YUV420P: case WEED_PALETTE_YVU420P: case WEED_PALETTE_YUV422P: rs[1] = rs[2] = bytes >> 1; break;
  case WEED_PALETTE_YUV444P: rs[1] = rs[2] = bytes; break;
  case WEED_PALETTE_YUVA4444P: rs[1] = rs[2] = rs[3] = bytes; break;
  default: break;
  }
  return nplanes;
}

// Spec "lgpu-polyphase-v1" (DESIGN.md).  One filter row per output sample:
//   ratio = srcn / dstn, scale = max(1, ratio), support = radius(kernel) * scale, ntaps = ceil(2 * support)
//   centre = (i + 0.5) * ratio - 0.5, first tap = floor(centre - support) + 1
//   weight_j = k((first + j - centre) / scale), quantised to Q14 so that each row sums to exactly 16384
//   (rounding residue goes to the largest tap).  Taps outside the image are resolved by edge replication
//   at fetch time.
int lgpu_make_filter(int srcn, int dstn, int kernel, int *ntaps_out, int32_t *pos, int16_t *coef, int maxtaps) {
  if (srcn < 1 || dstn < 1 || kernel < 0 || kernel > 2) return LGPU_E_BADARG;
  const double ratio = (double)srcn / (double)dstn, scale = ratio > 1. ? ratio : 1.;
  const double radius = kernel == 0 ? 1. : kernel == 1 ? 2. : 3.;
  const double support = radius * scale;
  const int ntaps = (int)ceil(2. * support);
  if (ntaps > maxtaps) return LGPU_E_UNSUPPORTED;
  std::vector<double> w(ntaps);
  std::vector<int> q(ntaps);
  *ntaps_out = ntaps;
  for (int i = 0; i < dstn; i++) {
    const double centre = ((double)i + 0.5) * ratio - 0.5;
    const int first = (int)floor(centre - support) + 1;
    double total = 0.;
    for (int j = 0; j < ntaps; j++) total += (w[j] = tap_weight(kernel, ((double)(first + j) - centre) / scale));
    int acc = 0, peak = 0;
    for (int j = 0; j < ntaps; j++) {
      q[j] = (int)floor(w[j] / total * 16384. + 0.5);
      acc += q[j];
      if (q[j] > q[peak]) peak = j;
    }
    q[peak] += 16384 - acc;
    pos[i] = first;
    for (int j = 0; j < ntaps; j++) coef[(size_t)i * ntaps + j] = (int16_t)q[j];
  }
  return LGPU_OK;
}

}  // extern "C"

/* the per-byte-position tables of the script effects (lives-plugins/weed-plugins/scripts/): kind 0 negate (negate.script <process>: colour
   bytes ^ 0xFF, alpha copied), 1 posterise (posterise.script: levmask = 128 + 128 >> 1 + ..., bytes 0..2 masked, byte 3 of a 4-byte pixel
   copied -- whatever the palette), 2 ccorrect (ccorrect.script: make_table(val): (int)(val * i + .5) capped at 255; r / g / b tables at the
   palette's colour positions, alpha copied).  p0..p2: posterise levels in p0; ccorrect red / green / blue factors.  luts_out[psize][256]. */
extern "C" int lgpu_fx_luts(int kind, int palette, double p0, double p1, double p2, uint8_t *luts_out) {
  if (!luts_out || palette < 1 || palette > 5) return 0;
  const int psize = palette <= 2 ? 3 : 4;
  const int alpha = palette == 5 ? 0 : psize == 4 ? 3 : -1;
  uint8_t id[256], t[3][256];
  for (int i = 0; i < 256; i++) id[i] = (uint8_t)i;
  if (kind == 0) {
    for (int c = 0; c < psize; c++)
      for (int i = 0; i < 256; i++) luts_out[c * 256 + i] = (c == alpha) ? id[i] : (uint8_t)(0xFF ^ i);
    return psize;
  }
  if (kind == 1) {
    if (palette == 5) return 0;                                  /* ALL_RGBX_PALETTES: no ARGB32 */
    const int levels = (int)p0;
    unsigned char levmask = 128;
    for (int i = 1; i < levels; i++) levmask += 128 >> i;
    for (int c = 0; c < psize; c++)
      for (int i = 0; i < 256; i++) luts_out[c * 256 + i] = (c == 3) ? id[i] : (uint8_t)(i & levmask);
    return psize;
  }
  if (kind == 2) {
    const double val[3] = {p0, p1, p2};
    for (int k = 0; k < 3; k++)
      for (int i = 0; i < 256; i++) {
        const int ival = (int)(val[k] * i + .5);
        t[k][i] = ival > 255 ? (uint8_t)255 : (uint8_t)ival;
      }
    const bool bgr = (palette == 2 || palette == 4);
    const int offs = palette == 5 ? 1 : 0;
    for (int c = 0; c < psize; c++) {
      const uint8_t *src = id;
      if (c != alpha) { const int k = c - offs; src = t[bgr ? 2 - k : k]; }
      memcpy(luts_out + c * 256, src, 256);
    }
    return psize;
  }
  return 0;
}

/* the dissolve mask (multi_transitions.c:41-69): width * height floats drawn from xorshift64 (libweed/weed-plugin-utils.c:666) chained on the
   instance's random seed, each (double)x / 0xFFFFFFFF / 0xFFFFFFFF rounded to float.  A serial chain: built once per instance on the host. */
extern "C" int lgpu_dissolve_mask(uint64_t seed, int width, int height, float *mask_out) {
  if (!mask_out || width < 1 || height < 1) return 0;
  static const double divd = (double)(0xFFFFFFFF);
  uint64_t x = seed;
  for (size_t i = 0; i < (size_t)width * height; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const double val = (double)x / divd;
    mask_out[i] = (float)(val / divd * 1.);
  }
  return 1;
}

/* init_unal (src/colourspace.c:1141-1160), the four tables alpha_premult uses on CLAMPED YUVA layers, as the bytes its loops store:
   out = unalcy, alcy, unalcuv, alcuv, each [256 alpha][256 value].  Float arithmetic as in the reference (built without -ffast-math). */
extern "C" int lgpu_premult_yuv_tables(uint8_t *unalcy, uint8_t *alcy, uint8_t *unalcuv, uint8_t *alcuv) {
  if (!unalcy || !alcy || !unalcuv || !alcuv) return 0;
  auto c255f = [](double a) -> int { return a >= 254.5 ? 255 : a < -0.5 ? 0 : (uint8_t)(a + .5); };
  for (int i = 0; i < 256; i++) {
    const float alpha = (float)255. / (float)i;
    for (int j = 0; j < 256; j++) {
      unalcuv[i * 256 + j] = (uint8_t)c255f((float)(j - 16.) * alpha + 16.);
      alcuv[i * 256 + j] = (uint8_t)c255f((float)(j - 128.) * alpha + 128.);
      unalcy[i * 256 + j] = (uint8_t)((int)((float)j / alpha + .5) > (235. - 16.) ? (int)235. : (int)((float)(j - 16.) / alpha + 16. + .5));
      alcy[i * 256 + j] = (uint8_t)((int)((float)j / alpha + .5) > (240. - 16.) ? (int)240. : (int)((float)(j - 16.) / alpha + 16. + .5));
    }
  }
  return 1;
}
