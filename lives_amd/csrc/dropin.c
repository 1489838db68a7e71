/* dropin.c -- liblivesgpu_dropin.so: the layer-op seam under the REFERENCE's own names (src/colourspace.h:377-423), one forwarding
 * function per name into liblivesgpu.so.  A LiVES build that drops its CPU bodies of these functions links this instead
 * (INTEGRATION.md); tests/test_dropin.py links a small C host against the names.  Kept out of liblivesgpu.so so that a process
 * which still carries the CPU bodies has no duplicate symbols. */
#include "../../include/lives_gpu_layer.h"

typedef lives_gpu_layer_t weed_layer_t;
typedef lives_gpu_boolean boolean;
typedef int LiVESInterpType;
#define EXPORT __attribute__((visibility("default")))

EXPORT boolean convert_layer_palette(weed_layer_t *l, int outpl, int op_clamping) { return lives_gpu_convert_layer_palette(l, outpl, op_clamping); }
EXPORT boolean convert_layer_palette_with_sampling(weed_layer_t *l, int outpl, int out_sampling) { return lives_gpu_convert_layer_palette_with_sampling(l, outpl, out_sampling); }
EXPORT boolean convert_layer_palette_full(weed_layer_t *l, int outpl, int oclamping, int osampling, int osubspace, int tgt_gamma) {
  return lives_gpu_convert_layer_palette_full(l, outpl, oclamping, osampling, osubspace, tgt_gamma);
}
EXPORT boolean gamma_convert_layer(int gamma_type, weed_layer_t *l) { return lives_gpu_gamma_convert_layer(gamma_type, l); }
EXPORT boolean gamma_convert_layer_variant(double file_gamma, int tgt_gamma, weed_layer_t *l) { return lives_gpu_gamma_convert_layer_variant(file_gamma, tgt_gamma, l); }
EXPORT boolean gamma_convert_sub_layer(int gamma_type, double fileg, weed_layer_t *l, int x, int y, int width, int height, boolean may_thread) {
  return lives_gpu_gamma_convert_sub_layer(gamma_type, fileg, l, x, y, width, height, may_thread);
}
EXPORT void alpha_premult(weed_layer_t *l, int direction) { lives_gpu_alpha_premult(l, direction); }
EXPORT boolean resize_layer_full(weed_layer_t *l, int width, int height, LiVESInterpType interp, int opal_hint, int oclamp_hint, int osamp_hint,
                                 int osubs_hint, int tgt_gamma) {
  return lives_gpu_resize_layer_full(l, width, height, interp, opal_hint, oclamp_hint, osamp_hint, osubs_hint, tgt_gamma);
}
EXPORT boolean resize_layer(weed_layer_t *l, int width, int height, LiVESInterpType interp, int opal_hint, int oclamp_hint) {
  return lives_gpu_resize_layer(l, width, height, interp, opal_hint, oclamp_hint);
}
EXPORT boolean letterbox_layer(weed_layer_t *l, int nwidth, int nheight, int width, int height, LiVESInterpType interp, int tpal, int tclamp) {
  return lives_gpu_letterbox_layer(l, nwidth, nheight, width, height, interp, tpal, tclamp);
}
EXPORT boolean unletterbox_layer(weed_layer_t *l, int opwidth, int opheight, int top, int bottom, int left, int right) {
  return lives_gpu_unletterbox_layer(l, opwidth, opheight, top, bottom, left, right);
}
EXPORT boolean compact_rowstrides(weed_layer_t *l) { return lives_gpu_compact_rowstrides(l); }
EXPORT boolean create_empty_pixel_data(weed_layer_t *l, boolean black_fill, boolean may_contig) { return lives_gpu_create_empty_pixel_data(l, black_fill, may_contig); }
EXPORT boolean weed_layer_clear_pixel_data(weed_layer_t *l) { return lives_gpu_weed_layer_clear_pixel_data(l); }
EXPORT int *calc_rowstrides(int width, int pal, weed_layer_t *l, int *nplanes) { return lives_gpu_calc_rowstrides(width, pal, l, nplanes); }
/* the planner's queries of the same header range (src/colourspace.h:400-407) */
EXPORT int get_resizable(int *ppalette, int *pxpal, int *oclamp_hint, int *opal, int *pxopal, boolean upscale) {
  return lives_gpu_get_resizable(ppalette, pxpal, oclamp_hint, opal, pxopal, upscale);
}
EXPORT int get_tgt_gamma(int ipal, int opal) { return lives_gpu_get_tgt_gamma(ipal, opal); }
EXPORT boolean can_inline_gamma(int inpl, int opal) { return lives_gpu_can_inline_gamma(inpl, opal); }
EXPORT boolean pconv_can_inplace(int inpl, int outpl) { return lives_gpu_pconv_can_inplace(inpl, outpl); }
