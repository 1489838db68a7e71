/* A tiny C "host" that calls the layer-op seam by the REFERENCE's names and prototypes (src/colourspace.h:377-423), linked against
 * lives_amd/liblivesgpu_dropin.so -- what a LiVES build without the CPU bodies of these functions does.
 * Built and driven by tests/test_dropin.py. */
#include <stddef.h>

typedef void weed_layer_t;
typedef int boolean;
typedef int LiVESInterpType;

boolean convert_layer_palette(weed_layer_t *, int outpl, int op_clamping);
boolean convert_layer_palette_with_sampling(weed_layer_t *, int outpl, int out_sampling);
boolean convert_layer_palette_full(weed_layer_t *, int outpl, int oclamping, int osampling, int osubspace, int tgt_gamma);
boolean gamma_convert_layer(int gamma_type, weed_layer_t *);
boolean gamma_convert_layer_variant(double file_gamma, int tgt_gamma, weed_layer_t *);
boolean gamma_convert_sub_layer(int gamma_type, double fileg, weed_layer_t *, int x, int y, int width, int height, boolean may_thread);
void alpha_premult(weed_layer_t *, int direction);
boolean resize_layer_full(weed_layer_t *layer, int width, int height, LiVESInterpType interp, int opal_hint, int oclamp_hint, int osamp_hint,
                          int osubs_hint, int tgt_gamma);
boolean resize_layer(weed_layer_t *, int width, int height, LiVESInterpType interp, int opal_hint, int oclamp_hint);
boolean letterbox_layer(weed_layer_t *, int nwidth, int nheight, int width, int height, LiVESInterpType interp, int tpal, int tclamp);
boolean unletterbox_layer(weed_layer_t *layer, int opwidth, int opheight, int top, int bottom, int left, int right);
boolean compact_rowstrides(weed_layer_t *);
boolean create_empty_pixel_data(weed_layer_t *, boolean black_fill, boolean may_contig);
boolean weed_layer_clear_pixel_data(weed_layer_t *);
int *calc_rowstrides(int width, int pal, weed_layer_t *, int *nplanes);

/* the CONVERT chain of one plan step the way src/nodemodel.c:1065-1253 strings these calls together */
int host_convert_chain(weed_layer_t *layer, int outpl, int gamma, int w, int h, int nw, int nh) {
  if (!convert_layer_palette(layer, outpl, 0)) return 1;
  if (!gamma_convert_layer(gamma, layer)) return 2;
  if (!resize_layer_full(layer, w, h, 3, outpl, 0, 0, 0, 0)) return 3;
  if (!letterbox_layer(layer, nw, nh, w, h, 3, outpl, 0)) return 4;
  return 0;
}

/* every name is referenced once so that the link fails if the shim lacks one */
const void *host_all_names[] = {
  (const void *)convert_layer_palette, (const void *)convert_layer_palette_with_sampling, (const void *)convert_layer_palette_full,
  (const void *)gamma_convert_layer, (const void *)gamma_convert_layer_variant, (const void *)gamma_convert_sub_layer, (const void *)alpha_premult,
  (const void *)resize_layer_full, (const void *)resize_layer, (const void *)letterbox_layer, (const void *)unletterbox_layer,
  (const void *)compact_rowstrides, (const void *)create_empty_pixel_data, (const void *)weed_layer_clear_pixel_data, (const void *)calc_rowstrides,
};
int host_name_count(void) { return (int)(sizeof host_all_names / sizeof host_all_names[0]); }
int host_rowstride(int width, int pal) {
  int n = 0;
  int *rs = calc_rowstrides(width, pal, NULL, &n);
  return (rs && n > 0) ? rs[0] : -1;
}
