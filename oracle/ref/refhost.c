/* refhost.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A minimal Weed *host* used to run the REFERENCE's own effect plugins
 * (simple_blend.c, multi_blends.c, mirrors.c, generated colorkey.c), built by
 * oracle/ref/build_ref.sh from the sources where they lie under /root/reference,
 * so that golden vectors can be generated (oracle/ref/gen_golden.py) and so the
 * repo's own drop-in plugin (livesgpu_fx.so) can be driven through exactly the
 * same host code path in tests.
 *
 * It follows the host side of the weed bootstrap the way LiVES does it:
 *   load_weed_plugin()           src/effects-weed.c:4468-4568  (dlopen, dlsym("weed_setup"), setup_fn(weed_bootstrap))
 *   weed_instance_from_filter()  (instance + channel + parameter plants)
 *   run_process_func()           src/effects-weed.c:2519-2548
 *   process_func_threaded()      src/effects-weed.c:1563-1758  (row-slice protocol: offset / height[2] / pre-offset pixel_data)
 *
 * Everything here is original code written for this repo; it links against the
 * reference's libweed (oracle/_ref/libweedall.so).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dlfcn.h>

#include <weed/weed-host.h>
#include <weed/weed.h>
#include <weed/weed-palettes.h>
#include <weed/weed-effects.h>
#include <weed/weed-utils.h>
#include <weed/weed-host-utils.h>

static int inited = 0;

int refhost_init(void) {
  if (!inited) {
    weed_error_t err = libweed_init(WEED_ABI_VERSION, 0);
    if (err != WEED_SUCCESS) return (int)err;
    inited = 1;
  }
  return 0;
}

/* returns the plugin_info plant, or NULL */
void *refhost_load(const char *so_path) {
  void *h;
  weed_setup_f setup;
  if (refhost_init()) return NULL;
  h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "refhost: dlopen %s: %s\n", so_path, dlerror()); return NULL; }
  setup = (weed_setup_f)dlsym(h, "weed_setup");
  if (!setup) { fprintf(stderr, "refhost: no weed_setup in %s\n", so_path); return NULL; }
  return (void *)(*setup)(weed_bootstrap);
}

int refhost_num_filters(void *pinfo) {
  int n = 0;
  weed_plant_t **f = weed_get_plantptr_array_counted((weed_plant_t *)pinfo, WEED_LEAF_FILTERS, &n);
  if (f) free(f);
  return n;
}

/* copies name of filter idx into buf; returns flags, or -1 */
int refhost_filter_info(void *pinfo, int idx, char *buf, int buflen, int *palettes, int maxpal,
                        int *n_in, int *n_out, int *n_params) {
  int n = 0, flags, np = 0, i;
  weed_plant_t **f = weed_get_plantptr_array_counted((weed_plant_t *)pinfo, WEED_LEAF_FILTERS, &n);
  weed_plant_t *filt;
  char *name;
  int *pl;
  if (!f || idx >= n) return -1;
  filt = f[idx];
  free(f);
  name = weed_get_string_value(filt, WEED_LEAF_NAME, NULL);
  snprintf(buf, buflen, "%s", name ? name : "");
  if (name) free(name);
  flags = weed_get_int_value(filt, WEED_LEAF_FLAGS, NULL);
  pl = weed_get_int_array_counted(filt, WEED_LEAF_PALETTE_LIST, &np);
  for (i = 0; i < maxpal; i++) palettes[i] = (i < np) ? pl[i] : 0;
  if (pl) free(pl);
  if (n_in) { weed_plant_t **t = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, n_in); if (t) free(t); }
  if (n_out) { weed_plant_t **t = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, n_out); if (t) free(t); }
  if (n_params) { weed_plant_t **t = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, n_params); if (t) free(t); }
  return flags;
}

typedef struct {
  int kind;        /* 0 = int, 1 = double, 2 = int array, 3 = boolean */
  int n;           /* elements for kind 2 */
  int ival[4];
  double dval;
} refhost_param_t;

static weed_plant_t *find_filter(weed_plant_t *pinfo, const char *fname) {
  int n = 0, i;
  weed_plant_t **f = weed_get_plantptr_array_counted(pinfo, WEED_LEAF_FILTERS, &n);
  weed_plant_t *ret = NULL;
  for (i = 0; i < n && !ret; i++) {
    char *name = weed_get_string_value(f[i], WEED_LEAF_NAME, NULL);
    if (name && !strcmp(name, fname)) ret = f[i];
    if (name) free(name);
  }
  if (f) free(f);
  return ret;
}

/* YUV_clamping leaf for the channels made from now on (-1 = none): "YUVdelay" reads it (RGBdelay.c:236) */
static int g_yuv_clamping = -1;
void refhost_set_yuv_clamping(int clamping) { g_yuv_clamping = clamping; }

/* "random_seed" leaf for the instances made from now on (0 = none): dissolve seeds its mask from it (multi_transitions.c:56) */
static int64_t g_random_seed = 0;
void refhost_set_random_seed(int64_t seed) { g_random_seed = seed; }

static weed_plant_t *mk_channel(weed_plant_t *tmpl, int pal, int w, int h, int stride, void *pd) {
  weed_plant_t *c = weed_plant_new(WEED_PLANT_CHANNEL);
  weed_set_plantptr_value(c, WEED_LEAF_TEMPLATE, tmpl);
  weed_set_int_value(c, WEED_LEAF_WIDTH, w);
  weed_set_int_value(c, WEED_LEAF_HEIGHT, h);
  weed_set_int_value(c, WEED_LEAF_CURRENT_PALETTE, pal);
  weed_set_int_value(c, WEED_LEAF_ROWSTRIDES, stride);
  weed_set_voidptr_value(c, WEED_LEAF_PIXEL_DATA, pd);
  if (g_yuv_clamping >= 0) weed_set_int_value(c, WEED_LEAF_YUV_CLAMPING, g_yuv_clamping);
  return c;
}

/* Run one filter once.
 *   nslices <= 1 : a single process_func call on the whole frame (no "offset" leaf)
 *   nslices  > 1 : emulate process_func_threaded (src/effects-weed.c:1563-1758): per slice a copy of
 *                  the out channel with offset, height = {slice_h, real_h} and pre-offset pixel_data;
 *                  state_updated handling for stateful filters (:1700-1725).
 * dst may alias src[0] (inplace). Returns the weed_error_t of the (last failing) process call.
 */
int refhost_run(void *pinfo_v, const char *fname, int pal, int w, int h,
                int nin, uint8_t **src, const int *istrides, uint8_t *dst, int ostride,
                int nparams, const refhost_param_t *params, int nslices) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst, *inch[4], *outch, **ictm, **octm, **iptm, *inpar[256];
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  int nict = 0, noct = 0, nipt = 0, i, ret = WEED_SUCCESS, flags;

  if (!filt) { fprintf(stderr, "refhost: filter '%s' not found\n", fname); return -100; }
  if (nin > 4 || nparams > 256) return -101;

  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  iptm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, &nipt);
  if (nict < nin || noct < 1 || nipt < nparams) {
    fprintf(stderr, "refhost: '%s' wants %d in / %d out / %d params\n", fname, nict, noct, nipt);
    return -102;
  }
  flags = weed_get_int_value(filt, WEED_LEAF_FLAGS, NULL);

  inst = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
  weed_set_plantptr_value(inst, WEED_LEAF_FILTER_CLASS, filt);
  if (g_random_seed) weed_set_int64_value(inst, WEED_LEAF_RANDOM_SEED, g_random_seed);
  for (i = 0; i < nin; i++) inch[i] = mk_channel(ictm[i], pal, w, h, istrides[i], src[i]);
  outch = mk_channel(octm[0], pal, w, h, ostride, dst);
  weed_set_plantptr_array(inst, WEED_LEAF_IN_CHANNELS, nin, inch);
  weed_set_plantptr_value(inst, WEED_LEAF_OUT_CHANNELS, outch);
  for (i = 0; i < nparams; i++) {
    inpar[i] = weed_plant_new(WEED_PLANT_PARAMETER);
    weed_set_plantptr_value(inpar[i], WEED_LEAF_TEMPLATE, iptm[i]);
    switch (params[i].kind) {
    case 0: weed_set_int_value(inpar[i], WEED_LEAF_VALUE, params[i].ival[0]); break;
    case 1: weed_set_double_value(inpar[i], WEED_LEAF_VALUE, params[i].dval); break;
    case 2: weed_set_int_array(inpar[i], WEED_LEAF_VALUE, params[i].n, (int32_t *)params[i].ival); break;
    case 3: weed_set_boolean_value(inpar[i], WEED_LEAF_VALUE, params[i].ival[0]); break;
    }
  }
  if (nparams) weed_set_plantptr_array(inst, WEED_LEAF_IN_PARAMETERS, nparams, inpar);

  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);

  if (init_func) {
    ret = (*init_func)(inst);
    if (ret != WEED_SUCCESS) goto done;
  }

  if (nslices <= 1) {
    ret = (*process_func)(inst, (weed_timecode_t)0);
  } else {
    /* slice height rule: CEIL(h / n, 4) rows (src/effects-weed.c:1633-1640 follows the colourspace rule) */
    int dh = ((h + nslices - 1) / nslices + 3) & ~3, off;
    int stateful = (flags & WEED_FILTER_HINT_STATEFUL) ? 1 : 0;
    if (stateful) weed_set_boolean_value(inst, WEED_LEAF_STATE_UPDATED, WEED_FALSE);
    for (off = 0; off < h; off += dh) {
      int sh = (off + dh > h) ? h - off : dh;
      int hh[2];
      weed_plant_t *xoutch = mk_channel(octm[0], pal, w, h, ostride, dst + (size_t)off * ostride);
      weed_plant_t *xinst = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
      void *internal;
      int r;
      hh[0] = sh; hh[1] = h;
      weed_set_int_array(xoutch, WEED_LEAF_HEIGHT, 2, hh);
      weed_set_int_value(xoutch, WEED_LEAF_OFFSET, off);
      weed_set_plantptr_value(xinst, WEED_LEAF_FILTER_CLASS, filt);
      weed_set_plantptr_array(xinst, WEED_LEAF_IN_CHANNELS, nin, inch);
      weed_set_plantptr_value(xinst, WEED_LEAF_OUT_CHANNELS, xoutch);
      if (nparams) weed_set_plantptr_array(xinst, WEED_LEAF_IN_PARAMETERS, nparams, inpar);
      if (weed_plant_has_leaf(inst, "plugin_internal")) {
        internal = weed_get_voidptr_value(inst, "plugin_internal", NULL);
        weed_set_voidptr_value(xinst, "plugin_internal", internal);
      }
      if (stateful)
        weed_set_boolean_value(xinst, WEED_LEAF_STATE_UPDATED, off == 0 ? WEED_FALSE : WEED_TRUE);
      r = (*process_func)(xinst, (weed_timecode_t)0);
      if (r != WEED_SUCCESS) ret = r;
      weed_plant_free(xinst);
      weed_plant_free(xoutch);
    }
  }

  if (deinit_func) (*deinit_func)(inst);

done:
  for (i = 0; i < nparams; i++) weed_plant_free(inpar[i]);
  for (i = 0; i < nin; i++) weed_plant_free(inch[i]);
  weed_plant_free(outch);
  weed_plant_free(inst);
  if (ictm) free(ictm);
  if (octm) free(octm);
  if (iptm) free(iptm);
  return ret;
}

/* One process_func call on a PLANAR frame (softlight.c wants YUV planes): a single in and out channel with
 * nplanes pixel_data / rowstrides entries and the YUV_clamping leaf (WEED_YUV_CLAMPING_CLAMPED 0 / UNCLAMPED 1). */
int refhost_run_planar(void *pinfo_v, const char *fname, int pal, int w, int h, int nplanes,
                       uint8_t **src, const int *istrides, uint8_t **dst, const int *ostrides, int clamping) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst, *inch, *outch, **ictm, **octm;
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  int nict = 0, noct = 0, ret = WEED_SUCCESS;
  if (!filt) { fprintf(stderr, "refhost: filter '%s' not found\n", fname); return -100; }
  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  if (nict < 1 || noct < 1) return -102;
  inst = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
  weed_set_plantptr_value(inst, WEED_LEAF_FILTER_CLASS, filt);
  if (g_random_seed) weed_set_int64_value(inst, WEED_LEAF_RANDOM_SEED, g_random_seed);
  inch = mk_channel(ictm[0], pal, w, h, istrides[0], src[0]);
  outch = mk_channel(octm[0], pal, w, h, ostrides[0], dst[0]);
  weed_set_voidptr_array(inch, WEED_LEAF_PIXEL_DATA, nplanes, (void **)src);
  weed_set_int_array(inch, WEED_LEAF_ROWSTRIDES, nplanes, (int32_t *)istrides);
  weed_set_voidptr_array(outch, WEED_LEAF_PIXEL_DATA, nplanes, (void **)dst);
  weed_set_int_array(outch, WEED_LEAF_ROWSTRIDES, nplanes, (int32_t *)ostrides);
  weed_set_int_value(inch, WEED_LEAF_YUV_CLAMPING, clamping);
  weed_set_int_value(outch, WEED_LEAF_YUV_CLAMPING, clamping);
  weed_set_plantptr_value(inst, WEED_LEAF_IN_CHANNELS, inch);
  weed_set_plantptr_value(inst, WEED_LEAF_OUT_CHANNELS, outch);
  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);
  if (init_func) ret = (*init_func)(inst);
  if (ret == WEED_SUCCESS) {
    ret = (*process_func)(inst, (weed_timecode_t)0);
    if (deinit_func) (*deinit_func)(inst);
  }
  weed_plant_free(inch);
  weed_plant_free(outch);
  weed_plant_free(inst);
  if (ictm) free(ictm);
  if (octm) free(octm);
  return ret;
}

/* The compositor class: nin in channels of their own sizes, all instances of in channel template 0 (max_repeats 0), some possibly disabled by the host;
 * parameters 0-4 are double arrays with one value per channel, 5 = int[3] colour, 6 = switch.  One process call on the whole frame. */
int refhost_run_compositor(void *pinfo_v, const char *fname, int pal, int nin, uint8_t **src, const int *ws, const int *hs, const int *istrides, const int *disabled,
                           uint8_t *dst, int ow, int oh, int ostride, const double *offsx, const double *offsy, const double *scalex, const double *scaley,
                           const double *alpha, const int *bgcol, int revz) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst, *inch[64], *outch, **ictm, **octm, **iptm, *inpar[7];
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  const double *arrs[5] = {offsx, offsy, scalex, scaley, alpha};
  int nict = 0, noct = 0, nipt = 0, i, ret = WEED_SUCCESS;
  if (!filt) return -100;
  if (nin > 64) return -101;
  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  iptm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, &nipt);
  if (nict < 1 || noct < 1 || nipt < 7) return -102;
  inst = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
  weed_set_plantptr_value(inst, WEED_LEAF_FILTER_CLASS, filt);
  for (i = 0; i < nin; i++) {
    inch[i] = mk_channel(ictm[0], pal, ws[i], hs[i], istrides[i], src[i]);
    if (disabled[i]) weed_set_boolean_value(inch[i], WEED_LEAF_DISABLED, WEED_TRUE);
  }
  outch = mk_channel(octm[0], pal, ow, oh, ostride, dst);
  weed_set_plantptr_array(inst, WEED_LEAF_IN_CHANNELS, nin, inch);
  weed_set_plantptr_value(inst, WEED_LEAF_OUT_CHANNELS, outch);
  for (i = 0; i < 7; i++) {
    inpar[i] = weed_plant_new(WEED_PLANT_PARAMETER);
    weed_set_plantptr_value(inpar[i], WEED_LEAF_TEMPLATE, iptm[i]);
    if (i < 5) weed_set_double_array(inpar[i], WEED_LEAF_VALUE, nin, (double *)arrs[i]);
    else if (i == 5) weed_set_int_array(inpar[i], WEED_LEAF_VALUE, 3, (int32_t *)bgcol);
    else weed_set_boolean_value(inpar[i], WEED_LEAF_VALUE, revz ? WEED_TRUE : WEED_FALSE);
  }
  weed_set_plantptr_array(inst, WEED_LEAF_IN_PARAMETERS, 7, inpar);
  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);
  if (init_func) ret = (*init_func)(inst);
  if (ret == WEED_SUCCESS) {
    ret = (*process_func)(inst, (weed_timecode_t)0);
    if (deinit_func) (*deinit_func)(inst);
  }
  for (i = 0; i < 7; i++) weed_plant_free(inpar[i]);
  for (i = 0; i < nin; i++) weed_plant_free(inch[i]);
  weed_plant_free(outch);
  weed_plant_free(inst);
  if (ictm) free(ictm);
  if (octm) free(octm);
  if (iptm) free(iptm);
  return ret;
}

/* A stateful filter over a SEQUENCE of frames on ONE instance (init once, process_func per frame, deinit once):
 * blurzoom.c keeps its background / feedback buffers in "plugin_internal".  src[i] / dst[i]: frame i. */
int refhost_run_seq(void *pinfo_v, const char *fname, int pal, int w, int h, int nframes,
                    uint8_t **src, int istride, uint8_t **dst, int ostride, int nparams, const refhost_param_t *params) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst, *inch, *outch, **ictm, **octm, **iptm, *inpar[256];
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  int nict = 0, noct = 0, nipt = 0, i, ret = WEED_SUCCESS;
  if (!filt) { fprintf(stderr, "refhost: filter '%s' not found\n", fname); return -100; }
  if (nparams > 256) return -101;
  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  iptm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, &nipt);
  if (nict < 1 || noct < 1 || nipt < nparams) return -102;
  inst = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
  weed_set_plantptr_value(inst, WEED_LEAF_FILTER_CLASS, filt);
  if (g_random_seed) weed_set_int64_value(inst, WEED_LEAF_RANDOM_SEED, g_random_seed);
  inch = mk_channel(ictm[0], pal, w, h, istride, src[0]);
  outch = mk_channel(octm[0], pal, w, h, ostride, dst[0]);
  weed_set_plantptr_value(inst, WEED_LEAF_IN_CHANNELS, inch);
  weed_set_plantptr_value(inst, WEED_LEAF_OUT_CHANNELS, outch);
  for (i = 0; i < nparams; i++) {
    inpar[i] = weed_plant_new(WEED_PLANT_PARAMETER);
    weed_set_plantptr_value(inpar[i], WEED_LEAF_TEMPLATE, iptm[i]);
    switch (params[i].kind) {
    case 0: weed_set_int_value(inpar[i], WEED_LEAF_VALUE, params[i].ival[0]); break;
    case 1: weed_set_double_value(inpar[i], WEED_LEAF_VALUE, params[i].dval); break;
    case 2: weed_set_int_array(inpar[i], WEED_LEAF_VALUE, params[i].n, (int32_t *)params[i].ival); break;
    case 3: weed_set_boolean_value(inpar[i], WEED_LEAF_VALUE, params[i].ival[0]); break;
    }
  }
  if (nparams) weed_set_plantptr_array(inst, WEED_LEAF_IN_PARAMETERS, nparams, inpar);
  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);
  if (init_func) ret = (*init_func)(inst);
  if (ret == WEED_SUCCESS) {
    for (i = 0; i < nframes && ret == WEED_SUCCESS; i++) {
      weed_set_voidptr_value(inch, WEED_LEAF_PIXEL_DATA, src[i]);
      weed_set_voidptr_value(outch, WEED_LEAF_PIXEL_DATA, dst[i]);
      ret = (*process_func)(inst, (weed_timecode_t)i);
    }
    if (deinit_func) (*deinit_func)(inst);
  }
  for (i = 0; i < nparams; i++) weed_plant_free(inpar[i]);
  weed_plant_free(inch);
  weed_plant_free(outch);
  weed_plant_free(inst);
  if (ictm) free(ictm);
  if (octm) free(octm);
  if (iptm) free(iptm);
  return ret;
}


/* n instances of one two-input filter class with one double parameter each (the transitions), as a plan step holds them for n tracks:
 * init each, then either process_func per instance (batch_hook NULL) or ONE call of the plugin's batch hook
 *   weed_error_t hook(weed_plant_t **instances, int n, weed_timecode_t tc)
 * (this repo's livesgpu_fx.so exports livesgpu_fx_process_batch; the reference's plugins have none), then deinit each.
 * src1[i] / src2[i] / dst[i]: the planes of instance i (dst[i] may alias src1[i]). */
typedef weed_error_t (*refhost_batch_f)(weed_plant_t **, int, weed_timecode_t);
int refhost_run_batch(void *pinfo_v, const char *fname, int pal, int w, int h, int n, uint8_t **src1, int istride1, uint8_t **src2, int istride2,
                      uint8_t **dst, int ostride, const double *amounts, int int_param, void *batch_hook) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst[64], *in1[64], *in2[64], *outc[64], *par[64], **ictm, **octm, **iptm;
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  int nict = 0, noct = 0, nipt = 0, i, ninit = 0, ret = WEED_SUCCESS;
  if (!filt) { fprintf(stderr, "refhost: filter '%s' not found\n", fname); return -100; }
  if (n < 1 || n > 64) return -101;
  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  iptm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, &nipt);
  if (nict < 2 || noct < 1 || nipt < 1) return -102;
  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);
  for (i = 0; i < n; i++) {
    weed_plant_t *ch[2];
    inst[i] = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
    weed_set_plantptr_value(inst[i], WEED_LEAF_FILTER_CLASS, filt);
    ch[0] = in1[i] = mk_channel(ictm[0], pal, w, h, istride1, src1[i]);
    ch[1] = in2[i] = mk_channel(ictm[1], pal, w, h, istride2, src2[i]);
    outc[i] = mk_channel(octm[0], pal, w, h, ostride, dst[i]);
    weed_set_plantptr_array(inst[i], WEED_LEAF_IN_CHANNELS, 2, ch);
    weed_set_plantptr_value(inst[i], WEED_LEAF_OUT_CHANNELS, outc[i]);
    par[i] = weed_plant_new(WEED_PLANT_PARAMETER);
    weed_set_plantptr_value(par[i], WEED_LEAF_TEMPLATE, iptm[0]);
    if (int_param) weed_set_int_value(par[i], WEED_LEAF_VALUE, (int)amounts[i]);
    else weed_set_double_value(par[i], WEED_LEAF_VALUE, amounts[i]);
    weed_set_plantptr_value(inst[i], WEED_LEAF_IN_PARAMETERS, par[i]);
  }
  for (ninit = 0; ninit < n && ret == WEED_SUCCESS; ninit++)
    if (init_func) { ret = (*init_func)(inst[ninit]); if (ret != WEED_SUCCESS) break; }
  if (ret == WEED_SUCCESS) {
    if (batch_hook) ret = (*(refhost_batch_f)batch_hook)(inst, n, (weed_timecode_t)0);
    else for (i = 0; i < n; i++) { int r = (*process_func)(inst[i], (weed_timecode_t)0); if (r != WEED_SUCCESS) ret = r; }
  }
  for (i = 0; i < ninit; i++) if (deinit_func) (*deinit_func)(inst[i]);
  for (i = 0; i < n; i++) { weed_plant_free(par[i]); weed_plant_free(in1[i]); weed_plant_free(in2[i]); weed_plant_free(outc[i]); weed_plant_free(inst[i]); }
  if (ictm) free(ictm);
  if (octm) free(octm);
  if (iptm) free(iptm);
  return ret;
}

/* n instances of a one-input PLANAR class (softlight.c): planes laid out frame after frame in src / dst ([i * nplanes + p]), one rowstride set for all.
 * batch_hook as in refhost_run_batch. */
int refhost_run_planar_batch(void *pinfo_v, const char *fname, int pal, int w, int h, int nplanes, int n,
                             uint8_t **src, const int *istrides, uint8_t **dst, const int *ostrides, int clamping, void *batch_hook) {
  weed_plant_t *pinfo = (weed_plant_t *)pinfo_v;
  weed_plant_t *filt = find_filter(pinfo, fname);
  weed_plant_t *inst[64], *inch[64], *outch[64], **ictm, **octm;
  weed_init_f init_func;
  weed_process_f process_func;
  weed_deinit_f deinit_func;
  int nict = 0, noct = 0, i, ninit, ret = WEED_SUCCESS;
  if (!filt) { fprintf(stderr, "refhost: filter '%s' not found\n", fname); return -100; }
  if (n < 1 || n > 64) return -101;
  ictm = weed_get_plantptr_array_counted(filt, WEED_LEAF_IN_CHANNEL_TEMPLATES, &nict);
  octm = weed_get_plantptr_array_counted(filt, WEED_LEAF_OUT_CHANNEL_TEMPLATES, &noct);
  if (nict < 1 || noct < 1) return -102;
  init_func = (weed_init_f)weed_get_funcptr_value(filt, WEED_LEAF_INIT_FUNC, NULL);
  process_func = (weed_process_f)weed_get_funcptr_value(filt, WEED_LEAF_PROCESS_FUNC, NULL);
  deinit_func = (weed_deinit_f)weed_get_funcptr_value(filt, WEED_LEAF_DEINIT_FUNC, NULL);
  for (i = 0; i < n; i++) {
    inst[i] = weed_plant_new(WEED_PLANT_FILTER_INSTANCE);
    weed_set_plantptr_value(inst[i], WEED_LEAF_FILTER_CLASS, filt);
    inch[i] = mk_channel(ictm[0], pal, w, h, istrides[0], src[i * nplanes]);
    outch[i] = mk_channel(octm[0], pal, w, h, ostrides[0], dst[i * nplanes]);
    weed_set_voidptr_array(inch[i], WEED_LEAF_PIXEL_DATA, nplanes, (void **)(src + i * nplanes));
    weed_set_int_array(inch[i], WEED_LEAF_ROWSTRIDES, nplanes, (int32_t *)istrides);
    weed_set_voidptr_array(outch[i], WEED_LEAF_PIXEL_DATA, nplanes, (void **)(dst + i * nplanes));
    weed_set_int_array(outch[i], WEED_LEAF_ROWSTRIDES, nplanes, (int32_t *)ostrides);
    weed_set_int_value(inch[i], WEED_LEAF_YUV_CLAMPING, clamping);
    weed_set_int_value(outch[i], WEED_LEAF_YUV_CLAMPING, clamping);
    weed_set_plantptr_value(inst[i], WEED_LEAF_IN_CHANNELS, inch[i]);
    weed_set_plantptr_value(inst[i], WEED_LEAF_OUT_CHANNELS, outch[i]);
  }
  for (ninit = 0; ninit < n; ninit++)
    if (init_func) { ret = (*init_func)(inst[ninit]); if (ret != WEED_SUCCESS) break; }
  if (ret == WEED_SUCCESS) {
    if (batch_hook) ret = (*(refhost_batch_f)batch_hook)(inst, n, (weed_timecode_t)0);
    else for (i = 0; i < n; i++) { int r = (*process_func)(inst[i], (weed_timecode_t)0); if (r != WEED_SUCCESS) ret = r; }
  }
  for (i = 0; i < ninit; i++) if (deinit_func) (*deinit_func)(inst[i]);
  for (i = 0; i < n; i++) { weed_plant_free(inch[i]); weed_plant_free(outch[i]); weed_plant_free(inst[i]); }
  if (ictm) free(ictm);
  if (octm) free(octm);
  return ret;
}
