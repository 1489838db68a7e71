/* fx_plugin.c -- livesgpu_fx.so: the drop-in weed effect plugin (plugin seam of SURVEY 8b.2).
 *
 * Exports weed_setup(weed_bootstrap_f) (libweed/weed-effects.h:174-186) and registers filter classes with the
 * SAME names, channel / parameter templates and palette lists as the reference plugins they replace:
 *   "chroma blend", "luma overlay", "luma underlay", "negative luma overlay", "averaged luma overlay"
 *                                                lives-plugins/weed-plugins/simple_blend.c:213-291
 *   "blend_multiply" .. "blend_burn"             lives-plugins/weed-plugins/multi_blends.c:205-300
 *   "colorkey"                                   lives-plugins/weed-plugins/scripts/colorkey.script
 *   "mirrorx", "mirrory", "mirrorxy"             lives-plugins/weed-plugins/mirrors.c:125-160
 *   "compositor"                                 lives-plugins/weed-plugins/gdk/compositor.c:127-351 (scaler: gdk_pixbuf_scale_simple, pinned -- pixbuf.hip)
 * process_func uploads the host channels, runs the liblivesgpu.so kernel and downloads the result (the host
 * owns pixel_data -- host memory; device residency across a chain is what the layer seam is for).
 * WEED_FILTER_HINT_MAY_THREAD is deliberately NOT advertised, so the host makes one call per frame
 * (can_thread(), src/effects-weed.c:1810); if a host slices anyway the offset / height[2] protocol of
 * process_func_threaded (:1563-1758) is honoured.
 *
 * Built against include/lives_gpu_weed_abi.h only (ids + signatures of the public ABI); the core accessors
 * are the function pointers the host hands over in weed_setup(), exactly as libweed/weed-plugin-utils.c:164-249.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "../../include/lives_gpu_weed_abi.h"
#include "../../include/lives_gpu.h"
#include "../../include/lives_gpu_layer.h"
#include "../../include/livesgpu_fx.h"

/* ---- host functions obtained at bootstrap ---- */
static weed_leaf_get_f w_get;
static weed_leaf_set_f w_set;
static weed_plant_new_f w_new;
static weed_leaf_num_elements_f w_nelems;
static weed_malloc_f w_malloc;
static weed_free_f w_free;

/* ---- tiny leaf helpers ---- */
static int g_int(weed_plant_t *p, const char *k, int idx, int dflt) { int32_t v = dflt; if (w_get(p, k, (weed_size_t)idx, &v) != WEED_SUCCESS) return dflt; return v; }
static double g_dbl(weed_plant_t *p, const char *k, double dflt) { double v = dflt; if (w_get(p, k, 0, &v) != WEED_SUCCESS) return dflt; return v; }
static void *g_ptr(weed_plant_t *p, const char *k, int idx) { void *v = NULL; if (w_get(p, k, (weed_size_t)idx, &v) != WEED_SUCCESS) return NULL; return v; }
static int has(weed_plant_t *p, const char *k) { return w_nelems(p, k) > 0; }
static void s_int(weed_plant_t *p, const char *k, int v) { int32_t x = v; w_set(p, k, WEED_SEED_INT, 1, &x); }
static void s_bool(weed_plant_t *p, const char *k, int v) { int32_t x = v; w_set(p, k, WEED_SEED_BOOLEAN, 1, &x); }
static void s_str(weed_plant_t *p, const char *k, const char *v) { w_set(p, k, WEED_SEED_STRING, 1, &v); }
static void s_dbl(weed_plant_t *p, const char *k, double v) { w_set(p, k, WEED_SEED_DOUBLE, 1, &v); }

static int psize_of(int pal) {
  switch (pal) {
  case WEED_PALETTE_RGB24: case WEED_PALETTE_BGR24: case WEED_PALETTE_YUV888: return 3;
  case WEED_PALETTE_RGBA32: case WEED_PALETTE_BGRA32: case WEED_PALETTE_ARGB32: case WEED_PALETTE_YUVA8888:
  case WEED_PALETTE_UYVY: case WEED_PALETTE_YUYV: return 4;
  default: return 0;
  }
}

/* ---- per-instance device buffers ("plugin_internal", like simple_blend.c:36-45) ---- */
typedef struct { void *last_stream; int has_stream; void *d[3]; size_t cap[3]; lgpu_blurzoom *bz; int bz_w, bz_h, bz_pal; int slide_dir; lgpu_rgbdelay *rd; float *mask_d; int mask_w, mask_h; int64_t mask_seed; } fxdata_t;

/* The effects are enqueued on the CALLING THREAD's stream (lives_gpu_thread_stream: the one the layer seam uses for this host thread; NULL -- the null stream --
   when the host has not bound the layer seam), so effects of different tracks, applied by different pool threads, overlap on the device.  An instance keeps
   device state across calls (staging buffers, the blurzoom / RGBdelay rings): when a call arrives on another thread than the previous one, this thread's stream
   first waits for everything the previous one enqueued. */
static __thread void *fx_stream_;
#define FXS fx_stream_
static fxdata_t *fx_data(weed_plant_t *inst);
static void fx_enter(fxdata_t *fx) {
  FXS = lives_gpu_thread_stream();
  if (!fx) return;
  if (fx->has_stream && fx->last_stream != FXS) lives_gpu_stream_follow(fx->last_stream);
  fx->last_stream = FXS; fx->has_stream = 1;
}
static fxdata_t *fx_data(weed_plant_t *inst) {
  fxdata_t *fx = (fxdata_t *)g_ptr(inst, "plugin_internal", 0);
  if (!fx) {
    fx = (fxdata_t *)w_malloc(sizeof(fxdata_t));
    if (!fx) return NULL;
    memset(fx, 0, sizeof *fx);
    void *v = fx;
    w_set(inst, "plugin_internal", WEED_SEED_VOIDPTR, 1, &v);
  }
  return fx;
}
static void *fx_buf(fxdata_t *fx, int i, size_t bytes) {
  if (fx->cap[i] < bytes) {
    if (fx->d[i]) lgpu_free(fx->d[i]);
    fx->d[i] = NULL; fx->cap[i] = 0;
    if (lgpu_malloc(&fx->d[i], bytes) != LGPU_OK) return NULL;
    fx->cap[i] = bytes;
  }
  return fx->d[i];
}
static weed_error_t fx_init(weed_plant_t *inst) {
  if (lgpu_init(0) != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }   /* no GPU: refuse loudly, no CPU path */
  return fx_data(inst) ? WEED_SUCCESS : WEED_ERROR_MEMORY_ALLOCATION;
}
static weed_error_t fx_deinit(weed_plant_t *inst) {
  fxdata_t *fx = (fxdata_t *)g_ptr(inst, "plugin_internal", 0);
  if (fx) {
    for (int i = 0; i < 3; i++) if (fx->d[i]) lgpu_free(fx->d[i]);
    if (fx->bz) lgpu_blurzoom_destroy(fx->bz);
    if (fx->rd) lgpu_rgbdelay_destroy(fx->rd);
    if (fx->mask_d) lgpu_free(fx->mask_d);
    w_free(fx);
    void *v = NULL;
    w_set(inst, "plugin_internal", WEED_SEED_VOIDPTR, 1, &v);
  }
  return WEED_SUCCESS;
}

/* ---- the common frame plumbing: channels -> device -> kernel -> out channel ---- */
typedef struct {
  uint8_t *src[2], *dst;        /* host */
  uint8_t *dsrc[2], *ddst;      /* device, already offset to the slice */
  int irow[2], orow, width, height, pal, psize, nin, inplace;
} fxframe_t;

typedef int (*fx_kernel_f)(const fxframe_t *f, weed_plant_t *inst, int kind);

static int k_simple(const fxframe_t *f, weed_plant_t *inst, int kind);
static int param_int(weed_plant_t *inst, int idx, int dflt);
static weed_error_t fx_run(weed_plant_t *inst, int nin, int kind, fx_kernel_f kernel, int whole_frame) {
  fxdata_t *fx = fx_data(inst);
  weed_plant_t *ochan = (weed_plant_t *)g_ptr(inst, WEED_LEAF_OUT_CHANNELS, 0);
  fxframe_t f;
  int offset = 0, real_h, slice_h, i;
  if (!fx || !ochan) return WEED_ERROR_FILTER_INVALID;
  fx_enter(fx);
  memset(&f, 0, sizeof f);
  f.nin = nin;
  f.pal = g_int(ochan, WEED_LEAF_CURRENT_PALETTE, 0, 0);
  f.psize = psize_of(f.pal);
  f.width = g_int(ochan, WEED_LEAF_WIDTH, 0, 0);
  slice_h = g_int(ochan, WEED_LEAF_HEIGHT, 0, 0);
  real_h = (w_nelems(ochan, WEED_LEAF_HEIGHT) > 1) ? g_int(ochan, WEED_LEAF_HEIGHT, 1, slice_h) : slice_h;
  f.orow = g_int(ochan, WEED_LEAF_ROWSTRIDES, 0, 0);
  f.dst = (uint8_t *)g_ptr(ochan, WEED_LEAF_PIXEL_DATA, 0);           /* pre-offset by the host when slicing */
  if (has(ochan, WEED_LEAF_OFFSET)) offset = g_int(ochan, WEED_LEAF_OFFSET, 0, 0);
  if (!f.psize || f.width <= 0 || slice_h <= 0 || !f.dst) return WEED_ERROR_FILTER_INVALID;
  if (whole_frame && (offset != 0 || slice_h != real_h)) return WEED_ERROR_FILTER_INVALID;   /* mirrors need the full frame */
  f.height = slice_h;
  for (i = 0; i < nin; i++) {
    weed_plant_t *ic = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_CHANNELS, i);
    if (!ic) return WEED_ERROR_FILTER_INVALID;
    f.irow[i] = g_int(ic, WEED_LEAF_ROWSTRIDES, 0, 0);
    f.src[i] = (uint8_t *)g_ptr(ic, WEED_LEAF_PIXEL_DATA, 0);
    if (!f.src[i]) return WEED_ERROR_FILTER_INVALID;
    f.src[i] += (size_t)offset * f.irow[i];                           /* inputs are NOT pre-offset (simple_blend.c:87-91) */
  }
  f.inplace = (f.src[0] == f.dst);
  /* "chroma blend" in place on a pinned layer whose plane is still a pending program of the layer seam (convert -> resize -> ... recorded, not yet run): the
     blend joins the program and the whole chain runs as one launch when the host flushes its tracks or needs the pixels (include/lives_gpu_layer.h) */
  if (kernel == k_simple && kind == 0 && nin == 2 && f.inplace && offset == 0 && slice_h == real_h &&
      lives_gpu_deferred_blend_chroma(f.dst, f.orow, f.width, f.height, f.pal, f.src[1], f.irow[1], param_int(inst, 0, 128))) return WEED_SUCCESS;
  /* stage to the device: in rows as they are (rowstride preserved so alignment-dependent paths match).  A channel whose pixel_data is the
     plane of a pinned layer (lives_gpu_layer_pin, include/lives_gpu_layer.h) is used where it lives in HBM: no upload, and for the out
     channel no download -- the device copy is the plane until lives_gpu_layer_sync() */
  {
    /* the ARGB chroma blend reads one byte past the last pixel of each row of layer 2 (reference quirk B1) */
    const size_t ob = (size_t)f.orow * f.height;
    int uploaded = 0;
    const void *rel[3] = {NULL, NULL, NULL};         /* resident planes to release once the effect is enqueued: [0] the out channel (write), [1..] in channels (read) */
    uint8_t *res_dst = (uint8_t *)lives_gpu_resident_acquire(f.dst - (size_t)offset * f.orow, (size_t)f.orow * real_h, 1);
    if (res_dst) rel[0] = f.dst - (size_t)offset * f.orow;
    if (res_dst) f.ddst = res_dst + (size_t)offset * f.orow;
    else f.ddst = (uint8_t *)fx_buf(fx, 2, ob + 16);
    if (!f.ddst) return WEED_ERROR_MEMORY_ALLOCATION;
    for (i = 0; i < nin; i++) {
      const size_t ib = (size_t)f.irow[i] * f.height;
      uint8_t *res_src;
      if (i == 0 && f.inplace) { f.dsrc[0] = f.ddst; if (!res_dst && lgpu_upload(f.ddst, f.dst, ob, FXS)) return WEED_ERROR_PLUGIN_INVALID; continue; }
      res_src = (uint8_t *)lives_gpu_resident_acquire(f.src[i] - (size_t)offset * f.irow[i], (size_t)f.irow[i] * real_h, 0);
      if (res_src) { rel[1 + i] = f.src[i] - (size_t)offset * f.irow[i]; f.dsrc[i] = res_src + (size_t)offset * f.irow[i]; continue; }
      f.dsrc[i] = (uint8_t *)fx_buf(fx, i, ib + 16);
      if (!f.dsrc[i]) return WEED_ERROR_MEMORY_ALLOCATION;
      if (lgpu_upload(f.dsrc[i], f.src[i], ib, FXS)) return WEED_ERROR_PLUGIN_INVALID;
      uploaded = 1;
    }
    if (!f.inplace && !res_dst && lgpu_upload(f.ddst, f.dst, ob, FXS)) return WEED_ERROR_PLUGIN_INVALID;   /* bytes the effect leaves alone */
    {
      const int krc = kernel(&f, inst, kind);
      lives_gpu_resident_release(rel[0], 1); lives_gpu_resident_release(rel[1], 0); lives_gpu_resident_release(rel[2], 0);
      if (krc != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
    }
    /* an out channel on a pinned layer: the effect is enqueued and the call returns (stream order carries it to whatever reads the plane next; the host
       bytes are stale by the pinning contract).  Otherwise the result is brought home and has to be complete on return. */
    if (!res_dst && (lgpu_download(f.dst, f.ddst, ob, FXS) || lgpu_sync(FXS))) return WEED_ERROR_PLUGIN_INVALID;
    /* resident out channel, but an input came from host memory: from a page-locked frame that is an asynchronous DMA out of the host's plane, which the host may
       reuse as soon as this call returns -- wait for the stream (uploads and, on this mixed path, the effect behind them) */
    if (res_dst && uploaded && lgpu_sync(FXS)) return WEED_ERROR_PLUGIN_INVALID;
  }
  return WEED_SUCCESS;
}

static int param_int(weed_plant_t *inst, int idx, int dflt) {
  weed_plant_t *p = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, idx);
  return p ? g_int(p, WEED_LEAF_VALUE, 0, dflt) : dflt;
}

/* ---- kernels per filter family ---- */
static int k_simple(const fxframe_t *f, weed_plant_t *inst, int kind) {
  const int v = param_int(inst, 0, 128);
  const int af = (f->pal == WEED_PALETTE_ARGB32);
  if (kind == 0)
    return lgpu_blend_chroma(f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->psize, af, v, FXS);
  if (af) return LGPU_E_UNSUPPORTED;
  return lgpu_blend_luma(kind, f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->psize,
                         (f->pal == WEED_PALETTE_BGR24 || f->pal == WEED_PALETTE_BGRA32) ? 1 : 0, v, FXS);
}
static int k_multi(const fxframe_t *f, weed_plant_t *inst, int kind) {
  return lgpu_blend_multi(kind, f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height,
                          f->pal == WEED_PALETTE_BGR24, param_int(inst, 0, 128), FXS);
}
static int k_ckey(const fxframe_t *f, weed_plant_t *inst, int kind) {
  weed_plant_t *pd = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 0), *po = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 1),
               *pc = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 2);
  (void)kind;
  if (!pd || !po || !pc) return LGPU_E_BADARG;
  return lgpu_colorkey(f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->pal == WEED_PALETTE_BGR24,
                       g_dbl(pd, WEED_LEAF_VALUE, .2), g_dbl(po, WEED_LEAF_VALUE, 1.), g_int(pc, WEED_LEAF_VALUE, 0, 0),
                       g_int(pc, WEED_LEAF_VALUE, 1, 0), g_int(pc, WEED_LEAF_VALUE, 2, 255), FXS);
}
static int k_transition(const fxframe_t *f, weed_plant_t *inst, int kind) {
  weed_plant_t *pa = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 0);
  return lgpu_transition(kind, f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->psize,
                         pa ? g_dbl(pa, WEED_LEAF_VALUE, 0.) : 0., FXS);
}
/* "slide over" (slide_over.c:40-51, :83-86): direction from the radio parameters 1..5; "random" is drawn once per instance */
static int param_bool(weed_plant_t *inst, int idx, int dflt) {
  weed_plant_t *p = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, idx);
  int32_t v = dflt;
  if (p) w_get(p, WEED_LEAF_VALUE, 0, &v);
  return v == WEED_TRUE;
}
static int k_slide(const fxframe_t *f, weed_plant_t *inst, int kind) {
  fxdata_t *fx = fx_data(inst);
  int dirn;
  (void)kind;
  if (!fx) return LGPU_E_NOMEM;
  if (param_bool(inst, 1, WEED_TRUE)) {
    if (!fx->slide_dir) {
      const uint64_t seed = (uint64_t)time(NULL) ^ (uint64_t)(uintptr_t)fx;
      fx->slide_dir = (int)(((seed * 6364136223846793005ull + 1442695040888963407ull) >> 24) & 3) + 1;
    }
    dirn = fx->slide_dir;
  } else {
    fx->slide_dir = 0;
    dirn = param_bool(inst, 2, 0) ? 1 : param_bool(inst, 3, 0) ? 2 : param_bool(inst, 4, 0) ? 3 : 4;
  }
  return lgpu_slide_over(f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->psize,
                         param_int(inst, 0, 0), dirn, param_bool(inst, 6, WEED_TRUE), param_bool(inst, 7, WEED_FALSE), FXS);
}
static int k_deint(const fxframe_t *f, weed_plant_t *inst, int kind) {
  (void)inst; (void)kind;
  return lgpu_deinterlace(f->dsrc[0], f->irow[0], f->ddst, f->orow, f->width, f->height, f->pal, FXS);
}
/* "RGBdelay" / "YUVdelay" (RGBdelay.c:135-416): stateful, the frame ring lives in the device handle kept in plugin_internal */
static double param_dbl(weed_plant_t *inst, int idx, double dflt) {
  weed_plant_t *p = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, idx);
  return p ? g_dbl(p, WEED_LEAF_VALUE, dflt) : dflt;
}
static int k_rgbdelay(const fxframe_t *f, weed_plant_t *inst, int kind) {
  fxdata_t *fx = fx_data(inst);
  weed_plant_t *ic = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_CHANNELS, 0);
  int on[153], j, clamped = 0;
  double strength[51];
  (void)kind;
  if (!fx) return LGPU_E_NOMEM;
  if (!fx->rd && lgpu_rgbdelay_create(&fx->rd) != LGPU_OK) return LGPU_E_NOMEM;
  for (j = 0; j < 51; j++) {
    on[3 * j] = param_bool(inst, 4 * j + 1, 0); on[3 * j + 1] = param_bool(inst, 4 * j + 2, 0); on[3 * j + 2] = param_bool(inst, 4 * j + 3, 0);
    strength[j] = param_dbl(inst, 4 * j + 4, 1.);
  }
  if (f->pal == WEED_PALETTE_YUV888 && ic) clamped = g_int(ic, WEED_LEAF_YUV_CLAMPING, 0, WEED_YUV_CLAMPING_CLAMPED) == WEED_YUV_CLAMPING_CLAMPED;
  return lgpu_rgbdelay_process(fx->rd, f->dsrc[0], f->irow[0], f->ddst, f->orow, f->width, f->height, f->pal, clamped, param_int(inst, 0, 20), on, strength, FXS);
}
/* negate / posterise / ccorrect (scripts of those names): per-byte-position tables built on the host, one gather launch */
static int k_scriptfx(const fxframe_t *f, weed_plant_t *inst, int kind) {
  uint8_t luts[4 * 256];
  double p0 = 0., p1 = 0., p2 = 0.;
  if (kind == 1) p0 = (double)param_int(inst, 0, 1);
  else if (kind == 2) { p0 = param_dbl(inst, 0, 1.); p1 = param_dbl(inst, 1, 1.); p2 = param_dbl(inst, 2, 1.); }
  if (lgpu_fx_luts(kind, f->pal, p0, p1, p2, luts) != f->psize) return LGPU_E_UNSUPPORTED;
  return lgpu_byte_luts(f->dsrc[0], f->irow[0], f->ddst, f->orow, f->width, f->height, f->psize, luts, FXS);
}
/* "triple split" (layout_blends.c:24-113) */
static int k_tsplit(const fxframe_t *f, weed_plant_t *inst, int kind) {
  weed_plant_t *pc = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 6);
  int bc[3] = {0, 0, 0};
  (void)kind;
  if (pc) { bc[0] = g_int(pc, WEED_LEAF_VALUE, 0, 0); bc[1] = g_int(pc, WEED_LEAF_VALUE, 1, 0); bc[2] = g_int(pc, WEED_LEAF_VALUE, 2, 0); }
  return lgpu_triple_split(f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->pal == WEED_PALETTE_BGR24,
                           param_dbl(inst, 0, 0.666667), param_bool(inst, 1, WEED_TRUE), param_dbl(inst, 3, 0.333333), param_bool(inst, 4, WEED_FALSE),
                           param_dbl(inst, 5, 0.), bc, FXS);
}
/* "dissolve" (multi_transitions.c:41-69, :208-212): the mask is drawn once per instance (and geometry) from the instance's "random_seed" leaf,
   as dissolve_init does, and kept in device memory */
static int k_dissolve(const fxframe_t *f, weed_plant_t *inst, int kind) {
  fxdata_t *fx = fx_data(inst);
  weed_plant_t *pa = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 0);
  int64_t seed = 0;
  (void)kind;
  if (!fx) return LGPU_E_NOMEM;
  if (has(inst, WEED_LEAF_RANDOM_SEED)) w_get(inst, WEED_LEAF_RANDOM_SEED, 0, &seed);
  if (!fx->mask_d || fx->mask_w != f->width || fx->mask_h != f->height || fx->mask_seed != seed) {
    const size_t n = (size_t)f->width * f->height;
    float *m = (float *)w_malloc(n * sizeof(float));
    int ok;
    if (!m) return LGPU_E_NOMEM;
    if (fx->mask_d) { lgpu_free(fx->mask_d); fx->mask_d = NULL; }
    ok = lgpu_dissolve_mask((uint64_t)seed, f->width, f->height, m) == 1 && lgpu_malloc((void **)&fx->mask_d, n * sizeof(float)) == LGPU_OK &&
         lgpu_upload(fx->mask_d, m, n * sizeof(float), FXS) == LGPU_OK && lgpu_sync(FXS) == LGPU_OK;
    w_free(m);
    if (!ok) return LGPU_E_NOMEM;
    fx->mask_w = f->width; fx->mask_h = f->height; fx->mask_seed = seed;
  }
  return lgpu_dissolve(f->dsrc[0], f->irow[0], f->dsrc[1], f->irow[1], f->ddst, f->orow, f->width, f->height, f->psize, fx->mask_d,
                       pa ? g_dbl(pa, WEED_LEAF_VALUE, 0.) : 0., FXS);
}
/* "rand replace" (multi_transitions.c:96-112, :213-220): per frame ONE draw of a uniform number decides whether the whole frame is the second input
   (draw < amount) or the first; in place and "first" means nothing to do.  The reference draws from libweed's time-seeded global generator
   (fastrnd_dbl, weed-plugin-utils.c): no sequence to reproduce, so this is an xorshift64* seeded from the clock the same way; the frame copy is the device work */
static uint64_t g_rr_state;
static double rr_draw(void) {
  if (!g_rr_state) { g_rr_state = ((uint64_t)time(NULL) << 20) ^ (uint64_t)clock() ^ 0x9E3779B97F4A7C15ull; }
  g_rr_state ^= g_rr_state >> 12; g_rr_state ^= g_rr_state << 25; g_rr_state ^= g_rr_state >> 27;
  return (double)((g_rr_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);       /* [0, 1) */
}
static int k_rreplace(const fxframe_t *f, weed_plant_t *inst, int kind) {
  weed_plant_t *pa = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, 0);
  const double bfd = pa ? g_dbl(pa, WEED_LEAF_VALUE, 0.) : 0.;
  const int cpy0 = rr_draw() < bfd;
  (void)kind;
  if (f->inplace && !cpy0) return LGPU_OK;
  return lgpu_copy_rows(f->ddst, f->orow, cpy0 ? f->dsrc[1] : f->dsrc[0], cpy0 ? f->irow[1] : f->irow[0], f->width * f->psize, f->height, FXS);
}
static int k_mirror(const fxframe_t *f, weed_plant_t *inst, int kind) {
  (void)inst;
  return lgpu_mirror(kind, f->dsrc[0], f->irow[0], f->ddst, f->orow, f->width, f->height, f->psize, FXS);
}

static int k_edge(const fxframe_t *f, weed_plant_t *inst, int kind) {
  (void)kind;
  return lgpu_edge(f->dsrc[0], f->irow[0], f->ddst, f->orow, f->width, f->height, f->pal, param_int(inst, 0, 0), FXS);
}

/* "softlight" (softlight.c:62-151): planar YUV in, planar YUV out, not in place; whole frame only */
static weed_error_t p_softlight(weed_plant_t *inst, weed_timecode_t tc) {
  fxdata_t *fx = fx_data(inst);
  weed_plant_t *ic = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_CHANNELS, 0), *oc = (weed_plant_t *)g_ptr(inst, WEED_LEAF_OUT_CHANNELS, 0);
  const uint8_t *dsrc[4] = {0, 0, 0, 0};
  uint8_t *ddst[4] = {0, 0, 0, 0}, *rdst[4] = {0, 0, 0, 0}, *hdst[4], *base;
  int rsrc[4] = {0, 0, 0, 0};
  int irow[4], orow[4], ph[4], i, nplanes, pal, w, h, clamping;
  size_t ioff[4], ooff[4], itot = 0, otot = 0;
  (void)tc;
  if (!fx || !ic || !oc) return WEED_ERROR_FILTER_INVALID;
  fx_enter(fx);
  pal = g_int(ic, WEED_LEAF_CURRENT_PALETTE, 0, 0);
  w = g_int(ic, WEED_LEAF_WIDTH, 0, 0); h = g_int(ic, WEED_LEAF_HEIGHT, 0, 0);
  clamping = g_int(ic, WEED_LEAF_YUV_CLAMPING, 0, WEED_YUV_CLAMPING_CLAMPED);
  if (pal != WEED_PALETTE_YUV444P && pal != WEED_PALETTE_YUVA4444P && pal != WEED_PALETTE_YUV422P && pal != WEED_PALETTE_YUV420P &&
      pal != WEED_PALETTE_YVU420P) return WEED_ERROR_FILTER_INVALID;
  if (has(oc, WEED_LEAF_OFFSET) && g_int(oc, WEED_LEAF_OFFSET, 0, 0) != 0) return WEED_ERROR_FILTER_INVALID;
  nplanes = pal == WEED_PALETTE_YUVA4444P ? 4 : 3;
  for (i = 0; i < nplanes; i++) {
    irow[i] = g_int(ic, WEED_LEAF_ROWSTRIDES, i, 0); orow[i] = g_int(oc, WEED_LEAF_ROWSTRIDES, i, 0);
    ph[i] = (i == 0 || i == 3) ? h : ((pal == WEED_PALETTE_YUV420P || pal == WEED_PALETTE_YVU420P) ? h >> 1 : h);
    ioff[i] = itot; itot += ((size_t)irow[i] * ph[i] + 15) & ~(size_t)15;
    ooff[i] = otot; otot += ((size_t)orow[i] * ph[i] + 15) & ~(size_t)15;
    hdst[i] = (uint8_t *)g_ptr(oc, WEED_LEAF_PIXEL_DATA, i);
    if (!g_ptr(ic, WEED_LEAF_PIXEL_DATA, i) || !hdst[i] || irow[i] <= 0 || orow[i] <= 0) return WEED_ERROR_FILTER_INVALID;
  }
  base = (uint8_t *)fx_buf(fx, 0, itot);
  if (!base) return WEED_ERROR_MEMORY_ALLOCATION;
  for (i = 0; i < nplanes; i++) {                       /* planes of a pinned layer are used where they live in HBM (see fx_run) */
    const void *hp = g_ptr(ic, WEED_LEAF_PIXEL_DATA, i);
    const uint8_t *res = (const uint8_t *)lives_gpu_resident_acquire(hp, (size_t)irow[i] * ph[i], 0);
    if (res) { dsrc[i] = res; rsrc[i] = 1; continue; }
    dsrc[i] = base + ioff[i];
    if (lgpu_upload(base + ioff[i], hp, (size_t)irow[i] * ph[i], FXS)) return WEED_ERROR_PLUGIN_INVALID;
  }
  base = (uint8_t *)fx_buf(fx, 2, otot);
  if (!base) return WEED_ERROR_MEMORY_ALLOCATION;
  for (i = 0; i < nplanes; i++) {
    rdst[i] = (uint8_t *)lives_gpu_resident_acquire(hdst[i], (size_t)orow[i] * ph[i], 1);
    if (rdst[i]) { ddst[i] = rdst[i]; continue; }
    ddst[i] = base + ooff[i];
    if (lgpu_upload(ddst[i], hdst[i], (size_t)orow[i] * ph[i], FXS)) return WEED_ERROR_PLUGIN_INVALID;   /* row padding stays as it was */
  }
  {
    const int krc = lgpu_softlight(dsrc, irow, ddst, orow, w, h, pal, clamping == WEED_YUV_CLAMPING_UNCLAMPED, FXS);
    for (i = 0; i < nplanes; i++) {                       /* the planes of pinned layers: a read / a write has been enqueued on this thread's stream */
      if (rsrc[i]) lives_gpu_resident_release(g_ptr(ic, WEED_LEAF_PIXEL_DATA, i), 0);
      if (rdst[i]) lives_gpu_resident_release(hdst[i], 1);
    }
    if (krc != LGPU_OK) {
      fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error());
      return WEED_ERROR_PLUGIN_INVALID;
    }
  }
  {
    int home = 0;
    for (i = 0; i < nplanes; i++)
      if (!rdst[i]) { home = 1; if (lgpu_download(hdst[i], ddst[i], (size_t)orow[i] * ph[i], FXS)) return WEED_ERROR_PLUGIN_INVALID; }
    if (home && lgpu_sync(FXS)) return WEED_ERROR_PLUGIN_INVALID;           /* planes of a pinned layer: enqueued, not waited for (see fx_run) */
  }
  return WEED_SUCCESS;
}

/* "blurzoom" (blurzoom.c:345-421): stateful -- the device handle lives in plugin_internal and is rebuilt when the frame
   geometry changes (the reference's in channel is REINIT_ON_SIZE_CHANGE) */
static int k_blurzoom(const fxframe_t *f, weed_plant_t *inst, int kind) {
  fxdata_t *fx = fx_data(inst);
  (void)kind;
  if (!fx) return LGPU_E_NOMEM;
  if (!fx->bz || fx->bz_w != f->width || fx->bz_h != f->height || fx->bz_pal != f->pal) {
    if (fx->bz) lgpu_blurzoom_destroy(fx->bz);
    if (fx->rd) lgpu_rgbdelay_destroy(fx->rd);
    if (fx->mask_d) lgpu_free(fx->mask_d);
    fx->bz = NULL;
    if (lgpu_blurzoom_create(f->width, f->height, f->pal, &fx->bz) != LGPU_OK) return LGPU_E_BADARG;
    fx->bz_w = f->width; fx->bz_h = f->height; fx->bz_pal = f->pal;
  }
  return lgpu_blurzoom_process(fx->bz, f->dsrc[0], f->irow[0], f->ddst, f->orow, param_int(inst, 0, 0), param_int(inst, 1, 0), FXS);
}

#define PROC(name, nin, kind, kern, whole) static weed_error_t name(weed_plant_t *inst, weed_timecode_t tc) { (void)tc; return fx_run(inst, nin, kind, kern, whole); }
PROC(p_chroma, 2, 0, k_simple, 0) PROC(p_lumo, 2, 1, k_simple, 0) PROC(p_lumu, 2, 2, k_simple, 0) PROC(p_nlumo, 2, 3, k_simple, 0) PROC(p_avlumo, 2, 4, k_simple, 0)
PROC(p_mpy, 2, 0, k_multi, 0) PROC(p_screen, 2, 1, k_multi, 0) PROC(p_darken, 2, 2, k_multi, 0) PROC(p_lighten, 2, 3, k_multi, 0)
PROC(p_overlay, 2, 4, k_multi, 0) PROC(p_dodge, 2, 5, k_multi, 0) PROC(p_burn, 2, 6, k_multi, 0)
PROC(p_ckey, 2, 0, k_ckey, 0)
PROC(p_mirrorx, 1, 0, k_mirror, 0) PROC(p_mirrory, 1, 1, k_mirror, 1) PROC(p_mirrorxy, 1, 2, k_mirror, 1)
PROC(p_edge, 1, 0, k_edge, 1) PROC(p_blurzoom, 1, 0, k_blurzoom, 1)
PROC(p_irisr, 2, 0, k_transition, 1) PROC(p_irisc, 2, 1, k_transition, 1) PROC(p_fourw, 2, 2, k_transition, 1)
PROC(p_slide, 2, 0, k_slide, 1) PROC(p_deint, 1, 0, k_deint, 1) PROC(p_rgbdelay, 1, 0, k_rgbdelay, 1)
PROC(p_tsplit, 2, 0, k_tsplit, 1) PROC(p_dissolve, 2, 0, k_dissolve, 1) PROC(p_rreplace, 2, 0, k_rreplace, 1)
PROC(p_negate, 1, 0, k_scriptfx, 0) PROC(p_posterise, 1, 1, k_scriptfx, 0) PROC(p_ccorrect, 1, 2, k_scriptfx, 0)

/* ---- "compositor" (lives-plugins/weed-plugins/gdk/compositor.c:127-292) -----------------------------------------------------------------------------
   any number of in channels of any size (WEED_FILTER_CHANNEL_SIZES_MAY_VARY, in channel template with max_repeats 0); per-channel x / y offset, x / y
   scale and alpha (variable-size parameters, one value per channel), background colour, z order.  Every enabled channel becomes a pixbuf (with alpha
   for 4-byte palettes, :83-90), is scaled to (((int)(owidth * scalex + 1.)) >> 1) << 1 by the same rule for the height (:218-219) with
   gdk_pixbuf_scale_simple -- GDK_INTERP_HYPER when a side grows, GDK_INTERP_BILINEAR otherwise (:262-266) -- and painted at (int)(offs * out size) by
   paint_pixel (:120-125).  Here: lgpu_pixbuf_scale (bit-exact to that library call) per channel into stream-ordered device frames, then ONE
   lgpu_composite launch.  A channel with an "inner_size" leaf is cropped first as :228-258 do.  At most LGPU_COMP_MAX_LAYERS enabled channels. */
/* ---- several instances of one filter class in ONE launch (an extension of this plugin, not of the weed API; a host finds it with dlsym) ----
   The reference applies the effects of a plan step one instance after another (src/effects-weed.c:1563-1758 per instance), and a 640x360 frame is a
   ramp-and-drain bound launch on 256 CUs: n instances of the same class on frames of one geometry go out as one lgpu_fx_batch launch (include/lives_gpu.h).
   Batched today: the three transitions of multi_transitions.c ("iris rectangle", "iris circle", "4 way split"), the five blends of simple_blend.c and the seven of
   multi_blends.c (ARGB32 frames excepted), each instance with its own amount / threshold, and "softlight" (planar YUV);
   any other class, mixed classes, mixed geometry, sliced channels or more than LGPU_FX_MAX_FRAMES instances fall back to process_func per instance, so the
   result is the same either way.  Channels on pinned layers are used where they live (no copy, no wait), the others are staged as fx_run does. */
/* the classes with a batch kernel: op of lgpu_fx_batch, its `kind`, and whether parameter 0 (the one value an instance adds to the launch) is an integer */
typedef struct { int op, kind, int_param, two_in; } batch_class_t;
static int batch_class(weed_plant_t *inst, batch_class_t *bc) {
  static const struct { weed_process_f f; batch_class_t c; } tab[] = {
    {p_irisr, {LGPU_FX_TRANSITION, 0, 0, 1}}, {p_irisc, {LGPU_FX_TRANSITION, 1, 0, 1}}, {p_fourw, {LGPU_FX_TRANSITION, 2, 0, 1}},
    {p_chroma, {LGPU_FX_BLEND_CHROMA, 0, 1, 1}},
    {p_lumo, {LGPU_FX_BLEND_LUMA, 1, 1, 1}}, {p_lumu, {LGPU_FX_BLEND_LUMA, 2, 1, 1}}, {p_nlumo, {LGPU_FX_BLEND_LUMA, 3, 1, 1}}, {p_avlumo, {LGPU_FX_BLEND_LUMA, 4, 1, 1}},
    {p_mpy, {LGPU_FX_BLEND_MULTI, 0, 1, 1}}, {p_screen, {LGPU_FX_BLEND_MULTI, 1, 1, 1}}, {p_darken, {LGPU_FX_BLEND_MULTI, 2, 1, 1}}, {p_lighten, {LGPU_FX_BLEND_MULTI, 3, 1, 1}},
    {p_overlay, {LGPU_FX_BLEND_MULTI, 4, 1, 1}}, {p_dodge, {LGPU_FX_BLEND_MULTI, 5, 1, 1}}, {p_burn, {LGPU_FX_BLEND_MULTI, 6, 1, 1}},
  };
  weed_plant_t *fc = (weed_plant_t *)g_ptr(inst, WEED_LEAF_FILTER_CLASS, 0);
  weed_process_f pf = NULL;
  if (!fc || w_get(fc, WEED_LEAF_PROCESS_FUNC, 0, &pf) != WEED_SUCCESS) return 0;
  for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++) if (tab[i].f == pf) { *bc = tab[i].c; return 1; }
  return 0;
}
static weed_error_t batch_fallback(weed_plant_t **insts, int n, weed_timecode_t tc) {
  weed_error_t ret = WEED_SUCCESS;
  for (int i = 0; i < n; i++) {
    weed_plant_t *fc = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_FILTER_CLASS, 0);
    weed_process_f pf = NULL;
    weed_error_t r;
    if (!fc || w_get(fc, WEED_LEAF_PROCESS_FUNC, 0, &pf) != WEED_SUCCESS || !pf) return WEED_ERROR_FILTER_INVALID;
    r = (*pf)(insts[i], tc);
    if (r != WEED_SUCCESS) ret = r;
  }
  return ret;
}
/* "softlight" (planar YUV, one in channel): the planes of n instances through ONE lgpu_fx_batch launch; staging per instance as p_softlight does it */
static weed_error_t batch_softlight(weed_plant_t **insts, int n, weed_timecode_t tc) {
  lgpu_fx_frame fr[LGPU_FX_MAX_FRAMES];
  lgpu_fx_params P;
  uint8_t *hdst[LGPU_FX_MAX_FRAMES][4], *ddst[LGPU_FX_MAX_FRAMES][4];
  const void *hsrc[LGPU_FX_MAX_FRAMES][4];
  int rsrc[LGPU_FX_MAX_FRAMES][4], rdst[LGPU_FX_MAX_FRAMES][4];
  int irow[4] = {0, 0, 0, 0}, orow[4] = {0, 0, 0, 0}, ph[4] = {0, 0, 0, 0}, i, k, nplanes = 0, pal = 0, w = 0, h = 0, clamping = 0, home = 0, uploaded = 0, krc;
  size_t ioff[4], ooff[4], itot = 0, otot = 0;
  weed_error_t ret = WEED_SUCCESS;
  for (i = 0; i < n; i++) {
    weed_plant_t *ic = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_CHANNELS, 0), *oc = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_OUT_CHANNELS, 0);
    if (!ic || !oc) return WEED_ERROR_FILTER_INVALID;
    if (has(oc, WEED_LEAF_OFFSET) || w_nelems(oc, WEED_LEAF_HEIGHT) > 1) return batch_fallback(insts, n, tc);
    if (i == 0) {
      pal = g_int(ic, WEED_LEAF_CURRENT_PALETTE, 0, 0); w = g_int(ic, WEED_LEAF_WIDTH, 0, 0); h = g_int(ic, WEED_LEAF_HEIGHT, 0, 0);
      clamping = g_int(ic, WEED_LEAF_YUV_CLAMPING, 0, WEED_YUV_CLAMPING_CLAMPED);
      if (pal != WEED_PALETTE_YUV444P && pal != WEED_PALETTE_YUVA4444P && pal != WEED_PALETTE_YUV422P && pal != WEED_PALETTE_YUV420P && pal != WEED_PALETTE_YVU420P)
        return batch_fallback(insts, n, tc);
      nplanes = pal == WEED_PALETTE_YUVA4444P ? 4 : 3;
      for (k = 0; k < nplanes; k++) {
        irow[k] = g_int(ic, WEED_LEAF_ROWSTRIDES, k, 0); orow[k] = g_int(oc, WEED_LEAF_ROWSTRIDES, k, 0);
        ph[k] = (k == 0 || k == 3) ? h : ((pal == WEED_PALETTE_YUV420P || pal == WEED_PALETTE_YVU420P) ? h >> 1 : h);
        ioff[k] = itot; itot += ((size_t)irow[k] * ph[k] + 15) & ~(size_t)15;
        ooff[k] = otot; otot += ((size_t)orow[k] * ph[k] + 15) & ~(size_t)15;
        if (irow[k] <= 0 || orow[k] <= 0) return WEED_ERROR_FILTER_INVALID;
      }
    } else if (pal != g_int(ic, WEED_LEAF_CURRENT_PALETTE, 0, 0) || w != g_int(ic, WEED_LEAF_WIDTH, 0, 0) || h != g_int(ic, WEED_LEAF_HEIGHT, 0, 0) ||
               clamping != g_int(ic, WEED_LEAF_YUV_CLAMPING, 0, WEED_YUV_CLAMPING_CLAMPED)) return batch_fallback(insts, n, tc);
    for (k = 0; k < nplanes; k++) {
      if (irow[k] != g_int(ic, WEED_LEAF_ROWSTRIDES, k, 0) || orow[k] != g_int(oc, WEED_LEAF_ROWSTRIDES, k, 0)) return batch_fallback(insts, n, tc);
      hsrc[i][k] = g_ptr(ic, WEED_LEAF_PIXEL_DATA, k); hdst[i][k] = (uint8_t *)g_ptr(oc, WEED_LEAF_PIXEL_DATA, k);
      if (!hsrc[i][k] || !hdst[i][k]) return WEED_ERROR_FILTER_INVALID;
    }
  }
  for (i = 0; i < n; i++)                                  /* independent instances only (see the caller) */
    for (k = 0; k < n; k++)
      if (k != i && (hdst[i][0] == hdst[k][0] || (const void *)hdst[i][0] == hsrc[k][0])) return batch_fallback(insts, n, tc);
  if (lgpu_init(0) != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
  memset(fr, 0, sizeof fr); memset(rsrc, 0, sizeof rsrc); memset(rdst, 0, sizeof rdst);
  for (i = 0; i < n && ret == WEED_SUCCESS; i++) {
    fxdata_t *fx = fx_data(insts[i]);
    uint8_t *ibase, *obase;
    if (!fx) { ret = WEED_ERROR_MEMORY_ALLOCATION; break; }
    fx_enter(fx);
    ibase = (uint8_t *)fx_buf(fx, 0, itot); obase = (uint8_t *)fx_buf(fx, 2, otot);
    if (!ibase || !obase) { ret = WEED_ERROR_MEMORY_ALLOCATION; break; }
    for (k = 0; k < nplanes; k++) {
      const uint8_t *res = (const uint8_t *)lives_gpu_resident_acquire(hsrc[i][k], (size_t)irow[k] * ph[k], 0);
      if (res) { fr[i].in0[k] = res; rsrc[i][k] = 1; }
      else {
        fr[i].in0[k] = ibase + ioff[k];
        if (lgpu_upload(ibase + ioff[k], hsrc[i][k], (size_t)irow[k] * ph[k], FXS)) ret = WEED_ERROR_PLUGIN_INVALID;
        uploaded = 1;
      }
      ddst[i][k] = (uint8_t *)lives_gpu_resident_acquire(hdst[i][k], (size_t)orow[k] * ph[k], 1);
      if (ddst[i][k]) rdst[i][k] = 1;
      else {
        ddst[i][k] = obase + ooff[k]; home = 1;
        if (lgpu_upload(ddst[i][k], hdst[i][k], (size_t)orow[k] * ph[k], FXS)) ret = WEED_ERROR_PLUGIN_INVALID;      /* row padding stays as it was */
      }
      fr[i].out[k] = ddst[i][k];
    }
  }
  krc = LGPU_OK;
  if (ret == WEED_SUCCESS) {
    memset(&P, 0, sizeof P);
    P.op = LGPU_FX_SOFTLIGHT; P.width = w; P.height = h; P.palette = pal; P.ip[0] = clamping == WEED_YUV_CLAMPING_UNCLAMPED;
    for (k = 0; k < nplanes; k++) { P.irow0[k] = irow[k]; P.orow[k] = orow[k]; }
    krc = lgpu_fx_batch(&P, fr, n, FXS);
  }
  for (i = 0; i < n; i++)
    for (k = 0; k < nplanes; k++) {
      if (rsrc[i][k]) lives_gpu_resident_release(hsrc[i][k], 0);
      if (rdst[i][k]) lives_gpu_resident_release(hdst[i][k], 1);
    }
  if (ret != WEED_SUCCESS) return ret;
  if (krc != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
  for (i = 0; i < n; i++)
    for (k = 0; k < nplanes; k++)
      if (!rdst[i][k] && lgpu_download(hdst[i][k], ddst[i][k], (size_t)orow[k] * ph[k], FXS)) return WEED_ERROR_PLUGIN_INVALID;
  if ((home || uploaded) && lgpu_sync(FXS)) return WEED_ERROR_PLUGIN_INVALID;
  return WEED_SUCCESS;
}
weed_error_t livesgpu_fx_process_batch(weed_plant_t **insts, int n, weed_timecode_t tc) {
  lgpu_fx_frame fr[LGPU_FX_MAX_FRAMES];
  lgpu_fx_params P;
  const void *rel[LGPU_FX_MAX_FRAMES][3];
  uint8_t *hdst[LGPU_FX_MAX_FRAMES], *ddst[LGPU_FX_MAX_FRAMES];
  batch_class_t bc, bci;
  int i, c, w = 0, h = 0, pal = 0, irow[2] = {0, 0}, orow = 0, psize, home = 0, uploaded = 0, krc;
  double amounts[LGPU_FX_MAX_FRAMES];
  weed_error_t ret = WEED_SUCCESS;
  if (!insts || n <= 0) return WEED_ERROR_FILTER_INVALID;
  for (i = 0; i < n; i++) if (!insts[i]) return WEED_ERROR_FILTER_INVALID;
  if (n > 1 && n <= LGPU_FX_MAX_FRAMES) {
    weed_plant_t *fc0 = (weed_plant_t *)g_ptr(insts[0], WEED_LEAF_FILTER_CLASS, 0);
    weed_process_f pf0 = NULL;
    if (fc0 && w_get(fc0, WEED_LEAF_PROCESS_FUNC, 0, &pf0) == WEED_SUCCESS && pf0 == p_softlight) {
      for (i = 1; i < n; i++) {
        weed_plant_t *fci = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_FILTER_CLASS, 0);
        weed_process_f pfi = NULL;
        if (!fci || w_get(fci, WEED_LEAF_PROCESS_FUNC, 0, &pfi) != WEED_SUCCESS || pfi != p_softlight) return batch_fallback(insts, n, tc);
      }
      return batch_softlight(insts, n, tc);
    }
  }
  if (!batch_class(insts[0], &bc) || n > LGPU_FX_MAX_FRAMES || n == 1) return batch_fallback(insts, n, tc);
  for (i = 0; i < n; i++) {
    weed_plant_t *oc = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_OUT_CHANNELS, 0), *pa = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_PARAMETERS, 0);
    amounts[i] = !pa ? (bc.int_param ? 128. : 0.) : bc.int_param ? (double)g_int(pa, WEED_LEAF_VALUE, 0, 128) : g_dbl(pa, WEED_LEAF_VALUE, 0.);
    if (!batch_class(insts[i], &bci) || bci.op != bc.op || bci.kind != bc.kind || !oc || has(oc, WEED_LEAF_OFFSET) || w_nelems(oc, WEED_LEAF_HEIGHT) > 1) return batch_fallback(insts, n, tc);
    if (i == 0) {
      w = g_int(oc, WEED_LEAF_WIDTH, 0, 0); h = g_int(oc, WEED_LEAF_HEIGHT, 0, 0); pal = g_int(oc, WEED_LEAF_CURRENT_PALETTE, 0, 0);
      orow = g_int(oc, WEED_LEAF_ROWSTRIDES, 0, 0);
    } else if (w != g_int(oc, WEED_LEAF_WIDTH, 0, 0) || h != g_int(oc, WEED_LEAF_HEIGHT, 0, 0) || pal != g_int(oc, WEED_LEAF_CURRENT_PALETTE, 0, 0) ||
               orow != g_int(oc, WEED_LEAF_ROWSTRIDES, 0, 0)) return batch_fallback(insts, n, tc);
    if (!g_ptr(oc, WEED_LEAF_PIXEL_DATA, 0)) return WEED_ERROR_FILTER_INVALID;
    for (c = 0; c < 2; c++) {
      weed_plant_t *ic = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_CHANNELS, c);
      if (!ic || !g_ptr(ic, WEED_LEAF_PIXEL_DATA, 0)) return WEED_ERROR_FILTER_INVALID;
      if (i == 0) irow[c] = g_int(ic, WEED_LEAF_ROWSTRIDES, 0, 0);
      else if (irow[c] != g_int(ic, WEED_LEAF_ROWSTRIDES, 0, 0)) return batch_fallback(insts, n, tc);
    }
  }
  /* one launch has no order between its frames: an instance that reads (or writes) the plane another one writes keeps the sequence of the per-instance loop */
  for (i = 0; i < n; i++) {
    const void *oi = g_ptr((weed_plant_t *)g_ptr(insts[i], WEED_LEAF_OUT_CHANNELS, 0), WEED_LEAF_PIXEL_DATA, 0);
    for (c = 0; c < n; c++) {
      if (c == i) continue;
      if (oi == g_ptr((weed_plant_t *)g_ptr(insts[c], WEED_LEAF_OUT_CHANNELS, 0), WEED_LEAF_PIXEL_DATA, 0) ||
          oi == g_ptr((weed_plant_t *)g_ptr(insts[c], WEED_LEAF_IN_CHANNELS, 0), WEED_LEAF_PIXEL_DATA, 0) ||
          oi == g_ptr((weed_plant_t *)g_ptr(insts[c], WEED_LEAF_IN_CHANNELS, 1), WEED_LEAF_PIXEL_DATA, 0)) return batch_fallback(insts, n, tc);
    }
  }
  psize = psize_of(pal);
  if (!psize || w <= 0 || h <= 0) return WEED_ERROR_FILTER_INVALID;
  if (bc.op == LGPU_FX_BLEND_CHROMA) {
    /* in-place chroma blends on planes that are still pending programs of the layer seam join those programs (as in fx_run); what cannot join runs below / instance by instance */
    weed_plant_t *rest[LGPU_FX_MAX_FRAMES];
    int nrec = 0, nrest = 0;
    for (i = 0; i < n; i++) {
      weed_plant_t *oc = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_OUT_CHANNELS, 0);
      const void *od = g_ptr(oc, WEED_LEAF_PIXEL_DATA, 0), *i0 = g_ptr((weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_CHANNELS, 0), WEED_LEAF_PIXEL_DATA, 0),
                 *i1 = g_ptr((weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_CHANNELS, 1), WEED_LEAF_PIXEL_DATA, 0);
      if (od == i0 && lives_gpu_deferred_blend_chroma(od, orow, w, h, pal, i1, irow[1], (int)amounts[i])) nrec++;
      else rest[nrest++] = insts[i];
    }
    if (nrec == n) return WEED_SUCCESS;
    if (nrec) return batch_fallback(rest, nrest, tc);
  }
  if (bc.op != LGPU_FX_TRANSITION && (pal == WEED_PALETTE_ARGB32 || (bc.op == LGPU_FX_BLEND_MULTI && psize != 3))) return batch_fallback(insts, n, tc);   /* as k_simple / k_multi serve them */
  if (lgpu_init(0) != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
  memset(fr, 0, sizeof fr); memset(rel, 0, sizeof rel);
  for (i = 0; i < n && ret == WEED_SUCCESS; i++) {
    fxdata_t *fx = fx_data(insts[i]);
    weed_plant_t *oc = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_OUT_CHANNELS, 0);
    const size_t ob = (size_t)orow * h;
    uint8_t *src0 = NULL;
    if (!fx) { ret = WEED_ERROR_MEMORY_ALLOCATION; break; }
    fx_enter(fx);
    hdst[i] = (uint8_t *)g_ptr(oc, WEED_LEAF_PIXEL_DATA, 0);
    ddst[i] = (uint8_t *)lives_gpu_resident_acquire(hdst[i], ob, 1);
    if (ddst[i]) rel[i][0] = hdst[i];
    else { ddst[i] = (uint8_t *)fx_buf(fx, 2, ob + 16); home = 1; }
    if (!ddst[i]) { ret = WEED_ERROR_MEMORY_ALLOCATION; break; }
    for (c = 0; c < 2; c++) {
      weed_plant_t *ic = (weed_plant_t *)g_ptr(insts[i], WEED_LEAF_IN_CHANNELS, c);
      uint8_t *hs = (uint8_t *)g_ptr(ic, WEED_LEAF_PIXEL_DATA, 0), *ds;
      const size_t ib = (size_t)irow[c] * h;
      if (c == 0) src0 = hs;
      if (c == 0 && hs == hdst[i]) {                    /* in place (the two iris classes may be) */
        if (!rel[i][0] && lgpu_upload(ddst[i], hs, ob, FXS)) ret = WEED_ERROR_PLUGIN_INVALID;
        fr[i].in0[0] = ddst[i];
        continue;
      }
      ds = (uint8_t *)lives_gpu_resident_acquire(hs, ib, 0);
      if (ds) rel[i][1 + c] = hs;
      else {
        ds = (uint8_t *)fx_buf(fx, c, ib + 16);
        if (!ds) { ret = WEED_ERROR_MEMORY_ALLOCATION; break; }
        if (lgpu_upload(ds, hs, ib, FXS)) ret = WEED_ERROR_PLUGIN_INVALID;
        uploaded = 1;
      }
      if (c == 0) fr[i].in0[0] = ds; else fr[i].in1[0] = ds;
    }
    if (src0 != hdst[i] && !rel[i][0] && ret == WEED_SUCCESS && lgpu_upload(ddst[i], hdst[i], ob, FXS)) ret = WEED_ERROR_PLUGIN_INVALID;   /* row padding */
    fr[i].out[0] = ddst[i];
  }
  krc = LGPU_OK;
  if (ret == WEED_SUCCESS) {
    memset(&P, 0, sizeof P);
    P.op = bc.op; P.width = w; P.height = h;
    P.irow0[0] = irow[0]; P.irow1[0] = irow[1]; P.orow[0] = orow;
    P.frame_dp0 = amounts;
    if (bc.op == LGPU_FX_TRANSITION) { P.ip[0] = bc.kind; P.ip[1] = psize; }
    else if (bc.op == LGPU_FX_BLEND_CHROMA) { P.ip[0] = psize; P.ip[1] = 0; }
    else if (bc.op == LGPU_FX_BLEND_LUMA) { P.ip[0] = bc.kind; P.ip[1] = psize; P.ip[2] = (pal == WEED_PALETTE_BGR24 || pal == WEED_PALETTE_BGRA32) ? 1 : 0; }
    else { P.ip[0] = bc.kind; P.ip[1] = pal == WEED_PALETTE_BGR24; }
    krc = lgpu_fx_batch(&P, fr, n, FXS);
  }
  for (i = 0; i < n; i++) { lives_gpu_resident_release(rel[i][0], 1); lives_gpu_resident_release(rel[i][1], 0); lives_gpu_resident_release(rel[i][2], 0); }
  if (ret != WEED_SUCCESS) return ret;
  if (krc != LGPU_OK) { fprintf(stderr, "livesgpu_fx: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
  for (i = 0; i < n; i++)
    if (!rel[i][0] && lgpu_download(hdst[i], ddst[i], (size_t)orow * h, FXS)) return WEED_ERROR_PLUGIN_INVALID;
  if ((home || uploaded) && lgpu_sync(FXS)) return WEED_ERROR_PLUGIN_INVALID;    /* as fx_run: host planes are complete / reusable on return */
  return WEED_SUCCESS;
}

static double g_dbl_at(weed_plant_t *p, const char *k, int idx, double dflt) { double v = dflt; if (w_get(p, k, (weed_size_t)idx, &v) != WEED_SUCCESS) return dflt; return v; }
static weed_error_t p_compositor(weed_plant_t *inst, weed_timecode_t tc) {
  fxdata_t *fx = fx_data(inst);
  weed_plant_t *ochan = (weed_plant_t *)g_ptr(inst, WEED_LEAF_OUT_CHANNELS, 0), *par[7];
  lgpu_comp_layer layers[LGPU_COMP_MAX_LAYERS];
  void *tofree[2 * LGPU_COMP_MAX_LAYERS + 1];
  const void *rel_in[LGPU_COMP_MAX_LAYERS];        /* resident in planes to release (read) once the work is enqueued */
  int nfree = 0, nrel = 0, uploaded = 0, nin, z, i, owidth, oheight, pal, psize, orow, bg[3], revz, rc = LGPU_OK;
  uint8_t *dst, *ddst = NULL, *res_dst = NULL;
  (void)tc;
  if (!fx || !ochan) return WEED_ERROR_FILTER_INVALID;
  fx_enter(fx);
  nin = (int)w_nelems(inst, WEED_LEAF_IN_CHANNELS);
  owidth = g_int(ochan, WEED_LEAF_WIDTH, 0, 0); oheight = g_int(ochan, WEED_LEAF_HEIGHT, 0, 0);
  pal = g_int(ochan, WEED_LEAF_CURRENT_PALETTE, 0, 0); psize = psize_of(pal);
  orow = g_int(ochan, WEED_LEAF_ROWSTRIDES, 0, 0);
  dst = (uint8_t *)g_ptr(ochan, WEED_LEAF_PIXEL_DATA, 0);
  if (!dst || owidth <= 0 || oheight <= 0 || !(pal == WEED_PALETTE_RGB24 || pal == WEED_PALETTE_BGR24 || pal == WEED_PALETTE_RGBA32 || pal == WEED_PALETTE_BGRA32)) return WEED_ERROR_FILTER_INVALID;
  if (nin > LGPU_COMP_MAX_LAYERS) { fprintf(stderr, "livesgpu_fx: compositor: more than %d in channels\n", LGPU_COMP_MAX_LAYERS); return WEED_ERROR_FILTER_INVALID; }
  for (i = 0; i < 7; i++) if (!(par[i] = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_PARAMETERS, i))) return WEED_ERROR_FILTER_INVALID;
  for (i = 0; i < 3; i++) bg[i] = g_int(par[5], WEED_LEAF_VALUE, i, 0);
  revz = g_int(par[6], WEED_LEAF_VALUE, 0, WEED_FALSE) == WEED_TRUE;
  memset(layers, 0, sizeof layers);
  for (z = 0; z < nin && rc == LGPU_OK; z++) {
    weed_plant_t *ic = (weed_plant_t *)g_ptr(inst, WEED_LEAF_IN_CHANNELS, z);
    const uint8_t *src, *frame_d;
    int in_width, in_height, full_height, irow, out_width, out_height, cutleft = 0, cuttop = 0, interp, srow;
    double scx, scy;
    void *dsrc = NULL, *dscaled = NULL;
    if (!ic || g_int(ic, WEED_LEAF_DISABLED, 0, WEED_FALSE) == WEED_TRUE) continue;                 /* :192-193 */
    src = (const uint8_t *)g_ptr(ic, WEED_LEAF_PIXEL_DATA, 0);
    if (!src) continue;
    scx = z < (int)w_nelems(par[2], WEED_LEAF_VALUE) ? g_dbl_at(par[2], WEED_LEAF_VALUE, z, 1.) : 1.;
    scy = z < (int)w_nelems(par[3], WEED_LEAF_VALUE) ? g_dbl_at(par[3], WEED_LEAF_VALUE, z, 1.) : 1.;
    out_width = (((int)(owidth * scx + 1.)) >> 1) << 1;
    out_height = (((int)(oheight * scy + 1.)) >> 1) << 1;
    if (out_width * out_height < 16) continue;                                                        /* :221 */
    in_width = g_int(ic, WEED_LEAF_WIDTH, 0, 0); in_height = g_int(ic, WEED_LEAF_HEIGHT, 0, 0); irow = g_int(ic, WEED_LEAF_ROWSTRIDES, 0, 0);
    if (in_width <= 0 || in_height <= 0 || irow < in_width * psize) { rc = LGPU_E_BADARG; break; }
    full_height = in_height;
    if (w_nelems(ic, WEED_LEAF_INNER_SIZE) >= 4) {                                                  /* letterboxed channel: :228-258 */
      const int lbx = g_int(ic, WEED_LEAF_INNER_SIZE, 0, 0), lby = g_int(ic, WEED_LEAF_INNER_SIZE, 1, 0);
      const int lbw = g_int(ic, WEED_LEAF_INNER_SIZE, 2, in_width), lbh = g_int(ic, WEED_LEAF_INNER_SIZE, 3, in_height);
      const int lbwidth = (int)(lbw / scx), lbheight = (int)(lbh / scy);
      if (lbwidth < in_width) {
        const int totlbw = in_width - lbw;
        const double lbscale = (double)(in_width - lbwidth) / (double)totlbw;
        const int extra = (int)((double)(in_width - lbx - lbw) * lbscale + .5);
        cutleft = (int)((double)lbx * (1. - lbscale) + .5);
        in_width = lbx - cutleft + lbw + extra;
      } else { cutleft = lbx; in_width = lbw; }
      if (lbheight < in_height) {
        const int totlbh = in_height - lbh;
        const double lbscale = ((double)in_height - (double)lbheight) / (double)totlbh;
        const int extra = (int)((double)(in_height - lby - lbh) * lbscale + .5);
        cuttop = (int)((double)lby * (1. - lbscale) + .5);
        in_height = lby - cuttop + lbh + extra;
      } else { cuttop = lby; in_height = lbh; }
      if (in_width <= 0 || in_height <= 0) continue;
    }
    interp = (out_width > in_width || out_height > in_height) ? LIVES_INTERP_BEST : LIVES_INTERP_NORMAL;   /* up_interp HYPER / down_interp BILINEAR, :154-155 */
    srow = (out_width * psize + 3) & ~3;                                                               /* the scaled pixbuf's rowstride */
    /* the plane of a pinned layer (lives_gpu_layer_pin) is scaled from where it lives in HBM -- its host bytes are stale by the pinning contract; otherwise the
       host plane is uploaded */
    frame_d = (const uint8_t *)lives_gpu_resident_acquire(src, (size_t)irow * full_height, 0);
    if (frame_d) { rel_in[nrel++] = src; frame_d += (size_t)cuttop * irow; }
    else {
      if ((rc = lgpu_malloc_ordered(&dsrc, (size_t)irow * in_height + 16, FXS))) break;
      tofree[nfree++] = dsrc;
      if ((rc = lgpu_upload(dsrc, src + (size_t)cuttop * irow, (size_t)irow * in_height, FXS))) break;
      uploaded = 1;
      frame_d = (const uint8_t *)dsrc;
    }
    if ((rc = lgpu_malloc_ordered(&dscaled, (size_t)srow * out_height + 16, FXS))) break;
    tofree[nfree++] = dscaled;
    if ((rc = lgpu_pixbuf_scale(frame_d + (size_t)cutleft * psize, irow, in_width, in_height, (uint8_t *)dscaled, srow, out_width, out_height, psize, interp, FXS))) break;
    layers[z].src_d = (const uint8_t *)dscaled; layers[z].irow = srow; layers[z].width = out_width; layers[z].height = out_height;
    layers[z].offs_x = z < (int)w_nelems(par[0], WEED_LEAF_VALUE) ? (int)(g_dbl_at(par[0], WEED_LEAF_VALUE, z, 0.) * (double)owidth) : 0;
    layers[z].offs_y = z < (int)w_nelems(par[1], WEED_LEAF_VALUE) ? (int)(g_dbl_at(par[1], WEED_LEAF_VALUE, z, 0.) * (double)oheight) : 0;
    layers[z].alpha = z < (int)w_nelems(par[4], WEED_LEAF_VALUE) ? g_dbl_at(par[4], WEED_LEAF_VALUE, z, 1.) : 1.;
  }
  /* the out channel: composited into the device copy when its layer is pinned (no download: the device copy IS the plane until lives_gpu_layer_sync()) */
  if (rc == LGPU_OK) res_dst = (uint8_t *)lives_gpu_resident_acquire(dst, (size_t)orow * oheight, 1);
  if (rc == LGPU_OK) {
    if (res_dst) ddst = res_dst;
    else {
      rc = lgpu_malloc_ordered((void **)&ddst, (size_t)orow * oheight + 16, FXS);
      if (rc == LGPU_OK) {
        tofree[nfree++] = ddst;
        if (orow != owidth * psize) rc = lgpu_upload(ddst, dst, (size_t)orow * oheight, FXS);           /* row padding keeps the host's bytes */
      }
    }
  }
  if (rc == LGPU_OK) rc = lgpu_composite(ddst, orow, owidth, oheight, psize, pal == WEED_PALETTE_BGR24 || pal == WEED_PALETTE_BGRA32, bg, layers, nin, revz, FXS);
  for (i = 0; i < nrel; i++) lives_gpu_resident_release(rel_in[i], 0);
  if (res_dst) lives_gpu_resident_release(dst, 1);
  if (rc == LGPU_OK && !res_dst) rc = lgpu_download(dst, ddst, (size_t)orow * oheight, FXS);
  /* the result has to be home on return unless it lives on the device; uploads out of host planes have to be over either way (the host may reuse them) */
  if (rc == LGPU_OK && (!res_dst || uploaded)) rc = lgpu_sync(FXS);
  for (i = 0; i < nfree; i++) lgpu_free_ordered(tofree[i], FXS);
  if (rc != LGPU_OK) { fprintf(stderr, "livesgpu_fx: compositor: %s\n", lgpu_last_error()); return WEED_ERROR_PLUGIN_INVALID; }
  return WEED_SUCCESS;
}

/* ---- class construction (same leaves as weed_filter_class_init & friends, weed-plugin-utils.c:258-420) ---- */
static weed_plant_t *chantmpl(const char *name, int flags) {
  weed_plant_t *t = w_new(WEED_PLANT_CHANNEL_TEMPLATE);
  s_str(t, WEED_LEAF_NAME, name);
  s_int(t, WEED_LEAF_FLAGS, flags);
  return t;
}
static weed_plant_t *paramtmpl_gui(weed_plant_t *pt, const char *label) {
  weed_plant_t *gui = w_new(WEED_PLANT_GUI);
  w_set(pt, WEED_LEAF_GUI, WEED_SEED_PLANTPTR, 1, &gui);
  s_str(gui, WEED_LEAF_LABEL, label);
  s_bool(gui, WEED_LEAF_USE_MNEMONIC, WEED_TRUE);
  return gui;
}
static weed_plant_t *int_param(const char *name, const char *label, int def, int mn, int mx, int transition) {
  weed_plant_t *pt = w_new(WEED_PLANT_PARAMETER_TEMPLATE);
  s_str(pt, WEED_LEAF_NAME, name); s_int(pt, WEED_LEAF_PARAM_TYPE, WEED_PARAM_INTEGER);
  s_int(pt, WEED_LEAF_DEFAULT, def); s_int(pt, WEED_LEAF_MIN, mn); s_int(pt, WEED_LEAF_MAX, mx);
  paramtmpl_gui(pt, label);
  if (transition) s_bool(pt, WEED_LEAF_IS_TRANSITION, WEED_TRUE);
  return pt;
}
static weed_plant_t *switch_param(const char *name, const char *label, int def) {   /* weed_switch_init, weed-plugin-utils.c:350-361 */
  weed_plant_t *pt = w_new(WEED_PLANT_PARAMETER_TEMPLATE);
  s_str(pt, WEED_LEAF_NAME, name); s_int(pt, WEED_LEAF_PARAM_TYPE, WEED_PARAM_SWITCH);
  s_bool(pt, WEED_LEAF_DEFAULT, def);
  paramtmpl_gui(pt, label);
  return pt;
}
static weed_plant_t *float_param(const char *name, const char *label, double def, double mn, double mx) {
  weed_plant_t *pt = w_new(WEED_PLANT_PARAMETER_TEMPLATE), *gui;
  s_str(pt, WEED_LEAF_NAME, name); s_int(pt, WEED_LEAF_PARAM_TYPE, WEED_PARAM_FLOAT);
  s_dbl(pt, WEED_LEAF_DEFAULT, def); s_dbl(pt, WEED_LEAF_MIN, mn); s_dbl(pt, WEED_LEAF_MAX, mx);
  gui = paramtmpl_gui(pt, label);
  s_int(gui, WEED_LEAF_DECIMALS, 2);
  return pt;
}
static weed_plant_t *rgb_param(const char *name, const char *label, int r, int g, int b) {
  weed_plant_t *pt = w_new(WEED_PLANT_PARAMETER_TEMPLATE);
  int32_t def[3] = {r, g, b};
  s_str(pt, WEED_LEAF_NAME, name); s_int(pt, WEED_LEAF_PARAM_TYPE, WEED_PARAM_COLOR); s_int(pt, WEED_LEAF_COLORSPACE, WEED_COLORSPACE_RGB);
  w_set(pt, WEED_LEAF_DEFAULT, WEED_SEED_INT, 3, def);
  s_int(pt, WEED_LEAF_MIN, 0); s_int(pt, WEED_LEAF_MAX, 255);
  paramtmpl_gui(pt, label);
  return pt;
}
static void add_filter(weed_plant_t *pinfo, const char *name, int flags, const int32_t *pals, int npals, weed_process_f proc,
                       int nin, const char *in0, const char *in1, const char *out0, weed_plant_t **params, int nparams) {
  weed_plant_t *fc = w_new(WEED_PLANT_FILTER_CLASS), *ins[2], *outs[1], **flt;
  const char *author = "lives-gfx950";
  weed_init_f fi = fx_init;
  weed_deinit_f fd = fx_deinit;
  weed_size_t n, i;
  s_str(fc, WEED_LEAF_NAME, name); s_str(fc, WEED_LEAF_AUTHOR, author); s_int(fc, WEED_LEAF_VERSION, 1); s_int(fc, WEED_LEAF_FLAGS, flags);
  w_set(fc, WEED_LEAF_INIT_FUNC, WEED_SEED_FUNCPTR, 1, &fi);
  w_set(fc, WEED_LEAF_PROCESS_FUNC, WEED_SEED_FUNCPTR, 1, &proc);
  w_set(fc, WEED_LEAF_DEINIT_FUNC, WEED_SEED_FUNCPTR, 1, &fd);
  ins[0] = chantmpl(in0, 0);
  if (nin > 1) ins[1] = chantmpl(in1, 0);
  outs[0] = chantmpl(out0, WEED_CHANNEL_CAN_DO_INPLACE);
  w_set(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, WEED_SEED_PLANTPTR, (weed_size_t)nin, ins);
  w_set(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, WEED_SEED_PLANTPTR, 1, outs);
  w_set(fc, WEED_LEAF_IN_PARAMETER_TEMPLATES, WEED_SEED_PLANTPTR, (weed_size_t)nparams, nparams ? params : NULL);
  w_set(fc, WEED_LEAF_OUT_PARAMETER_TEMPLATES, WEED_SEED_PLANTPTR, 0, NULL);
  w_set(fc, WEED_LEAF_PALETTE_LIST, WEED_SEED_INT, (weed_size_t)npals, (void *)pals);
  /* weed_plugin_info_add_filter_class */
  n = has(pinfo, WEED_LEAF_FILTERS) ? w_nelems(pinfo, WEED_LEAF_FILTERS) : 0;
  flt = (weed_plant_t **)w_malloc((n + 1) * sizeof(weed_plant_t *));
  for (i = 0; i < n; i++) w_get(pinfo, WEED_LEAF_FILTERS, i, &flt[i]);
  flt[n] = fc;
  w_set(pinfo, WEED_LEAF_FILTERS, WEED_SEED_PLANTPTR, n + 1, flt);
  w_set(fc, WEED_LEAF_PLUGIN_INFO, WEED_SEED_PLANTPTR, 1, &pinfo);
  w_free(flt);
}

weed_plant_t *weed_setup(weed_bootstrap_f weed_boot) {
  static const int32_t rgb_all[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_ARGB32};
  static const int32_t rgb_noargb[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32};
  static const int32_t rgb24[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24};
  static const int32_t packed[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_ARGB32,
                                   WEED_PALETTE_YUV888, WEED_PALETTE_YUVA8888, WEED_PALETTE_UYVY, WEED_PALETTE_YUYV};
  weed_default_getter_f dget;
  weed_plant_t *host_info, *pinfo = NULL, *p[205];
  int32_t filter_api = 0;
  int i;
  if (!weed_boot) return NULL;
  host_info = (*weed_boot)(&dget, WEED_API_VERSION_MIN, 203, 200, WEED_FILTER_API_VERSION);
  if (!host_info) return NULL;
  if (dget(host_info, WEED_LEAF_GET_FUNC, (void *)&w_get) != WEED_SUCCESS) return NULL;
  if (dget(host_info, WEED_LEAF_MALLOC_FUNC, (void *)&w_malloc) != WEED_SUCCESS) return NULL;
  if (dget(host_info, WEED_LEAF_FREE_FUNC, (void *)&w_free) != WEED_SUCCESS) return NULL;
  if (w_get(host_info, WEED_LEAF_SET_FUNC, 0, &w_set) != WEED_SUCCESS) return NULL;
  if (w_get(host_info, WEED_PLANT_NEW_FUNC, 0, &w_new) != WEED_SUCCESS) return NULL;
  if (w_get(host_info, WEED_LEAF_NUM_ELEMENTS_FUNC, 0, &w_nelems) != WEED_SUCCESS) return NULL;
  w_get(host_info, WEED_LEAF_FILTER_API_VERSION, 0, &filter_api);
  if (has(host_info, WEED_LEAF_PLUGIN_INFO)) w_get(host_info, WEED_LEAF_PLUGIN_INFO, 0, &pinfo);
  if (!pinfo) pinfo = w_new(WEED_PLANT_PLUGIN_INFO);
  if (!pinfo) return NULL;
  w_set(pinfo, WEED_LEAF_HOST_INFO, WEED_SEED_PLANTPTR, 1, &host_info);

  /* simple_blend.c:234-291 -- PREF_LINEAR_GAMMA kept where the reference asks for it; STATEFUL / MAY_THREAD dropped (no shared state, one call) */
  p[0] = int_param("amount", "Blend _amount", 128, 0, 255, 1);
  add_filter(pinfo, "chroma blend", WEED_FILTER_PREF_LINEAR_GAMMA, rgb_all, 5, p_chroma, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
  {
    static const struct { const char *n; weed_process_f f; int flags; } lum[] = {
        {"luma overlay", p_lumo, 0}, {"luma underlay", p_lumu, 0}, {"negative luma overlay", p_nlumo, 0}, {"averaged luma overlay", p_avlumo, WEED_FILTER_PREF_LINEAR_GAMMA}};
    for (i = 0; i < 4; i++) {
      p[0] = int_param("threshold", "luma _threshold", 64, 0, 255, 1);
      add_filter(pinfo, lum[i].n, lum[i].flags, rgb_noargb, 4, lum[i].f, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
    }
  }
  {
    static const struct { const char *n; weed_process_f f; } mb[] = {{"blend_multiply", p_mpy}, {"blend_screen", p_screen}, {"blend_darken", p_darken},
        {"blend_lighten", p_lighten}, {"blend_overlay", p_overlay}, {"blend_dodge", p_dodge}, {"blend_burn", p_burn}};
    for (i = 0; i < 7; i++) {
      p[0] = int_param("amount", "Blend _amount", 128, 0, 255, 1);
      add_filter(pinfo, mb[i].n, WEED_FILTER_PREF_LINEAR_GAMMA, rgb24, 2, mb[i].f, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
    }
  }
  p[0] = float_param("delta", "_Delta", .2, 0., 1.);
  p[1] = float_param("opac", "_Opacity", 1., 0., 1.);
  p[2] = rgb_param("col", "_Colour", 0, 0, 255);
  add_filter(pinfo, "colorkey", 0, rgb24, 2, p_ckey, 2, "in_channel0", "in_channel1", "out_channel0", p, 3);
  add_filter(pinfo, "mirrorx", 0, packed, 9, p_mirrorx, 1, "in channel 0", NULL, "out channel 0", NULL, 0);
  add_filter(pinfo, "mirrory", 0, packed, 9, p_mirrory, 1, "in channel 0", NULL, "out channel 0", NULL, 0);
  add_filter(pinfo, "mirrorxy", 0, packed, 9, p_mirrorxy, 1, "in channel 0", NULL, "out channel 0", NULL, 0);
  /* edge.c:251-263: "edge detect", string-list parameter "mode", in channel REINIT_ON_SIZE_CHANGE, out channel CAN_DO_INPLACE */
  {
    static const char *modes[] = {"normal", "monochrome", "supercolour"};
    weed_plant_t *gui = NULL, *fc = NULL, *ict = NULL;
    p[0] = int_param("mode", "Edge _Mode", 0, 0, 2, 0);
    w_get(p[0], WEED_LEAF_GUI, 0, &gui);
    if (gui) w_set(gui, WEED_LEAF_CHOICES, WEED_SEED_STRING, 3, modes);
    add_filter(pinfo, "edge detect", WEED_FILTER_PREF_LINEAR_GAMMA, rgb_all, 5, p_edge, 1, "in channel 0", NULL, "out channel 0", p, 1);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) w_get(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, 0, &ict);
    if (ict) s_int(ict, WEED_LEAF_FLAGS, WEED_CHANNEL_REINIT_ON_SIZE_CHANGE);
  }
  /* multi_transitions.c:262-296: "iris rectangle", "iris circle" (out channel CAN_DO_INPLACE), "4 way split" (not in place);
     float parameter "amount" flagged as the transition parameter; MAY_THREAD dropped (whole-frame geometry, one call) */
  {
    static const int32_t pk[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_YUV888, WEED_PALETTE_YUVA8888};
    static const struct { const char *n; weed_process_f f; int inplace; } tr[] = {{"iris rectangle", p_irisr, 1}, {"iris circle", p_irisc, 1}, {"4 way split", p_fourw, 0}};
    for (i = 0; i < 3; i++) {
      weed_plant_t *fc = NULL, *oct = NULL;
      p[0] = float_param("amount", "_Transition", 0., 0., 1.);
      s_bool(p[0], WEED_LEAF_IS_TRANSITION, WEED_TRUE);
      add_filter(pinfo, tr[i].n, 0, pk, 6, tr[i].f, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
      if (!tr[i].inplace) {
        w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
        if (fc) w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct);
        if (oct) s_int(oct, WEED_LEAF_FLAGS, 0);
      }
    }
  }
  /* slide_over.c:148-196: "slide over": integer transition parameter, five direction radios (group 1, REINIT_ON_VALUE_CHANGE),
     two switches; out channel not in place; packed palettes of 3 / 4 bytes */
  {
    static const int32_t pk[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_ARGB32,
                                 WEED_PALETTE_YUV888, WEED_PALETTE_YUVA8888, WEED_PALETTE_UYVY, WEED_PALETTE_YUYV};   /* macropixels of 4 bytes */
    static const char *rn[] = {"dir_rand", "dir_r2l", "dir_l2r", "dir_b2t", "dir_t2b"},
                      *rl[] = {"_Random", "_Right to left", "_Left to right", "_Bottom to top", "_Top to bottom"};
    weed_plant_t *fc = NULL, *oct = NULL;
    p[0] = int_param("amount", "Transition _value", 0, 0, 255, 1);
    for (i = 0; i < 5; i++) {
      p[1 + i] = switch_param(rn[i], rl[i], i == 0 ? WEED_TRUE : WEED_FALSE);
      s_int(p[1 + i], WEED_LEAF_GROUP, 1);
      s_int(p[1 + i], WEED_LEAF_FLAGS, WEED_PARAMETER_REINIT_ON_VALUE_CHANGE);
    }
    p[6] = switch_param("mlower", "_Slide lower clip", WEED_TRUE);
    p[7] = switch_param("mupper", "_Slide upper clip", WEED_FALSE);
    add_filter(pinfo, "slide over", 0, pk, 9, p_slide, 2, "in channel 0", "in channel 1", "out channel 0", p, 8);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct);
    if (oct) s_int(oct, WEED_LEAF_FLAGS, 0);
  }
  /* deinterlace.c:312-327: "deinterlace", no parameters, out channel CAN_DO_INPLACE.  The planar palettes of the reference's list are
     left out: its loop does nothing for them (pixel_size() == 0) or crashes (YUV444P) */
  {
    static const int32_t pk[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_YUV888, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_ARGB32,
                                 WEED_PALETTE_YUVA8888, WEED_PALETTE_UYVY, WEED_PALETTE_YUYV};
    add_filter(pinfo, "deinterlace", 0, pk, 9, p_deint, 1, "in channel 0", NULL, "out channel 0", p, 0);
  }
  /* RGBdelay.c:435-528: "RGBdelay" (RGB24 / BGR24) and "YUVdelay" (YUV888): parameter 0 = cache size (REINIT_ON_VALUE_CHANGE), then 51 groups of
     three switches + one strength (defaults: R of frame 0, G of frame -4, B of frame -8); in channel REINIT_ON_SIZE_CHANGE, out channel in place */
  {
    static const int32_t prgb[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24}, pyuv[] = {WEED_PALETTE_YUV888};
    int which, g, c;
    for (which = 0; which < 2; which++) {
      weed_plant_t *fc = NULL, *ict = NULL;
      char label[64];
      p[0] = int_param("fcsize", "Frame _Cache Size (max)", 20, 0, 50, 0);
      s_int(p[0], WEED_LEAF_FLAGS, WEED_PARAMETER_REINIT_ON_VALUE_CHANGE);
      for (g = 0; g < 51; g++) {
        for (c = 0; c < 3; c++) {
          const int idx = 4 * g + 1 + c;
          if (c == 2) snprintf(label, sizeof(label), "        Frame -%-2d       ", g); else label[0] = 0;
          p[idx] = switch_param("", label, (idx == 1 || idx == 18 || idx == 35) ? WEED_TRUE : WEED_FALSE);
        }
        p[4 * g + 4] = float_param("", "", 1., 0., 1.);
      }
      add_filter(pinfo, which ? "YUVdelay" : "RGBdelay", 0, which ? pyuv : prgb, which ? 1 : 2, p_rgbdelay, 1, "in channel 0", NULL, "out channel 0", p, 205);
      w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
      if (fc) w_get(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, 0, &ict);
      if (ict) s_int(ict, WEED_LEAF_FLAGS, WEED_CHANNEL_REINIT_ON_SIZE_CHANGE);
    }
  }
  /* scripts/negate.script, posterise.script, ccorrect.script (<palette_list>, <params>): per-pixel table effects; out channel in place */
  {
    static const int32_t rgbx[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_ARGB32};
    add_filter(pinfo, "negate", 0, rgbx, 5, p_negate, 1, "in_channel0", NULL, "out_channel0", p, 0);
    p[0] = int_param("levels", "Colour _levels", 1, 1, 8, 0);
    add_filter(pinfo, "posterise", 0, rgbx, 4, p_posterise, 1, "in_channel0", NULL, "out_channel0", p, 1);
    p[0] = float_param("red", "_Red factor", 1., 0., 4.);
    p[1] = float_param("green", "_Green factor", 1., 0., 4.);
    p[2] = float_param("blue", "_Blue factor", 1., 0., 4.);
    add_filter(pinfo, "ccorrect", 0, rgbx, 5, p_ccorrect, 1, "in_channel0", NULL, "out_channel0", p, 3);
  }
  /* layout_blends.c:121-153: "triple split", RGB24 / BGR24, seven parameters (two of them radios of group 1), out channel in place */
  {
    static const int32_t p24[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24};
    p[0] = float_param("start", "_Start", 0.666667, 0., 1.);
    p[1] = switch_param("sym", "Make s_ymmetrical", WEED_TRUE); s_int(p[1], WEED_LEAF_GROUP, 1);
    p[2] = switch_param("usend", "Use _end value", WEED_FALSE); s_int(p[2], WEED_LEAF_GROUP, 1);
    p[3] = float_param("end", "_End", 0.333333, 0., 1.);
    p[4] = switch_param("vert", "Split _horizontally", WEED_FALSE);
    p[5] = float_param("borderw", "Border _width", 0., 0., 0.5);
    p[6] = rgb_param("borderc", "Border _colour", 0, 0, 0);
    add_filter(pinfo, "triple split", 0, p24, 2, p_tsplit, 2, "in channel 0", "in channel 1", "out channel 0", p, 7);
  }
  /* multi_transitions.c:298-309: "dissolve": same templates as the iris transitions, out channel in place + REINIT_ON_SIZE_CHANGE */
  {
    static const int32_t pk[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32, WEED_PALETTE_YUV888, WEED_PALETTE_YUVA8888};
    weed_plant_t *fc = NULL, *oct = NULL;
    p[0] = float_param("amount", "_Transition", 0., 0., 1.);
    s_bool(p[0], WEED_LEAF_IS_TRANSITION, WEED_TRUE);
    add_filter(pinfo, "dissolve", 0, pk, 6, p_dissolve, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct);
    if (oct) s_int(oct, WEED_LEAF_FLAGS, WEED_CHANNEL_CAN_DO_INPLACE | WEED_CHANNEL_REINIT_ON_SIZE_CHANGE);
    /* multi_transitions.c:314-326: "rand replace", the fifth class of that plugin: same templates, out channel in place */
    p[0] = float_param("amount", "_Transition", 0., 0., 1.);
    s_bool(p[0], WEED_LEAF_IS_TRANSITION, WEED_TRUE);
    add_filter(pinfo, "rand replace", 0, pk, 6, p_rreplace, 2, "in channel 0", "in channel 1", "out channel 0", p, 1);
    fc = NULL; oct = NULL;
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct);
    if (oct) s_int(oct, WEED_LEAF_FLAGS, WEED_CHANNEL_CAN_DO_INPLACE);
  }
  /* blurzoom.c:424-446: "blurzoom" by effectTV, string-list parameters "mode" and "color", BGRA32 / RGBA32, out channel NOT in place */
  {
    static const char *modes[] = {"normal", "strobe", "strobe2", "trigger"}, *patterns[] = {"blue", "green", "red", "white"};
    static const int32_t bzpal[] = {WEED_PALETTE_BGRA32, WEED_PALETTE_RGBA32};
    weed_plant_t *gui = NULL, *fc = NULL, *ict = NULL, *oct = NULL;
    p[0] = int_param("mode", "Trigger _Mode", 0, 0, 3, 0);
    w_get(p[0], WEED_LEAF_GUI, 0, &gui);
    if (gui) w_set(gui, WEED_LEAF_CHOICES, WEED_SEED_STRING, 4, modes);
    p[1] = int_param("color", "_Color", 0, 0, 3, 0);
    gui = NULL;
    w_get(p[1], WEED_LEAF_GUI, 0, &gui);
    if (gui) w_set(gui, WEED_LEAF_CHOICES, WEED_SEED_STRING, 4, patterns);
    add_filter(pinfo, "blurzoom", WEED_FILTER_PREF_LINEAR_GAMMA, bzpal, 2, p_blurzoom, 1, "in channel 0", NULL, "out channel 0", p, 2);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) { w_get(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, 0, &ict); w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct); }
    if (ict) s_int(ict, WEED_LEAF_FLAGS, WEED_CHANNEL_REINIT_ON_SIZE_CHANGE);
    if (oct) s_int(oct, WEED_LEAF_FLAGS, 0);
  }
  /* softlight.c:160-176: planar YUV palettes, out channel NOT in place, in channel prefers unclamped luma */
  {
    static const int32_t yuvp[] = {WEED_PALETTE_YUV444P, WEED_PALETTE_YUVA4444P, WEED_PALETTE_YUV422P, WEED_PALETTE_YUV420P, WEED_PALETTE_YVU420P};
    weed_plant_t *fc = NULL, *ict = NULL, *oct = NULL;
    add_filter(pinfo, "softlight", 0, yuvp, 5, p_softlight, 1, "in channel 0", NULL, "out channel 0", NULL, 0);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) { w_get(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, 0, &ict); w_get(fc, WEED_LEAF_OUT_CHANNEL_TEMPLATES, 0, &oct); }
    if (ict) s_int(ict, WEED_LEAF_YUV_CLAMPING, WEED_YUV_CLAMPING_UNCLAMPED);
    if (oct) s_int(oct, WEED_LEAF_FLAGS, 0);
  }
  /* gdk/compositor.c:295-351: "compositor": one in channel template repeated without limit (max_repeats 0), five per-channel float parameters
     (VARIABLE_SIZE | VALUE_PER_CHANNEL, new_default), background colour, z switch; WEED_FILTER_CHANNEL_SIZES_MAY_VARY.  The palette list is the reference's
     (RGBA32 twice there; BGRA32 is what its paint loop's r / b swap is for and is accepted here too) */
  {
    static const int32_t pk[] = {WEED_PALETTE_RGB24, WEED_PALETTE_BGR24, WEED_PALETTE_RGBA32, WEED_PALETTE_BGRA32};
    static const char *rfx[] = {"layout|p6|", "layout|p0|p1|", "layout|p2|p3|", "layout|p4|", "layout|hseparator|", "layout|p5|", "special|framedraw|multirect|0|1|2|3|4|"};
    static const double nd[5] = {0., 0., 1., 1., 1.};
    weed_plant_t *fc = NULL, *ict = NULL, *gui;
    p[0] = float_param("xoffs", "_X offset", 0., 0., 1.); p[1] = float_param("yoffs", "_Y offset", 0., 0., 1.);
    p[2] = float_param("scalex", "Scale _width", 1., 0., 1.); p[3] = float_param("scaley", "Scale _height", 1., 0., 1.);
    p[4] = float_param("alpha", "_Alpha", 1.0, 0.0, 1.0);
    p[5] = rgb_param("bgcol", "_Background color", 0, 0, 0);
    p[6] = switch_param("revz", "Invert _Z Index", WEED_FALSE);
    for (i = 0; i < 5; i++) { s_int(p[i], WEED_LEAF_FLAGS, WEED_PARAMETER_VARIABLE_SIZE | WEED_PARAMETER_VALUE_PER_CHANNEL); s_dbl(p[i], WEED_LEAF_NEW_DEFAULT, nd[i]); }
    s_str(p[6], WEED_LEAF_DESCRIPTION, "If checked, the rear frames overlay the front ones.");
    add_filter(pinfo, "compositor", WEED_FILTER_CHANNEL_SIZES_MAY_VARY, pk, 4, p_compositor, 1, "in channel 0", NULL, "out channel 0", p, 7);
    w_get(pinfo, WEED_LEAF_FILTERS, w_nelems(pinfo, WEED_LEAF_FILTERS) - 1, &fc);
    if (fc) {
      w_get(fc, WEED_LEAF_IN_CHANNEL_TEMPLATES, 0, &ict);
      if (ict) s_int(ict, WEED_LEAF_MAX_REPEATS, 0);
      gui = w_new(WEED_PLANT_GUI);
      w_set(fc, WEED_LEAF_GUI, WEED_SEED_PLANTPTR, 1, &gui);
      s_str(gui, WEED_LEAF_LAYOUT_SCHEME, "RFX");
      s_str(gui, "layout_rfx_delim", "|");
      w_set(gui, "layout_rfx_strings", WEED_SEED_STRING, 7, rfx);
    }
  }
  s_int(pinfo, WEED_LEAF_VERSION, 1);
  return pinfo;
}
