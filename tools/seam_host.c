/* seam_host.c -- a C "render host" that drives the two seams the way LiVES' node model drives them (bench.py's seam_chain leg, tests/test_seam_host.py).
 *
 * Per tick and track, ON ONE HOST THREAD PER TRACK (src/nodemodel.c:2027-2101 runs the plan steps of the tracks on pool threads), by the REFERENCE's names out of
 * liblivesgpu_dropin.so and the "chroma blend" class of livesgpu_fx.so:
 *     convert_layer_palette(layer, RGBA32, 0)                          pconv substep   src/nodemodel.c:1093
 *     resize_layer(layer, dw, dh, LIVES_INTERP_BEST, RGBA32, 0)        res substep     src/nodemodel.c:1187
 *     process_func(chroma blend instance: in0 = out = the layer, in1 = the track's second layer)      weed_apply_instance, src/effects-weed.c:1850-2425
 *     gamma_convert_layer(WEED_GAMMA_BT709, layer)                     gamma substep   src/nodemodel.c:1138 (the layer is SRGB after the gdk-pixbuf body)
 * and, once per tick on the collecting thread, lives_gpu_layers_flush(layers, ntracks) -- the library's one addition to the host's code (include/lives_gpu_layer.h).
 * The layers are genuine weed plants (tools/miniweed.c is the weed host here); their frames ALREADY lie in HBM (lives_gpu_layer_pin_device: the bench's resident
 * synthetic frames standing where a hardware decoder's surfaces would), as the inputs of bench.py's `value` do.  Nothing of oracle/ is linked or loaded. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include "../include/lives_gpu.h"
#include "../include/lives_gpu_layer.h"
#include "miniweed.h"

/* the reference's prototypes (src/colourspace.h:377-423), resolved from liblivesgpu_dropin.so */
typedef weed_plant_t weed_layer_t;
typedef int boolean;
boolean convert_layer_palette(weed_layer_t *, int outpl, int op_clamping);
boolean resize_layer(weed_layer_t *, int width, int height, int interp, int opal_hint, int oclamp_hint);
boolean gamma_convert_layer(int gamma_type, weed_layer_t *);

#define MAXT 64
typedef struct {
  int ntracks, sw, sh, dw, dh, bf, gamma;
  const void *src_d[MAXT], *l2_d[MAXT];
  weed_plant_t *layer[MAXT], *layer2[MAXT], *inst[MAXT], *chan[MAXT][3];
  weed_process_f process;
  /* host frame memory: one block per track and role, reused every tick (what LiVES' bigblock pool does, src/memory.c) */
  void *host_src[MAXT], *host_out[MAXT][4];
  int out_rr[MAXT];
  /* tick machinery */
  pthread_t th[MAXT];
  atomic_int go, done, quit, failed;
  int tick_of[MAXT];
} host_t;

static host_t H;
/* where the host time goes (seam_host_profile): per call kind, summed over tracks and ticks, in ns; [6] the collecting thread's wait for its tracks, [7] its flush */
static _Atomic long long g_prof[8];
static inline long long now_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 1000000000ll + t.tv_nsec; }
#define PROF(i, stmt) do { const long long t0_ = now_ns(); stmt; atomic_fetch_add_explicit(&g_prof[i], now_ns() - t0_, memory_order_relaxed); } while (0)
void seam_host_profile(double us[8], int reset) {
  for (int i = 0; i < 8; i++) { us[i] = (double)atomic_load(&g_prof[i]) / 1e3; if (reset) atomic_store(&g_prof[i], 0); }
}
static weed_plant_t *g_pinfo;
static __thread int t_track = -1;

/* the frame allocator handed to the library (lives_gpu_weed_api.pixel_alloc / pixel_free): the calling track's blocks, round robin -- no malloc, no page faults
   in the timed region (pinned layers never touch their host bytes) */
static void *pix_alloc(size_t bytes) {
  if (t_track >= 0 && bytes <= (size_t)H.sw * 4 * H.sh + 4096) { void *p = H.host_out[t_track][H.out_rr[t_track] & 3]; H.out_rr[t_track]++; return p; }
  return calloc(1, bytes ? bytes : 1);
}
static int is_pool_block(void *p) {
  for (int t = 0; t < H.ntracks; t++) { if (p == H.host_src[t]) return 1; for (int k = 0; k < 4; k++) if (p == H.host_out[t][k]) return 1; }
  return 0;
}
static void pix_free(void *p) { if (p && !is_pool_block(p)) free(p); }

static void set_i(weed_plant_t *p, const char *k, int v) { int32_t x = v; mw_leaf_set(p, k, WEED_SEED_INT, 1, &x); }
static void set_p(weed_plant_t *p, const char *k, void *v) { mw_leaf_set(p, k, WEED_SEED_VOIDPTR, 1, &v); }
static int get_i(weed_plant_t *p, const char *k) { int32_t v = 0; mw_leaf_get(p, k, 0, &v); return v; }
static void *get_p(weed_plant_t *p, const char *k) { void *v = NULL; mw_leaf_get(p, k, 0, &v); return v; }

static void fill_layer(weed_plant_t *l, int pal, int w, int h, void *host_plane) {
  set_i(l, WEED_LEAF_CURRENT_PALETTE, pal); set_i(l, WEED_LEAF_WIDTH, w); set_i(l, WEED_LEAF_HEIGHT, h);
  set_i(l, WEED_LEAF_ROWSTRIDES, w * 4); set_p(l, WEED_LEAF_PIXEL_DATA, host_plane); set_i(l, WEED_LEAF_GAMMA_TYPE, WEED_GAMMA_SRGB);
}
/* weed_apply_instance's channel set-up (src/effects-weed.c:2200-2330): the channels carry the layers' pixel_data pointers and geometry */
static void channel_from_layer(weed_plant_t *c, weed_plant_t *l) {
  set_i(c, WEED_LEAF_CURRENT_PALETTE, get_i(l, WEED_LEAF_CURRENT_PALETTE)); set_i(c, WEED_LEAF_WIDTH, get_i(l, WEED_LEAF_WIDTH));
  set_i(c, WEED_LEAF_HEIGHT, get_i(l, WEED_LEAF_HEIGHT)); set_i(c, WEED_LEAF_ROWSTRIDES, get_i(l, WEED_LEAF_ROWSTRIDES));
  set_p(c, WEED_LEAF_PIXEL_DATA, get_p(l, WEED_LEAF_PIXEL_DATA));
}

/* one track's plan step of one tick */
static int track_step(int t) {
  weed_plant_t *l = H.layer[t];
  const void *pl[1] = {H.src_d[t]};
  t_track = t;
  /* the frame source: a fresh BGRA32 layer whose frame already is in HBM */
  int ok = 1;
  PROF(0, { if (mw_leaf_num_elements(l, "host_gpu_resident")) lives_gpu_layer_forget(l);
            fill_layer(l, WEED_PALETTE_BGRA32, H.sw, H.sh, H.host_src[t]);
            ok = lives_gpu_layer_pin_device(l, pl, 1, NULL, 1) == LGPU_OK; });
  if (!ok) return 1;
  PROF(1, ok = convert_layer_palette(l, WEED_PALETTE_RGBA32, 0));
  if (!ok) return 2;
  PROF(2, ok = resize_layer(l, H.dw, H.dh, LIVES_INTERP_BEST, WEED_PALETTE_RGBA32, 0));
  if (!ok) return 3;
  PROF(3, { channel_from_layer(H.chan[t][0], l); channel_from_layer(H.chan[t][1], H.layer2[t]); channel_from_layer(H.chan[t][2], l);
            ok = H.process(H.inst[t], 0) == WEED_SUCCESS; });
  if (!ok) return 4;
  if (H.gamma) PROF(4, ok = gamma_convert_layer(H.gamma, l));
  if (!ok) return 5;
  return 0;
}

/* Pool threads SLEEP between ticks (futex), as LiVES' do (src/threading.c): a host that spins seventeen threads at 100 % starves itself wherever the process has a
   CPU quota -- measured on the pool's boxes: every seam call 50-400 x slower, tools/seam_profile.py -- and a render host has other work for its cores anyway.
   No spin first (g_spin below). */
static long futex(atomic_int *addr, int op, int val) { return syscall(SYS_futex, addr, op, val, NULL, NULL, 0); }
/* SEAM_SPIN: pause iterations before a waiter parks.  0 = park at once, the default: measured on the pool's boxes (16 CPUs by quota, 17 threads here plus the
   interpreter's) 0 -> 101-107 k frames/s, 200 -> 87-104 k, 2000 -> 65 k, 10000 -> 72-82 k, 50000 -> 61-66 k (profiles/r06/seam_spin_sweep.txt): every waiter that
   spins is a runnable thread that can push the holder of the library's table lock off its CPU */
static int g_spin = 0;
static void wait_change(atomic_int *addr, int seen) {
  for (int i = 0; i < g_spin; i++) { if (atomic_load_explicit(addr, memory_order_acquire) != seen) return; __builtin_ia32_pause(); }
  while (atomic_load_explicit(addr, memory_order_acquire) == seen) futex(addr, FUTEX_WAIT_PRIVATE, seen);
}
static void *worker(void *arg) {
  const int t = (int)(intptr_t)arg;
  int seen = 0;
  for (;;) {
    wait_change(&H.go, seen);
    if (atomic_load(&H.quit)) return NULL;
    seen = atomic_load_explicit(&H.go, memory_order_acquire);
    const int rc = track_step(t);
    if (rc) atomic_store(&H.failed, rc * 100 + t);
    if (atomic_fetch_add_explicit(&H.done, 1, memory_order_acq_rel) + 1 == H.ntracks) futex(&H.done, FUTEX_WAKE_PRIVATE, 1);      /* the last one in wakes the collector */
  }
}

static weed_plant_t *find_filter(const char *name) {
  const weed_size_t n = mw_leaf_num_elements(g_pinfo, WEED_LEAF_FILTERS);
  for (weed_size_t i = 0; i < n; i++) {
    weed_plant_t *f = NULL;
    mw_leaf_get(g_pinfo, WEED_LEAF_FILTERS, i, &f);
    const char *s = f ? mw_string(f, WEED_LEAF_NAME) : NULL;
    if (s && !strcmp(s, name)) return f;
  }
  return NULL;
}

/* bind the layer seam to this host's plants, load the plugin through weed_setup(weed_bootstrap) as load_weed_plugin does (src/effects-weed.c:4468-4568) */
int seam_host_init(const char *fx_so_path) {
  if (getenv("SEAM_SPIN")) g_spin = atoi(getenv("SEAM_SPIN"));
  lives_gpu_weed_api api = {mw_leaf_get, mw_leaf_set, mw_leaf_num_elements, mw_leaf_delete, pix_alloc, pix_free};
  if (lives_gpu_bind_weed(&api) != 0) return -1;          /* every time: a test process may have bound another weed host in between */
  if (g_pinfo) return 0;
  void *h = dlopen(fx_so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "seam_host: %s\n", dlerror()); return -2; }
  weed_plant_t *(*setup)(weed_bootstrap_f) = (weed_plant_t *(*)(weed_bootstrap_f))dlsym(h, "weed_setup");
  if (!setup) return -3;
  g_pinfo = setup(mw_bootstrap);
  return g_pinfo ? 0 : -4;
}

/* ntracks tracks of sw x sh BGRA32 frames at src_d[] and dw x dh RGBA32 second layers at l2_d[] (device memory); `ticks` timed ticks after `warm` untimed ones.
   threads != 0: one host thread per track; 0: the calling thread walks the tracks.  *ms_total: wall-clock from the first timed tick to the completion of the last
   launch.  out_d (optional): the device planes of the LAST tick's result layers, track by track (valid until the next call).  Returns 0, or the failed step. */
int seam_host_run(int ntracks, int sw, int sh, int dw, int dh, const void *const *src_d, const void *const *l2_d, int bf, int gamma, int ticks, int warm, int threads,
                  double *ms_total, void **out_d, int *out_row) {
  if (!g_pinfo || ntracks < 1 || ntracks > MAXT) return -1;
  weed_plant_t *filt = find_filter("chroma blend");
  if (!filt) return -2;
  weed_funcptr_t fp = NULL;
  mw_leaf_get(filt, WEED_LEAF_PROCESS_FUNC, 0, &fp);
  weed_init_f init = NULL;
  { weed_funcptr_t ip = NULL; mw_leaf_get(filt, WEED_LEAF_INIT_FUNC, 0, &ip); init = (weed_init_f)ip; }
  memset(&H, 0, sizeof H);
  H.process = (weed_process_f)fp;
  H.ntracks = ntracks; H.sw = sw; H.sh = sh; H.dw = dw; H.dh = dh; H.bf = bf; H.gamma = gamma;
  weed_plant_t *ptmpl = NULL;
  mw_leaf_get(filt, WEED_LEAF_IN_PARAMETER_TEMPLATES, 0, &ptmpl);
  for (int t = 0; t < ntracks; t++) {
    H.src_d[t] = src_d[t]; H.l2_d[t] = l2_d[t];
    H.host_src[t] = malloc((size_t)sw * 4 * sh + 4096);
    for (int k = 0; k < 4; k++) H.host_out[t][k] = malloc((size_t)sw * 4 * sh + 4096);
    H.layer[t] = mw_plant_new(128);                      /* WEED_PLANT_LAYER (src/layers.h:14) */
    H.layer2[t] = mw_plant_new(128);
    fill_layer(H.layer2[t], WEED_PALETTE_RGBA32, dw, dh, malloc((size_t)dw * 4 * dh + 4096));
    const void *pl[1] = {l2_d[t]};
    if (lives_gpu_layer_pin_device(H.layer2[t], pl, 1, NULL, 1) != LGPU_OK) return -3;
    /* weed_instance_from_filter: instance, channels, one parameter */
    weed_plant_t *inst = mw_plant_new(WEED_PLANT_FILTER_INSTANCE), *par = mw_plant_new(WEED_PLANT_PARAMETER);
    for (int k = 0; k < 3; k++) H.chan[t][k] = mw_plant_new(WEED_PLANT_CHANNEL);
    mw_leaf_set(inst, WEED_LEAF_FILTER_CLASS, WEED_SEED_PLANTPTR, 1, &filt);
    mw_leaf_set(inst, WEED_LEAF_IN_CHANNELS, WEED_SEED_PLANTPTR, 2, H.chan[t]);
    mw_leaf_set(inst, WEED_LEAF_OUT_CHANNELS, WEED_SEED_PLANTPTR, 1, &H.chan[t][2]);
    mw_leaf_set(par, WEED_LEAF_TEMPLATE, WEED_SEED_PLANTPTR, 1, &ptmpl);
    set_i(par, WEED_LEAF_VALUE, bf);
    mw_leaf_set(inst, WEED_LEAF_IN_PARAMETERS, WEED_SEED_PLANTPTR, 1, &par);
    if (init && init(inst) != WEED_SUCCESS) return -4;
    H.inst[t] = inst;
  }
  if (threads) for (int t = 0; t < ntracks; t++) pthread_create(&H.th[t], NULL, worker, (void *)(intptr_t)t);
  struct timespec t0, t1;
  int rc = 0;
  for (int tick = 0; tick < warm + ticks && !rc; tick++) {
    if (tick == warm) { lgpu_sync(lives_gpu_thread_stream()); for (int i = 0; i < 8; i++) atomic_store(&g_prof[i], 0); clock_gettime(CLOCK_MONOTONIC, &t0); }
    if (threads) {
      atomic_store(&H.done, 0);
      atomic_fetch_add_explicit(&H.go, 1, memory_order_release);
      futex(&H.go, FUTEX_WAKE_PRIVATE, ntracks);
      PROF(6, { int d; while ((d = atomic_load_explicit(&H.done, memory_order_acquire)) < ntracks) wait_change(&H.done, d); });
      rc = atomic_load(&H.failed);
    } else
      for (int t = 0; t < ntracks && !rc; t++) { rc = track_step(t); if (rc) rc = rc * 100 + t; }
    t_track = -1;
    if (!rc) PROF(7, { if (lives_gpu_layers_flush(H.layer, ntracks) != LGPU_OK) rc = 9900; });
  }
  lgpu_sync(lives_gpu_thread_stream());
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (ms_total) *ms_total = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6;
  if (threads) {
    atomic_store(&H.quit, 1);
    atomic_fetch_add(&H.go, 1);
    futex(&H.go, FUTEX_WAKE_PRIVATE, ntracks);
    for (int t = 0; t < ntracks; t++) pthread_join(H.th[t], NULL);
  }
  if (!rc && out_d)
    for (int t = 0; t < ntracks; t++) {
      void *hp = get_p(H.layer[t], WEED_LEAF_PIXEL_DATA);
      out_d[t] = lives_gpu_resident_lookup(hp, 1);
      if (out_row) *out_row = get_i(H.layer[t], WEED_LEAF_ROWSTRIDES);
    }
  return rc;
}
/* the result layers' leaves after the last tick (tests): palette, width, height, gamma of track t */
int seam_host_layer_info(int t, int info[4]) {
  if (t < 0 || t >= H.ntracks || !H.layer[t]) return -1;
  info[0] = get_i(H.layer[t], WEED_LEAF_CURRENT_PALETTE); info[1] = get_i(H.layer[t], WEED_LEAF_WIDTH); info[2] = get_i(H.layer[t], WEED_LEAF_HEIGHT);
  info[3] = get_i(H.layer[t], WEED_LEAF_GAMMA_TYPE);
  return 0;
}
/* release what the last run holds (device entries, host blocks) */
void seam_host_release(void) {
  for (int t = 0; t < H.ntracks; t++) {
    if (H.layer[t]) lives_gpu_layer_forget(H.layer[t]);
    if (H.layer2[t]) { lives_gpu_layer_forget(H.layer2[t]); free(get_p(H.layer2[t], WEED_LEAF_PIXEL_DATA)); }
    free(H.host_src[t]);
    for (int k = 0; k < 4; k++) free(H.host_out[t][k]);
  }
  memset(&H, 0, sizeof H);
}
