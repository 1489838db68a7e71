// layer_seam.cpp -- the weed_layer_t seam (include/lives_gpu_layer.h): host-side logic of
//   convert_layer_palette_full   src/colourspace.c:12190-13928
//   gamma_convert_sub_layer      src/colourspace.c:14069-14143
//   alpha_premult                src/colourspace.c:11968-12105
//   resize_layer_full            src/colourspace.c:14759-15328
//   letterbox_layer              src/colourspace.c:15343-15567
//   create_empty_pixel_data      src/colourspace.c:11434-11700
//   calc_rowstrides              src/colourspace.c:11252-11366
// re-written around the frame-level C ABI (lives_gpu.h).  All pixel work happens in the HIP kernels; this
// file only reads / writes leaves through the accessors the host bound, moves planes over PCIe and keeps the
// reference's bookkeeping (palette, size, rowstrides, gamma, premult flag, YUV leaves, failure = untouched layer).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/lives_gpu.h"
#include "../../include/lives_gpu_layer.h"

namespace {

lives_gpu_weed_api g_api = {};
weed_leaf_get_flags_f g_leaf_get_flags = nullptr;      // optional, lives_gpu_bind_leaf_get_flags
lives_gpu_prefs g_prefs = {1, 0, 2, 1.4, 0};

constexpr const char *kLeafHostFlags = "host_flags";          // LIVES_LEAF_HOST_FLAGS (src/colourspace.h:37)
constexpr const char *kLeafContiguous = "host_contiguous";    // LIVES_LEAF_PIXEL_DATA_CONTIGUOUS (:33)
constexpr const char *kLeafResident = "host_gpu_resident";    // this library's private leaf (host_* convention, src/effects-weed.h:80-119)

bool bound() { return g_api.leaf_get && g_api.leaf_set && g_api.leaf_num_elements && g_api.leaf_delete; }
void *palloc(size_t n) { return g_api.pixel_alloc ? g_api.pixel_alloc(n) : calloc(1, n ? n : 1); }
void pfree(void *p) { if (!p) return; if (g_api.pixel_free) g_api.pixel_free(p); else free(p); }

bool has_leaf(weed_plant_t *p, const char *k) { return g_api.leaf_num_elements(p, k) > 0; }
int get_int(weed_plant_t *p, const char *k, int dflt, int idx = 0) {
  int32_t v = dflt;
  if (g_api.leaf_get(p, k, (weed_size_t)idx, &v) != WEED_SUCCESS) return dflt;
  return v;
}
void set_int(weed_plant_t *p, const char *k, int v) { int32_t x = v; g_api.leaf_set(p, k, WEED_SEED_INT, 1, &x); }
void set_bool(weed_plant_t *p, const char *k, int v) { int32_t x = v; g_api.leaf_set(p, k, WEED_SEED_BOOLEAN, 1, &x); }

bool pal_is_rgb(int p) { return p >= WEED_PALETTE_RGB24 && p <= WEED_PALETTE_ARGB32; }
bool pal_is_planar_yuv(int p) { return p == WEED_PALETTE_YUV420P || p == WEED_PALETTE_YVU420P || p == WEED_PALETTE_YUV422P || p == WEED_PALETTE_YUV444P || p == WEED_PALETTE_YUVA4444P; }
bool pal_is_444(int p) { return p == WEED_PALETTE_YUV444P || p == WEED_PALETTE_YUVA4444P; }
bool pal_alpha_first(int p) { return p == WEED_PALETTE_ARGB32; }
bool pal_alpha_last(int p) { return p == WEED_PALETTE_RGBA32 || p == WEED_PALETTE_BGRA32; }
bool pal_has_alpha(int p) { return pal_alpha_first(p) || pal_alpha_last(p) || p == WEED_PALETTE_YUVA8888 || p == WEED_PALETTE_YUVA4444P; }
bool pal_red_first(int p) { return p == WEED_PALETTE_RGB24 || p == WEED_PALETTE_RGBA32 || p == WEED_PALETTE_ARGB32; }
int pal_psize(int p) { return (p == WEED_PALETTE_RGB24 || p == WEED_PALETTE_BGR24) ? 3 : pal_is_rgb(p) ? 4 : pal_is_planar_yuv(p) ? 1 : 0; }

// ---- a layer as this file sees it ------------------------------------------------------------------------------
struct Layer {
  weed_plant_t *plant;
  int pal, width, height, nplanes;
  int rs[4];
  uint8_t *pd[4];
  int clamping, subspace, sampling, gamma, flags;
  bool contiguous;
};

int plane_w(const Layer &l, int p) { return (p == 0 || pal_is_444(l.pal)) ? l.width : l.width >> 1; }
int plane_h(const Layer &l, int p) { return (p == 0 || pal_is_444(l.pal) || l.pal == WEED_PALETTE_YUV422P) ? l.height : l.height >> 1; }

bool read_layer(weed_plant_t *plant, Layer *l) {
  if (!plant || !bound()) return false;
  l->plant = plant;
  l->pal = get_int(plant, WEED_LEAF_CURRENT_PALETTE, 0);
  l->width = get_int(plant, WEED_LEAF_WIDTH, 0);
  l->height = get_int(plant, WEED_LEAF_HEIGHT, 0);
  l->nplanes = (int)g_api.leaf_num_elements(plant, WEED_LEAF_PIXEL_DATA);
  if (l->nplanes < 1 || l->nplanes > 4 || (int)g_api.leaf_num_elements(plant, WEED_LEAF_ROWSTRIDES) < l->nplanes) return false;
  for (int i = 0; i < 4; i++) { l->rs[i] = 0; l->pd[i] = nullptr; }
  for (int i = 0; i < l->nplanes; i++) {
    l->rs[i] = get_int(plant, WEED_LEAF_ROWSTRIDES, 0, i);
    void *v = nullptr;
    g_api.leaf_get(plant, WEED_LEAF_PIXEL_DATA, (weed_size_t)i, &v);
    l->pd[i] = (uint8_t *)v;
  }
  if (!l->pd[0] || l->width <= 0 || l->height <= 0) return false;
  l->clamping = get_int(plant, WEED_LEAF_YUV_CLAMPING, -1);
  l->subspace = get_int(plant, WEED_LEAF_YUV_SUBSPACE, WEED_YUV_SUBSPACE_YUV);
  l->sampling = get_int(plant, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);
  l->gamma = get_int(plant, WEED_LEAF_GAMMA_TYPE, WEED_GAMMA_UNKNOWN);
  l->flags = get_int(plant, kLeafHostFlags, 0);
  l->contiguous = get_int(plant, kLeafContiguous, 0) != 0;
  return true;
}

// ---- device residency of pinned layers ---------------------------------------------------------------------------------
// A layer the host pinned (lives_gpu_layer_pin) keeps the authoritative copy of its planes in HBM, keyed by the HOST plane
// pointer: uploads of such a plane become device-to-device copies, downloads replace the device copy and leave the host
// bytes stale until lives_gpu_layer_sync().  The CONVERT chain of one plan step (pconv -> gamma -> resize -> letterbox) then
// crosses PCIe once in each direction instead of eight times.
struct ResEntry { void *d; size_t bytes; };
std::mutex g_res_mu;
std::unordered_map<const void *, ResEntry> g_res;
unsigned long long g_h2d = 0, g_d2h = 0;        // PCIe byte counters (tests check the residency contract with them)
thread_local bool t_pinned = false;             // the call in progress works on a pinned layer

// device buffers of dropped planes are recycled (hipMalloc / hipFree cost ~0.1 ms each and synchronise the device): a pinned layer going
// through a chain of seam calls allocates a new plane per call.  Callers hold g_res_mu.  A pooled buffer is only handed out again to work that is
// enqueued after the work that last used it (everything here runs on the null stream of the calling thread's device), so stream order protects it.
struct PoolEntry { void *d; size_t cap; };
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = 1u << 30, kPoolMaxEntries = 32;
void *pool_take(size_t n, size_t *cap_out) {
  int best = -1;
  for (int i = 0; i < (int)g_pool.size(); i++)
    if (g_pool[i].cap >= n && g_pool[i].cap <= 2 * n + (1u << 20) && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
  if (best >= 0) {
    void *d = g_pool[best].d;
    *cap_out = g_pool[best].cap;
    g_pool_bytes -= g_pool[best].cap;
    g_pool.erase(g_pool.begin() + best);
    return d;
  }
  void *d = nullptr;
  if (lgpu_malloc(&d, n + 64) != LGPU_OK) return nullptr;
  *cap_out = n;
  return d;
}
void pool_give(void *d, size_t cap) {
  if (!d) return;
  if (g_pool.size() >= kPoolMaxEntries || g_pool_bytes + cap > kPoolMaxBytes) { lgpu_free(d); return; }
  g_pool.push_back({d, cap});
  g_pool_bytes += cap;
}

void res_drop(const void *h) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_res.find(h);
  if (it == g_res.end()) return;
  pool_give(it->second.d, it->second.bytes);
  g_res.erase(it);
}
struct PinScope {
  bool prev;
  explicit PinScope(weed_plant_t *layer) : prev(t_pinned) { t_pinned = layer && bound() && has_leaf(layer, kLeafResident); }
  ~PinScope() { t_pinned = prev; }
};

void free_planes(const Layer &l) {
  for (int i = 0; i < l.nplanes; i++) res_drop(l.pd[i]);
  if (l.contiguous) pfree(l.pd[0]);
  else for (int i = 0; i < l.nplanes; i++) pfree(l.pd[i]);
}

// fixed rowstrides (:11268-11275): a "new_rowstrides" leaf, or rowstrides flagged LIVES_FLAG_CONST_VALUE (1 << 16, src/main.h:127); taken per
// plane when computed <= fixed < 2 * computed (the reference's integer `constrs[i] / rs[i]` between .75 and 1.25, :11358-11363)
void apply_const_rowstrides(weed_plant_t *layer, int n, int *rs) {
  if (!layer || !bound()) return;
  const char *key = nullptr;
  if (has_leaf(layer, "new_rowstrides")) key = WEED_LEAF_ROWSTRIDES;        // the reference reads the rowstrides leaf in both cases (:11269, :11273)
  else if (g_leaf_get_flags && has_leaf(layer, WEED_LEAF_ROWSTRIDES) && (g_leaf_get_flags(layer, WEED_LEAF_ROWSTRIDES) & (1 << 16))) key = WEED_LEAF_ROWSTRIDES;
  if (!key) return;
  const int have = (int)g_api.leaf_num_elements(layer, key);
  for (int i = 0; i < n && i < have; i++) {
    int32_t c = 0;
    if (g_api.leaf_get(layer, key, (weed_size_t)i, &c) != WEED_SUCCESS || rs[i] <= 0) continue;
    const int q = c / rs[i];
    if (q >= .75 && q <= 1.25) rs[i] = c;
  }
}

// new host planes for (pal, width, height) with the reference's rowstride rule; one block for planar ("contiguous")
struct NewPlanes { int n; int rs[4]; uint8_t *pd[4]; size_t sz[4]; };
bool alloc_planes(int pal, int width, int height, int alignment, NewPlanes *np, weed_plant_t *fixed_from = nullptr) {
  np->n = lgpu_calc_rowstrides(width, pal, alignment, np->rs);
  if (np->n < 1) return false;
  apply_const_rowstrides(fixed_from, np->n, np->rs);
  size_t tot = 0;
  for (int i = 0; i < np->n; i++) {
    const int h = (i == 0 || pal_is_444(pal) || pal == WEED_PALETTE_YUV422P) ? height : height >> 1;
    np->sz[i] = (size_t)np->rs[i] * h;
    tot += np->sz[i];
  }
  uint8_t *blk = (uint8_t *)palloc(tot + 64);     // + EXTRA_BYTES-style slack (reference loops read a few bytes past the end)
  if (!blk) return false;
  size_t off = 0;
  for (int i = 0; i < np->n; i++) { np->pd[i] = blk + off; off += np->sz[i]; }
  return true;
}
void commit_planes(weed_plant_t *plant, int pal, int width, int height, const NewPlanes &np) {
  set_int(plant, WEED_LEAF_CURRENT_PALETTE, pal);
  set_int(plant, WEED_LEAF_WIDTH, width);
  set_int(plant, WEED_LEAF_HEIGHT, height);
  int32_t rs[4];
  void *pd[4];
  for (int i = 0; i < np.n; i++) { rs[i] = np.rs[i]; pd[i] = np.pd[i]; }
  g_api.leaf_set(plant, WEED_LEAF_ROWSTRIDES, WEED_SEED_INT, (weed_size_t)np.n, rs);
  g_api.leaf_set(plant, WEED_LEAF_PIXEL_DATA, WEED_SEED_VOIDPTR, (weed_size_t)np.n, pd);
  if (np.n > 1) set_bool(plant, kLeafContiguous, WEED_TRUE); else g_api.leaf_delete(plant, kLeafContiguous);
}

// ---- device scratch (per calling thread; grown on demand) ------------------------------------------------------------
struct Scratch {
  void *p[8] = {nullptr};
  size_t cap[8] = {0};
  ~Scratch() { for (auto q : p) if (q) lgpu_free(q); }
  uint8_t *get(int i, size_t bytes) {
    if (cap[i] < bytes) {
      if (p[i]) lgpu_free(p[i]);
      p[i] = nullptr; cap[i] = 0;
      if (lgpu_malloc(&p[i], bytes + 64) != LGPU_OK) return nullptr;
      cap[i] = bytes;
    }
    return (uint8_t *)p[i];
  }
};
thread_local Scratch t_scr;

bool ready() { return bound() && lgpu_init(g_prefs.device) == LGPU_OK; }
bool up(uint8_t *d, const uint8_t *h, size_t n) {
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    auto it = g_res.find(h);
    if (it != g_res.end() && it->second.bytes >= n) return lgpu_copy(d, it->second.d, n, nullptr) == LGPU_OK;   // resident plane: stays in HBM
  }
  g_h2d += n;
  return lgpu_upload(d, h, n, nullptr) == LGPU_OK;
}
// a freshly allocated (zeroed) host plane that a kernel is about to fill: nothing worth sending for a pinned layer
bool up_fresh(uint8_t *d, const uint8_t *h, size_t n) {
  if (t_pinned) return lgpu_fill(d, 0, n, nullptr) == LGPU_OK;
  return up(d, h, n);
}
bool down(uint8_t *h, const uint8_t *d, size_t n) {
  if (t_pinned) {                       // the device copy becomes the plane; the host bytes go stale until lives_gpu_layer_sync()
    std::lock_guard<std::mutex> lk(g_res_mu);
    ResEntry &e = g_res[h];
    if (e.bytes < n) {
      pool_give(e.d, e.bytes);
      e.d = nullptr; e.bytes = 0;
      size_t cap = 0;
      e.d = pool_take(n, &cap);
      if (!e.d) { g_res.erase(h); return false; }
      e.bytes = cap;
    }
    return lgpu_copy(e.d, d, n, nullptr) == LGPU_OK;
  }
  g_d2h += n;
  return lgpu_download(h, d, n, nullptr) == LGPU_OK;
}
bool sync() { return lgpu_sync(nullptr) == LGPU_OK; }

int rgb_swizzle_op(int inpl, int outpl, int *alpha_first_arg) {
  // the selector tree of src/colourspace.c:12370-12556
  const bool swap = pal_red_first(inpl) != pal_red_first(outpl);
  *alpha_first_arg = 0;
  if (!pal_alpha_first(inpl)) {
    if (!pal_alpha_last(inpl)) {                        // RGB24 / BGR24 in
      if (!pal_alpha_first(outpl)) {
        if (!pal_alpha_last(outpl)) return LGPU_SWAP3;
        return swap ? LGPU_SWAP3ADDPOST : LGPU_ADDPOST;
      }
      return swap ? LGPU_SWAP3ADDPRE : LGPU_ADDPRE;     // -> ARGB
    }
    if (!pal_alpha_first(outpl)) {                      // RGBA / BGRA in
      if (!pal_alpha_last(outpl)) return swap ? LGPU_SWAP3DELPOST : LGPU_DELPOST;
      return LGPU_SWAP3POSTALPHA;
    }
    return swap ? LGPU_SWAP4 : LGPU_SWAPPREPOST;        // -> ARGB (alpha_first = FALSE)
  }
  *alpha_first_arg = 1;                                 // ARGB in
  if (!pal_alpha_first(outpl)) {
    if (!pal_alpha_last(outpl)) return swap ? LGPU_SWAP3DELPRE : LGPU_DELPRE;
    return swap ? LGPU_SWAP4 : LGPU_SWAPPREPOST;
  }
  return LGPU_SWAP3PREALPHA;
}

// K5 on a layer: switch_yuv_clamping_and_subspace (:10929-11090), in place on the layer's own planes
bool switch_layer_clamping(weed_plant_t *layer, const Layer &l, int oclamping) {
  uint8_t *d[4] = {nullptr, nullptr, nullptr, nullptr};
  int rs[4] = {0, 0, 0, 0};
  size_t bytes[4] = {0, 0, 0, 0};
  const bool planar = pal_is_planar_yuv(l.pal);
  bool ok = true;
  for (int p = 0; p < l.nplanes && ok; p++) {
    const int ph = (!planar || p == 0 || p == 3 || pal_is_444(l.pal) || l.pal == WEED_PALETTE_YUV422P) ? l.height : l.height >> 1;
    bytes[p] = (size_t)l.rs[p] * ph;
    d[p] = t_scr.get(p == 0 ? 0 : p + 3, bytes[p]);
    rs[p] = l.rs[p];
    ok = d[p] && up(d[p], l.pd[p], bytes[p]);
  }
  ok = ok && lgpu_yuv_switch_clamping(d, rs, l.pal, l.height, oclamping == WEED_YUV_CLAMPING_UNCLAMPED, nullptr) == LGPU_OK;
  for (int p = 0; p < l.nplanes && ok; p++) ok = down(l.pd[p], d[p], bytes[p]);
  ok = ok && sync();
  if (ok) set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);
  return ok;
}

int k3_fmt(int pal) {
  switch (pal) {
  case WEED_PALETTE_YUV888: case WEED_PALETTE_YUVA8888: return 0;
  case WEED_PALETTE_YUV444P: case WEED_PALETTE_YUVA4444P: return 1;
  case WEED_PALETTE_UYVY: return 2;
  case WEED_PALETTE_YUYV: return 3;
  default: return -1;
  }
}
int k4_fmt(int pal) {
  switch (pal) {
  case WEED_PALETTE_YUV420P: case WEED_PALETTE_YVU420P: return 4;
  case WEED_PALETTE_YUV422P: return 5;
  default: return k3_fmt(pal);
  }
}

// K4 on a layer: the RGB24 / BGR24 / RGBA32 / BGRA32 / ARGB32 cases of src/colourspace.c:12559-12935 plus conv_done (:13860-13893)
// K4b on a layer (:12627-12632 and the same case under each RGB input): width leaf becomes width >> 2 macropixels, the new frame is
// written as compact macropixel rows whatever rowstride it was given (the reference passes none)
lives_gpu_boolean rgb_layer_to_yuv411(weed_plant_t *layer, const Layer &l, int oclamping) {
  const int order = pal_alpha_first(l.pal) ? 2 : pal_red_first(l.pal) ? 0 : 1;
  const int in_alpha = pal_has_alpha(l.pal) ? 1 : 0, wm = l.width >> 2;
  if (wm < 1 || l.height < 1) return 0;
  NewPlanes np;
  if (!alloc_planes(WEED_PALETTE_YUV411, wm, l.height, 0, &np)) return 0;
  const size_t ibytes = (size_t)l.rs[0] * l.height;
  uint8_t *d_in = t_scr.get(0, ibytes), *d_out = t_scr.get(3, np.sz[0]);
  const bool ok = d_in && d_out && up(d_in, l.pd[0], ibytes) && up_fresh(d_out, np.pd[0], np.sz[0]) &&
                  lgpu_rgb_to_yuv411(d_in, l.rs[0], l.width, l.height, order, in_alpha, d_out, oclamping == WEED_YUV_CLAMPING_UNCLAMPED, nullptr) == LGPU_OK &&
                  down(np.pd[0], d_out, np.sz[0]) && sync();
  if (!ok) { res_drop(np.pd[0]); pfree(np.pd[0]); return 0; }
  int flags = l.flags;
  if (in_alpha) flags &= ~LIVES_LAYER_ALPHA_PREMULT;
  free_planes(l);
  commit_planes(layer, WEED_PALETTE_YUV411, wm, l.height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);                         // conv_done, as for the other RGB -> YUV cases below
  set_int(layer, WEED_LEAF_YUV_SUBSPACE, l.gamma == WEED_GAMMA_BT709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR);
  if (!has_leaf(layer, WEED_LEAF_YUV_SAMPLING)) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);
  return 1;
}

lives_gpu_boolean rgb_layer_to_yuv(weed_plant_t *layer, const Layer &l, int outpl, int oclamping, int osubspace, int tgt_gamma) {
  if (outpl == WEED_PALETTE_YUV411) {
    if (g_prefs.apply_gamma && l.gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != l.gamma) return 0;
    return rgb_layer_to_yuv411(layer, l, oclamping);
  }
  const int fmt = k4_fmt(outpl);
  if (fmt < 0) return 0;
  if (g_prefs.apply_gamma && l.gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != l.gamma) return 0;   // LUT16 variants: CPU body
  const int order = pal_alpha_first(l.pal) ? 2 : pal_red_first(l.pal) ? 0 : 1;
  if (order == 2 && fmt >= 4) return 0;                                     // reference-broken (:6353), declined
  const int in_alpha = pal_has_alpha(l.pal) ? 1 : 0, out_alpha = pal_has_alpha(outpl) ? 1 : 0;
  int width = l.width, height = l.height;
  if (fmt >= 2 && (width & 1)) return 0;
  if (fmt == 4) { width = (width >> 1) << 1; height = (height >> 1) << 1; }       // create_empty_pixel_data :11601-11603
  if (width < 2 || height < 1) return 0;
  // subspace argument as the dispatcher passes it: some cases hand WEED_YUV_SAMPLING_DEFAULT (= 0 -> YCbCr) to the subspace slot
  const bool use_osub = (fmt == 4 && l.pal != WEED_PALETTE_RGB24) || (fmt == 5 && l.pal == WEED_PALETTE_RGB24);
  const int which = (oclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | ((fmt >= 4 && use_osub && osubspace == WEED_YUV_SUBSPACE_BT709) ? 2 : 0);
  const int lwidth = (fmt == 2 || fmt == 3) ? width >> 1 : width;                   // UYVY / YUYV layers count macropixels
  NewPlanes np;
  if (!alloc_planes(outpl, lwidth, height, 0, &np)) return 0;
  const size_t ibytes = (size_t)l.rs[0] * l.height;
  uint8_t *d_in = t_scr.get(0, ibytes);
  bool ok = d_in && up(d_in, l.pd[0], ibytes);
  uint8_t *ddst[4] = {nullptr, nullptr, nullptr, nullptr};
  int ors[4] = {0, 0, 0, 0};
  for (int p = 0; p < np.n && ok; p++) {
    ddst[p] = t_scr.get(3 + p, np.sz[p]);
    ors[p] = np.rs[p];
    ok = ddst[p] && up_fresh(ddst[p], np.pd[p], np.sz[p]);     // calloc'd padding stays as the host made it
  }
  ok = ok && lgpu_rgb_to_yuv(d_in, l.rs[0], width, height, order, in_alpha, ddst, ors, fmt, out_alpha, which, nullptr) == LGPU_OK;
  for (int p = 0; p < np.n && ok; p++) ok = down(np.pd[p], ddst[p], np.sz[p]);
  ok = ok && sync();
  if (!ok) { for (int q = 0; q < np.n; q++) res_drop(np.pd[q]); pfree(np.pd[0]); return 0; }
  int flags = l.flags;
  if (in_alpha && !out_alpha) flags &= ~LIVES_LAYER_ALPHA_PREMULT;
  free_planes(l);
  if (outpl == WEED_PALETTE_YVU420P) { uint8_t *t = np.pd[1]; np.pd[1] = np.pd[2]; np.pd[2] = t; }   // swap_chroma_planes (:13890)
  commit_planes(layer, outpl, lwidth, height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);
  set_int(layer, WEED_LEAF_YUV_SUBSPACE, l.gamma == WEED_GAMMA_BT709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR);
  if (fmt >= 4 || !has_leaf(layer, WEED_LEAF_YUV_SAMPLING)) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);
  return 1;
}

}  // namespace

extern "C" {

int lives_gpu_bind_weed(const lives_gpu_weed_api *api) {
  if (!api || !api->leaf_get || !api->leaf_set || !api->leaf_num_elements || !api->leaf_delete) return LGPU_E_BADARG;
  g_api = *api;
  return LGPU_OK;
}

int lives_gpu_bind_leaf_get_flags(weed_leaf_get_flags_f leaf_get_flags) {
  g_leaf_get_flags = leaf_get_flags;
  return LGPU_OK;
}

int lives_gpu_set_prefs(const lives_gpu_prefs *prefs) {
  if (!prefs) return LGPU_E_BADARG;
  g_prefs = *prefs;
  return LGPU_OK;
}

int *lives_gpu_calc_rowstrides(int width, int pal, lives_gpu_layer_t *layer, int *nplanes) {
  if (pal == WEED_PALETTE_NONE) { if (!layer || !bound()) return nullptr; pal = get_int(layer, WEED_LEAF_CURRENT_PALETTE, 0); }
  if (!width) { if (!layer || !bound()) return nullptr; width = get_int(layer, WEED_LEAF_WIDTH, 0); }
  int rs[4];
  const int n = lgpu_calc_rowstrides(width, pal, 0, rs);
  if (nplanes) *nplanes = n;
  if (!n) return nullptr;
  apply_const_rowstrides(layer, n, rs);
  int *out = (int *)calloc((size_t)n, sizeof(int));      // caller frees with lives_free, like the reference
  for (int i = 0; i < n; i++) out[i] = rs[i];
  return out;
}

// YUV -> YUV repack of a layer (:12937-13750, the non-RGB half of the dispatcher) through lgpu_yuv_repack; 0 = not taken, the
// caller's CPU body runs (pairs / layouts listed in include/lives_gpu.h)
lives_gpu_boolean yuv_layer_repack(weed_plant_t *layer, const Layer &l, int outpl, int iclamping) {
  const int inpl = l.pal;
  const bool inpk = (inpl == WEED_PALETTE_UYVY || inpl == WEED_PALETTE_YUYV), outpk = (outpl == WEED_PALETTE_UYVY || outpl == WEED_PALETTE_YUYV);
  const int width = inpk ? l.width * 2 : l.width, height = l.height;              // pixels
  if (width < 1 || height < 1) return 0;
  const int unclamped = iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0;
  const uint8_t *dsrc[4] = {nullptr, nullptr, nullptr, nullptr};
  uint8_t *ddst[4] = {nullptr, nullptr, nullptr, nullptr};
  int irs[4] = {0, 0, 0, 0}, ors[4] = {0, 0, 0, 0};
  bool ok = true;
  for (int p = 0; p < l.nplanes && ok; p++) {
    const size_t b = (size_t)l.rs[p] * plane_h(l, p);
    uint8_t *d = t_scr.get(p == 0 ? 0 : p, b);     // slots 0..2 (+ 7 for a fourth plane)
    if (p == 3) d = t_scr.get(7, b);
    ok = d && up(d, l.pd[p], b);
    dsrc[p] = d; irs[p] = l.rs[p];
  }
  if (!ok) return 0;
  if (inpl == WEED_PALETTE_YVU420P) { const uint8_t *t = dsrc[1]; dsrc[1] = dsrc[2]; dsrc[2] = t; const int r = irs[1]; irs[1] = irs[2]; irs[2] = r; }
  if (inpk && outpk) {
    // convert_swab_frame (:13139): in place, the layer keeps its pixel data
    uint8_t *dd[4] = {const_cast<uint8_t *>(dsrc[0]), nullptr, nullptr, nullptr};
    ok = lgpu_yuv_repack(inpl, outpl, dsrc, irs, dd, irs, width, height, unclamped, 0, nullptr) == LGPU_OK &&
         down(l.pd[0], dd[0], (size_t)l.rs[0] * height) && sync();
    if (!ok) return 0;
    set_int(layer, WEED_LEAF_CURRENT_PALETTE, outpl);
    return 1;
  }
  const int lwidth = outpk ? width >> 1 : width;
  NewPlanes np;
  if (!alloc_planes(outpl, lwidth, height, 0, &np)) return 0;
  for (int p = 0; p < np.n && ok; p++) {
    ddst[p] = t_scr.get(3 + p, np.sz[p]);
    ors[p] = np.rs[p];
    ok = ddst[p] && up_fresh(ddst[p], np.pd[p], np.sz[p]);
  }
  ok = ok && lgpu_yuv_repack(inpl, outpl, dsrc, irs, ddst, ors, width, height, unclamped, 0, nullptr) == LGPU_OK;
  for (int p = 0; p < np.n && ok; p++) ok = down(np.pd[p], ddst[p], np.sz[p]);
  ok = ok && sync();
  if (!ok) { for (int q = 0; q < np.n; q++) res_drop(np.pd[q]); pfree(np.pd[0]); return 0; }
  int flags = l.flags;
  if (pal_has_alpha(inpl) && !pal_has_alpha(outpl)) flags &= ~LIVES_LAYER_ALPHA_PREMULT;
  free_planes(l);
  if (outpl == WEED_PALETTE_YVU420P) { uint8_t *t = np.pd[1]; np.pd[1] = np.pd[2]; np.pd[2] = t; }   // swap_chroma_planes (:13890)
  commit_planes(layer, outpl, lwidth, height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  if (outpl == WEED_PALETTE_YUV420P || outpl == WEED_PALETTE_YVU420P) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);   // :13022
  return 1;
}

lives_gpu_boolean lives_gpu_create_empty_pixel_data(lives_gpu_layer_t *layer, lives_gpu_boolean black_fill, lives_gpu_boolean may_contig) {
  (void)may_contig;
  if (!layer || !bound()) return 0;
  const int pal = get_int(layer, WEED_LEAF_CURRENT_PALETTE, 0);
  int width = get_int(layer, WEED_LEAF_WIDTH, 0), height = get_int(layer, WEED_LEAF_HEIGHT, 0);
  if (width <= 0 || height <= 0 || !pal_psize(pal)) return 0;
  if (pal == WEED_PALETTE_YUV420P || pal == WEED_PALETTE_YVU420P) { width = (width >> 1) << 1; height = (height >> 1) << 1; }   // :11601-11603
  Layer old;
  const bool had = read_layer(layer, &old);
  NewPlanes np;
  if (!alloc_planes(pal, width, height, 0, &np, layer)) return 0;
  if (black_fill) {
    // opaque black: RGB 0,0,0 (alpha 255); YUV 16 (clamped) or 0, 128, 128 (src/colourspace.c:11448-11460)
    const int clamping = get_int(layer, WEED_LEAF_YUV_CLAMPING, WEED_YUV_CLAMPING_UNCLAMPED);
    if (pal_is_planar_yuv(pal)) {
      memset(np.pd[0], clamping == WEED_YUV_CLAMPING_CLAMPED ? 16 : 0, np.sz[0]);
      memset(np.pd[1], 128, np.sz[1]);
      memset(np.pd[2], 128, np.sz[2]);
    } else if (pal_has_alpha(pal)) {
      const int a = pal_alpha_first(pal) ? 0 : 3;
      for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) np.pd[0][(size_t)y * np.rs[0] + x * 4 + a] = 255;
    }
  }
  if (had) free_planes(old);
  commit_planes(layer, pal, width, height, np);
  return 1;
}

lives_gpu_boolean lives_gpu_convert_layer_palette_full(lives_gpu_layer_t *layer, int outpl, int oclamping, int osampling,
                                                       int osubspace, int tgt_gamma) {
  (void)osampling;
  PinScope pin(layer);
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  const int inpl = l.pal;
  if (!pal_is_rgb(inpl) && !pal_is_rgb(outpl) && l.clamping >= 0 && (l.clamping != oclamping || l.subspace != osubspace)) {
    // YUV -> YUV with a different range (:12241-12262): same subspace = in-place table switch; a subspace change goes through
    // RGB in the reference -- left to the caller's CPU body
    if (l.subspace != osubspace) return 0;
    if (!switch_layer_clamping(layer, l, oclamping)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  if (inpl == outpl) return 1;                                           // :12265
  if (!pal_is_rgb(outpl)) {
    if (pal_is_rgb(inpl)) return rgb_layer_to_yuv(layer, l, outpl, oclamping, osubspace, tgt_gamma);
    return yuv_layer_repack(layer, l, outpl, l.clamping >= 0 ? l.clamping : oclamping);   // YUV -> YUV repacks (K5b)
  }
  const int iclamping = l.clamping >= 0 ? l.clamping : oclamping;        // :12216-12218

  // gamma decision (:12311-12332): only an explicit target changes the transfer function here
  int new_gamma = l.gamma;
  uint8_t lut[256];
  const uint8_t *lutp = nullptr;
  if (g_prefs.apply_gamma && l.gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != WEED_GAMMA_UNKNOWN && tgt_gamma != l.gamma) {
    new_gamma = tgt_gamma;
    if (lgpu_gamma_lut8(1.0, l.gamma, new_gamma, g_prefs.screen_gamma, lut)) lutp = lut;
  }
  // premultiplied-alpha bookkeeping (:12290-12306)
  int flags = l.flags;
  if (g_prefs.alpha_post) {
    if ((flags & LIVES_LAYER_ALPHA_PREMULT) && pal_has_alpha(inpl) && !pal_has_alpha(outpl)) {
      lives_gpu_alpha_premult(layer, LIVES_DIRECTION_REVERSE);
      if (!read_layer(layer, &l)) return 0;
      flags = l.flags;
    }
  } else if (!pal_has_alpha(inpl) && pal_has_alpha(outpl)) flags |= LIVES_LAYER_ALPHA_PREMULT;
  if (pal_has_alpha(inpl) && !pal_has_alpha(outpl)) flags &= ~LIVES_LAYER_ALPHA_PREMULT;

  NewPlanes np;
  const int owidth = (inpl == WEED_PALETTE_UYVY || inpl == WEED_PALETTE_YUYV) ? l.width * 2 : inpl == WEED_PALETTE_YUV411 ? l.width * 4 : l.width;   // macropixels -> pixels (:13010, :13759)
  if (!alloc_planes(outpl, owidth, l.height, 0, &np)) return 0;
  const size_t obytes = (size_t)np.rs[0] * l.height;
  uint8_t *d_out = t_scr.get(3, obytes);
  bool ok = d_out != nullptr;
  if (ok && pal_is_rgb(inpl)) {
    int af = 0;
    const int op = rgb_swizzle_op(inpl, outpl, &af);
    const size_t ibytes = (size_t)l.rs[0] * l.height;
    uint8_t *d_in = t_scr.get(0, ibytes);
    ok = d_in && up(d_in, l.pd[0], ibytes) &&
         lgpu_swizzle(op, af, d_in, l.rs[0], d_out, np.rs[0], l.width, l.height, lutp, nullptr) == LGPU_OK;
  } else if (ok && (inpl == WEED_PALETTE_YUV420P || inpl == WEED_PALETTE_YVU420P || inpl == WEED_PALETTE_YUV422P)) {
    const int iu = (inpl == WEED_PALETTE_YVU420P) ? 2 : 1, iv = (inpl == WEED_PALETTE_YVU420P) ? 1 : 2;   // swap_chroma_planes (:12353)
    const int ch = plane_h(l, 1);
    const size_t yb = (size_t)l.rs[0] * l.height, ub = (size_t)l.rs[iu] * ch, vb = (size_t)l.rs[iv] * ch;
    uint8_t *dy = t_scr.get(0, yb), *du = t_scr.get(1, ub), *dv = t_scr.get(2, vb);
    const int strides[3] = {l.rs[0], l.rs[iu], l.rs[iv]};
    const int which = (iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | (l.subspace == WEED_YUV_SUBSPACE_BT709 ? 2 : 0);
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    ok = dy && du && dv && up(dy, l.pd[0], yb) && up(du, l.pd[iu], ub) && up(dv, l.pd[iv], vb);
    if (ok && lutp) {
      // with a target gamma the reference fuses the 16-bit indexed LUT of create_gamma_lut into the conversion (:3274-3283)
      static thread_local std::vector<uint16_t> h16(65536);
      uint8_t *d16 = t_scr.get(7, 65536 * 2);
      ok = d16 && lgpu_gamma_lut16(1.0, l.gamma, new_gamma, g_prefs.screen_gamma, h16.data()) && up(d16, (const uint8_t *)h16.data(), 65536 * 2) &&
           lgpu_yuv420p_to_rgb_lut16(dy, du, dv, strides, (long)ub, (long)vb, d_out, np.rs[0], l.width, l.height, pal_psize(outpl), order,
                                     inpl == WEED_PALETTE_YUV422P, which, g_prefs.pb_quality, (const uint16_t *)d16, 0, nullptr) == LGPU_OK;
    } else if (ok)
      ok = lgpu_yuv420p_to_rgb(dy, du, dv, strides, (long)ub, (long)vb, d_out, np.rs[0], l.width, l.height, pal_psize(outpl), order,
                               inpl == WEED_PALETTE_YUV422P, which, g_prefs.pb_quality, nullptr, 0, nullptr) == LGPU_OK;
  } else if (ok && k3_fmt(inpl) >= 0) {
    // K3: packed / planar 4:4:4, UYVY, YUYV -> RGB family (src/colourspace.c:12937-13860 cases); no inline gamma on these paths
    const int fmt = k3_fmt(inpl), in_alpha = (inpl == WEED_PALETTE_YUVA8888 || inpl == WEED_PALETTE_YUVA4444P);
    const int pxw = (fmt >= 2) ? l.width * 2 : l.width;                     // UYVY / YUYV layers count macropixels
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    const int which = (iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | ((fmt == 0 && l.subspace == WEED_YUV_SUBSPACE_BT709) ? 2 : 0);
    ok = !lutp && (fmt < 2 || np.rs[0] >= pxw * pal_psize(outpl));
    const uint8_t *dsrc[4] = {nullptr, nullptr, nullptr, nullptr};
    int irs[4] = {0, 0, 0, 0};
    for (int p = 0; p < l.nplanes && ok; p++) {
      const size_t b = (size_t)l.rs[p] * l.height;
      uint8_t *d = t_scr.get(p == 0 ? 0 : p + 3, b);
      ok = d && up(d, l.pd[p], b);
      dsrc[p] = d; irs[p] = l.rs[p];
    }
    ok = ok && lgpu_yuv_to_rgb(dsrc, irs, pxw, l.height, fmt, in_alpha, d_out, np.rs[0], order, pal_has_alpha(outpl) ? 1 : 0, which, nullptr) == LGPU_OK;
  } else if (ok && inpl == WEED_PALETTE_YUV411) {
    // K3b (:13755-13795): the reference walks the source as compact rows of `width` macropixels and leaves some alpha bytes of the
    // new (zeroed, create_empty_pixel_data) frame unwritten -- the device frame starts zeroed too
    const size_t ibytes = (size_t)l.width * 6 * l.height;
    uint8_t *d_in = t_scr.get(0, ibytes);
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    ok = !lutp && d_in && (size_t)l.rs[0] * l.height >= ibytes && up(d_in, l.pd[0], ibytes) && lgpu_fill(d_out, 0, obytes, nullptr) == LGPU_OK &&
         lgpu_yuv411_to_rgb(d_in, l.width, l.height, d_out, np.rs[0], order, pal_has_alpha(outpl) ? 1 : 0,
                            iclamping == WEED_YUV_CLAMPING_UNCLAMPED, nullptr) == LGPU_OK;
  } else ok = false;
  ok = ok && down(np.pd[0], d_out, obytes) && sync();
  if (!ok) { for (int q = 0; q < np.n; q++) res_drop(np.pd[q]); pfree(np.pd[0]); return 0; }                                  // memfail: layer untouched
  free_planes(l);
  commit_planes(layer, outpl, owidth, l.height, np);
  if (new_gamma != l.gamma) set_int(layer, WEED_LEAF_GAMMA_TYPE, new_gamma);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  g_api.leaf_delete(layer, WEED_LEAF_YUV_CLAMPING);                          // conv_done (:13881-13884)
  g_api.leaf_delete(layer, WEED_LEAF_YUV_SUBSPACE);
  g_api.leaf_delete(layer, WEED_LEAF_YUV_SAMPLING);
  return 1;
}

lives_gpu_boolean lives_gpu_convert_layer_palette(lives_gpu_layer_t *layer, int outpl, int op_clamping) {
  return lives_gpu_convert_layer_palette_full(layer, outpl, op_clamping, WEED_YUV_SAMPLING_DEFAULT, WEED_YUV_SUBSPACE_YUV, WEED_GAMMA_UNKNOWN);   // :13931
}

lives_gpu_boolean lives_gpu_gamma_convert_sub_layer(int gamma_type, double fileg, lives_gpu_layer_t *layer, int x, int y, int width,
                                                    int height, lives_gpu_boolean may_thread) {
  (void)may_thread;
  PinScope pin(layer);
  if (!g_prefs.apply_gamma) return 1;
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  if (!pal_is_rgb(l.pal)) return 0;
  if (gamma_type == l.gamma && fileg == 1.0) return 1;
  uint8_t lut[256];
  if (!lgpu_gamma_lut8(gamma_type == LIVES_GAMMA_VARIANT ? fileg : 1.0, l.gamma, gamma_type, g_prefs.screen_gamma, lut)) return 1;
  if (x < 0 || y < 0 || x + width > l.width || y + height > l.height) return 0;
  const size_t bytes = (size_t)l.rs[0] * l.height;
  uint8_t *d = t_scr.get(0, bytes);
  const bool ok = d && up(d, l.pd[0], bytes) &&
                  lgpu_gamma_apply(d, l.rs[0], x, y, width, height, pal_psize(l.pal), pal_alpha_first(l.pal), lut, nullptr) == LGPU_OK &&
                  down(l.pd[0], d, bytes) && sync();
  if (!ok) return 0;
  if (gamma_type != LIVES_GAMMA_VARIANT) set_int(layer, WEED_LEAF_GAMMA_TYPE, gamma_type);
  return 1;
}

lives_gpu_boolean lives_gpu_gamma_convert_layer(int gamma_type, lives_gpu_layer_t *layer) {
  Layer l;
  if (!bound() || !read_layer(layer, &l)) return 0;
  return lives_gpu_gamma_convert_sub_layer(gamma_type, 1.0, layer, 0, 0, l.width, l.height, 1);   // :14146-14155
}

void lives_gpu_alpha_premult(lives_gpu_layer_t *layer, int direction) {
  PinScope pin(layer);
  Layer l;
  if (!ready() || !read_layer(layer, &l) || !pal_has_alpha(l.pal)) return;
  bool ok = true;
  if (l.pal == WEED_PALETTE_YUVA8888 || l.pal == WEED_PALETTE_YUVA4444P) {
    // :11982, :12005-12047, :12063-12096: clamped layers go through the alcy / alcuv / unalcy / unalcuv tables, unclamped ones through al / unal
    const int clamped = (l.clamping < 0 || l.clamping == WEED_YUV_CLAMPING_CLAMPED) ? 1 : 0;      // weed_layer_get_yuv_clamping(): CLAMPED (0) when the leaf is missing
    uint8_t *dp[4] = {nullptr, nullptr, nullptr, nullptr};
    int rs[4] = {0, 0, 0, 0};
    size_t nb[4] = {0, 0, 0, 0};
    for (int p = 0; p < l.nplanes && ok; p++) {
      nb[p] = (size_t)l.rs[p] * l.height;
      dp[p] = t_scr.get(p == 3 ? 7 : p, nb[p]);
      rs[p] = l.rs[p];
      ok = dp[p] && up(dp[p], l.pd[p], nb[p]);
    }
    ok = ok && lgpu_alpha_premult_yuva(dp, rs, l.width, l.height, l.pal, clamped, direction == LIVES_DIRECTION_REVERSE, nullptr) == LGPU_OK;
    for (int p = 0; p < 3 && p < l.nplanes && ok; p++) ok = down(l.pd[p], dp[p], nb[p]);           // the alpha plane is only read
    ok = ok && sync();
  } else {
    const size_t bytes = (size_t)l.rs[0] * l.height;
    uint8_t *d = t_scr.get(0, bytes);
    ok = d && up(d, l.pd[0], bytes) &&
         lgpu_alpha_premult(d, l.rs[0], l.width, l.height, pal_alpha_first(l.pal), direction == LIVES_DIRECTION_REVERSE, nullptr) == LGPU_OK &&
         down(l.pd[0], d, bytes) && sync();
  }
  if (!ok) return;
  int flags = l.flags;
  if (direction == LIVES_DIRECTION_FORWARD) flags |= LIVES_LAYER_ALPHA_PREMULT; else flags &= ~LIVES_LAYER_ALPHA_PREMULT;   // :12098-12102
  set_int(layer, kLeafHostFlags, flags);
}

// resize every plane of `l` into freshly allocated planes of (width x height); returns device-side success
static bool resize_into(const Layer &l, int width, int height, int interp, int alignment, NewPlanes *np) {
  if (!alloc_planes(l.pal, width, height, alignment, np)) return false;
  bool ok = true;
  Layer nl = l;
  nl.width = width; nl.height = height;
  for (int p = 0; p < np->n && ok; p++) {
    const int ps = pal_is_planar_yuv(l.pal) ? 1 : pal_psize(l.pal);
    const int sw = plane_w(l, p), sh = plane_h(l, p), dw = plane_w(nl, p), dh = plane_h(nl, p);
    const size_t ib = (size_t)l.rs[p] * sh, ob = (size_t)np->rs[p] * dh;
    uint8_t *d_in = t_scr.get(0, ib), *d_out = t_scr.get(3, ob);
    ok = d_in && d_out && up(d_in, l.pd[p], ib) &&
         lgpu_resize(d_in, l.rs[p], sw, sh, d_out, np->rs[p], dw, dh, ps, interp, nullptr, nullptr) == LGPU_OK &&
         down(np->pd[p], d_out, ob) && sync();
  }
  if (!ok) { for (int q = 0; q < np->n; q++) res_drop(np->pd[q]); pfree(np->pd[0]); }
  return ok;
}

lives_gpu_boolean lives_gpu_resize_layer(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint) {
  (void)oclamp_hint;
  PinScope pin(layer);
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  // opal_hint / oclamp_hint "may be ignored ... layer palette should be checked on return" (:14746-14751): a frame whose palette this path
  // resizes keeps it (the caller's following convert_layer_palette does the rest); a packed-YUV frame is first taken to the hinted
  // palette when that one is resizable and the conversion is served here
  if (!(pal_is_rgb(l.pal) || pal_is_planar_yuv(l.pal))) {
    if (opal_hint == WEED_PALETTE_NONE || opal_hint == l.pal || !(pal_is_rgb(opal_hint) || pal_is_planar_yuv(opal_hint))) return 0;
    const int cl = l.clamping >= 0 ? l.clamping : WEED_YUV_CLAMPING_CLAMPED;
    if (!lives_gpu_convert_layer_palette_full(layer, opal_hint, cl, WEED_YUV_SAMPLING_DEFAULT, l.subspace, WEED_GAMMA_UNKNOWN)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  int iwidth = (l.width >> 1) << 1, iheight = (l.height >> 1) << 1;      // :14854-14863
  if (width < 4) width = 4;
  if (height < 4) height = 4;
  if (iwidth != width || iheight != height) height = (height >> 1) << 1;
  if (iwidth == width && iheight == height) return 1;
  if (pal_is_planar_yuv(l.pal)) width = (width >> 1) << 1;
  Layer src = l;
  src.width = iwidth; src.height = iheight;
  NewPlanes np;
  if (!resize_into(src, width, height, interp, 16, &np)) return 0;       // rowstride_alignment_hint = 16 (:14989)
  free_planes(l);
  commit_planes(layer, l.pal, width, height, np);
  return 1;
}

lives_gpu_boolean lives_gpu_letterbox_layer(lives_gpu_layer_t *layer, int nwidth, int nheight, int width, int height, int interp,
                                            int tpal, int tclamp) {
  PinScope pin(layer);
  if (!width || !height || !nwidth || !nheight) return 1;                 // :15377
  if (nwidth < width) nwidth = width;
  if (nheight < height) nheight = height;
  if (nheight == height && nwidth == width) { lives_gpu_resize_layer(layer, width, height, interp, tpal, tclamp); return 1; }
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  if (l.width != width || l.height != height) {
    if (!lives_gpu_resize_layer(layer, width, height, interp, tpal, tclamp)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  width = l.width; height = l.height;
  if (nwidth < width || nheight < height) return 0;
  Layer canvas = l;
  canvas.width = nwidth; canvas.height = nheight;
  NewPlanes np;
  if (!alloc_planes(l.pal, nwidth, nheight, 0, &np)) return 0;
  bool ok = true;
  const int offs_x = (nwidth - width + 1) >> 1, offs_y = (nheight - height + 1) >> 1;     // :15522-15523
  for (int p = 0; p < np.n && ok; p++) {
    const int ps = pal_is_planar_yuv(l.pal) ? 1 : pal_psize(l.pal);
    // chroma planes: offsets scaled by the plane ratio and truncated (:15553-15556)
    const int px = (plane_w(l, p) == l.width) ? offs_x : (int)(offs_x * 0.5), py = (plane_h(l, p) == l.height) ? offs_y : (int)(offs_y * 0.5);
    uint8_t black[4] = {0, 0, 0, 0};
    if (pal_is_planar_yuv(l.pal)) black[0] = (p == 0) ? (l.clamping == WEED_YUV_CLAMPING_UNCLAMPED ? 0 : 16) : 128;
    else if (pal_alpha_first(l.pal)) black[0] = 255;
    else if (pal_alpha_last(l.pal)) black[3] = 255;
    const int sw = plane_w(l, p), sh = plane_h(l, p), cw = plane_w(canvas, p), chh = plane_h(canvas, p);
    const size_t ib = (size_t)l.rs[p] * sh, ob = (size_t)np.rs[p] * chh;
    uint8_t *d_in = t_scr.get(0, ib), *d_out = t_scr.get(3, ob);
    // the canvas keeps zeroed row padding (calloc on the host side); upload it so the kernel's untouched bytes stay zero
    ok = d_in && d_out && up(d_in, l.pd[p], ib) && up_fresh(d_out, np.pd[p], ob) &&
         lgpu_letterbox_at(d_in, l.rs[p], sw, sh, d_out, np.rs[p], cw, chh, ps, black, px, py, nullptr) == LGPU_OK && down(np.pd[p], d_out, ob) && sync();
  }
  if (!ok) { for (int q = 0; q < np.n; q++) res_drop(np.pd[q]); pfree(np.pd[0]); return 0; }
  free_planes(l);
  commit_planes(layer, l.pal, nwidth, nheight, np);
  return 1;
}

// ---- device residency API (seam 2) ---------------------------------------------------------------------------
static size_t plane_bytes(const Layer &l, int p) {
  const bool planar = pal_is_planar_yuv(l.pal);
  const int h = (!planar || p == 0 || p == 3 || pal_is_444(l.pal) || l.pal == WEED_PALETTE_YUV422P) ? l.height : l.height >> 1;
  return (size_t)l.rs[p] * h;
}
int lives_gpu_layer_pin(lives_gpu_layer_t *layer) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return LGPU_E_BADARG;
  if (has_leaf(layer, kLeafResident)) return LGPU_OK;
  for (int p = 0; p < l.nplanes; p++) {
    const size_t n = plane_bytes(l, p);
    void *d = nullptr;
    size_t cap = 0;
    { std::lock_guard<std::mutex> lk(g_res_mu); d = pool_take(n, &cap); }
    if (!d) { for (int q = 0; q < p; q++) res_drop(l.pd[q]); return LGPU_E_NOMEM; }
    g_h2d += n;
    if (lgpu_upload(d, l.pd[p], n, nullptr) != LGPU_OK) {
      { std::lock_guard<std::mutex> lk(g_res_mu); pool_give(d, cap); }
      for (int q = 0; q < p; q++) res_drop(l.pd[q]);
      return LGPU_E_HIP;
    }
    std::lock_guard<std::mutex> lk(g_res_mu);
    ResEntry &e = g_res[l.pd[p]];
    pool_give(e.d, e.bytes);
    e.d = d; e.bytes = cap;
  }
  if (!sync()) return LGPU_E_HIP;
  set_int(layer, kLeafResident, 1);
  return LGPU_OK;
}
int lives_gpu_layer_sync(lives_gpu_layer_t *layer) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return LGPU_E_BADARG;
  for (int p = 0; p < l.nplanes; p++) {
    const size_t n = plane_bytes(l, p);
    void *d = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_res_mu);
      auto it = g_res.find(l.pd[p]);
      if (it != g_res.end() && it->second.bytes >= n) d = it->second.d;
    }
    if (!d) continue;                                   // this plane's host bytes are current
    g_d2h += n;
    if (lgpu_download(l.pd[p], d, n, nullptr) != LGPU_OK) return LGPU_E_HIP;
  }
  return sync() ? LGPU_OK : LGPU_E_HIP;
}
int lives_gpu_layer_unpin(lives_gpu_layer_t *layer) {
  const int rc = lives_gpu_layer_sync(layer);
  Layer l;
  if (read_layer(layer, &l)) for (int p = 0; p < l.nplanes; p++) res_drop(l.pd[p]);
  if (bound() && layer) g_api.leaf_delete(layer, kLeafResident);
  return rc;
}
// optional frame allocator pair for lives_gpu_weed_api.pixel_alloc / pixel_free: page-locked, zeroed host memory, so the planes the seam creates
// (and any frame the host allocates through it) cross PCIe by DMA at link rate instead of through the staging chunks
void *lives_gpu_pinned_calloc(size_t bytes) { return lgpu_pinned_calloc(bytes); }
void lives_gpu_pinned_free(void *p) { lgpu_pinned_free(p); }

// residency bridge for the weed plugin (same library, other seam): the device copy of a pinned layer's plane, looked up by the host plane
// pointer the channel carries; NULL when the plane is not resident (or smaller than asked)
void *lives_gpu_resident_lookup(const void *host_plane, size_t min_bytes) {
  if (!host_plane) return nullptr;
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_res.find(host_plane);
  if (it == g_res.end() || it->second.bytes < min_bytes) return nullptr;
  return it->second.d;
}
void lives_gpu_transfer_stats(unsigned long long *h2d_bytes, unsigned long long *d2h_bytes) {
  if (h2d_bytes) *h2d_bytes = g_h2d;
  if (d2h_bytes) *d2h_bytes = g_d2h;
}

}  // extern "C"
