// layer_seam.cpp -- the weed_layer_t seam (include/lives_gpu_layer.h): host-side logic of
//   convert_layer_palette_full   src/colourspace.c:12190-13928
//   gamma_convert_sub_layer      src/colourspace.c:14069-14143
//   alpha_premult                src/colourspace.c:11968-12105
//   resize_layer_full            src/colourspace.c:14759-15328
//   letterbox_layer              src/colourspace.c:15343-15567
//   create_empty_pixel_data      src/colourspace.c:11434-11700
//   calc_rowstrides              src/colourspace.c:11252-11366
//   weed_layer_clear_pixel_data  src/colourspace.c:11229-11244
//   compact_rowstrides           src/colourspace.c:14439-14496
//   unletterbox_layer            src/colourspace.c:15570-15628
// re-written around the frame-level C ABI (lives_gpu.h).  All pixel work happens in the HIP kernels; this
// file only reads / writes leaves through the accessors the host bound, moves planes over PCIe and keeps the
// reference's bookkeeping (palette, size, rowstrides, gamma, premult flag, YUV leaves, failure = untouched layer).
//
// Data movement (struct Work below): an ordinary layer's planes are uploaded into per-thread scratch, the kernels run,
// the results are downloaded into freshly allocated host planes and the call synchronises ONCE before it returns (host
// bytes must be valid then).  A pinned layer (lives_gpu_layer_pin) keeps its planes in HBM: the kernels read and write
// the resident buffers directly, new planes are pool buffers registered under the new host pointer, nothing is copied and
// nothing synchronises -- the calls of a chain only enqueue work, the one synchronisation is lives_gpu_layer_sync().
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sched.h>
#include <string.h>
#include <atomic>
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
constexpr const char *kLeafOpaque = "host_gpu_opaque";        // the host's word that every pixel of the layer's frame has alpha 255 (lives_gpu_layer_set_opaque)
constexpr const char *kLeafResident = "host_gpu_resident";    // this library's private leaf (host_* convention, src/effects-weed.h:80-119)

bool bound() { return g_api.leaf_get && g_api.leaf_set && g_api.leaf_num_elements && g_api.leaf_delete; }
// zeroed = false: every byte of the block is about to be overwritten (a download of whole planes, or -- pinned layer -- the bytes are stale by contract
// until lives_gpu_layer_sync() writes whole planes): glibc's calloc memsets recycled heap memory, ~60 us for a 1080p RGBA plane, which was the largest
// single item of a seam call's host time (tools/bench_seam.py; profiles/r02/seam_bench.txt)
void *palloc(size_t n, bool zeroed = true) {
  if (g_api.pixel_alloc) return g_api.pixel_alloc(n);
  return zeroed ? calloc(1, n ? n : 1) : malloc(n ? n : 1);
}
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
bool pal_is_yuv(int p) {
  return pal_is_planar_yuv(p) || p == WEED_PALETTE_UYVY || p == WEED_PALETTE_YUYV || p == WEED_PALETTE_YUV888 || p == WEED_PALETTE_YUVA8888 || p == WEED_PALETTE_YUV411;
}
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
// Streams: every host thread enqueues on a stream of its own (LiVES runs plan steps and conversions on pool threads, src/threading.c; one shared
// stream would run the small kernels of different tracks one after the other).  Non-blocking streams: a launch on a blocking one costs twice the host
// time (ordering against the null stream is checked per launch).  livesgpu_fx.so enqueues on the same per-thread streams (lives_gpu_resident_acquire /
// _release); a caller on the null stream is ordered at its hand-over (lives_gpu_resident_lookup) like any other stream.  Never destroyed (a thread_local destructor of the main thread runs after HIP's teardown), but recycled:
// A thread that ends hands its stream (and its hand-over event) to a spare list, and a new thread takes one from there before it creates one: a host that
// runs short-lived threads does not pile up streams.  No HIP call in the destructor (it may run after HIP's teardown); the list itself is never destroyed.
struct SpareGpuObjects { std::atomic<bool> held{false}; std::vector<void *> streams, events; };
SpareGpuObjects &spares() { static SpareGpuObjects *p = new SpareGpuObjects; return *p; }
struct ThreadGpu {
  void *stream = nullptr, *event = nullptr;
  bool tried = false;
  ~ThreadGpu() {
    if (!stream && !event) return;
    SpareGpuObjects &sp = spares();
    while (sp.held.exchange(true, std::memory_order_acquire)) sched_yield();
    if (stream) sp.streams.push_back(stream);
    if (event) sp.events.push_back(event);
    sp.held.store(false, std::memory_order_release);
  }
};
thread_local ThreadGpu t_gpu;
void *S() {
  if (!t_gpu.tried) {
    t_gpu.tried = true;
    SpareGpuObjects &sp = spares();
    while (sp.held.exchange(true, std::memory_order_acquire)) sched_yield();
    if (!sp.streams.empty()) { t_gpu.stream = sp.streams.back(); sp.streams.pop_back(); }
    if (!sp.events.empty()) { t_gpu.event = sp.events.back(); sp.events.pop_back(); }
    sp.held.store(false, std::memory_order_release);
    if (!t_gpu.stream && lgpu_stream_create(&t_gpu.stream, 1) != LGPU_OK) t_gpu.stream = nullptr;     // the null stream then: everything is ordered, nothing overlaps
  }
  return t_gpu.stream;
}
#define t_event (t_gpu.event)

// A device buffer and the stream its last use was enqueued on.  Ownership of a resident plane moves from call to call; a call on ANOTHER thread's
// stream orders itself behind the previous owner by recording an event on that stream at the moment of the hand-over (everything enqueued there
// so far, which includes the buffer's last use) and waiting for it -- nothing is recorded while a layer stays on one thread (an event per plane and
// call measured 4 - 8 us each on these streams: 9 -> 17 us per seam call).
// `stream` is where the last WRITE (or exclusive use) was enqueued; `rs` are other streams that have enqueued READS since (two effect instances on two pool
// threads may read one layer at the same time): a reader waits for the writer only, a writer for the writer and all readers.
struct Lazy;
// lazy: the plane is not computed yet -- it stands for a pending program (deferred execution, below; d is null then).  external: caller-owned device memory
// (lives_gpu_layer_pin_device) that never enters the pool.  lazy_readers: pending programs of OTHER planes read this one (their layer 2): they run before it is
// written, replaced or dropped.
struct Dev { void *d = nullptr; size_t bytes = 0; void *stream = nullptr; void *rs[4] = {nullptr, nullptr, nullptr, nullptr}; int nr = 0;
             Lazy *lazy = nullptr; bool external = false; int lazy_readers = 0; };
static char g_idle_tag;
void *const kIdle = &g_idle_tag;                  // Dev::stream of a buffer whose last use is known to be complete (the stream was synchronised since)
// The table lock: a dozen sub-microsecond critical sections per seam call.  A futex mutex here made sixteen host threads run slower than one (every
// contended acquisition a sleep / wake cycle, ~10 us; tools/bench_seam_mt.py: 500 us per call at 16 threads against 10 us alone), so: spin briefly, then yield.
struct SpinLock {
  std::atomic<bool> held{false};
  void lock() {
    for (int spins = 0;; spins++) {
      if (!held.load(std::memory_order_relaxed) && !held.exchange(true, std::memory_order_acquire)) return;
      if (spins < 4000) __builtin_ia32_pause(); else { sched_yield(); spins = 0; }
    }
  }
  void unlock() { held.store(false, std::memory_order_release); }
};
// Resident planes by HOST plane pointer, in 64 SHARDS with a lock each (round 6): sixteen host threads, one per track, each enter the table ~15 times per plan step
// of their track -- behind ONE lock a recorded seam call cost 6-30 us instead of 0.4 (tools/seam_profile.py); the tracks' planes are different allocations and
// land in different shards.  No function holds two shard locks at once.  The buffer pool has a lock of its own.
struct alignas(128) ResShard { SpinLock mu; std::unordered_map<const void *, Dev> m; };
constexpr int kResShards = 64;
ResShard g_shards[kResShards];
ResShard &shard_of(const void *h) { const uintptr_t a = (uintptr_t)h; return g_shards[((a >> 12) ^ (a >> 18) ^ (a >> 25)) & (kResShards - 1)]; }
SpinLock g_pool_mu;                               // guards g_pool
std::atomic<unsigned long long> g_h2d{0}, g_d2h{0};   // PCIe byte counters (tests check the residency contract with them)
thread_local bool t_pinned = false;             // the call in progress works on a pinned layer
// t_gpu.event: this thread's hand-over event (re-recorded at every hand-over; a wait holds the record it saw)

// No HIP call is made under a table lock: the functions below copy what they need out of the tables and do the stream work afterwards.
// The calling thread's stream waits for everything enqueued so far on the stream of the buffer's last use, if that is another stream.
void follow(void *other) {                        // other: a stream (nullptr = the null stream) or kIdle
  if (other == S() || other == kIdle) return;
  if (!t_event && lgpu_event_create(&t_event) != LGPU_OK) t_event = nullptr;
  if (t_event && lgpu_event_record(t_event, other) == LGPU_OK && lgpu_stream_wait_event(S(), t_event) == LGPU_OK) return;
  lgpu_sync(other);                               // no event to be had: wait for that stream on the host instead
}
void await(const Dev &b, bool write = true) {
  if (!b.d) return;
  follow(b.stream);
  if (write) for (int i = 0; i < b.nr; i++) follow(b.rs[i]);
}
// (the plane's shard lock held) the calling thread's stream has enqueued a read / a write of the buffer
void note_use(Dev &e, bool write) {
  void *s = S();
  if (write) { e.stream = s; e.nr = 0; return; }
  if (e.stream == s) return;
  for (int i = 0; i < e.nr; i++) if (e.rs[i] == s) return;
  if (e.nr < 4) e.rs[e.nr++] = s;
  else e.rs[0] = s;                               // a fifth concurrent reader stream of one plane: not tracked (LiVES copies a layer that fans out)
}

// Device buffers of resident planes: a pinned layer going through a chain of seam calls takes a new plane per call and drops the old one, and neither
// side may wait for the device.  First level: a list of dropped buffers (a hit costs nothing).  Behind it the device's stream-ordered pool
// (lgpu_malloc_ordered: hipMallocAsync / hipFreeAsync on the calling thread's stream; ~10 us each, no synchronisation -- hipMalloc costs ~0.1 ms
// and hipFree waits for the device to drain, tools/alloc_probe.hip).  A buffer from the list carries the stream of its last use; a taker on another
// stream waits for that one.
// Buffers come in size classes ({1, 1.25, 1.5, 1.75} x 2^k, at least 64 KB): planes of similar size -- a 960 x 540 frame and its 960 x 600 letterboxed
// canvas -- recycle each other's buffers instead of each going to the device pool.  The list keeps up to 8 GB (a 288 GB device; what a few dozen 4K
// layers in flight hand back and forth) and is bucketed by class, so in steady state no call reaches hipMallocAsync / hipFreeAsync at all: under
// sixteen host threads those two were where the seam calls queued (tools/bench_seam_mt.py).
std::unordered_map<size_t, std::vector<Dev>> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = 8ull << 30;
size_t pool_class(size_t n) {
  if (n <= (64u << 10)) return 64u << 10;
  size_t base = 64u << 10;
  while (base * 2 <= n) base *= 2;
  const size_t step = base / 4;
  return base + (n - base + step - 1) / step * step;
}
bool pool_take(size_t n, Dev *out) {
  const size_t cls = pool_class(n + 64);
  bool hit = false;
  {
    std::lock_guard<SpinLock> lk(g_pool_mu);
    auto it = g_pool.find(cls);
    if (it != g_pool.end() && !it->second.empty()) {
      std::vector<Dev> &v = it->second;
      int best = (int)v.size() - 1;                                // newest first; one whose last use needs no cross-stream wait if there is one near the top
      for (int i = best, k = 0; i >= 0 && k < 8; i--, k++)
        if ((v[i].stream == S() || v[i].stream == kIdle) && v[i].nr == 0) { best = i; break; }
      *out = v[best];
      v[best] = v.back();
      v.pop_back();
      g_pool_bytes -= cls;
      hit = true;
    }
  }
  if (hit) { await(*out); out->stream = S(); out->nr = 0; return true; }
  Dev b;
  if (lgpu_malloc_ordered(&b.d, cls, S()) != LGPU_OK) return false;
  b.bytes = cls; b.stream = S();
  *out = b;
  return true;
}
void lazy_discard(Lazy *z);
void lazy_run_readers_of(const void *h);
void pool_give(Dev b) {
  if (b.lazy) { lazy_discard(b.lazy); return; }       // a pending program nobody will ever look at: its source frame goes back, nothing runs
  if (!b.d || b.external) return;
  {
    std::lock_guard<SpinLock> lk(g_pool_mu);
    if (g_pool_bytes + b.bytes <= kPoolMaxBytes) {
      g_pool[b.bytes].push_back(b);
      g_pool_bytes += b.bytes;
      return;
    }
  }
  await(b);                                       // the free is ordered on this thread's stream, which first waits for the last use
  lgpu_free_ordered(b.d, S());
}
// the resident copy registered under host plane h becomes b (last used on the calling thread's stream); whatever was there goes back to the pool
void res_put(const void *h, Dev b) {
  Dev old;
  b.stream = S(); b.nr = 0;
  lazy_run_readers_of(h);
  {
    ResShard &sh = shard_of(h);
    std::lock_guard<SpinLock> lk(sh.mu);
    Dev &e = sh.m[h];
    old = e;
    e = b;
  }
  pool_give(old);
}

void res_drop(const void *h) {
  Dev b;
  lazy_run_readers_of(h);
  {
    ResShard &sh = shard_of(h);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(h);
    if (it == sh.m.end()) return;
    b = it->second;
    sh.m.erase(it);
  }
  pool_give(b);
}
// every resident plane whose host pointer lies inside [base, base + bytes): a contiguous planar block is freed through its base pointer, but its planes 1..3
// are registered under interior pointers and must not survive it (a later block at the same address would otherwise inherit their device copies)
void res_drop_range(const void *base, size_t bytes) {
  std::vector<Dev> gone;
  {
    std::vector<const void *> read;
    for (ResShard &sh : g_shards) {
      std::lock_guard<SpinLock> lk(sh.mu);
      for (auto &kv : sh.m) if ((uintptr_t)kv.first >= (uintptr_t)base && (uintptr_t)kv.first < (uintptr_t)base + bytes && kv.second.lazy_readers > 0) read.push_back(kv.first);
    }
    for (const void *h : read) lazy_run_readers_of(h);
  }
  for (ResShard &sh : g_shards) {
    std::lock_guard<SpinLock> lk(sh.mu);
    for (auto it = sh.m.begin(); it != sh.m.end();) {
      const uintptr_t h = (uintptr_t)it->first;
      if (h >= (uintptr_t)base && h < (uintptr_t)base + bytes) { gone.push_back(it->second); it = sh.m.erase(it); } else ++it;
    }
  }
  for (auto &b : gone) pool_give(b);
}
int lives_gpu_layer_unpin_impl(weed_plant_t *layer);
struct PinScope {
  bool prev;
  weed_plant_t *layer;
  explicit PinScope(weed_plant_t *layer_) : prev(t_pinned), layer(layer_) { t_pinned = layer && bound() && has_leaf(layer, kLeafResident); }
  ~PinScope() { t_pinned = prev; }
  // every FALSE a seam call returns on a PINNED layer -- declined, or failed after it started (allocation, launch, copy) -- leaves the layer
  // synchronised and unpinned (lives_gpu_layer.h: "the CPU body reads current bytes"); decline() does the same on the paths that know they decline
  int settle(int rc) {
    if (!rc && t_pinned && layer && has_leaf(layer, kLeafResident)) lives_gpu_layer_unpin_impl(layer);
    return rc;
  }
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
// THREADVAR(rowstride_alignment_hint) of the calling thread as calc_rowstrides consumes it (:11285-11297): a value >= 4 applies to the next
// allocation only, -1 (compact rows: what v1 playback plugins and the transcoder ask for, src/player.c:1355-1356, src/transcode.c:42) stays until
// the caller resets it.  The host forwards its own thread variable with lives_gpu_set_rowstride_alignment_hint(); the reference also remembers
// the last alignment used as the thread's new default (:11295) -- not mirrored: that default belongs to LiVES' own allocations.
thread_local int t_rs_hint = 0;
int take_alignment(int forced) {
  if (forced) { t_rs_hint = 0; return forced; }      // resize_layer overwrites the hint with 16 and the allocation consumes it (:14989)
  const int h = t_rs_hint;
  if (h != -1) t_rs_hint = 0;
  return h;
}
bool alloc_planes(int pal, int width, int height, int alignment, NewPlanes *np, weed_plant_t *fixed_from = nullptr, bool zeroed = false) {
  np->n = lgpu_calc_rowstrides(width, pal, take_alignment(alignment), np->rs);
  if (np->n < 1) return false;
  apply_const_rowstrides(fixed_from, np->n, np->rs);
  size_t tot = 0;
  for (int i = 0; i < np->n; i++) {
    const int h = (i == 0 || pal_is_444(pal) || pal == WEED_PALETTE_YUV422P) ? height : height >> 1;
    np->sz[i] = (size_t)np->rs[i] * h;
    tot += np->sz[i];
  }
  uint8_t *blk = (uint8_t *)palloc(tot + 64, zeroed);     // + EXTRA_BYTES-style slack (reference loops read a few bytes past the end)
  if (!blk) return false;
  if (!zeroed) memset(blk + tot, 0, 64);
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
// A thread that ends leaves its buffers on a spare list (no HIP call in a thread_local destructor: the main thread's runs after HIP's teardown) and the next
// new thread starts from them.
struct ScratchSet { void *p[8]; size_t cap[8]; void *stream; };      // stream: where the thread that owned the set enqueued its last use
struct SpareScratch { std::mutex mu; std::vector<ScratchSet> sets; };
SpareScratch &spare_scratch() { static SpareScratch *sp = new SpareScratch; return *sp; }
struct Scratch {
  void *p[8] = {nullptr};
  size_t cap[8] = {0};
  bool adopted = false;
  void *last_stream = nullptr;
  uint8_t *get(int i, size_t bytes) {
    if (!adopted) {
      adopted = true;
      void *prev = kIdle;
      {
        SpareScratch &sp = spare_scratch();
        std::lock_guard<std::mutex> lk(sp.mu);
        if (!sp.sets.empty()) {
          for (int k = 0; k < 8; k++) { p[k] = sp.sets.back().p[k]; cap[k] = sp.sets.back().cap[k]; }
          prev = sp.sets.back().stream;
          sp.sets.pop_back();
        }
      }
      // the set's previous owner never synchronised on a pinned layer: its last kernels may still read these buffers on ITS stream, which some third thread may
      // have adopted meanwhile -- this thread's stream orders itself behind that stream once, at adoption
      follow(prev);
    }
    last_stream = S();
    if (cap[i] < bytes) {
      if (p[i]) lgpu_free(p[i]);
      p[i] = nullptr; cap[i] = 0;
      if (lgpu_malloc(&p[i], bytes + 64) != LGPU_OK) return nullptr;
      cap[i] = bytes;
    }
    return (uint8_t *)p[i];
  }
  ~Scratch() {
    bool any = false;
    for (int k = 0; k < 8; k++) any = any || p[k];
    if (!any) return;
    SpareScratch &sp = spare_scratch();
    std::lock_guard<std::mutex> lk(sp.mu);
    ScratchSet st;
    for (int k = 0; k < 8; k++) { st.p[k] = p[k]; st.cap[k] = cap[k]; }
    st.stream = last_stream;
    sp.sets.push_back(st);
  }
};
thread_local Scratch t_scr;

bool ready() { return bound() && lgpu_init(g_prefs.device) == LGPU_OK; }
bool sync() { return lgpu_sync(S()) == LGPU_OK; }

// resident device copy of a plane of the layer the call in progress works on (only pinned layers have one: the table is
// keyed by host pointer, and a host pointer proves nothing about a layer that was never pinned), ready for work on the calling
// thread's stream.  The caller marks the plane (touch_done) once its work is enqueued.
bool lazy_materialise(const void *h);
uint8_t *acquire(const void *h, size_t n, bool write) {
  Dev b;
  for (int pass = 0;; pass++) {
    {
      ResShard &sh = shard_of(h);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(h);
      if (it == sh.m.end() || it->second.bytes < n) return nullptr;
      b = it->second;
    }
    if (!b.lazy && !(write && b.lazy_readers > 0)) break;
    if (pass > 3) return nullptr;
    if (b.lazy && !lazy_materialise(h)) return nullptr;       // whoever needs the pixels themselves gets them: the pending program runs now, on this thread's stream
    if (write && b.lazy_readers > 0) lazy_run_readers_of(h);    // programs that still read this plane see it as it is now
  }
  await(b, write);
  return (uint8_t *)b.d;
}
uint8_t *resident(const void *h, size_t n, bool write) {
  if (!t_pinned || !h) return nullptr;
  return acquire(h, n, write);
}
void touch_done(const void *h, bool write) {
  ResShard &sh = shard_of(h);
  std::lock_guard<SpinLock> lk(sh.mu);
  auto it = sh.m.find(h);
  if (it != sh.m.end()) note_use(it->second, write);
}

// One seam call's device-side work, enqueued on the calling thread's stream.  in(): the current bytes of an existing plane.  out(): a plane
// the call creates (for a new host plane).  inout(): a plane modified in place.  finish() makes the results the planes' contents: downloads + ONE
// sync for an ordinary layer; for a pinned layer the pool buffers the kernels wrote become the resident copies, nothing moves, nothing waits.
// A Work that is dropped without finish() (any failure) gives its buffers back: the layer is as it was.  Either way every resident plane the
// call touched is marked with the calling thread's stream, which is what a later call on another thread's stream orders itself behind.
struct Work {
  struct Out { uint8_t *host; Dev b; size_t n; bool pooled; };
  Out outs[8];
  const void *touched[12];
  bool twrite[12];
  int nout = 0, ntouched = 0;
  bool ok = true, done = false;
  uint8_t *use(const void *h, size_t n, bool write) {
    uint8_t *r = resident(h, n, write);
    if (r && ntouched < 12) { touched[ntouched] = h; twrite[ntouched++] = write; }
    else if (r) { touch_done(h, true); }                      // cannot happen with <= 4 planes in and out; marked early (as a write) rather than lost
    return r;
  }
  const uint8_t *in(const uint8_t *h, size_t n, int slot) {
    if (!ok) return nullptr;
    if (uint8_t *r = use(h, n, false)) return r;
    uint8_t *d = t_scr.get(slot, n);
    g_h2d += n;
    ok = d && lgpu_upload(d, h, n, S()) == LGPU_OK;
    return ok ? d : nullptr;
  }
  // zero: the kernel does not write every byte (row padding, skipped alpha bytes): start from the zeros of the fresh host plane
  uint8_t *out(uint8_t *h, size_t n, int slot, bool zero) {
    if (!ok || nout >= 8) { ok = false; return nullptr; }
    Out o;
    o.host = h; o.n = n; o.pooled = false;
    if (t_pinned) {
      if (!pool_take(n, &o.b)) { ok = false; return nullptr; }
      o.pooled = true;
    } else o.b.d = t_scr.get(slot, n);
    ok = o.b.d && (!zero || lgpu_fill(o.b.d, 0, n, S()) == LGPU_OK);
    if (o.b.d) outs[nout++] = o;
    return ok ? (uint8_t *)o.b.d : nullptr;
  }
  uint8_t *inout(uint8_t *h, size_t n, int slot) {
    if (!ok || nout >= 8) { ok = false; return nullptr; }
    if (uint8_t *r = use(h, n, true)) return r;              // modified where it lives
    uint8_t *d = const_cast<uint8_t *>(in(h, n, slot));
    if (d) { Out o; o.host = h; o.b.d = d; o.n = n; o.pooled = false; outs[nout++] = o; }
    return d;
  }
  void mark_touched() {
    for (int i = 0; i < ntouched; i++) touch_done(touched[i], twrite[i]);
    ntouched = 0;
  }
  bool finish() {
    if (!ok) return false;
    bool moved = false;
    for (int i = 0; i < nout && ok; i++) {
      Out &o = outs[i];
      if (o.pooled) {                                         // the buffer becomes the resident copy of the new host plane
        res_put(o.host, o.b);
        o.b = Dev();
      } else {
        g_d2h += o.n;
        ok = lgpu_download(o.host, o.b.d, o.n, S()) == LGPU_OK;
        moved = true;
      }
    }
    mark_touched();
    if (ok && moved) ok = sync();
    done = ok;
    return ok;
  }
  ~Work() {
    if (done) return;
    mark_touched();
    for (int i = 0; i < nout; i++)
      if (outs[i].pooled && outs[i].b.d) { outs[i].b.stream = S(); outs[i].b.nr = 0; pool_give(outs[i].b); }
  }
};

// A call that this library does not serve returns FALSE and the caller's CPU body takes over.  The host bytes of a
// pinned layer are stale by contract, so a pinned layer is first brought home and unpinned: the CPU body then sees the current
// pixels, and no stale device copy survives it.
int lives_gpu_layer_unpin_impl(weed_plant_t *layer);
lives_gpu_boolean decline(weed_plant_t *layer) {
  if (t_pinned && layer) lives_gpu_layer_unpin_impl(layer);
  return 0;
}

// ---- deferred execution on pinned layers ------------------------------------------------------------------------------------------------
// The host bytes of a pinned layer are stale by contract until lives_gpu_layer_sync(), so nothing obliges a seam call on a pinned layer to have RUN when it
// returns -- only to have happened, in order, before anybody sees the pixels.  The calls of one track's plan step (src/nodemodel.c:1065-1253: pconv, resize,
// letterbox substeps; src/effects-weed.c:1850-2425: the filter instance; the gamma substep) on an RGBA32 / BGRA32 frame are therefore RECORDED on the plane
// instead of launched one by one: R <-> B swizzle, gdk-pixbuf scale, letterbox canvas, "chroma blend" with a second layer, gamma LUT -- in that order, each at
// most once.  Every leaf of the layer changes exactly as in the eager call (palette, size, rowstrides, a fresh host plane, gamma tag); the plane's entry in the
// residency table holds the program instead of pixels.  A program runs
//   * when someone needs the pixels (any other seam call or effect on the plane, lives_gpu_layer_sync / _unpin: through acquire()), by itself;
//   * when lives_gpu_layers_flush() is handed the layers of a tick: programs of equal shape become ONE launch of the fused chain kernel (lgpu_chain /
//     lgpu_chain_canvas, one track per layer) -- the launch bench.py times, reached through the reference's own calls;
//   * before a plane it reads (layer 2 of its blend) is written, replaced or released.
// What a stage could refuse is checked when it is recorded (palette, geometry, the scaler's range), so a recorded call cannot turn into a FALSE later; a device
// failure at run time (allocation, launch) surfaces at the flush / sync that runs the program, which then stays pending.
// lives_gpu_set_deferred(0) switches the recording off (every call launches its own kernels, as before round 6); tests compare the two.
enum { LZ_NONE = 0, LZ_SWAP = 1, LZ_SCALE = 2, LZ_CANVAS = 3, LZ_BLEND = 4, LZ_LUT = 5 };
struct Lazy {
  Dev src;                          // the frame the program starts from (owned: goes back to the pool when the program has run, unless external)
  int sw = 0, sh = 0, srs = 0;
  int stage = LZ_NONE;              // the last stage recorded
  bool swap = false;
  bool scale = false; int dw = 0, dh = 0, interp = 0; bool opaque = false;     // opaque: the layer carried the host's word when the scale was recorded
  bool canvas = false; int nw = 0, nh = 0, ox = 0, oy = 0;
  bool blend = false; int bf = 0; const void *l2h = nullptr; Dev l2; int l2rs = 0;
  bool lut = false; uint8_t lut8[256];
  int w = 0, h = 0, rs = 0;         // the plane the program stands for
};
std::atomic<int> g_deferred{1};
std::atomic<unsigned long long> g_lz_recorded{0}, g_lz_chain_launches{0}, g_lz_chain_tracks{0}, g_lz_staged{0};      // lives_gpu_deferred_stats
std::mutex g_lazy_mu;               // one program (group) runs at a time; never taken with a table lock held
bool lazy_pal(int pal) { return pal == WEED_PALETTE_RGBA32 || pal == WEED_PALETTE_BGRA32; }

void lazy_unread(Lazy *z) {          // (no lock held) the program no longer reads its layer 2
  if (!z->blend || !z->l2h) return;
  ResShard &sh = shard_of(z->l2h);
  std::lock_guard<SpinLock> lk(sh.mu);
  auto it = sh.m.find(z->l2h);
  if (it != sh.m.end() && it->second.lazy_readers > 0) it->second.lazy_readers--;
  z->l2h = nullptr;
}
void lazy_discard(Lazy *z) {
  if (!z) return;
  lazy_unread(z);
  Dev src = z->src;
  delete z;
  pool_give(src);
}
bool lazy_same_shape(const Lazy *a, const Lazy *b) {
  return a->sw == b->sw && a->sh == b->sh && a->srs == b->srs && a->swap == b->swap && a->scale == b->scale && a->dw == b->dw && a->dh == b->dh &&
         a->interp == b->interp && a->opaque == b->opaque && a->canvas == b->canvas && a->nw == b->nw && a->nh == b->nh && a->ox == b->ox && a->oy == b->oy && a->blend == b->blend &&
         a->l2rs == b->l2rs &&          /* (not the blend amount: every track of a launch has its own, lgpu_chain_amounts) */ a->lut == b->lut && (!a->lut || !memcmp(a->lut8, b->lut8, 256)) && a->w == b->w && a->h == b->h && a->rs == b->rs;
}
// one program, stage by stage through stream-ordered scratch frames, into out (the plane's rowstride)
int lazy_run_staged(const Lazy *z, uint8_t *out) {
  const uint8_t *cur = (const uint8_t *)z->src.d;
  int cw = z->sw, ch = z->sh, crs = z->srs, rc = LGPU_OK;
  void *tmp[3] = {nullptr, nullptr, nullptr};
  int nt = 0;
  const int last_geo = z->canvas ? LZ_CANVAS : z->scale ? LZ_SCALE : LZ_SWAP;
  auto target = [&](int stage, int w, int h, uint8_t **dst, int *drs) -> int {       // where a geometric stage writes: the plane itself for the last one
    if (stage == last_geo) { *dst = out; *drs = z->rs; return LGPU_OK; }
    const int r = lgpu_malloc_ordered(&tmp[nt], (size_t)w * 4 * h + 64, S());
    if (r) return r;
    *dst = (uint8_t *)tmp[nt++]; *drs = w * 4;
    return LGPU_OK;
  };
  uint8_t *dst = nullptr;
  int drs = 0;
  if (!z->swap && !z->scale && !z->canvas) rc = lgpu_copy_rows(out, z->rs, cur, crs, cw * 4, ch, S());      // no geometric stage: the in-place stages work on a copy
  if (!rc && z->swap) {
    if (!(rc = target(LZ_SWAP, cw, ch, &dst, &drs))) rc = lgpu_swizzle(LGPU_SWAP3POSTALPHA, 0, cur, crs, dst, drs, cw, ch, nullptr, S());
    cur = dst; crs = drs;
  }
  if (!rc && z->scale) {
    if (!(rc = target(LZ_SCALE, z->dw, z->dh, &dst, &drs))) rc = lgpu_pixbuf_scale(cur, crs, cw, ch, dst, drs, z->dw, z->dh, 4, z->interp | (z->opaque ? LGPU_INTERP_OPAQUE : 0), S());
    cur = dst; crs = drs; cw = z->dw; ch = z->dh;
  }
  if (!rc && z->canvas) {
    const uint8_t black[4] = {0, 0, 0, 255};
    if (!(rc = target(LZ_CANVAS, z->nw, z->nh, &dst, &drs))) rc = lgpu_letterbox_at(cur, crs, cw, ch, dst, drs, z->nw, z->nh, 4, black, z->ox, z->oy, S());
    cur = dst; crs = drs; cw = z->nw; ch = z->nh;
  }
  if (!rc && z->blend) rc = lgpu_blend_chroma(out, z->rs, (const uint8_t *)z->l2.d, z->l2rs, out, z->rs, z->w, z->h, 4, 0, z->bf, S());
  if (!rc && z->lut) rc = lgpu_gamma_apply(out, z->rs, 0, 0, z->w, z->h, 4, 0, z->lut8, S());
  for (int i = 0; i < nt; i++) lgpu_free_ordered(tmp[i], S());
  return rc;
}
// run n pending programs of ONE shape (g_lazy_mu held by the caller; hs[i]: the host plane zs[i] is registered under).  Afterwards the planes are ordinary
// resident planes whose last writer is the calling thread's stream.
int lazy_run_group(Lazy *const *zs, const void *const *hs, int n) {
  const Lazy *z0 = zs[0];
  const size_t bytes = (size_t)z0->rs * z0->h;
  std::vector<Dev> outs((size_t)n);
  int rc = LGPU_OK;
  for (int i = 0; i < n; i++)
    if (!pool_take(bytes, &outs[(size_t)i])) { for (int k = 0; k < i; k++) pool_give(outs[(size_t)k]); return LGPU_E_NOMEM; }
  for (int i = 0; i < n; i++) {
    await(zs[i]->src, false);
    if (zs[i]->blend) await(zs[i]->l2, false);
    if (z0->rs != z0->w * 4) rc = rc ? rc : lgpu_fill(outs[(size_t)i].d, 0, bytes, S());      // the row padding of a fresh plane is zero (calloc in the eager path)
  }
  bool done = false;
  // every shape: with or without a resize stage, with or without a blend (lgpu_chain_amounts); LGPU_SEAM_STAGED / lgpu_tuning_set("SEAM_STAGED", 1): the fallback walk, for tests
  if (!rc && n <= LGPU_CHAIN_MAX_TRACKS && lgpu_tuning_get("SEAM_STAGED") <= 0) {
    lgpu_chain_params pr;
    memset(&pr, 0, sizeof pr);
    pr.sw = z0->sw; pr.sh = z0->sh; pr.irow = z0->srs; pr.dw = z0->scale ? z0->dw : z0->sw; pr.dh = z0->scale ? z0->dh : z0->sh; pr.irow2 = z0->l2rs; pr.orow = z0->rs;
    pr.swap_rb = z0->swap ? 1 : 0; pr.interp = (z0->scale ? z0->interp : 0) | LGPU_INTERP_PIXBUF | (z0->blend ? 0 : LGPU_INTERP_NOBLEND) | (z0->scale && z0->opaque ? LGPU_INTERP_OPAQUE : 0); pr.do_blur = 0; pr.bf = z0->bf; pr.use_lut = z0->lut ? 1 : 0;
    if (!z0->blend) pr.irow2 = z0->rs;
    if (z0->lut) memcpy(pr.lut8, z0->lut8, 256);
    std::vector<lgpu_chain_track> tr((size_t)n);
    for (int i = 0; i < n; i++) { tr[(size_t)i].src_d = (const uint8_t *)zs[i]->src.d; tr[(size_t)i].layer2_d = (const uint8_t *)zs[i]->l2.d; tr[(size_t)i].dst_d = (uint8_t *)outs[(size_t)i].d; }
    std::vector<uint8_t> amounts((size_t)n);
    for (int i = 0; i < n; i++) amounts[(size_t)i] = (uint8_t)zs[i]->bf;
    const lgpu_canvas cv = {z0->nw, z0->nh, z0->ox, z0->oy};
    const int crc = lgpu_chain_amounts(&pr, z0->canvas ? &cv : nullptr, tr.data(), n, amounts.data(), S());
    if (crc == LGPU_OK) { done = true; g_lz_chain_launches++; g_lz_chain_tracks += (unsigned long long)n; }
    else if (crc != LGPU_E_BADARG && crc != LGPU_E_UNSUPPORTED) rc = crc;          // a shape the fused kernel does not take runs stage by stage below
  }
  if (!rc && !done)
    for (int i = 0; i < n && !rc; i++) { rc = lazy_run_staged(zs[i], (uint8_t *)outs[(size_t)i].d); g_lz_staged++; }
  if (rc) {                                                                            // the programs stay pending; what was enqueued wrote scratch only
    for (int i = 0; i < n; i++) { outs[(size_t)i].stream = S(); pool_give(outs[(size_t)i]); }
    return rc;
  }
  for (int i = 0; i < n; i++) {
    Lazy *z = zs[i];
    {
      ResShard &sh = shard_of(hs[i]);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(hs[i]);
      if (it != sh.m.end() && it->second.lazy == z) {
        Dev &e = it->second;
        const int readers = e.lazy_readers;
        e = outs[(size_t)i];
        e.stream = S(); e.nr = 0; e.lazy = nullptr; e.lazy_readers = readers;
        outs[(size_t)i] = Dev();
      }
    }
    if (z->blend && z->l2h) {
      ResShard &sh = shard_of(z->l2h);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto l2 = sh.m.find(z->l2h);
      if (l2 != sh.m.end()) { note_use(l2->second, false); if (l2->second.lazy_readers > 0) l2->second.lazy_readers--; }
      z->l2h = nullptr;
    }
    if (outs[(size_t)i].d) { outs[(size_t)i].stream = S(); pool_give(outs[(size_t)i]); }          // (the plane vanished meanwhile: cannot happen under the host's own ordering)
    Dev src = z->src;
    src.stream = S(); src.nr = 0;                                                       // its last use is the launch just enqueued
    delete z;
    pool_give(src);
  }
  return LGPU_OK;
}
// the pending program of plane h, if any, runs now (on the calling thread's stream)
bool lazy_materialise(const void *h) {
  std::lock_guard<std::mutex> run(g_lazy_mu);
  Lazy *z = nullptr;
  {
    ResShard &sh = shard_of(h);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(h);
    if (it == sh.m.end()) return false;
    z = it->second.lazy;
  }
  if (!z) return true;                      // somebody else ran it meanwhile
  return lazy_run_group(&z, &h, 1) == LGPU_OK;
}
// programs that read plane h as their layer 2 run before h changes
void lazy_run_readers_of(const void *h) {
  for (int guard = 0; guard < 64; guard++) {
    const void *reader = nullptr;
    {
      ResShard &sh = shard_of(h);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(h);
      if (it == sh.m.end() || it->second.lazy_readers <= 0) return;
    }
    for (ResShard &sh : g_shards) {                  // (rare: a plane that pending programs read is about to change) look for one of them, shard by shard
      std::lock_guard<SpinLock> lk(sh.mu);
      for (auto &kv : sh.m) if (kv.second.lazy && kv.second.lazy->blend && kv.second.lazy->l2h == h) { reader = kv.first; break; }
      if (reader) break;
    }
    if (!reader) {
      ResShard &sh = shard_of(h);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(h);
      if (it != sh.m.end()) it->second.lazy_readers = 0;
      return;
    }
    if (!lazy_materialise(reader)) return;
  }
}
// The plane of layer l becomes (or stays) the subject of a pending program that can still take `stage`.  Returns the program DETACHED from the table (the caller
// records its stage and registers it under the layer's new host plane with lazy_attach, or puts it back under the old one on failure), or nullptr: not deferrable.
Lazy *lazy_detach(const Layer &l, int stage) {
  if (!g_deferred.load(std::memory_order_relaxed) || !t_pinned || !lazy_pal(l.pal) || l.nplanes != 1 || (l.rs[0] & 3) || ((uintptr_t)l.pd[0] & 3)) return nullptr;
  const size_t n = (size_t)l.rs[0] * l.height;
  for (int pass = 0; pass < 2; pass++) {
    Dev e;
    bool need_run = false;
    {
      ResShard &sh = shard_of(l.pd[0]);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(l.pd[0]);
      if (it == sh.m.end() || it->second.bytes < n) return nullptr;
      if (it->second.lazy_readers > 0) return nullptr;                     // someone's layer 2: it keeps its pixels
      if (it->second.lazy && it->second.lazy->stage >= stage) need_run = true;
      else { e = it->second; sh.m.erase(it); }
    }
    if (need_run) { if (!lazy_materialise(l.pd[0])) return nullptr; continue; }    // the recorded program cannot take this stage: it runs, a new one starts from its result
    if (e.lazy) return e.lazy;
    Lazy *z = new Lazy;
    z->src = e; z->sw = l.width; z->sh = l.height; z->srs = l.rs[0];
    z->w = l.width; z->h = l.height; z->rs = l.rs[0];
    return z;
  }
  return nullptr;
}
bool plane_is_lazy(const void *h) {
  ResShard &sh = shard_of(h);
  std::lock_guard<SpinLock> lk(sh.mu);
  auto it = sh.m.find(h);
  return it != sh.m.end() && it->second.lazy != nullptr;
}
void lazy_attach(const void *h, Lazy *z) {
  Dev e;
  g_lz_recorded++;
  e.lazy = z; e.bytes = (size_t)z->rs * z->h; e.stream = kIdle;
  Dev old;
  {
    ResShard &sh = shard_of(h);
    std::lock_guard<SpinLock> lk(sh.mu);
    Dev &slot = sh.m[h];
    old = slot;
    slot = e;
  }
  if (old.d || old.lazy) pool_give(old);
}
// put a detached program / plane back under the host plane it came from (a stage could not be recorded after all)
void lazy_reattach(const void *h, Lazy *z) {
  if (z->stage == LZ_NONE) {              // it was an ordinary resident plane
    Dev e = z->src;
    delete z;
    ResShard &sh = shard_of(h);
    std::lock_guard<SpinLock> lk(sh.mu);
    sh.m[h] = e;
    return;
  }
  lazy_attach(h, z);
}
// new host plane(s) for the plane a recorded stage produces; the old host plane is released, the program moves under the new pointer
bool lazy_commit(weed_plant_t *layer, const Layer &l, Lazy *z, int pal, int width, int height, int alignment) {
  NewPlanes np;
  if (!alloc_planes(pal, width, height, alignment, &np)) return false;
  z->w = width; z->h = height; z->rs = np.rs[0];
  lazy_attach(np.pd[0], z);
  if (l.contiguous) pfree(l.pd[0]); else for (int i = 0; i < l.nplanes; i++) pfree(l.pd[i]);      // free_planes without the table (the entry has moved)
  commit_planes(layer, pal, width, height, np);
  return true;
}

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
  Work w;
  uint8_t *d[4] = {nullptr, nullptr, nullptr, nullptr};
  int rs[4] = {0, 0, 0, 0};
  const bool planar = pal_is_planar_yuv(l.pal);
  for (int p = 0; p < l.nplanes; p++) {
    const int ph = (!planar || p == 0 || p == 3 || pal_is_444(l.pal) || l.pal == WEED_PALETTE_YUV422P) ? l.height : l.height >> 1;
    d[p] = w.inout(l.pd[p], (size_t)l.rs[p] * ph, p == 0 ? 0 : p + 3);
    rs[p] = l.rs[p];
  }
  if (!w.ok || lgpu_yuv_switch_clamping(d, rs, l.pal, l.height, oclamping == WEED_YUV_CLAMPING_UNCLAMPED, S()) != LGPU_OK) return false;
  if (!w.finish()) return false;
  set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);
  return true;
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

void drop_new_planes(const NewPlanes &np) { pfree(np.pd[0]); }       // one block (alloc_planes)

// K4b on a layer (:12627-12632 and the same case under each RGB input): width leaf becomes width >> 2 macropixels, the new frame is
// written as compact macropixel rows whatever rowstride it was given (the reference passes none).  `flags` = host_flags after the
// reference's premultiplied-alpha bookkeeping (:12290-12306), done by the caller for every palette pair.
lives_gpu_boolean rgb_layer_to_yuv411(weed_plant_t *layer, const Layer &l, int oclamping, int flags) {
  const int order = pal_alpha_first(l.pal) ? 2 : pal_red_first(l.pal) ? 0 : 1;
  const int in_alpha = pal_has_alpha(l.pal) ? 1 : 0, wm = l.width >> 2;
  if (wm < 1 || l.height < 1) return decline(layer);
  NewPlanes np;
  if (!alloc_planes(WEED_PALETTE_YUV411, wm, l.height, 0, &np)) return 0;
  Work w;
  const uint8_t *d_in = w.in(l.pd[0], (size_t)l.rs[0] * l.height, 0);
  uint8_t *d_out = w.out(np.pd[0], np.sz[0], 3, true);
  const bool ok = w.ok && lgpu_rgb_to_yuv411(d_in, l.rs[0], l.width, l.height, order, in_alpha, d_out, oclamping == WEED_YUV_CLAMPING_UNCLAMPED, S()) == LGPU_OK &&
                  w.finish();
  if (!ok) { drop_new_planes(np); return 0; }
  free_planes(l);
  commit_planes(layer, WEED_PALETTE_YUV411, wm, l.height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);                         // conv_done, as for the other RGB -> YUV cases below
  set_int(layer, WEED_LEAF_YUV_SUBSPACE, l.gamma == WEED_GAMMA_BT709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR);
  if (!has_leaf(layer, WEED_LEAF_YUV_SAMPLING)) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);
  return 1;
}

// K4 on a layer: the RGB24 / BGR24 / RGBA32 / BGRA32 / ARGB32 cases of src/colourspace.c:12559-12935 plus conv_done (:13860-13893)
const uint16_t *device_lut16(int from, int to);
lives_gpu_boolean rgb_layer_to_yuv(weed_plant_t *layer, const Layer &l_in, int outpl, int oclamping, int osubspace, int tgt_gamma, int flags) {
  // gamma on the way (:12311-12332): an RGB layer of known gamma that goes to YUV ends up SRGB (BT709 for a BT.709 target subspace) unless the caller names a
  // target.  The UYVY / YUYV entry points take the 16-bit LUT inline (can_inline_gamma :12136-12142, rgb2uyvy_with_gamma); for every other YUV palette the
  // reference converts the layer's gamma first (gamma_convert_layer, :12326-12329) and then the palette
  Layer l = l_in;
  int new_gamma = WEED_GAMMA_UNKNOWN;
  const bool inline_gamma = (outpl == WEED_PALETTE_UYVY || outpl == WEED_PALETTE_YUYV);
  const uint16_t *lut16 = nullptr;
  if (g_prefs.apply_gamma && l.gamma != WEED_GAMMA_UNKNOWN) {
    new_gamma = tgt_gamma != WEED_GAMMA_UNKNOWN ? tgt_gamma : osubspace == WEED_YUV_SUBSPACE_BT709 ? WEED_GAMMA_BT709 : WEED_GAMMA_SRGB;
    if (!inline_gamma) {
      if (new_gamma != l.gamma && !lives_gpu_gamma_convert_layer(new_gamma, layer)) return decline(layer);
      if (!read_layer(layer, &l)) return 0;
      new_gamma = l.gamma;
    } else if (new_gamma != l.gamma) {
      lut16 = device_lut16(l.gamma, new_gamma);          // nullptr when create_gamma_lut makes none for the pair: the plain entry point runs, as in the reference
    }
  }
  if (outpl == WEED_PALETTE_YUV411) return rgb_layer_to_yuv411(layer, l, oclamping, flags);
  const int fmt = k4_fmt(outpl);
  if (fmt < 0) return decline(layer);
  const int order = pal_alpha_first(l.pal) ? 2 : pal_red_first(l.pal) ? 0 : 1;
  if (order == 2 && fmt >= 4) return decline(layer);                        // reference-broken (:6353)
  const int in_alpha = pal_has_alpha(l.pal) ? 1 : 0, out_alpha = pal_has_alpha(outpl) ? 1 : 0;
  int width = l.width, height = l.height;
  if (fmt >= 2 && (width & 1)) return decline(layer);
  if (fmt == 4) { width = (width >> 1) << 1; height = (height >> 1) << 1; }       // create_empty_pixel_data :11601-11603
  if (width < 2 || height < 1) return decline(layer);
  // subspace argument as the dispatcher passes it: some cases hand WEED_YUV_SAMPLING_DEFAULT (= 0 -> YCbCr) to the subspace slot
  const bool use_osub = (fmt == 4 && l.pal != WEED_PALETTE_RGB24) || (fmt == 5 && l.pal == WEED_PALETTE_RGB24);
  const int which = (oclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | ((fmt >= 4 && use_osub && osubspace == WEED_YUV_SUBSPACE_BT709) ? 2 : 0);
  const int lwidth = (fmt == 2 || fmt == 3) ? width >> 1 : width;                   // UYVY / YUYV layers count macropixels
  NewPlanes np;
  if (!alloc_planes(outpl, lwidth, height, 0, &np)) return 0;
  Work w;
  const uint8_t *d_in = w.in(l.pd[0], (size_t)l.rs[0] * l.height, 0);
  uint8_t *ddst[4] = {nullptr, nullptr, nullptr, nullptr};
  int ors[4] = {0, 0, 0, 0};
  for (int p = 0; p < np.n; p++) { ddst[p] = w.out(np.pd[p], np.sz[p], 3 + p, true); ors[p] = np.rs[p]; }   // calloc'd padding stays as the host made it
  const bool ok = w.ok &&
                  (lut16 ? lgpu_rgb_to_yuv_lut16(d_in, l.rs[0], width, height, order, in_alpha, ddst[0], ors[0], fmt, which & 1, lut16, S())
                         : lgpu_rgb_to_yuv(d_in, l.rs[0], width, height, order, in_alpha, ddst, ors, fmt, out_alpha, which, S())) == LGPU_OK &&
                  w.finish();
  if (!ok) { drop_new_planes(np); return 0; }
  free_planes(l);
  if (outpl == WEED_PALETTE_YVU420P) { uint8_t *t = np.pd[1]; np.pd[1] = np.pd[2]; np.pd[2] = t; }   // swap_chroma_planes (:13890)
  commit_planes(layer, outpl, lwidth, height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  set_int(layer, WEED_LEAF_YUV_CLAMPING, oclamping);
  int final_gamma = l.gamma;
  if (inline_gamma && new_gamma != WEED_GAMMA_UNKNOWN) { set_int(layer, WEED_LEAF_GAMMA_TYPE, new_gamma); final_gamma = new_gamma; }      // :13873-13876
  set_int(layer, WEED_LEAF_YUV_SUBSPACE, final_gamma == WEED_GAMMA_BT709 ? WEED_YUV_SUBSPACE_BT709 : WEED_YUV_SUBSPACE_YCBCR);              // :13884-13888
  if (fmt >= 4 || !has_leaf(layer, WEED_LEAF_YUV_SAMPLING)) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);
  return 1;
}

// a pinned layer's LUT16 (create_gamma_lut, 65536 x uint16) per (device, from, to, screen gamma): built once, not per frame.  Entries are never evicted -- a
// pointer handed to one host thread may not yet be in a launch when another thread asks for a seventeenth table -- and there are few of them: the gamma pairs
// LiVES uses times the screen gammas a session sets, 128 KB each (a cap of 256 entries = 32 MB is a failure, not an eviction).
struct Lut16Key { int dev, from, to; double screen; bool operator==(const Lut16Key &o) const { return dev == o.dev && from == o.from && to == o.to && screen == o.screen; } };
std::mutex g_l16_mu;
std::vector<std::pair<Lut16Key, void *>> g_l16;
const uint16_t *device_lut16(int from, int to) {
  const Lut16Key k = {g_prefs.device, from, to, g_prefs.screen_gamma};
  std::lock_guard<std::mutex> lk(g_l16_mu);
  for (auto &e : g_l16) if (e.first == k) return (const uint16_t *)e.second;
  if (g_l16.size() >= 256) return nullptr;
  std::vector<uint16_t> h(65536);
  if (!lgpu_gamma_lut16(1.0, from, to, g_prefs.screen_gamma, h.data())) return nullptr;
  void *d = nullptr;
  if (lgpu_malloc(&d, 65536 * 2) != LGPU_OK) return nullptr;
  if (lgpu_upload(d, h.data(), 65536 * 2, S()) != LGPU_OK || lgpu_sync(S()) != LGPU_OK) { lgpu_free(d); return nullptr; }     // the calling thread's stream; complete before the pointer is shared
  g_l16.push_back({k, d});
  return (const uint16_t *)d;
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

void lives_gpu_set_rowstride_alignment_hint(int hint) { t_rs_hint = hint; }
int lives_gpu_get_rowstride_alignment_hint(void) { return t_rs_hint; }

int *lives_gpu_calc_rowstrides(int width, int pal, lives_gpu_layer_t *layer, int *nplanes) {
  if (pal == WEED_PALETTE_NONE) { if (!layer || !bound()) return nullptr; pal = get_int(layer, WEED_LEAF_CURRENT_PALETTE, 0); }
  if (!width) { if (!layer || !bound()) return nullptr; width = get_int(layer, WEED_LEAF_WIDTH, 0); }
  int rs[4];
  const int n = lgpu_calc_rowstrides(width, pal, take_alignment(0), rs);
  if (nplanes) *nplanes = n;
  if (!n) return nullptr;
  apply_const_rowstrides(layer, n, rs);
  int *out = (int *)calloc((size_t)n, sizeof(int));      // caller frees with lives_free, like the reference
  for (int i = 0; i < n; i++) out[i] = rs[i];
  return out;
}

// YUV -> YUV repack of a layer (:12937-13750, the non-RGB half of the dispatcher) through lgpu_yuv_repack; 0 = not taken, the
// caller's CPU body runs (pairs / layouts listed in include/lives_gpu.h)
static lives_gpu_boolean yuv_layer_repack(weed_plant_t *layer, const Layer &l, int outpl, int iclamping, int flags) {
  const int inpl = l.pal;
  const bool inpk = (inpl == WEED_PALETTE_UYVY || inpl == WEED_PALETTE_YUYV), outpk = (outpl == WEED_PALETTE_UYVY || outpl == WEED_PALETTE_YUYV);
  const bool in411 = inpl == WEED_PALETTE_YUV411, out411 = outpl == WEED_PALETTE_YUV411;
  const int width = inpk ? l.width * 2 : in411 ? l.width * 4 : l.width, height = l.height;              // pixels
  if (width < 1 || height < 1) return decline(layer);
  const int unclamped = iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0;
  const uint8_t *dsrc[4] = {nullptr, nullptr, nullptr, nullptr};
  uint8_t *ddst[4] = {nullptr, nullptr, nullptr, nullptr};
  int irs[4] = {0, 0, 0, 0}, ors[4] = {0, 0, 0, 0};
  Work w;
  if (inpk && outpk) {
    // convert_swab_frame (:13139): in place, the layer keeps its pixel data
    uint8_t *dd[4] = {w.inout(l.pd[0], (size_t)l.rs[0] * height, 0), nullptr, nullptr, nullptr};
    dsrc[0] = dd[0]; irs[0] = l.rs[0];
    if (!w.ok) return 0;
    const int rc = lgpu_yuv_repack(inpl, outpl, dsrc, irs, dd, irs, width, height, unclamped, 0, S());
    if (rc == LGPU_E_UNSUPPORTED) return decline(layer);
    if (rc != LGPU_OK || !w.finish()) return 0;
    set_int(layer, WEED_LEAF_CURRENT_PALETTE, outpl);
    return 1;
  }
  for (int p = 0; p < l.nplanes; p++) { dsrc[p] = w.in(l.pd[p], (size_t)l.rs[p] * plane_h(l, p), p == 3 ? 7 : p); irs[p] = l.rs[p]; }
  if (!w.ok) return 0;
  if (inpl == WEED_PALETTE_YVU420P) { const uint8_t *t = dsrc[1]; dsrc[1] = dsrc[2]; dsrc[2] = t; const int r = irs[1]; irs[1] = irs[2]; irs[2] = r; }
  // K5c, the 4:1:1 pairs (:13024-13029 ..., :13793-13846): the new layer gets its ordinary (aligned, zeroed) planes and the reference functions then walk
  // them as compact streams; lgpu_yuv_repack does the same
  if ((in411 || out411) && (width & 3)) return decline(layer);
  if (in411 && (outpl == WEED_PALETTE_YUV420P || outpl == WEED_PALETTE_YVU420P) && (height & 1)) return decline(layer);
  const int lwidth = outpk ? width >> 1 : out411 ? width >> 2 : width;
  NewPlanes np;
  if (!alloc_planes(outpl, lwidth, height, 0, &np)) return 0;
  for (int p = 0; p < np.n; p++) { ddst[p] = w.out(np.pd[p], np.sz[p], 3 + p, true); ors[p] = np.rs[p]; }
  const int rc = w.ok ? lgpu_yuv_repack(inpl, outpl, dsrc, irs, ddst, ors, width, height, unclamped, l.sampling, S()) : LGPU_E_NOMEM;   // isampling: read by K5d only
  if (rc == LGPU_E_UNSUPPORTED) { drop_new_planes(np); return decline(layer); }
  if (rc != LGPU_OK || !w.finish()) { drop_new_planes(np); return 0; }
  free_planes(l);
  if (outpl == WEED_PALETTE_YVU420P) { uint8_t *t = np.pd[1]; np.pd[1] = np.pd[2]; np.pd[2] = t; }   // swap_chroma_planes (:13890)
  commit_planes(layer, outpl, lwidth, height, np);
  if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
  if ((outpl == WEED_PALETTE_YUV420P || outpl == WEED_PALETTE_YVU420P) && !in411) set_int(layer, WEED_LEAF_YUV_SAMPLING, WEED_YUV_SAMPLING_DEFAULT);   // :13022
  return 1;
}

// black of a palette as the reference paints it (blank_pixel / blank_row, src/colourspace.c:11123-11210) into a HOST plane set
static void host_black_fill(int pal, int width, int height, int clamping, const NewPlanes &np) {
  const uint8_t yb = clamping == WEED_YUV_CLAMPING_UNCLAMPED ? 0 : 16;
  if (pal_is_planar_yuv(pal)) {
    for (int p = 0; p < np.n; p++) memset(np.pd[p], p == 0 ? yb : p == 3 ? 255 : 128, np.sz[p]);
  } else if (pal_has_alpha(pal)) {
    const int a = pal_alpha_first(pal) ? 0 : 3;
    for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) np.pd[0][(size_t)y * np.rs[0] + x * 4 + a] = 255;
  }
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
  if (!alloc_planes(pal, width, height, 0, &np, layer, true)) return 0;
  // opaque black: RGB 0,0,0 (alpha 255); YUV 16 (clamped) or 0, 128, 128 (src/colourspace.c:11448-11460)
  if (black_fill) host_black_fill(pal, width, height, get_int(layer, WEED_LEAF_YUV_CLAMPING, WEED_YUV_CLAMPING_UNCLAMPED), np);
  if (had) free_planes(old);
  commit_planes(layer, pal, width, height, np);
  return 1;
}

static lives_gpu_boolean convert_layer_palette_full_body(lives_gpu_layer_t *layer, int outpl, int oclamping, int osampling, int osubspace, int tgt_gamma);
lives_gpu_boolean lives_gpu_convert_layer_palette_full(lives_gpu_layer_t *layer, int outpl, int oclamping, int osampling,
                                                       int osubspace, int tgt_gamma) {
  PinScope pin(layer);
  return pin.settle(convert_layer_palette_full_body(layer, outpl, oclamping, osampling, osubspace, tgt_gamma));
}
static lives_gpu_boolean convert_layer_palette_full_body(lives_gpu_layer_t *layer, int outpl, int oclamping, int osampling, int osubspace, int tgt_gamma) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  const int inpl = l.pal;
  // The reference's own first steps (range switch :12241-12262, un-premultiply :12290-12306) are done first here too.  If the
  // conversion proper is then declined, the layer is in the state the reference has at that point and its CPU body continues from
  // there (both steps are no-ops the second time: the leaves they test have been updated).
  if (pal_is_yuv(inpl) && pal_is_yuv(outpl) && ((l.clamping >= 0 && l.clamping != oclamping) || l.subspace != osubspace)) {      // :12241 (iclamping = oclamping without the leaf, :12216-12218)
    // YUV -> YUV with a different range: same subspace = in-place table switch (:12242-12247); a subspace change goes through RGB(A) (:12248-12262) -- the
    // frame is taken to RGB24 / RGBA32 with its own tables and converted from there with the target's range and subspace, as the reference does it
    if (l.subspace != osubspace) {
      if (!lives_gpu_convert_layer_palette(layer, pal_has_alpha(inpl) ? WEED_PALETTE_RGBA32 : WEED_PALETTE_RGB24, 0)) return 0;
      return convert_layer_palette_full_body(layer, outpl, oclamping, osampling, osubspace, tgt_gamma);
    }
    if (!switch_layer_clamping(layer, l, oclamping)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  if (inpl == outpl) {                                                     // :12265
    // Nothing left to do, exactly as in the reference: its 4:2:0 JPEG <-> MPEG chroma-siting pass (switch_yuv_sampling, :10876-10925) is called
    // from inside `if (isampling == osampling && ...)` under the condition `isampling != osampling` (:12268-12274), i.e. never (quirk SS1): a request
    // that differs in sampling only returns TRUE with the pixels and the YUV_sampling leaf as they were.
    return 1;
  }
  // premultiplied-alpha bookkeeping (:12290-12306), for every in / out palette pair
  int flags = l.flags;
  if (g_prefs.alpha_post) {
    if ((flags & LIVES_LAYER_ALPHA_PREMULT) && pal_has_alpha(inpl) && !pal_has_alpha(outpl)) {
      lives_gpu_alpha_premult(layer, LIVES_DIRECTION_REVERSE);
      if (!read_layer(layer, &l)) return 0;
      flags = l.flags;
    }
  } else if (!pal_has_alpha(inpl) && pal_has_alpha(outpl)) flags |= LIVES_LAYER_ALPHA_PREMULT;
  if (pal_has_alpha(inpl) && !pal_has_alpha(outpl)) flags &= ~LIVES_LAYER_ALPHA_PREMULT;

  if (!pal_is_rgb(outpl)) {
    if (pal_is_rgb(inpl)) return rgb_layer_to_yuv(layer, l, outpl, oclamping, osubspace, tgt_gamma, flags);
    return yuv_layer_repack(layer, l, outpl, l.clamping >= 0 ? l.clamping : oclamping, flags);   // YUV -> YUV repacks (K5b)
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
  const bool in_planar_sub = (inpl == WEED_PALETTE_YUV420P || inpl == WEED_PALETTE_YVU420P || inpl == WEED_PALETTE_YUV422P);
  if (!pal_is_rgb(inpl) && !in_planar_sub && k3_fmt(inpl) < 0 && inpl != WEED_PALETTE_YUV411) return decline(layer);
  if (lutp && !pal_is_rgb(inpl) && !in_planar_sub) return decline(layer);   // no inline gamma on the K3 / 4:1:1 paths

  if (!lutp && lazy_pal(inpl) && lazy_pal(outpl)) {
    // RGBA32 <-> BGRA32 on a pinned layer: recorded, not launched (deferred execution, above)
    if (Lazy *z = lazy_detach(l, LZ_SWAP)) {
      const int prev = z->stage;
      z->swap = true; z->stage = LZ_SWAP;
      if (!lazy_commit(layer, l, z, outpl, l.width, l.height, 0)) { z->swap = false; z->stage = prev; lazy_reattach(l.pd[0], z); return 0; }
      if (new_gamma != l.gamma) set_int(layer, WEED_LEAF_GAMMA_TYPE, new_gamma);
      if (flags != l.flags) set_int(layer, kLeafHostFlags, flags);
      g_api.leaf_delete(layer, WEED_LEAF_YUV_CLAMPING);
      g_api.leaf_delete(layer, WEED_LEAF_YUV_SUBSPACE);
      g_api.leaf_delete(layer, WEED_LEAF_YUV_SAMPLING);
      return 1;
    }
  }
  NewPlanes np;
  const int owidth = (inpl == WEED_PALETTE_UYVY || inpl == WEED_PALETTE_YUYV) ? l.width * 2 : inpl == WEED_PALETTE_YUV411 ? l.width * 4 : l.width;   // macropixels -> pixels (:13010, :13759)
  if (!alloc_planes(outpl, owidth, l.height, 0, &np)) return 0;
  const size_t obytes = (size_t)np.rs[0] * l.height;
  Work w;
  bool ok = true;
  if (pal_is_rgb(inpl)) {
    int af = 0;
    const int op = rgb_swizzle_op(inpl, outpl, &af);
    const uint8_t *d_in = w.in(l.pd[0], (size_t)l.rs[0] * l.height, 0);
    uint8_t *d_out = w.out(np.pd[0], obytes, 3, true);
    ok = w.ok && lgpu_swizzle(op, af, d_in, l.rs[0], d_out, np.rs[0], l.width, l.height, lutp, S()) == LGPU_OK;
  } else if (in_planar_sub) {
    const int iu = (inpl == WEED_PALETTE_YVU420P) ? 2 : 1, iv = (inpl == WEED_PALETTE_YVU420P) ? 1 : 2;   // swap_chroma_planes (:12353)
    const int ch = plane_h(l, 1);
    const size_t yb = (size_t)l.rs[0] * l.height, ub = (size_t)l.rs[iu] * ch, vb = (size_t)l.rs[iv] * ch;
    const uint8_t *dy = w.in(l.pd[0], yb, 0), *du = w.in(l.pd[iu], ub, 1), *dv = w.in(l.pd[iv], vb, 2);
    uint8_t *d_out = w.out(np.pd[0], obytes, 3, np.rs[0] != l.width * pal_psize(outpl));     // the kernel writes every byte of every pixel: only row padding needs the zeros
    const int strides[3] = {l.rs[0], l.rs[iu], l.rs[iv]};
    const int which = (iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | (l.subspace == WEED_YUV_SUBSPACE_BT709 ? 2 : 0);
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    ok = w.ok;
    if (ok && lutp) {
      // with a target gamma the reference fuses the 16-bit indexed LUT of create_gamma_lut into the conversion (:3274-3283)
      const uint16_t *d16 = device_lut16(l.gamma, new_gamma);
      ok = d16 && lgpu_yuv420p_to_rgb_lut16(dy, du, dv, strides, (long)ub, (long)vb, d_out, np.rs[0], l.width, l.height, pal_psize(outpl), order,
                                            inpl == WEED_PALETTE_YUV422P, which, g_prefs.pb_quality, d16, 0, S()) == LGPU_OK;
    } else if (ok)
      ok = lgpu_yuv420p_to_rgb(dy, du, dv, strides, (long)ub, (long)vb, d_out, np.rs[0], l.width, l.height, pal_psize(outpl), order,
                               inpl == WEED_PALETTE_YUV422P, which, g_prefs.pb_quality, nullptr, 0, S()) == LGPU_OK;
  } else if (k3_fmt(inpl) >= 0) {
    // K3: packed / planar 4:4:4, UYVY, YUYV -> RGB family (src/colourspace.c:12937-13860 cases); no inline gamma on these paths
    const int fmt = k3_fmt(inpl), in_alpha = (inpl == WEED_PALETTE_YUVA8888 || inpl == WEED_PALETTE_YUVA4444P);
    const int pxw = (fmt >= 2) ? l.width * 2 : l.width;                     // UYVY / YUYV layers count macropixels
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    const int which = (iclamping == WEED_YUV_CLAMPING_UNCLAMPED ? 1 : 0) | ((fmt == 0 && l.subspace == WEED_YUV_SUBSPACE_BT709) ? 2 : 0);
    if (fmt >= 2 && np.rs[0] < pxw * pal_psize(outpl)) { drop_new_planes(np); return decline(layer); }
    const uint8_t *dsrc[4] = {nullptr, nullptr, nullptr, nullptr};
    int irs[4] = {0, 0, 0, 0};
    for (int p = 0; p < l.nplanes; p++) { dsrc[p] = w.in(l.pd[p], (size_t)l.rs[p] * l.height, p == 0 ? 0 : p + 3); irs[p] = l.rs[p]; }
    uint8_t *d_out = w.out(np.pd[0], obytes, 3, true);
    ok = w.ok && lgpu_yuv_to_rgb(dsrc, irs, pxw, l.height, fmt, in_alpha, d_out, np.rs[0], order, pal_has_alpha(outpl) ? 1 : 0, which, S()) == LGPU_OK;
  } else {
    // K3b, YUV411 (:13755-13795): the reference walks the source as compact rows of `width` macropixels and leaves some alpha bytes of the
    // new (zeroed, create_empty_pixel_data) frame unwritten -- the device frame starts zeroed too
    const size_t ibytes = (size_t)l.width * 6 * l.height;
    const int order = pal_alpha_first(outpl) ? 2 : pal_red_first(outpl) ? 0 : 1;
    if ((size_t)l.rs[0] * l.height < ibytes) { drop_new_planes(np); return decline(layer); }
    const uint8_t *d_in = w.in(l.pd[0], ibytes, 0);
    uint8_t *d_out = w.out(np.pd[0], obytes, 3, true);
    ok = w.ok && lgpu_yuv411_to_rgb(d_in, l.width, l.height, d_out, np.rs[0], order, pal_has_alpha(outpl) ? 1 : 0,
                                   iclamping == WEED_YUV_CLAMPING_UNCLAMPED, S()) == LGPU_OK;
  }
  ok = ok && w.finish();
  if (!ok) { drop_new_planes(np); return 0; }                                // memfail: layer untouched
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

// src/colourspace.c:13935-13938
lives_gpu_boolean lives_gpu_convert_layer_palette_with_sampling(lives_gpu_layer_t *layer, int outpl, int out_sampling) {
  return lives_gpu_convert_layer_palette_full(layer, outpl, WEED_YUV_CLAMPING_UNCLAMPED, out_sampling, WEED_YUV_SUBSPACE_YUV, WEED_GAMMA_UNKNOWN);
}

static lives_gpu_boolean gamma_convert_sub_layer_body(int gamma_type, double fileg, lives_gpu_layer_t *layer, int x, int y, int width, int height);
lives_gpu_boolean lives_gpu_gamma_convert_sub_layer(int gamma_type, double fileg, lives_gpu_layer_t *layer, int x, int y, int width,
                                                    int height, lives_gpu_boolean may_thread) {
  (void)may_thread;
  PinScope pin(layer);
  return pin.settle(gamma_convert_sub_layer_body(gamma_type, fileg, layer, x, y, width, height));
}
static lives_gpu_boolean gamma_convert_sub_layer_body(int gamma_type, double fileg, lives_gpu_layer_t *layer, int x, int y, int width, int height) {
  if (!g_prefs.apply_gamma) return 1;
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  if (!pal_is_rgb(l.pal)) return decline(layer);
  if (gamma_type == l.gamma && fileg == 1.0) return 1;
  uint8_t lut[256];
  {
    // the calling thread's last table is kept, as the reference keeps its last one (:664-670): 255 powf pairs per call otherwise
    thread_local struct { double fg, sg; int from, to, made; uint8_t t[256]; } memo = {0., 0., 0, 0, -1, {0}};
    const double fg = gamma_type == LIVES_GAMMA_VARIANT ? fileg : 1.0;
    if (!(memo.made >= 0 && memo.fg == fg && memo.sg == g_prefs.screen_gamma && memo.from == l.gamma && memo.to == gamma_type)) {
      memo.made = lgpu_gamma_lut8(fg, l.gamma, gamma_type, g_prefs.screen_gamma, memo.t);
      memo.fg = fg; memo.sg = g_prefs.screen_gamma; memo.from = l.gamma; memo.to = gamma_type;
    }
    if (!memo.made) return 1;
    memcpy(lut, memo.t, 256);
  }
  if (x < 0 || y < 0 || x + width > l.width || y + height > l.height) return 0;
  if (x == 0 && y == 0 && width == l.width && height == l.height && lazy_pal(l.pal) && plane_is_lazy(l.pd[0])) {
    // the whole frame of a plane that is still a pending program: the table becomes its last stage
    if (Lazy *z = lazy_detach(l, LZ_LUT)) {
      if (z->stage != LZ_NONE) {
        z->lut = true; memcpy(z->lut8, lut, 256); z->stage = LZ_LUT;
        lazy_attach(l.pd[0], z);
        if (gamma_type != LIVES_GAMMA_VARIANT) set_int(layer, WEED_LEAF_GAMMA_TYPE, gamma_type);
        return 1;
      }
      lazy_reattach(l.pd[0], z);          // the program had a table already and has run: a table pass alone is one kernel, in place, below
    }
  }
  Work w;
  uint8_t *d = w.inout(l.pd[0], (size_t)l.rs[0] * l.height, 0);
  const bool ok = w.ok && lgpu_gamma_apply(d, l.rs[0], x, y, width, height, pal_psize(l.pal), pal_alpha_first(l.pal), lut, S()) == LGPU_OK && w.finish();
  if (!ok) return 0;
  if (gamma_type != LIVES_GAMMA_VARIANT) set_int(layer, WEED_LEAF_GAMMA_TYPE, gamma_type);
  return 1;
}

lives_gpu_boolean lives_gpu_gamma_convert_layer(int gamma_type, lives_gpu_layer_t *layer) {
  Layer l;
  if (!bound() || !read_layer(layer, &l)) return 0;
  return lives_gpu_gamma_convert_sub_layer(gamma_type, 1.0, layer, 0, 0, l.width, l.height, 1);   // :14146-14155
}

// gamma_convert_layer_variant (src/colourspace.c:14157-14168): the layer is tagged LINEAR, then taken to tgt_gamma through
// gamma_convert_sub_layer(tgt_gamma, file_gamma, ...): as there, file_gamma enters the table only when tgt_gamma is WEED_GAMMA_VARIANT
// (:14099-14102) -- kept
lives_gpu_boolean lives_gpu_gamma_convert_layer_variant(double file_gamma, int tgt_gamma, lives_gpu_layer_t *layer) {
  Layer l;
  if (!bound() || !read_layer(layer, &l)) return 0;
  set_int(layer, WEED_LEAF_GAMMA_TYPE, WEED_GAMMA_LINEAR);
  return lives_gpu_gamma_convert_sub_layer(tgt_gamma, file_gamma, layer, 0, 0, l.width, l.height, 1);
}

void lives_gpu_alpha_premult(lives_gpu_layer_t *layer, int direction) {
  PinScope pin(layer);
  Layer l;
  if (!ready() || !read_layer(layer, &l) || !pal_has_alpha(l.pal)) return;
  Work w;
  bool ok = true;
  if (l.pal == WEED_PALETTE_YUVA8888 || l.pal == WEED_PALETTE_YUVA4444P) {
    // :11982, :12005-12047, :12063-12096: clamped layers go through the alcy / alcuv / unalcy / unalcuv tables, unclamped ones through al / unal
    const int clamped = (l.clamping < 0 || l.clamping == WEED_YUV_CLAMPING_CLAMPED) ? 1 : 0;      // weed_layer_get_yuv_clamping(): CLAMPED (0) when the leaf is missing
    uint8_t *dp[4] = {nullptr, nullptr, nullptr, nullptr};
    int rs[4] = {0, 0, 0, 0};
    for (int p = 0; p < l.nplanes; p++) {
      const size_t nb = (size_t)l.rs[p] * l.height;
      dp[p] = (p == 3) ? const_cast<uint8_t *>(w.in(l.pd[p], nb, 7)) : w.inout(l.pd[p], nb, p);           // the alpha plane is only read
      rs[p] = l.rs[p];
    }
    ok = w.ok && lgpu_alpha_premult_yuva(dp, rs, l.width, l.height, l.pal, clamped, direction == LIVES_DIRECTION_REVERSE, S()) == LGPU_OK;
  } else {
    uint8_t *d = w.inout(l.pd[0], (size_t)l.rs[0] * l.height, 0);
    ok = w.ok && lgpu_alpha_premult(d, l.rs[0], l.width, l.height, pal_alpha_first(l.pal), direction == LIVES_DIRECTION_REVERSE, S()) == LGPU_OK;
  }
  if (!ok || !w.finish()) return;
  int flags = l.flags;
  if (direction == LIVES_DIRECTION_FORWARD) flags |= LIVES_LAYER_ALPHA_PREMULT; else flags &= ~LIVES_LAYER_ALPHA_PREMULT;   // :12098-12102
  set_int(layer, kLeafHostFlags, flags);
}

// resize every plane of `l` into freshly allocated planes of (width x height); lut8: the fused post-pass of :14718-14720 (RGB only)
static bool resize_into(const Layer &l, int width, int height, int interp, int alignment, const uint8_t *lut8, NewPlanes *np) {
  if (!alloc_planes(l.pal, width, height, alignment, np)) return false;
  Work w;
  Layer nl = l;
  nl.width = width; nl.height = height;
  bool ok = true;
  for (int p = 0; p < np->n && ok; p++) {
    const int ps = pal_is_planar_yuv(l.pal) ? 1 : pal_psize(l.pal);
    const int sw = plane_w(l, p), sh = plane_h(l, p), dw = plane_w(nl, p), dh = plane_h(nl, p);
    const uint8_t *d_in = w.in(l.pd[p], (size_t)l.rs[p] * sh, p == 3 ? 7 : p);          // scratch slots (ordinary layers): planes in 0, 1, 2, 7; planes out 3 .. 6
    uint8_t *d_out = w.out(np->pd[p], (size_t)np->rs[p] * dh, 3 + p, np->rs[p] != dw * ps);          // zeros for the row padding only
    ok = w.ok && lgpu_resize(d_in, l.rs[p], sw, sh, d_out, np->rs[p], dw, dh, ps, interp, lut8, S()) == LGPU_OK;
  }
  ok = ok && w.finish();
  if (!ok) drop_new_planes(*np);
  return ok;
}

// What resize_layer_full decides before it scales (shared with letterbox_layer, which scales straight into its canvas): the layer converted to a resizable
// palette if need be, the even source size, the adjusted target size, the fused gamma table.  Returns 1 = scale, 0 = failed / declined (*rc says which
// value the seam call returns), 2 = nothing to do.
struct ResizePlan { Layer l, src; int width, height, new_gamma; bool use_lut; uint8_t lut[256]; };
static int plan_resize(weed_plant_t *layer, int width, int height, int opal_hint, int osubs_hint, int tgt_gamma, ResizePlan *rp, int *rc) {
  Layer &l = rp->l;
  *rc = 0;
  if (!ready() || !read_layer(layer, &l)) return 0;
  // opal_hint / oclamp_hint "may be ignored ... layer palette should be checked on return" (:14746-14751): a frame whose palette this path
  // resizes keeps it (the caller's following convert_layer_palette does the rest); a packed-YUV frame is first taken to the hinted
  // palette when that one is resizable and the conversion is served here
  if (!(pal_is_rgb(l.pal) || pal_is_planar_yuv(l.pal))) {
    if (opal_hint == WEED_PALETTE_NONE || opal_hint == l.pal || !(pal_is_rgb(opal_hint) || pal_is_planar_yuv(opal_hint))) { *rc = decline(layer); return 0; }
    const int cl = l.clamping >= 0 ? l.clamping : WEED_YUV_CLAMPING_CLAMPED;
    if (!lives_gpu_convert_layer_palette_full(layer, opal_hint, cl, WEED_YUV_SAMPLING_DEFAULT, l.subspace, WEED_GAMMA_UNKNOWN)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  int iwidth = (l.width >> 1) << 1, iheight = (l.height >> 1) << 1;      // :14854-14863
  if (width < 4) width = 4;
  if (height < 4) height = 4;
  if (iwidth != width || iheight != height) height = (height >> 1) << 1;
  if (iwidth == width && iheight == height) { *rc = 1; return 2; }
  if (pal_is_planar_yuv(l.pal)) width = (width >> 1) << 1;
  // target gamma (:14890-14899, :15119-15127): applied as a LUT8 after the scaler when the output is RGB and the layer's gamma is known
  const int opal = (opal_hint == WEED_PALETTE_NONE || opal_hint == WEED_PALETTE_ANY) ? l.pal : opal_hint;
  if (tgt_gamma == WEED_GAMMA_UNKNOWN && !pal_is_rgb(opal) && osubs_hint == WEED_YUV_SUBSPACE_BT709) tgt_gamma = WEED_GAMMA_BT709;
  if (tgt_gamma == WEED_GAMMA_UNKNOWN && pal_is_rgb(l.pal) && !pal_is_rgb(opal)) tgt_gamma = WEED_GAMMA_SRGB;       // get_tgt_gamma :14733
  if (tgt_gamma == WEED_GAMMA_UNKNOWN) tgt_gamma = l.gamma;
  rp->use_lut = false;
  rp->new_gamma = l.gamma;
  if (tgt_gamma != WEED_GAMMA_UNKNOWN && pal_is_rgb(opal) && pal_is_rgb(l.pal)) {
    if (l.gamma != WEED_GAMMA_UNKNOWN && l.gamma != tgt_gamma && lgpu_gamma_lut8(1.0, l.gamma, tgt_gamma, g_prefs.screen_gamma, rp->lut)) rp->use_lut = true;
    rp->new_gamma = tgt_gamma;
  }
  rp->src = l;
  rp->src.width = iwidth; rp->src.height = iheight;
  rp->width = width; rp->height = height;
  return 1;
}

// ---- resize backend ------------------------------------------------------------------------------------------------------------------
// resize_layer_full has two bodies in the reference: the swscale one (:14940-15259; un-vendored, version unpinned -> this library's own
// polyphase spec) and the gdk-pixbuf one (:15262-15322: layer_to_pixbuf, lives_pixbuf_scale_simple, pixbuf_to_layer).
// The second is pinned byte for byte (pixbuf.hip) and is the DEFAULT: a host that links the seam and calls resize_layer / letterbox_layer as the
// reference does gets arithmetic a reference binary pins.  The polyphase body is the opt-in of a host "built with USE_SWSCALE".
static std::atomic<int> g_resize_backend{LIVES_GPU_RESIZE_PIXBUF};
int lives_gpu_set_resize_backend(int backend) {
  if (backend != LIVES_GPU_RESIZE_POLYPHASE && backend != LIVES_GPU_RESIZE_PIXBUF) return -1;
  g_resize_backend.store(backend);
  return 0;
}
int lives_gpu_get_resize_backend(void) { return g_resize_backend.load(); }

static bool pal_is_pixbuf(int pal) {      // the cases of the switch at :15275-15290 (3 or 4 channels, alpha last)
  return pal == WEED_PALETTE_RGB24 || pal == WEED_PALETTE_BGR24 || pal == WEED_PALETTE_RGBA32 || pal == WEED_PALETTE_BGRA32 ||
         pal == WEED_PALETTE_YUV888 || pal == WEED_PALETTE_YUVA8888;
}

// ---- the palette resolution in front of the resize bodies, and the planner's capability queries (src/colourspace.h:400-407) ------------------------------
// The reference answers these from its CPU rules; exported under the reference's names by liblivesgpu_dropin.so, they describe the bodies of THIS library, so
// that a planner which swapped the bodies plans for what will run (callers: src/nodemodel.c:131, :143, :210, :2961).
// weed_palette_is_resizable (:2647-2654) for the body in force.  PIXBUF: the switch of a build without swscale (weed_palette_conv_resizable, :2619-2643), which is
// also the switch of the gdk-pixbuf body itself (:15275-15290).  POLYPHASE: what that body scales as it comes (packed RGB, planar YUV).
static bool pal_resizable(int pal) {
  if (g_resize_backend.load() == LIVES_GPU_RESIZE_PIXBUF) return pal_is_pixbuf(pal);
  return pal_is_rgb(pal) || pal_is_planar_yuv(pal);
}
static int masq_pal(int pal) {                                         // get_masq_pal (:14500-14513)
  if (g_resize_backend.load() != LIVES_GPU_RESIZE_PIXBUF) return WEED_PALETTE_NONE;      // the polyphase body scales a palette as what it is or not at all
  if (pal == WEED_PALETTE_RGBA32 || pal == WEED_PALETTE_BGRA32 || pal == WEED_PALETTE_YUVA8888) return WEED_PALETTE_RGBA32;
  if (pal == WEED_PALETTE_RGB24 || pal == WEED_PALETTE_BGR24 || pal == WEED_PALETTE_YUV888) return WEED_PALETTE_RGB24;
  if (pal == WEED_PALETTE_YVU420P) return WEED_PALETTE_YUV420P;
  return WEED_PALETTE_NONE;
}
static int inter_pal(int inpal, int outpal, bool upscale) {            // get_inter_pal (:14516-14575)
  const bool both_alpha = pal_has_alpha(inpal) && pal_has_alpha(outpal), any_planar = pal_is_planar_yuv(inpal) || pal_is_planar_yuv(outpal);
  const int rgb = both_alpha ? WEED_PALETTE_RGBA32 : WEED_PALETTE_RGB24;
  const int yuv = any_planar ? (both_alpha ? WEED_PALETTE_YUVA4444P : WEED_PALETTE_YUV444P) : (both_alpha ? WEED_PALETTE_YUVA8888 : WEED_PALETTE_YUV888);
  if (pal_is_rgb(inpal) && pal_is_rgb(outpal)) return rgb;
  if (pal_is_yuv(inpal) && pal_is_yuv(outpal)) return yuv;
  // rgb <-> yuv (or a hint that is neither: ANY): convert before an upscale, after a downscale
  if (pal_is_rgb(inpal)) return upscale ? yuv : rgb;
  return upscale ? rgb : yuv;
}
// get_resizable (:14577-14669).  LIVES_RESULT_SUCCESS = 1, LIVES_RESULT_FAIL = 0 (src/defs.h:211-212).  Where the reference ends in LIVES_FATAL ("Unable to
// convert from palette ...": neither the intermediate palette nor a masquerade of it is resizable) this returns FAIL with the arguments as they came.
int lives_gpu_get_resizable(int *ppalette, int *pxpal, int *oclamp_hint, int *opal, int *pxopal, lives_gpu_boolean upscale) {
  if (!ppalette || !opal) return 0;
  int resolved = WEED_PALETTE_NONE;
  const int palette = *ppalette;
  int xpalette = palette, opal_hint = *opal, xopal_hint = opal_hint;
  const bool in_resizable = pal_resizable(palette), out_resizable = pal_resizable(opal_hint);
  if (in_resizable) {
    if (opal_hint != WEED_PALETTE_ANY) {
      if (out_resizable) resolved = palette;
      else if (upscale) {
        const int omasq = masq_pal(opal_hint);
        if (omasq != WEED_PALETTE_NONE) { resolved = opal_hint; xpalette = xopal_hint = omasq; }
      }
    }
    if (resolved == WEED_PALETTE_NONE) resolved = xopal_hint = opal_hint = xpalette = palette;
  } else if (out_resizable) {
    if (!upscale || opal_hint == WEED_PALETTE_ANY) {
      const int imasq = masq_pal(palette);
      if (imasq) { resolved = opal_hint = palette; xpalette = xopal_hint = imasq; }
    }
    if (resolved == WEED_PALETTE_NONE) resolved = xpalette = xopal_hint = opal_hint;
  } else {
    int imasq = resolved = inter_pal(palette, opal_hint, upscale != 0);
    if (resolved == WEED_PALETTE_NONE || !pal_resizable(resolved)) {
      if (resolved != WEED_PALETTE_NONE) imasq = masq_pal(resolved);
      if (imasq == WEED_PALETTE_NONE) return 0;                        // LIVES_FATAL in the reference
    }
    opal_hint = resolved;
    xpalette = xopal_hint = imasq;
  }
  if (resolved == WEED_PALETTE_NONE) return 0;
  *ppalette = resolved;
  *opal = opal_hint;
  if (pxpal) *pxpal = xpalette;
  if (pxopal) *pxopal = xopal_hint;
  if (oclamp_hint && pal_is_yuv(resolved) && pal_is_rgb(xpalette)) *oclamp_hint = WEED_YUV_CLAMPING_UNCLAMPED;
  return 1;
}
int lives_gpu_get_tgt_gamma(int ipal, int opal) { return (pal_is_rgb(ipal) && pal_is_yuv(opal)) ? WEED_GAMMA_SRGB : WEED_GAMMA_UNKNOWN; }      // :14736-14740
// can_inline_gamma (:12128-12145) for the conversions of this library: a target gamma is folded into the conversion kernel for RGB <-> RGB (LUT8 in the
// swizzle), 4:2:0 / 4:2:2 planar -> RGB (the 16-bit LUT of :3274-3283) and RGB -> UYVY / YUYV; every other pair takes its gamma as a separate pass (the
// reference's rule answers TRUE for more pairs than its own conversions honour).
lives_gpu_boolean lives_gpu_can_inline_gamma(int inpl, int opal) {
  if (pal_is_rgb(inpl) && pal_is_rgb(opal)) return 1;
  if ((inpl == WEED_PALETTE_YUV420P || inpl == WEED_PALETTE_YVU420P || inpl == WEED_PALETTE_YUV422P) && pal_is_rgb(opal)) return 1;
  if (pal_is_rgb(inpl) && (opal == WEED_PALETTE_UYVY || opal == WEED_PALETTE_YUYV)) return 1;
  return 0;
}
// pconv_can_inplace (:12148-12157): TRUE where the conversion leaves the layer's pixel_data where it is.  Here every conversion writes new planes (the old ones go
// back through the host's allocator) except the byte swap between the two packed 4:2:2 orders; the planner books one more frame for the others (src/nodemodel.c:2961).
lives_gpu_boolean lives_gpu_pconv_can_inplace(int inpl, int outpl) {
  return (inpl == WEED_PALETTE_UYVY && outpl == WEED_PALETTE_YUYV) || (inpl == WEED_PALETTE_YUYV && outpl == WEED_PALETTE_UYVY);
}

// The gdk-pixbuf form of resize_layer_full (src/colourspace.c:14759-14937 + :15262-15322): the value it returns.
//   * size rules (:14854-14868): even source size only for the "nothing to do" test, width / height >= 4, even target height
//   * the palette resolution every build runs BEFORE its body (:14869-14912): get_resizable picks the palette the frame is scaled in -- a palette outside the body's
//     switch is first converted (to the hinted palette when that one is resizable, else to an intermediate one), with the target-gamma decision of :14890-14899
//     handed to that conversion; the call ends FALSE where the reference does: no resizable route (its LIVES_FATAL), or the layer does not have the resolved
//     palette / the hinted clamping afterwards (:14916-14923).  Quirk R2 (docs/QUIRKS.md): weed_layer_get_yuv_clamping() answers 0 = CLAMPED for a layer
//     without the leaf, which is every RGB layer, so an RGB frame with oclamp_hint UNCLAMPED fails that test -- as in the reference (unletterbox_layer :15628 passes
//     exactly that)
//   * clamped YUV888 / YUVA8888 is switched to unclamped inside the body (:15277-15284)
//   * the WHOLE layer (odd sizes included: the sizes are re-read at :15263-15264) is scaled; 4-byte palettes weight colours by alpha
//   * the new frame has the pixbuf's rowstride, ALIGN4(width * channels), and RGB layers come back tagged WEED_GAMMA_SRGB (pixbuf_to_layer :14378-14379,
//     :14405-14406), whatever they were tagged before; no gamma LUT runs in this body
static int width_pixels(const Layer &l) {       // weed_layer_get_width_pixels: the width leaf counts macropixels
  return (l.pal == WEED_PALETTE_UYVY || l.pal == WEED_PALETTE_YUYV) ? l.width * 2 : l.pal == WEED_PALETTE_YUV411 ? l.width * 4 : l.width;
}
// lgpu_pixbuf_scale_check with the calling thread's last positive answer remembered: the tracks of a render ask for the same geometry tick after tick, and
// sixteen host threads queueing for the scaler's table cache lock every tick is what the answer would otherwise cost
static bool scale_is_served(int sw, int sh, int dw, int dh, int interp) {
  thread_local int last[5] = {0, 0, 0, 0, -1};
  if (last[0] == sw && last[1] == sh && last[2] == dw && last[3] == dh && last[4] == interp) return true;
  if (lgpu_pixbuf_scale_check(sw, sh, dw, dh, 4, interp, S()) != LGPU_OK) return false;
  last[0] = sw; last[1] = sh; last[2] = dw; last[3] = dh; last[4] = interp;
  return true;
}
static int resize_pixbuf_body(weed_plant_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint, int osamp_hint, int osubs_hint, int tgt_gamma) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  if (opal_hint == WEED_PALETTE_NONE) opal_hint = WEED_PALETTE_ANY;                  // :14822
  if (width <= 0 || height <= 0) return 0;
  int iwidth = (width_pixels(l) >> 1) << 1, iheight = (l.height >> 1) << 1;
  if (width < 4) width = 4;
  if (height < 4) height = 4;
  if (iwidth != width || iheight != height) height = (height >> 1) << 1;
  if (iwidth == width && iheight == height) return 1;
  const int palette = l.pal;
  int iclamping = l.clamping < 0 ? 0 : l.clamping;                                   // weed_layer_get_yuv_clamping: 0 without the leaf
  int resolved = palette, xpalette, xopal_hint;
  const int upscale = (long)width * height > (long)iwidth * iheight;
  if (!lives_gpu_get_resizable(&resolved, &xpalette, &oclamp_hint, &opal_hint, &xopal_hint, upscale)) {
    fprintf(stderr, "Unable to convert from palette %d to palette %d\n", palette, opal_hint);
    return decline(layer);
  }
  if (tgt_gamma == WEED_GAMMA_UNKNOWN && pal_is_yuv(opal_hint) && osubs_hint == WEED_YUV_SUBSPACE_BT709) tgt_gamma = WEED_GAMMA_BT709;      // :14890-14899
  if (tgt_gamma == WEED_GAMMA_UNKNOWN) tgt_gamma = lives_gpu_get_tgt_gamma(palette, opal_hint);
  if (tgt_gamma == WEED_GAMMA_UNKNOWN) tgt_gamma = l.gamma;
  if (tgt_gamma == WEED_GAMMA_BT709 && pal_is_yuv(opal_hint)) osubs_hint = WEED_YUV_SUBSPACE_BT709;
  if (resolved != palette || oclamp_hint != iclamping) {
    lives_gpu_convert_layer_palette_full(layer, resolved, oclamp_hint, osamp_hint, osubs_hint, tgt_gamma);      // :14907 (its value is not looked at there either)
    if (!read_layer(layer, &l)) return 0;
  }
  iclamping = l.clamping < 0 ? 0 : l.clamping;
  if (l.pal != resolved || iclamping != oclamp_hint) return decline(layer);          // :14916-14923
  iwidth = (width_pixels(l) >> 1) << 1; iheight = (l.height >> 1) << 1;
  if (iwidth == width && iheight == height) return 1;
  // ---- the body (:15262-15322)
  if (width_pixels(l) == width && l.height == height) return 1;                     // "no resize needed" (:15265-15270) comes before the switch, for every palette
  if (!pal_is_pixbuf(l.pal) || (interp != LIVES_INTERP_FAST && interp != LIVES_INTERP_NORMAL && interp != LIVES_INTERP_BEST)) {
    if (!pal_is_pixbuf(l.pal)) fprintf(stderr, "Warning: resizing unknown palette %d\n", l.pal);
    fprintf(stderr, "unable to scale layer to %d x %d for palette %d\n", width, height, l.pal);
    return decline(layer);
  }
  if ((l.pal == WEED_PALETTE_YUV888 || l.pal == WEED_PALETTE_YUVA8888) && iclamping == WEED_YUV_CLAMPING_CLAMPED) {
    if (!lives_gpu_convert_layer_palette(layer, l.pal, WEED_YUV_CLAMPING_UNCLAMPED)) return 0;
    if (!read_layer(layer, &l)) return 0;
  }
  if (l.width == width && l.height == height) return 1;
  const int ch = (l.pal == WEED_PALETTE_RGB24 || l.pal == WEED_PALETTE_BGR24 || l.pal == WEED_PALETTE_YUV888) ? 3 : 4;
  if (lazy_pal(l.pal)) {
    // a pinned RGBA32 / BGRA32 frame: the scale is recorded (deferred execution) once the scaler has said it takes the geometry
    if (Lazy *z = lazy_detach(l, LZ_SCALE)) {
      if (scale_is_served(l.width, l.height, width, height, interp)) {
        const int prev = z->stage;
        z->scale = true; z->dw = width; z->dh = height; z->interp = interp; z->stage = LZ_SCALE; z->opaque = has_leaf(layer, kLeafOpaque);
        if (!lazy_commit(layer, l, z, l.pal, width, height, 4)) { z->scale = false; z->opaque = false; z->stage = prev; lazy_reattach(l.pd[0], z); return 0; }
        if (l.gamma != WEED_GAMMA_SRGB) set_int(layer, WEED_LEAF_GAMMA_TYPE, WEED_GAMMA_SRGB);
        return 1;
      }
      lazy_reattach(l.pd[0], z);          // the eager path below answers (and declines what the scaler does not cover)
    }
  }
  NewPlanes np;
  if (!alloc_planes(l.pal, width, height, 4, &np)) return 0;
  Work w;
  const uint8_t *d_in = w.in(l.pd[0], (size_t)l.rs[0] * l.height, 0);
  uint8_t *d_out = w.out(np.pd[0], (size_t)np.rs[0] * height, 3, np.rs[0] != width * ch);
  const int rc = w.ok ? lgpu_pixbuf_scale(d_in, l.rs[0], l.width, l.height, d_out, np.rs[0], width, height, ch, interp | ((ch == 4 && has_leaf(layer, kLeafOpaque)) ? LGPU_INTERP_OPAQUE : 0), S()) : LGPU_E_HIP;
  if (rc != LGPU_OK || !w.finish()) {
    drop_new_planes(np);
    return rc == LGPU_E_UNSUPPORTED ? decline(layer) : 0;      // reductions past the library's one-step range: the host's own body takes them
  }
  free_planes(l);
  commit_planes(layer, l.pal, width, height, np);
  if (pal_is_rgb(l.pal) && l.gamma != WEED_GAMMA_SRGB) set_int(layer, WEED_LEAF_GAMMA_TYPE, WEED_GAMMA_SRGB);
  return 1;
}

// resize_layer_full (src/colourspace.c:14759-15328).  PIXBUF backend: the reference's own sequence (palette resolution, pre-conversion, gdk-pixbuf body).
// POLYPHASE backend (the opt-in standing where the swscale body stands): the layer keeps its palette when the scaler takes it as it is (packed RGB, planar YUV;
// "layer palette should be checked on return", :14746-14751), a packed-YUV frame is first taken to the hinted palette, and osamp_hint / osubs_hint take part in
// the target-gamma decision only (:14890-14899).
static lives_gpu_boolean resize_layer_full_body(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint, int osamp_hint, int osubs_hint, int tgt_gamma);
lives_gpu_boolean lives_gpu_resize_layer_full(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint,
                                              int osamp_hint, int osubs_hint, int tgt_gamma) {
  PinScope pin(layer);
  return pin.settle(resize_layer_full_body(layer, width, height, interp, opal_hint, oclamp_hint, osamp_hint, osubs_hint, tgt_gamma));
}
static lives_gpu_boolean resize_layer_full_body(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint, int osamp_hint, int osubs_hint, int tgt_gamma) {
  if (g_resize_backend.load() == LIVES_GPU_RESIZE_PIXBUF) return resize_pixbuf_body(layer, width, height, interp, opal_hint, oclamp_hint, osamp_hint, osubs_hint, tgt_gamma);
  ResizePlan rp;
  int rc;
  if (plan_resize(layer, width, height, opal_hint, osubs_hint, tgt_gamma, &rp, &rc) != 1) return rc;
  NewPlanes np;
  if (!resize_into(rp.src, rp.width, rp.height, interp, 16, rp.use_lut ? rp.lut : nullptr, &np)) return 0;       // rowstride_alignment_hint = 16 (:14989)
  free_planes(rp.l);
  commit_planes(layer, rp.l.pal, rp.width, rp.height, np);
  if (rp.new_gamma != rp.l.gamma) set_int(layer, WEED_LEAF_GAMMA_TYPE, rp.new_gamma);
  return 1;
}

lives_gpu_boolean lives_gpu_resize_layer(lives_gpu_layer_t *layer, int width, int height, int interp, int opal_hint, int oclamp_hint) {
  return lives_gpu_resize_layer_full(layer, width, height, interp, opal_hint, oclamp_hint, WEED_YUV_SAMPLING_DEFAULT,
                                     WEED_YUV_SUBSPACE_YCBCR, WEED_GAMMA_UNKNOWN);                        // :15331-15335
}

static lives_gpu_boolean letterbox_layer_body(lives_gpu_layer_t *layer, int nwidth, int nheight, int width, int height, int interp, int tpal, int tclamp);
lives_gpu_boolean lives_gpu_letterbox_layer(lives_gpu_layer_t *layer, int nwidth, int nheight, int width, int height, int interp,
                                            int tpal, int tclamp) {
  PinScope pin(layer);
  return pin.settle(letterbox_layer_body(layer, nwidth, nheight, width, height, interp, tpal, tclamp));
}
static lives_gpu_boolean letterbox_layer_body(lives_gpu_layer_t *layer, int nwidth, int nheight, int width, int height, int interp, int tpal, int tclamp) {
  if (!width || !height || !nwidth || !nheight) return 1;                 // :15377
  if (nwidth < width) nwidth = width;
  if (nheight < height) nheight = height;
  if (nheight == height && nwidth == width) { lives_gpu_resize_layer(layer, width, height, interp, tpal, tclamp); return 1; }
  // The inner frame: what resize_layer(layer, width, height, interp, tpal, tclamp) would leave (:15389) -- scaled STRAIGHT INTO the canvas when it has to
  // be scaled (the resized frame never exists on its own: one plane-sized write and read less per plane), blitted when it already has its size.
  ResizePlan rp;
  int rc, todo = 2;
  if (!ready() || !read_layer(layer, &rp.l)) return 0;
  if (g_resize_backend.load() == LIVES_GPU_RESIZE_PIXBUF && (width_pixels(rp.l) != width || rp.l.height != height)) {
    // the pixbuf body as the reference runs it: resize_layer first (:15389), then the blit of the frame it left
    if (!lives_gpu_resize_layer(layer, width, height, interp, tpal, tclamp)) return 0;
    if (!read_layer(layer, &rp.l)) return 0;
    width = rp.l.width; height = rp.l.height;
  }
  if (!(pal_is_rgb(rp.l.pal) || pal_is_planar_yuv(rp.l.pal)) && pal_psize(rp.l.pal) == 0) return decline(layer);   // packed YUV the blit has no pixel size for
  if (rp.l.width != width || rp.l.height != height) {                                     // resize_layer is only called for a frame of another size
    todo = plan_resize(layer, width, height, tpal, WEED_YUV_SUBSPACE_YUV, WEED_GAMMA_UNKNOWN, &rp, &rc);
    if (todo == 0) return rc;
  }
  const Layer &l = rp.l;
  Layer inner = l;                                       // the frame as it sits in the canvas
  if (todo == 1) { inner.width = rp.width; inner.height = rp.height; }
  width = inner.width; height = inner.height;
  if (nwidth < width || nheight < height) {              // cannot hold it (the reference asserts its sizes earlier): leave the layer resized, as the two calls did
    if (todo == 1) {
      NewPlanes rnp;
      if (!resize_into(rp.src, rp.width, rp.height, interp, 16, rp.use_lut ? rp.lut : nullptr, &rnp)) return 0;
      free_planes(l);
      commit_planes(layer, l.pal, rp.width, rp.height, rnp);
      if (rp.new_gamma != l.gamma) set_int(layer, WEED_LEAF_GAMMA_TYPE, rp.new_gamma);
    }
    return 0;
  }
  if (todo == 2 && lazy_pal(l.pal) && plane_is_lazy(l.pd[0])) {
    // the frame is a pending program (it has just been scaled): the canvas becomes its next stage
    if (Lazy *z = lazy_detach(l, LZ_CANVAS)) {
      const int prev = z->stage;
      z->canvas = true; z->nw = nwidth; z->nh = nheight; z->ox = (nwidth - width + 1) >> 1; z->oy = (nheight - height + 1) >> 1; z->stage = LZ_CANVAS;
      if (!lazy_commit(layer, l, z, l.pal, nwidth, nheight, 0)) { z->canvas = false; z->stage = prev; lazy_reattach(l.pd[0], z); return 0; }
      return 1;
    }
  }
  Layer canvas = inner;
  canvas.width = nwidth; canvas.height = nheight;
  NewPlanes np;
  if (!alloc_planes(l.pal, nwidth, nheight, 0, &np)) return 0;
  Work w;
  bool ok = true;
  const int offs_x = (nwidth - width + 1) >> 1, offs_y = (nheight - height + 1) >> 1;     // :15522-15523
  for (int p = 0; p < np.n && ok; p++) {
    const int ps = pal_is_planar_yuv(l.pal) ? 1 : pal_psize(l.pal);
    // chroma planes: offsets scaled by the plane ratio and truncated (:15553-15556)
    const int px = (plane_w(inner, p) == inner.width) ? offs_x : (int)(offs_x * 0.5), py = (plane_h(inner, p) == inner.height) ? offs_y : (int)(offs_y * 0.5);
    uint8_t black[4] = {0, 0, 0, 0};
    if (pal_is_planar_yuv(l.pal)) black[0] = (p == 0) ? (l.clamping == WEED_YUV_CLAMPING_UNCLAMPED ? 0 : 16) : 128;
    else if (pal_alpha_first(l.pal)) black[0] = 255;
    else if (pal_alpha_last(l.pal)) black[3] = 255;
    const int iw = plane_w(inner, p), ih = plane_h(inner, p), cw = plane_w(canvas, p), chh = plane_h(canvas, p);
    const Layer &from = todo == 1 ? rp.src : l;
    const int sw = plane_w(from, p), sh = plane_h(from, p);
    const uint8_t *d_in = w.in(l.pd[p], (size_t)l.rs[p] * plane_h(l, p), p == 3 ? 7 : p);
    uint8_t *d_out = w.out(np.pd[p], (size_t)np.rs[p] * chh, 3 + p, np.rs[p] != cw * ps);     // the canvas keeps its zeroed row padding (the kernels paint every pixel)
    if (!w.ok) { ok = false; break; }
    if (todo == 1)
      ok = lgpu_letterbox_bars(d_out, np.rs[p], cw, chh, ps, black, px, py, iw, ih, S()) == LGPU_OK &&
           lgpu_resize(d_in, l.rs[p], sw, sh, d_out + (size_t)py * np.rs[p] + (size_t)px * ps, np.rs[p], iw, ih, ps, interp, rp.use_lut ? rp.lut : nullptr, S()) == LGPU_OK;
    else
      ok = lgpu_letterbox_at(d_in, l.rs[p], sw, sh, d_out, np.rs[p], cw, chh, ps, black, px, py, S()) == LGPU_OK;
  }
  ok = ok && w.finish();
  if (!ok) { drop_new_planes(np); return 0; }
  free_planes(l);
  commit_planes(layer, l.pal, nwidth, nheight, np);
  if (todo == 1 && rp.new_gamma != l.gamma) set_int(layer, WEED_LEAF_GAMMA_TYPE, rp.new_gamma);
  return 1;
}

// unletterbox_layer (src/colourspace.c:15570-15628): cut the borders off, then resize to (opwidth, opheight) (0 = keep, -1 = the outer
// size).  Packed palettes only, as in the reference.  Quirk U1: the reference's row copy moves `xwidth` BYTES, not pixels
// (:15614), so only the first xwidth / psize pixels of every row arrive and the rest of the new (black, opaque) frame stays black: kept.
static lives_gpu_boolean unletterbox_layer_body(lives_gpu_layer_t *layer, int opwidth, int opheight, int top, int bottom, int left, int right);
lives_gpu_boolean lives_gpu_unletterbox_layer(lives_gpu_layer_t *layer, int opwidth, int opheight, int top, int bottom, int left, int right) {
  PinScope pin(layer);
  return pin.settle(unletterbox_layer_body(layer, opwidth, opheight, top, bottom, left, right));
}
static lives_gpu_boolean unletterbox_layer_body(lives_gpu_layer_t *layer, int opwidth, int opheight, int top, int bottom, int left, int right) {
  Layer l;
  if (!layer || !ready() || !read_layer(layer, &l)) return 0;
  if (top < 0) top = 0;
  if (bottom < 0) bottom = 0;
  if (left < 0) left = 0;
  if (right < 0) right = 0;
  if (!pal_is_rgb(l.pal)) return decline(layer);
  const int width = l.width, height = l.height, ps = pal_psize(l.pal);
  const int xwidth = width - left - right, xheight = height - top - bottom;
  if ((xwidth == width && xheight == height) || xwidth <= 0 || xheight <= 0) return 1;
  NewPlanes np;
  if (!alloc_planes(l.pal, xwidth, xheight, 0, &np)) return 0;
  uint8_t black[4] = {0, 0, 0, 0};
  if (pal_alpha_first(l.pal)) black[0] = 255; else if (pal_alpha_last(l.pal)) black[3] = 255;
  {
    Work w;
    const uint8_t *d_in = w.in(l.pd[0], (size_t)l.rs[0] * height, 0);
    uint8_t *d_out = w.out(np.pd[0], np.sz[0], 3, true);
    // the copied part of a row is xwidth bytes = xwidth / ps whole pixels (+ xwidth % ps bytes of the next one: a byte-granular blit)
    const bool ok = w.ok && lgpu_fill_pattern(d_out, np.rs[0], black, ps, xwidth, xheight, S()) == LGPU_OK &&          // create_empty_pixel_data(black_fill)
                    lgpu_copy_rows(d_out, np.rs[0], d_in + (size_t)top * l.rs[0] + (size_t)left * ps, l.rs[0], xwidth, xheight, S()) == LGPU_OK && w.finish();
    if (!ok) { drop_new_planes(np); return 0; }
  }
  free_planes(l);
  commit_planes(layer, l.pal, xwidth, xheight, np);
  if (opwidth == -1) opwidth = width; else if (!opwidth) opwidth = xwidth;
  if (opheight == -1) opheight = height; else if (!opheight) opheight = xheight;
  if (opwidth == xwidth && opheight == xheight) return 1;
  return lives_gpu_resize_layer(layer, opwidth, opheight, LIVES_INTERP_BEST, WEED_PALETTE_ANY, WEED_YUV_CLAMPING_UNCLAMPED);
}

// compact_rowstrides (src/colourspace.c:14439-14496): new pixel data whose rowstrides are exactly width * bytes per (macro)pixel * plane ratio
static lives_gpu_boolean compact_rowstrides_body(lives_gpu_layer_t *layer);
lives_gpu_boolean lives_gpu_compact_rowstrides(lives_gpu_layer_t *layer) {
  PinScope pin(layer);
  return pin.settle(compact_rowstrides_body(layer));
}
static lives_gpu_boolean compact_rowstrides_body(lives_gpu_layer_t *layer) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return 0;
  int bpm = pal_psize(l.pal);                                               // bytes per macropixel of plane 0
  if (l.pal == WEED_PALETTE_UYVY || l.pal == WEED_PALETTE_YUYV || l.pal == WEED_PALETTE_YUVA8888) bpm = 4;
  else if (l.pal == WEED_PALETTE_YUV888) bpm = 3;
  else if (l.pal == WEED_PALETTE_YUV411) bpm = 6;
  if (!bpm) return decline(layer);
  NewPlanes np;
  np.n = l.nplanes;
  bool change = false;
  size_t tot = 0;
  for (int p = 0; p < l.nplanes; p++) {
    np.rs[p] = plane_w(l, p) * bpm;
    np.sz[p] = (size_t)np.rs[p] * plane_h(l, p);
    tot += np.sz[p];
    if (np.rs[p] != l.rs[p]) change = true;
  }
  if (!change) return 1;
  uint8_t *blk = (uint8_t *)palloc(tot + 64, false);
  if (!blk) return 0;
  memset(blk + tot, 0, 64);
  size_t off = 0;
  for (int p = 0; p < np.n; p++) { np.pd[p] = blk + off; off += np.sz[p]; }
  {
    Work w;
    bool ok = true;
    for (int p = 0; p < np.n && ok; p++) {
      const uint8_t *d_in = w.in(l.pd[p], (size_t)l.rs[p] * plane_h(l, p), p == 3 ? 7 : p);
      uint8_t *d_out = w.out(np.pd[p], np.sz[p], 3 + p, false);
      ok = w.ok && lgpu_copy_rows(d_out, np.rs[p], d_in, l.rs[p], np.rs[p], plane_h(l, p), S()) == LGPU_OK;
    }
    if (!ok || !w.finish()) { pfree(blk); return 0; }
  }
  free_planes(l);
  commit_planes(layer, l.pal, l.width, l.height, np);
  return 1;
}

// weed_layer_clear_pixel_data (src/colourspace.c:11229-11244): the frame painted black in place (blank_frame :11212-11226): host
// bytes of an ordinary layer (a fill of host memory is the host's business), the resident planes of a pinned one.  Packed palettes
// keep the reference's per-pixel pattern (opaque alpha); YUYV keeps its quirk (blank_pixel :11150-11154 never advances: only the first
// macropixel of a row is written).
static lives_gpu_boolean weed_layer_clear_pixel_data_body(lives_gpu_layer_t *layer);
lives_gpu_boolean lives_gpu_weed_layer_clear_pixel_data(lives_gpu_layer_t *layer) {
  PinScope pin(layer);
  return pin.settle(weed_layer_clear_pixel_data_body(layer));
}
static lives_gpu_boolean weed_layer_clear_pixel_data_body(lives_gpu_layer_t *layer) {
  Layer l;
  if (!layer || !bound() || !read_layer(layer, &l)) return 0;
  const int clamping = l.clamping >= 0 ? l.clamping : WEED_YUV_CLAMPING_CLAMPED;       // weed_layer_get_palette_yuv: CLAMPED when the leaf is missing
  const uint8_t yb = clamping == WEED_YUV_CLAMPING_UNCLAMPED ? 0 : 16;
  uint8_t pat[8];
  int plen = 0, nmp = l.width;                    // pattern of one (macro)pixel of plane 0, macropixels per row
  switch (l.pal) {
  case WEED_PALETTE_RGB24: case WEED_PALETTE_BGR24: pat[0] = pat[1] = pat[2] = 0; plen = 3; break;
  case WEED_PALETTE_RGBA32: case WEED_PALETTE_BGRA32: pat[0] = pat[1] = pat[2] = 0; pat[3] = 255; plen = 4; break;
  case WEED_PALETTE_ARGB32: pat[0] = 255; pat[1] = pat[2] = pat[3] = 0; plen = 4; break;
  case WEED_PALETTE_UYVY: pat[0] = pat[2] = 128; pat[1] = pat[3] = yb; plen = 4; break;
  case WEED_PALETTE_YUYV: pat[0] = pat[2] = yb; pat[1] = pat[3] = 128; plen = 4; nmp = 1; break;
  case WEED_PALETTE_YUV888: pat[0] = yb; pat[1] = pat[2] = 128; plen = 3; break;
  case WEED_PALETTE_YUVA8888: pat[0] = yb; pat[1] = pat[2] = 128; pat[3] = 255; plen = 4; break;
  case WEED_PALETTE_YUV411: pat[0] = pat[3] = 128; pat[1] = pat[2] = pat[4] = pat[5] = yb; plen = 6; break;
  default: break;
  }
  if (!plen && !pal_is_planar_yuv(l.pal)) return 0;
  const bool on_device = t_pinned && ready();
  for (int p = 0; p < l.nplanes; p++) {
    const int ph = plane_h(l, p), pw = plane_w(l, p);
    uint8_t one[1] = {(uint8_t)(p == 0 ? yb : p == 3 ? 255 : 128)};
    const uint8_t *pp = plen ? pat : one;
    const int pl = plen ? plen : 1, n = plen ? nmp : pw;
    uint8_t *d = on_device ? resident(l.pd[p], (size_t)l.rs[p] * ph, true) : nullptr;
    if (d) {
      const int rc = lgpu_fill_pattern(d, l.rs[p], pp, pl, n, ph, S());
      touch_done(l.pd[p], true);
      if (rc != LGPU_OK) return 0;
      continue;
    }
    for (int y = 0; y < ph; y++) {
      uint8_t *row = l.pd[p] + (size_t)y * l.rs[p];
      if (pl == 1) memset(row, pp[0], (size_t)n);
      else for (int x = 0; x < n; x++) memcpy(row + (size_t)x * pl, pp, (size_t)pl);
    }
  }
  return 1;
}

// ---- device residency API (seam 2) ---------------------------------------------------------------------------
static size_t plane_bytes(const Layer &l, int p) {
  const bool planar = pal_is_planar_yuv(l.pal);
  const int h = (!planar || p == 0 || p == 3 || pal_is_444(l.pal) || l.pal == WEED_PALETTE_YUV422P) ? l.height : l.height >> 1;
  return (size_t)l.rs[p] * h;
}
// this thread's stream has just been synchronised: planes whose last use was enqueued on it have no work pending, whoever touches them next need not wait
static void settle(const Layer &l) {
  for (int p = 0; p < l.nplanes; p++) {
    ResShard &sh = shard_of(l.pd[p]);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(l.pd[p]);
    if (it == sh.m.end()) continue;
    Dev &e = it->second;
    if (e.stream == S()) e.stream = kIdle;
    int k = 0;
    for (int i = 0; i < e.nr; i++) if (e.rs[i] != S()) e.rs[k++] = e.rs[i];
    e.nr = k;
  }
}
int lives_gpu_layer_pin(lives_gpu_layer_t *layer) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return LGPU_E_BADARG;
  if (has_leaf(layer, kLeafResident)) return LGPU_OK;
  for (int p = 0; p < l.nplanes; p++) {
    const size_t n = plane_bytes(l, p);
    Dev b;
    if (!pool_take(n, &b)) { for (int q = 0; q < p; q++) res_drop(l.pd[q]); return LGPU_E_NOMEM; }
    g_h2d += n;
    if (lgpu_upload(b.d, l.pd[p], n, S()) != LGPU_OK) {
      pool_give(b);
      for (int q = 0; q < p; q++) res_drop(l.pd[q]);
      return LGPU_E_HIP;
    }
    res_put(l.pd[p], b);
  }
  if (!sync()) return LGPU_E_HIP;             // the host may change or free its bytes once pin returns
  settle(l);
  set_int(layer, kLeafResident, 1);
  return LGPU_OK;
}
// A layer whose planes ALREADY are in device memory the caller owns (a hardware decoder's output surfaces, the frame a previous pass left in HBM): the layer is
// pinned with those buffers as its resident planes -- no upload, no copy.  The library reads them where they lie (in-place calls write them) and never frees them;
// they must stay valid until the layer's pixel_data has been replaced by a seam call, or the layer is synchronised / unpinned / forgotten.  producer_stream: the
// stream their contents were produced on (NULL = the null stream; pass lives_gpu_thread_stream() or any stream already synchronised for "complete").
int lives_gpu_layer_pin_device(lives_gpu_layer_t *layer, const void *const *planes_d, int nplanes, void *producer_stream, int producer_done) {
  Layer l;
  if (!ready() || !read_layer(layer, &l) || !planes_d || nplanes != l.nplanes) return LGPU_E_BADARG;
  if (has_leaf(layer, kLeafResident)) return LGPU_E_BADARG;
  for (int p = 0; p < l.nplanes; p++) if (!planes_d[p]) return LGPU_E_BADARG;
  for (int p = 0; p < l.nplanes; p++) {
    Dev b;
    b.d = const_cast<void *>(planes_d[p]); b.bytes = plane_bytes(l, p); b.external = true;
    b.stream = producer_done ? kIdle : producer_stream;
    lazy_run_readers_of(l.pd[p]);
    Dev old;
    {
      ResShard &sh = shard_of(l.pd[p]);
      std::lock_guard<SpinLock> lk(sh.mu);
      Dev &slot = sh.m[l.pd[p]];
      old = slot;
      slot = b;
    }
    if (old.d || old.lazy) pool_give(old);
  }
  set_int(layer, kLeafResident, 1);
  return LGPU_OK;
}
// weed_layer_copy(NULL, slayer) -- the deep copy (src/layers.c:755-846: a new layer, copy_pixel_data_slice) -- copies HOST bytes, which are stale while slayer is
// pinned.  Called on its result, this gives the copy the pixels: dlayer (the same palette, size and rowstrides, host planes of its own) becomes a pinned layer whose
// device planes are device-to-device copies of slayer's (a pending program of slayer runs first); no byte crosses PCIe.  slayer not pinned: nothing to do.
int lives_gpu_layer_copy(lives_gpu_layer_t *dlayer, lives_gpu_layer_t *slayer) {
  Layer s, d;
  if (!ready() || !read_layer(slayer, &s) || !read_layer(dlayer, &d)) return LGPU_E_BADARG;
  if (!has_leaf(slayer, kLeafResident)) return LGPU_OK;
  if (has_leaf(dlayer, kLeafResident) || s.nplanes != d.nplanes || s.pal != d.pal || s.width != d.width || s.height != d.height) return LGPU_E_BADARG;
  for (int p = 0; p < s.nplanes; p++)
    if (s.rs[p] != d.rs[p] || plane_bytes(s, p) != plane_bytes(d, p) || s.pd[p] == d.pd[p]) return LGPU_E_BADARG;      // (a shallow copy shares the planes and their device copies already)
  for (int p = 0; p < s.nplanes; p++) {
    const size_t n = plane_bytes(s, p);
    const uint8_t *src = acquire(s.pd[p], n, false);
    Dev b;
    if (!src || !pool_take(n, &b)) { for (int q = 0; q < p; q++) res_drop(d.pd[q]); return src ? LGPU_E_NOMEM : LGPU_E_BADARG; }
    if (lgpu_copy(b.d, src, n, S()) != LGPU_OK) { pool_give(b); for (int q = 0; q < p; q++) res_drop(d.pd[q]); return LGPU_E_HIP; }
    touch_done(s.pd[p], false);
    b.stream = S();
    res_put(d.pd[p], b);
  }
  set_int(dlayer, kLeafResident, 1);
  return LGPU_OK;
}
// The host's word that every pixel of the layer's frame has alpha 255 (decoded video, a frame that was RGB24 / YUV before): the scalers then run their all-opaque
// instantiations (LGPU_INTERP_OPAQUE: same bytes on such frames, 25-30 % less time for enlargements).  The word is the host's to keep true: it stays on the layer
// through the seam's own calls (a scale, letterbox, R <-> B, gamma or chroma blend of an opaque frame is opaque) until the host takes it back (on = 0) or hands
// the layer a new frame.  A frame that is not opaque gets wrong colours.
int lives_gpu_layer_set_opaque(lives_gpu_layer_t *layer, int on) {
  if (!layer || !bound()) return LGPU_E_BADARG;
  if (on) set_int(layer, kLeafOpaque, 1);
  else if (has_leaf(layer, kLeafOpaque)) g_api.leaf_delete(layer, kLeafOpaque);
  return LGPU_OK;
}
// deferred execution (see "deferred execution on pinned layers" above): on (default) / off; returns the previous setting
int lives_gpu_set_deferred(int on) { return g_deferred.exchange(on ? 1 : 0); }
// counters since the library was loaded: [0] stages recorded, [1] fused chain launches made for pending programs, [2] tracks (programs) those launches carried,
// [3] programs run stage by stage (shapes the fused kernel does not take)
void lives_gpu_deferred_stats(unsigned long long out[4]) {
  if (!out) return;
  out[0] = g_lz_recorded.load(); out[1] = g_lz_chain_launches.load(); out[2] = g_lz_chain_tracks.load(); out[3] = g_lz_staged.load();
}
// Run the pending programs of these layers now, on the calling thread's stream: programs of equal shape (the tracks of one plan step) share ONE launch of the
// fused chain kernel.  The layers stay pinned, nothing is downloaded, the host does not wait.  What a host calls once per tick when the plan steps of its tracks
// have returned (src/nodemodel.c:2027-2101 runs them on pool threads and collects them); without it every program still runs, by itself, when its pixels are needed.
int lives_gpu_layers_flush(lives_gpu_layer_t *const *layers, int nlayers) {
  if (!ready() || (nlayers > 0 && !layers)) return LGPU_E_BADARG;
  std::vector<const void *> hs;
  for (int i = 0; i < nlayers; i++) {
    Layer l;
    if (!layers[i] || !read_layer(layers[i], &l)) continue;
    for (int p = 0; p < l.nplanes; p++) hs.push_back(l.pd[p]);
  }
  std::lock_guard<std::mutex> run(g_lazy_mu);
  std::vector<Lazy *> zs(hs.size(), nullptr);
  for (size_t i = 0; i < hs.size(); i++) {
    ResShard &sh = shard_of(hs[i]);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(hs[i]);
    if (it != sh.m.end()) zs[i] = it->second.lazy;
  }
  int rc = LGPU_OK;
  std::vector<char> done(hs.size(), 0);
  for (size_t i = 0; i < hs.size(); i++) {
    if (!zs[i] || done[i]) continue;
    std::vector<Lazy *> gz;
    std::vector<const void *> gh;
    for (size_t k = i; k < hs.size() && (int)gz.size() < LGPU_CHAIN_MAX_TRACKS; k++)
      if (zs[k] && !done[k] && lazy_same_shape(zs[i], zs[k])) {
        bool dup = false;
        for (Lazy *q : gz) dup = dup || q == zs[k];
        done[k] = 1;
        if (!dup) { gz.push_back(zs[k]); gh.push_back(hs[k]); }
      }
    const int r = lazy_run_group(gz.data(), gh.data(), (int)gz.size());
    if (r && !rc) rc = r;
  }
  return rc;
}
// livesgpu_fx.so's "chroma blend" on a plane that is a pending program (in place: out channel = in channel 0): the blend with layer 2 is recorded as the program's
// next stage.  1 = recorded, the effect has nothing left to do; 0 = not applicable, the effect runs its kernel (acquiring the planes runs what is pending).
int lives_gpu_deferred_blend_chroma(const void *dst_host, int orow, int width, int height, int palette, const void *layer2_host, int irow2, int bf) {
  if (!g_deferred.load(std::memory_order_relaxed) || !dst_host || !layer2_host || dst_host == layer2_host || !lazy_pal(palette) || (irow2 & 3) || !ready()) return 0;
  {
    ResShard &sh = shard_of(dst_host);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(dst_host);
    if (it == sh.m.end() || !it->second.lazy || it->second.lazy_readers > 0) return 0;
    const Lazy *z = it->second.lazy;
    if (z->stage >= LZ_BLEND || z->w != width || z->h != height || z->rs != orow) return 0;
  }
  const size_t n2 = (size_t)irow2 * height;
  uint8_t *l2d = acquire(layer2_host, n2, false);          // layer 2 itself may be pending: it runs; this thread's stream is behind its writer
  if (!l2d) return 0;
  // layer 2 first (its own shard): it becomes a plane a pending program reads; then the program takes the stage -- or, should the plane have changed hands in
  // between (it cannot under the host's own ordering), layer 2 is let go again
  Dev l2copy;
  {
    ResShard &sh = shard_of(layer2_host);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto l2 = sh.m.find(layer2_host);
    if (l2 == sh.m.end() || !l2->second.d) return 0;
    l2copy = l2->second;
    l2->second.lazy_readers++;
  }
  bool taken = false;
  {
    ResShard &sh = shard_of(dst_host);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(dst_host);
    if (it != sh.m.end() && it->second.lazy && it->second.lazy->stage < LZ_BLEND) {
      Lazy *z = it->second.lazy;
      g_lz_recorded++;
      z->blend = true; z->bf = bf & 0xFF; z->l2h = layer2_host; z->l2 = l2copy; z->l2rs = irow2; z->stage = LZ_BLEND;
      taken = true;
    }
  }
  if (!taken) {
    ResShard &sh = shard_of(layer2_host);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto l2 = sh.m.find(layer2_host);
    if (l2 != sh.m.end() && l2->second.lazy_readers > 0) l2->second.lazy_readers--;
    return 0;
  }
  return 1;
}
int lives_gpu_layer_sync(lives_gpu_layer_t *layer) {
  Layer l;
  if (!ready() || !read_layer(layer, &l)) return LGPU_E_BADARG;
  if (!has_leaf(layer, kLeafResident)) return LGPU_OK;          // never pinned: the host bytes are the planes
  for (int p = 0; p < l.nplanes; p++) {
    const size_t n = plane_bytes(l, p);
    Dev b;
    if (plane_is_lazy(l.pd[p]) && !lazy_materialise(l.pd[p])) return LGPU_E_HIP;        // a pending program runs now
    {
      ResShard &sh = shard_of(l.pd[p]);
      std::lock_guard<SpinLock> lk(sh.mu);
      auto it = sh.m.find(l.pd[p]);
      if (it != sh.m.end() && it->second.bytes >= n) { b = it->second; note_use(it->second, false); }
    }
    if (!b.d) continue;                                 // this plane's host bytes are current
    await(b, false);                                    // behind the work of whichever thread wrote it last
    g_d2h += n;
    if (lgpu_download(l.pd[p], b.d, n, S()) != LGPU_OK) return LGPU_E_HIP;
  }
  if (!sync()) return LGPU_E_HIP;
  settle(l);
  return LGPU_OK;
}
}  // extern "C"
namespace {
int lives_gpu_layer_unpin_impl(weed_plant_t *layer) {
  const int rc = lives_gpu_layer_sync(layer);
  Layer l;
  if (read_layer(layer, &l)) for (int p = 0; p < l.nplanes; p++) res_drop(l.pd[p]);
  if (bound() && layer) g_api.leaf_delete(layer, kLeafResident);
  return rc;
}
}  // namespace
extern "C" {
int lives_gpu_layer_unpin(lives_gpu_layer_t *layer) { return lives_gpu_layer_unpin_impl(layer); }
// the host is about to free or replace this layer's pixel_data itself (weed_layer_pixel_data_free, an error path, a CPU body that
// allocates new planes): drop the device copies WITHOUT bringing them home.  After this no table entry refers to the layer's host
// pointers, so a later allocation at the same address cannot be mistaken for a resident plane.
int lives_gpu_layer_forget(lives_gpu_layer_t *layer) {
  Layer l;
  if (!bound() || !layer) return LGPU_E_BADARG;
  if (read_layer(layer, &l)) for (int p = 0; p < l.nplanes; p++) res_drop(l.pd[p]);
  g_api.leaf_delete(layer, kLeafResident);
  return LGPU_OK;
}
// optional frame allocator pair for lives_gpu_weed_api.pixel_alloc / pixel_free: page-locked, zeroed host memory, so the planes the seam creates
// (and any frame the host allocates through it) cross PCIe by DMA at link rate instead of through the staging chunks.  Freeing a plane through it
// also drops a device copy registered under that address.
static std::mutex g_pinned_mu;
static std::unordered_map<const void *, size_t> g_pinned_blocks;      // blocks handed out by lives_gpu_pinned_calloc: base -> bytes (for the range drop on free)
void *lives_gpu_pinned_calloc(size_t bytes) {
  void *p = lgpu_pinned_calloc(bytes);
  if (p) { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned_blocks[p] = bytes ? bytes : 1; }
  return p;
}
void lives_gpu_pinned_free(void *p) {
  size_t bytes = 1;
  if (p) {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto it = g_pinned_blocks.find(p);
    if (it != g_pinned_blocks.end()) { bytes = it->second; g_pinned_blocks.erase(it); }
  }
  if (p) res_drop_range(p, bytes);
  lgpu_pinned_free(p);
}

// residency bridge for the weed plugin (same library, other seam): the device copy of a pinned layer's plane, looked up by the host plane
// pointer the channel carries; NULL when the plane is not resident (or smaller than asked).  Entries exist only between lives_gpu_layer_pin
// and unpin / forget / the release of the plane through this library (free_planes, lives_gpu_pinned_free).
void *lives_gpu_resident_lookup(const void *host_plane, size_t min_bytes) {
  if (!host_plane) return nullptr;
  Dev b;
  if (plane_is_lazy(host_plane) && !lazy_materialise(host_plane)) return nullptr;
  lazy_run_readers_of(host_plane);                    // the caller may write the plane
  {
    ResShard &sh = shard_of(host_plane);
    std::lock_guard<SpinLock> lk(sh.mu);
    auto it = sh.m.find(host_plane);
    if (it == sh.m.end() || it->second.bytes < min_bytes) return nullptr;
    b = it->second;
    it->second.stream = nullptr; it->second.nr = 0;
  }
  // the caller works on the NULL stream: hand the plane over to it (the null stream waits for the plane's writer and readers; the next seam call on a
  // thread's stream will in turn wait for the null stream)
  void *users[5] = {b.stream, b.rs[0], b.rs[1], b.rs[2], b.rs[3]};
  for (int i = 0; i < 1 + b.nr; i++) {
    if (users[i] == nullptr || users[i] == kIdle) continue;
    if (!t_event && lgpu_event_create(&t_event) != LGPU_OK) t_event = nullptr;
    if (!(t_event && lgpu_event_record(t_event, users[i]) == LGPU_OK && lgpu_stream_wait_event(nullptr, t_event) == LGPU_OK)) lgpu_sync(users[i]);
  }
  return b.d;
}
// The same for callers that enqueue on the CALLING THREAD's stream (lives_gpu_thread_stream; what livesgpu_fx.so does): acquire orders that stream behind the
// plane's writer (and, for a write, its readers), release records the use once the caller's work is enqueued.
void *lives_gpu_thread_stream(void) { return ready() ? S() : nullptr; }
void lives_gpu_stream_follow(void *other) { if (ready()) follow(other); }
void *lives_gpu_resident_acquire(const void *host_plane, size_t min_bytes, int write) {
  if (!host_plane || !ready()) return nullptr;
  return acquire(host_plane, min_bytes, write != 0);
}
void lives_gpu_resident_release(const void *host_plane, int write) {
  if (host_plane) touch_done(host_plane, write != 0);
}
void lives_gpu_transfer_stats(unsigned long long *h2d_bytes, unsigned long long *d2h_bytes) {
  if (h2d_bytes) *h2d_bytes = g_h2d.load();
  if (d2h_bytes) *d2h_bytes = g_d2h.load();
}

}  // extern "C"
