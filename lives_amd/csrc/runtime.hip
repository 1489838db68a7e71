// runtime.hip -- device init, error string, memory helpers, resident conversion tables.
#include "lgpu_common.h"
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>
#include <string.h>
#include <stdlib.h>

namespace lgpu {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static std::atomic<bool> g_inited[64];     // release after the tables are published, acquire before they are read
static DeviceTables g_tables[64];

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

static int current_device() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) return -1;
  return d;
}

static int init_device(int dev) {
  // the common case -- an initialised device that is already the calling thread's current one -- takes no lock and makes no call that does:
  // every seam call comes through here, and sixteen host threads queueing for g_mu around hipGetDeviceCount + hipSetDevice was measurable
  if (dev >= 0 && dev < 64 && g_inited[dev].load(std::memory_order_acquire) && current_device() == dev) return LGPU_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device available (this library has no CPU fallback)");
    return LGPU_E_NODEVICE;
  }
  if (dev < 0 || dev >= n || dev >= 64) { set_error("device %d out of range (%d devices)", dev, n); return LGPU_E_BADARG; }
  LGPU_HIP(hipSetDevice(dev));
  if (g_inited[dev].load(std::memory_order_acquire)) return LGPU_OK;
  hipDeviceProp_t prop;
  LGPU_HIP(hipGetDeviceProperties(&prop, dev));
  if (!strstr(prop.gcnArchName, "gfx950")) {
    set_error("device %d is %s; this build carries gfx950 (MI355X) code objects only", dev, prop.gcnArchName);
    return LGPU_E_NODEVICE;
  }
  // all tables in ONE allocation: a failure part of the way through leaves nothing behind
  constexpr size_t kR2Y = 9 * 256, kY2R = 5 * 256, kLuma = 3 * 256, kTotal = 4 * (kR2Y + kY2R) + kLuma;
  std::vector<int32_t> host(kTotal);
  for (int which = 0; which < 4; which++) lgpu_conversion_tables(which, host.data() + which * (kR2Y + kY2R), host.data() + which * (kR2Y + kY2R) + kR2Y);
  {
    // luma weights of calc_luma(): myround(k * i * 65536.) per channel
    int32_t *lw = host.data() + 4 * (kR2Y + kY2R);
    for (int i = 0; i < 256; i++) {
      const double v = (double)i;
      const double r = 0.299 * v * 65536., g = (1. - 0.299 - 0.114) * v * 65536., b = 0.114 * v * 65536.;
      lw[i] = (int32_t)(r + 0.5); lw[256 + i] = (int32_t)(g + 0.5); lw[512 + i] = (int32_t)(b + 0.5);
    }
  }
  int32_t *d = nullptr;
  if (hipMalloc((void **)&d, kTotal * sizeof(int32_t)) != hipSuccess) { set_error("hipMalloc of the conversion tables failed"); return LGPU_E_NOMEM; }
  if (hipMemcpy(d, host.data(), kTotal * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); set_error("upload of the conversion tables failed"); return LGPU_E_HIP; }
  for (int which = 0; which < 4; which++) { g_tables[dev].rgb2yuv[which] = d + which * (kR2Y + kY2R); g_tables[dev].yuv2rgb[which] = d + which * (kR2Y + kY2R) + kR2Y; }
  g_tables[dev].luma = d + 4 * (kR2Y + kY2R);
  {
    // lgpu_malloc_ordered / lgpu_free_ordered: the device's default stream-ordered pool keeps up to 4 GB of freed blocks instead of returning them to the
    // driver at every synchronisation (hipMalloc + hipFree of a frame cost ~190 us on an idle device and hipFree waits for the device to drain:
    // tools/alloc_probe.hip measured 9 ms behind a busy stream, against 7 us for the pooled pair)
    hipMemPool_t mp = nullptr;
    uint64_t keep = 4ull << 30;
    if (hipDeviceGetDefaultMemPool(&mp, dev) != hipSuccess || hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess) (void)hipGetLastError();
  }
  g_inited[dev].store(true, std::memory_order_release);
  return LGPU_OK;
}

int ensure_init() {
  int d = current_device();
  if (d < 0) {
    set_error("no HIP device available (this library has no CPU fallback)");
    return LGPU_E_NODEVICE;
  }
  if (d < 64 && g_inited[d].load(std::memory_order_acquire)) return LGPU_OK;
  return init_device(d);
}

const DeviceTables *device_tables() {
  int d = current_device();
  return (d >= 0 && d < 64 && g_inited[d].load(std::memory_order_acquire)) ? &g_tables[d] : nullptr;
}

// CUs of the current device (persistent grids are sized from it); one property query per device and process
int device_cus() {
  static std::atomic<int> cus[64];
  const int d = current_device();
  if (d < 0 || d >= 64) return 256;
  int c = cus[d].load(std::memory_order_relaxed);
  if (!c) {
    hipDeviceProp_t prop;
    c = hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus[d].store(c, std::memory_order_relaxed);
  }
  return c;
}

static const char *const kTuneNames[TUNE_COUNT] = {
  "PBH_ALIGNED", "PBH_TH", "PB_NO_DOUBLE", "PB_NO_HALF3", "PB_NO_PAIRS", "PB_NO_GATHER", "PB_NO_UP", "PB_UP_RB",
  "GCK_TH", "CHAIN_SPARE_WGS", "SEP2_LDS_KB", "NO_SEP2P", "NO_SEP2P_MFMA", "PLAN_DEBUG",
  "SEP2P_FORCE", "PB_CACHE_MAX", "K2_WGS", "SOFT_NO_S", "SOFT_RB", "EDGE_NO_S", "EDGE_TH", "PBH_ORDER", "PBH_OCC", "PBH_GROUP", "G5_MFMA", "RGB2YUV_NO_S", "UYVY_NO_S", "REPACK_NO_S", "DISABLE_HALF8", "NO_SEP2", "SEP2P_TH", "G5_CLASSIC", "GAUSS5_NO_ROWS", "PB_NO_PRE", "PB_LDS_KB", "PHASE_PROFILE", "PB_TILE_ORDER", "PB_CHAIN_GROUP", "SEAM_STAGED"};
static std::atomic<int> g_tune[TUNE_COUNT];
static std::once_flag g_tune_once;
static void tune_init() {
  std::call_once(g_tune_once, [] {
    for (int i = 0; i < TUNE_COUNT; i++) {
      char name[64];
      snprintf(name, sizeof name, "LGPU_%s", kTuneNames[i]);
      const char *e = getenv(name);           // once per process, before the first launch this library makes
      g_tune[i].store(e ? (*e ? atoi(e) : 1) : -1, std::memory_order_relaxed);
    }
  });
}
int tune(Tune t) {
  tune_init();
  return g_tune[t].load(std::memory_order_relaxed);
}

}  // namespace lgpu

extern "C" {

// launch-shape / ablation switches by name (the LGPU_<NAME> environment variables without the prefix); value < 0 clears a switch.  Results never depend on them.
int lgpu_tuning_set(const char *name, int value) {
  if (!name) { lgpu::set_error("lgpu_tuning_set: null name"); return LGPU_E_BADARG; }
  lgpu::tune_init();
  for (int i = 0; i < lgpu::TUNE_COUNT; i++)
    if (!strcmp(name, lgpu::kTuneNames[i])) { lgpu::g_tune[i].store(value < 0 ? -1 : value, std::memory_order_relaxed); return LGPU_OK; }
  lgpu::set_error("lgpu_tuning_set: unknown switch %s", name);
  return LGPU_E_BADARG;
}
int lgpu_tuning_get(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < lgpu::TUNE_COUNT; i++)
    if (!strcmp(name, lgpu::kTuneNames[i])) return lgpu::tune((lgpu::Tune)i);
  return -1;
}

int lgpu_abi_version(void) { return LGPU_ABI_VERSION; }

int lgpu_init(int device) { return lgpu::init_device(device); }

const char *lgpu_last_error(void) { return lgpu::g_err; }

int lgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// fault injection for the failure-contract tests: the n-th lgpu_malloc from now fails as if the device were out of memory
static std::atomic<int> g_fail_alloc{0};
int lgpu_debug_fail_alloc(int nth) { g_fail_alloc.store(nth > 0 ? nth : 0); return LGPU_OK; }

int lgpu_malloc(void **ptr_d, size_t bytes) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!ptr_d) return LGPU_E_BADARG;
  if (g_fail_alloc.load() > 0 && g_fail_alloc.fetch_sub(1) == 1) { *ptr_d = nullptr; lgpu::set_error("hipMalloc(%zu) failed (injected)", bytes); return LGPU_E_NOMEM; }
  if (hipMalloc(ptr_d, bytes ? bytes : 1) != hipSuccess) { lgpu::set_error("hipMalloc(%zu) failed", bytes); return LGPU_E_NOMEM; }
  return LGPU_OK;
}

int lgpu_free(void *ptr_d) {
  if (!ptr_d) return LGPU_OK;
  LGPU_HIP(hipFree(ptr_d));
  return LGPU_OK;
}

int lgpu_malloc_ordered(void **ptr_d, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!ptr_d) return LGPU_E_BADARG;
  if (g_fail_alloc.load() > 0 && g_fail_alloc.fetch_sub(1) == 1) { *ptr_d = nullptr; lgpu::set_error("hipMallocAsync(%zu) failed (injected)", bytes); return LGPU_E_NOMEM; }
  if (hipMallocAsync(ptr_d, bytes ? bytes : 1, (hipStream_t)stream) != hipSuccess) {
    (void)hipGetLastError();
    *ptr_d = nullptr;
    lgpu::set_error("hipMallocAsync(%zu) failed", bytes);
    return LGPU_E_NOMEM;
  }
  return LGPU_OK;
}

int lgpu_free_ordered(void *ptr_d, void *stream) {
  if (!ptr_d) return LGPU_OK;
  LGPU_HIP(hipFreeAsync(ptr_d, (hipStream_t)stream));
  return LGPU_OK;
}

// streams and events for hosts that enqueue from several threads (the layer seam: one stream per host thread, events on the resident planes)
int lgpu_stream_create(void **stream_out, int nonblocking) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!stream_out) return LGPU_E_BADARG;
  hipStream_t s = nullptr;
  // default flags: ordered against the null stream, which plain callers use -- and every launch pays for that (measured through the layer seam:
  // 18 us per call on a blocking stream against 9 us on the null stream or a non-blocking one)
  LGPU_HIP(hipStreamCreateWithFlags(&s, nonblocking ? hipStreamNonBlocking : hipStreamDefault));
  *stream_out = (void *)s;
  return LGPU_OK;
}
int lgpu_stream_destroy(void *stream) {
  if (stream) LGPU_HIP(hipStreamDestroy((hipStream_t)stream));
  return LGPU_OK;
}
int lgpu_event_create(void **event_out) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!event_out) return LGPU_E_BADARG;
  hipEvent_t e = nullptr;
  LGPU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event_out = (void *)e;
  return LGPU_OK;
}
int lgpu_event_destroy(void *event) {
  if (event) LGPU_HIP(hipEventDestroy((hipEvent_t)event));
  return LGPU_OK;
}
int lgpu_event_record(void *event, void *stream) {
  if (!event) return LGPU_E_BADARG;
  LGPU_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return LGPU_OK;
}
int lgpu_stream_wait_event(void *stream, void *event) {
  if (!event) return LGPU_E_BADARG;
  LGPU_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return LGPU_OK;
}

// Pageable host memory (what LiVES' frame allocator hands out) goes TO the device at ~2.7 GB/s through hipMemcpyAsync.  Uploads therefore go through two
// pinned staging chunks per host thread: the CPU copies chunk k + 1 into one while the DMA engine moves chunk k out of the other (8.3 MB: 3.1 ms plain,
// 0.37 ms staged = 22 GB/s; hipHostRegister around a direct DMA costs 0.33 - 0.5 ms for the registration alone; tools/pcie_probe.hip).  Memory the host
// pinned or registered itself takes the direct path.
namespace {
constexpr size_t kStageChunk = 4u << 20, kStageMin = 256u << 10;
struct StageSet { void *buf[2]; hipEvent_t ev[2]; bool busy[2]; int dev; };
struct SpareStages { std::mutex mu; std::vector<StageSet> sets; };
extern "C++" SpareStages &spare_stages() { static SpareStages *p = new SpareStages; return *p; }      // never destroyed: thread destructors may run late
struct Stage {
  void *buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool busy[2] = {false, false};
  int dev = -1;
  bool ready(int device) {
    if (buf[0] && dev == device) return true;
    if (buf[0]) return false;                            // this thread staged for another device before: leave that setup alone, take the plain path
    {                                                    // chunks a finished thread left behind (8 MB of page-locked memory and ~1.3 ms of hipHostMalloc a set)
      SpareStages &sp = spare_stages();
      std::lock_guard<std::mutex> lk(sp.mu);
      for (size_t i = 0; i < sp.sets.size(); i++)
        if (sp.sets[i].dev == device) {
          const StageSet st = sp.sets[i];
          sp.sets[i] = sp.sets.back(); sp.sets.pop_back();
          for (int k = 0; k < 2; k++) { buf[k] = st.buf[k]; ev[k] = st.ev[k]; busy[k] = st.busy[k]; }
          dev = device;
          return true;
        }
    }
    for (int i = 0; i < 2; i++)
      if (hipHostMalloc(&buf[i], kStageChunk, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        for (int k = 0; k < 2; k++) { if (buf[k]) (void)hipHostFree(buf[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); buf[k] = nullptr; ev[k] = nullptr; }
        return false;
      }
    dev = device;
    return true;
  }
  // the destructor makes no HIP call (thread_local objects of the main thread die after the HIP runtime has shut down): the chunks go to a spare list for the
  // next thread that stages, the busy flags with them (its first use waits for a DMA that may still read a chunk)
  ~Stage() {
    if (!buf[0] || dev < 0) return;
    SpareStages &sp = spare_stages();
    std::lock_guard<std::mutex> lk(sp.mu);
    StageSet st;
    for (int k = 0; k < 2; k++) { st.buf[k] = buf[k]; st.ev[k] = ev[k]; st.busy[k] = busy[k]; }
    st.dev = dev;
    sp.sets.push_back(st);
  }
};
thread_local Stage t_stage;
bool host_is_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged;
}
}  // namespace

int lgpu_upload(void *dst_d, const void *src_h, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  if (bytes >= kStageMin && hipGetDevice(&dev) == hipSuccess && !host_is_pinned(src_h) && t_stage.ready(dev)) {
    int k = 0;
    for (size_t off = 0; off < bytes; off += kStageChunk, k ^= 1) {
      const size_t n = bytes - off < kStageChunk ? bytes - off : kStageChunk;
      if (t_stage.busy[k]) LGPU_HIP(hipEventSynchronize(t_stage.ev[k]));      // the DMA that last read this chunk has finished
      memcpy(t_stage.buf[k], (const char *)src_h + off, n);
      LGPU_HIP(hipMemcpyAsync((char *)dst_d + off, t_stage.buf[k], n, hipMemcpyHostToDevice, st));
      LGPU_HIP(hipEventRecord(t_stage.ev[k], st));
      t_stage.busy[k] = true;
    }
    return LGPU_OK;                                     // src_h is reusable (as with a pageable hipMemcpyAsync); the copy is stream-ordered
  }
  LGPU_HIP(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, st));
  return LGPU_OK;
}

// Downloads go straight to the destination, pageable or not: the runtime's own pageable path moves 8.3 MB in 157 us into memory the host has touched
// (53 GB/s) and in 0.8 ms into a fresh block -- the new host plane of a seam call -- where the staging-chunk scheme above took 2.6 ms, most of it first-touch
// page faults of the CPU copy out of the chunk (tools/pcie_probe.hip).  Only the UPLOAD direction is slow for pageable memory (2.7 GB/s plain, 22 GB/s staged).
int lgpu_download(void *dst_h, const void *src_d, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  LGPU_HIP(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return LGPU_OK;
}

void *lgpu_pinned_calloc(size_t bytes) {
  void *p = nullptr;
  if (lgpu::ensure_init() != LGPU_OK) return calloc(1, bytes ? bytes : 1);          // no device: plain memory, same contract
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  memset(p, 0, bytes);
  return p;
}
void lgpu_pinned_free(void *p) {
  if (!p) return;
  if (host_is_pinned(p)) { (void)hipHostFree(p); return; }
  free(p);
}

int lgpu_copy(void *dst_d, const void *src_d, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  LGPU_HIP(hipMemcpyAsync(dst_d, src_d, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return LGPU_OK;
}

int lgpu_fill(void *dst_d, int byte, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  LGPU_HIP(hipMemsetAsync(dst_d, byte, bytes, (hipStream_t)stream));
  return LGPU_OK;
}

// rows of row_bytes bytes between two pitched device buffers (compact_rowstrides, the cut of unletterbox_layer)
int lgpu_copy_rows(void *dst_d, int orow, const void *src_d, int irow, int row_bytes, int rows, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!dst_d || !src_d || row_bytes < 0 || rows < 0 || orow < row_bytes || irow < row_bytes) { lgpu::set_error("lgpu_copy_rows: bad geometry"); return LGPU_E_BADARG; }
  if (!row_bytes || !rows) return LGPU_OK;
  LGPU_HIP(hipMemcpy2DAsync(dst_d, (size_t)orow, src_d, (size_t)irow, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return LGPU_OK;
}

namespace lgpu {
struct Pat8 { uint8_t b[8]; };
struct Set64 { int32_t w[64]; };
__global__ void k_set64(int32_t *dst, const Set64 v, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x]; }
__global__ void k_set4(int32_t *blk, int32_t a, int32_t b, int32_t c, int32_t d) { if (threadIdx.x == 0) { blk[0] = a; blk[1] = b; blk[2] = c; blk[3] = d; } }
__global__ __launch_bounds__(kBlock) void k_fill_pattern(uint8_t *dst, int rowstride, Pat8 pat, int plen, int nbytes, int rows) {
  const int x = blockIdx.x * kBlock + threadIdx.x;            // byte within the row
  if (x >= nbytes) return;
  const uint8_t v = pat.b[x % plen];
  for (int y = blockIdx.y; y < rows; y += gridDim.y) dst[(size_t)y * rowstride + x] = v;
}
}  // namespace lgpu

// n repetitions of a plen-byte (1..8) pattern at the start of every row: the black of a palette (blank_pixel / blank_row,
// src/colourspace.c:11123-11210) on device-resident planes; bytes past n * plen in a row are left alone
int lgpu_fill_pattern(void *dst_d, int rowstride, const uint8_t *pattern, int plen, int n, int rows, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!dst_d || !pattern || plen < 1 || plen > 8 || n < 0 || rows < 0 || (long)n * plen > rowstride) { lgpu::set_error("lgpu_fill_pattern: bad geometry"); return LGPU_E_BADARG; }
  if (!n || !rows) return LGPU_OK;
  lgpu::Pat8 p = {};
  for (int i = 0; i < plen; i++) p.b[i] = pattern[i];
  const int nbytes = n * plen;
  dim3 grid(lgpu::cdiv((unsigned)nbytes, lgpu::kBlock), (unsigned)(rows > 512 ? 512 : rows));
  hipLaunchKernelGGL(lgpu::k_fill_pattern, grid, dim3(lgpu::kBlock), 0, (hipStream_t)stream, (uint8_t *)dst_d, rowstride, p, plen, nbytes, rows);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

int lgpu_sync(void *stream) {
  LGPU_HIP(hipStreamSynchronize((hipStream_t)stream));
  return LGPU_OK;
}

// the calling thread's current HIP device (helper threads of this library take their creator's)
int lgpu_current_device(int *device) {
  if (!device) return LGPU_E_BADARG;
  LGPU_HIP(hipGetDevice(device));
  return LGPU_OK;
}
int lgpu_set_device(int device) {
  LGPU_HIP(hipSetDevice(device));
  return LGPU_OK;
}

// 1: everything enqueued on the stream so far has completed, 0: not yet, negative: a HIP error (no blocking: hipStreamQuery)
int lgpu_stream_query(void *stream) {
  const hipError_t e = hipStreamQuery((hipStream_t)stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  lgpu::set_error("hipStreamQuery failed: %s", hipGetErrorString(e));
  return LGPU_E_HIP;
}

// the control rank writes the shared transition parameter block (int32[4], device memory) from four host values: they travel as kernel arguments, so
// the call costs one launch and no host -> device copy (a 16-byte hipMemcpyAsync from pageable memory stages and blocks)
int lgpu_params_set_n(int32_t *param_blocks_d, const int32_t *values, int nblocks, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!param_blocks_d || !values || nblocks < 1) { lgpu::set_error("lgpu_params_set_n: bad argument"); return LGPU_E_BADARG; }
  for (int i = 0; i < nblocks; i += 16) {        // up to 16 blocks (256 bytes) per launch, as kernel arguments
    lgpu::Set64 v;
    const int n = nblocks - i < 16 ? nblocks - i : 16;
    memset(&v, 0, sizeof v);
    memcpy(v.w, values + 4 * i, (size_t)n * 16);
    hipLaunchKernelGGL(lgpu::k_set64, dim3(1), dim3(64), 0, (hipStream_t)stream, param_blocks_d + 4 * i, v, 4 * n);
    LGPU_CHECK_LAUNCH();
  }
  return LGPU_OK;
}

int lgpu_params_set(int32_t *param_block_d, const int32_t values[4], void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!param_block_d || !values) { lgpu::set_error("lgpu_params_set: null argument"); return LGPU_E_BADARG; }
  hipLaunchKernelGGL(lgpu::k_set4, dim3(1), dim3(64), 0, (hipStream_t)stream, param_block_d, values[0], values[1], values[2], values[3]);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

}  // extern "C"
