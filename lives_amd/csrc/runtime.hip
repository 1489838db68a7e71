// runtime.hip -- device init, error string, memory helpers, resident conversion tables.
#include "lgpu_common.h"
#include <stdarg.h>
#include <mutex>
#include <string.h>

namespace lgpu {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static bool g_inited[64];
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
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device available (this library has no CPU fallback)");
    return LGPU_E_NODEVICE;
  }
  if (dev < 0 || dev >= n || dev >= 64) { set_error("device %d out of range (%d devices)", dev, n); return LGPU_E_BADARG; }
  LGPU_HIP(hipSetDevice(dev));
  if (g_inited[dev]) return LGPU_OK;
  hipDeviceProp_t prop;
  LGPU_HIP(hipGetDeviceProperties(&prop, dev));
  if (!strstr(prop.gcnArchName, "gfx950")) {
    set_error("device %d is %s; this build carries gfx950 (MI355X) code objects only", dev, prop.gcnArchName);
    return LGPU_E_NODEVICE;
  }
  for (int which = 0; which < 4; which++) {
    int32_t r2y[9 * 256], y2r[5 * 256];
    lgpu_conversion_tables(which, r2y, y2r);
    LGPU_HIP(hipMalloc((void **)&g_tables[dev].rgb2yuv[which], sizeof r2y));
    LGPU_HIP(hipMalloc((void **)&g_tables[dev].yuv2rgb[which], sizeof y2r));
    LGPU_HIP(hipMemcpy(g_tables[dev].rgb2yuv[which], r2y, sizeof r2y, hipMemcpyHostToDevice));
    LGPU_HIP(hipMemcpy(g_tables[dev].yuv2rgb[which], y2r, sizeof y2r, hipMemcpyHostToDevice));
  }
  {
    // luma weights of calc_luma(): myround(k * i * 65536.) per channel
    int32_t lw[3 * 256];
    for (int i = 0; i < 256; i++) {
      const double v = (double)i;
      const double r = 0.299 * v * 65536., g = (1. - 0.299 - 0.114) * v * 65536., b = 0.114 * v * 65536.;
      lw[i] = (int32_t)(r + 0.5); lw[256 + i] = (int32_t)(g + 0.5); lw[512 + i] = (int32_t)(b + 0.5);
    }
    LGPU_HIP(hipMalloc((void **)&g_tables[dev].luma, sizeof lw));
    LGPU_HIP(hipMemcpy(g_tables[dev].luma, lw, sizeof lw, hipMemcpyHostToDevice));
  }
  g_inited[dev] = true;
  return LGPU_OK;
}

int ensure_init() {
  int d = current_device();
  if (d < 0) {
    set_error("no HIP device available (this library has no CPU fallback)");
    return LGPU_E_NODEVICE;
  }
  if (d < 64 && g_inited[d]) return LGPU_OK;
  return init_device(d);
}

const DeviceTables *device_tables() {
  int d = current_device();
  return (d >= 0 && d < 64 && g_inited[d]) ? &g_tables[d] : nullptr;
}

}  // namespace lgpu

extern "C" {

int lgpu_abi_version(void) { return LGPU_ABI_VERSION; }

int lgpu_init(int device) { return lgpu::init_device(device); }

const char *lgpu_last_error(void) { return lgpu::g_err; }

int lgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lgpu_malloc(void **ptr_d, size_t bytes) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  if (!ptr_d) return LGPU_E_BADARG;
  if (hipMalloc(ptr_d, bytes ? bytes : 1) != hipSuccess) { lgpu::set_error("hipMalloc(%zu) failed", bytes); return LGPU_E_NOMEM; }
  return LGPU_OK;
}

int lgpu_free(void *ptr_d) {
  if (!ptr_d) return LGPU_OK;
  LGPU_HIP(hipFree(ptr_d));
  return LGPU_OK;
}

int lgpu_upload(void *dst_d, const void *src_h, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  LGPU_HIP(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return LGPU_OK;
}

int lgpu_download(void *dst_h, const void *src_d, size_t bytes, void *stream) {
  int rc = lgpu::ensure_init();
  if (rc) return rc;
  LGPU_HIP(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return LGPU_OK;
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

int lgpu_sync(void *stream) {
  LGPU_HIP(hipStreamSynchronize((hipStream_t)stream));
  return LGPU_OK;
}

}  // extern "C"
