// stencil.hip -- the 3x3 stencil effects of the reference plugin set:
//   "softlight"     lives-plugins/weed-plugins/softlight.c:62-141   (planar YUV, luma plane only)
//   "edge detect"   lives-plugins/weed-plugins/edge.c:129-248       (packed RGB, global Otsu threshold)
// Both are HBM-bound byte work: one LDS tile with its halo per workgroup, lane = 4 consecutive pixels.
#include "lgpu_common.h"
#include <map>
#include <mutex>
#include <tuple>

namespace lgpu {

constexpr int kStW = 64, kStH = 16;      // output tile of a 256-thread workgroup: thread = 4 pixels of one row

// ---------------------------------------------------------------------------------------------------------------------
// softlight
// ---------------------------------------------------------------------------------------------------------------------
// floor(sqrt(n)) for n < 2^24 (softlight.c:34-47 computes it digit by digit; n <= 1020^2 + 1275^2 here)
__device__ __forceinline__ uint32_t isqrt24(uint32_t n) {
  uint32_t r = (uint32_t)__fsqrt_rn((float)n);
  if (r * r > n) r--;
  if ((r + 1) * (r + 1) <= n) r++;
  return r;
}

__global__ __launch_bounds__(kBlock) void k_softlight(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height,
                                                        int ymin, int ymax) {
  __shared__ uint8_t s[(kStH + 2) * (kStW + 8)];              // rows y0-1 .. y0+kStH, columns x0-4 .. x0+kStW+3
  constexpr int P = kStW + 8;
  const int x0 = blockIdx.x * kStW, y0 = blockIdx.y * kStH;
  for (int i = threadIdx.x; i < (kStH + 2) * P; i += kBlock) {
    const int r = i / P, c = i - r * P;
    int sy = y0 - 1 + r, sx = x0 - 4 + c;
    sy = sy < 0 ? 0 : sy >= height ? height - 1 : sy;          // clamped fetches only feed border outputs, which are copies
    sx = sx < 0 ? 0 : sx >= width ? width - 1 : sx;
    s[i] = src[(size_t)sy * irow + sx];
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4;
  const int y = y0 + ly;
  if (y >= height) return;
  const uint8_t *c0 = s + (ly + 1) * P + lx + 4;
  uint8_t out[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int x = x0 + lx + j;
    const uint8_t *c = c0 + j;
    int v = c[0];
    if (y > 0 && y < height - 1 && x > 0 && x < width - 1) {
      // softlight.c:115-120 as written: the third term of row0 pairs the lower-right with the lower-LEFT sample, the
      // third term of row1 is a sum
      const int row0 = (c[P - 1] - c[-P - 1]) + ((c[P] - c[-P]) << 1) + (c[P + 1] - c[P - 1]);
      const int row1 = (c[-P + 1] - c[-P - 1]) + ((c[1] - c[-1]) << 1) + (c[P + 1] + c[P - 1]);
      int sum = (int)(((3 * isqrt24((uint32_t)(row0 * row0 + row1 * row1)) / 2) * 384u) >> 8);
      sum = sum < ymin ? ymin : sum > ymax ? ymax : sum;
      sum = (64 * sum + 192 * v) >> 8;
      v = sum < ymin ? ymin : sum > ymax ? ymax : sum;
    }
    out[j] = (uint8_t)v;
  }
  uint8_t *d = dst + (size_t)y * orow + x0 + lx;
  const int n = width - (x0 + lx);
  if (n >= 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0))
    *reinterpret_cast<uint32_t *>(d) = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
  else
    for (int j = 0; j < 4 && j < n; j++) d[j] = out[j];
}

// ---------------------------------------------------------------------------------------------------------------------
// edge detect
// ---------------------------------------------------------------------------------------------------------------------
struct EdgeState {            // the function-scope accumulators of edge_process (edge.c:146-149): carried over the passes of mode 2
  unsigned long long bh, bl, nbh, nbl;
  double difmax;
  unsigned int threshmax, thresh;
  unsigned long long sum;     // sum of the map values of the current pass (added to bh before the Otsu scan)
  unsigned int hist[1024];
};

__global__ void k_edge_reset(EdgeState *st) {
  const int i = threadIdx.x;
  if (i == 0) { st->bh = st->bl = st->nbh = st->nbl = 0; st->difmax = 0.; st->threshmax = 0; st->thresh = 0; st->sum = 0; }
  for (int k = i; k < 1024; k += blockDim.x) st->hist[k] = 0;
}

// luma (pass 0) or byte pass-1 of the pixel; gradient magnitude map + histogram
template <int PS>
__global__ __launch_bounds__(kBlock) void k_edge_map(const uint8_t *src, int irow, int width, int height, int order, int pass,
                                                       const int32_t *gluma, uint16_t *map, EdgeState *st) {
  __shared__ uint8_t l[(kStH + 4) * (kStW + 4)];
  __shared__ int32_t s_luma[768];
  __shared__ unsigned int s_hist[1024];
  __shared__ unsigned int s_sum;
  constexpr int P = kStW + 4;
  const int x0 = blockIdx.x * kStW, y0 = blockIdx.y * kStH;
  if (pass == 0) for (int i = threadIdx.x; i < 768; i += kBlock) s_luma[i] = gluma[i];
  for (int i = threadIdx.x; i < 1024; i += kBlock) s_hist[i] = 0;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < (kStH + 4) * P; i += kBlock) {
    const int r = i / P, c = i - r * P;
    const int sy = y0 - 2 + r, sx = x0 - 2 + c;
    uint8_t v = 0;
    if (sy >= 0 && sy < height && sx >= 0 && sx < width) {
      const uint8_t *p = src + (size_t)sy * irow + (size_t)sx * PS;
      if (pass == 0) {        // calc_luma(), libweed/weed-plugin-utils.c:924-934
        const int c0 = order == 2 ? p[1] : p[0], c1 = order == 2 ? p[2] : p[1], c2 = order == 2 ? p[3] : p[2];
        const int32_t t = order == 1 ? (s_luma[c2] + s_luma[256 + c1] + s_luma[512 + c0]) : (s_luma[c0] + s_luma[256 + c1] + s_luma[512 + c2]);
        v = (uint8_t)(t >> 16);
      } else v = p[pass - 1];
    }
    l[i] = v;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4;
  const int y = y0 + ly;
  unsigned int lsum = 0;
  if (y < height) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = x0 + lx + j;
      if (x >= width) break;
      unsigned int val = 0;
      if (y >= 2 && y < height - 2 && x >= 2 && x < width - 2) {
        const uint8_t *c = l + (ly + 2) * P + lx + j + 2;
        // v0 = sum over rows y-1..y+1 of (l[x+1] - l[x-1]); v1 = sum over columns x-1..x+1 of (l[y+1] - l[y-1])
        const int v0 = (c[-P + 1] - c[-P - 1]) + (c[1] - c[-1]) + (c[P + 1] - c[P - 1]);
        const int v1 = (c[P - 1] - c[-P - 1]) + (c[P] - c[-P]) + (c[P + 1] - c[-P + 1]);
        const float m = __fmul_rn(__fsqrt_rn(__fadd_rn((float)(v0 * v0), (float)(v1 * v1))), 0.94f);
        val = (unsigned int)m & 0xFFFFu;
        atomicAdd(&s_hist[val & 1023], 1u);
        lsum += val;
      }
      map[(size_t)y * width + x] = (uint16_t)val;
    }
  }
  if (lsum) atomicAdd(&s_sum, lsum);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += kBlock)
    if (s_hist[i]) atomicAdd(&st->hist[i], s_hist[i]);
  if (threadIdx.x == 0 && s_sum) atomicAdd(&st->sum, (unsigned long long)s_sum);
}

// the Otsu scan of edge.c:184-205, one thread, IEEE double arithmetic in the reference's operation order
__global__ void k_edge_otsu(EdgeState *st, unsigned long long count) {
  if (threadIdx.x != 0) return;
  unsigned long long bh = st->bh + st->sum, nbh = st->nbh + count, bl = st->bl, nbl = st->nbl;
  double difmax = st->difmax;
  unsigned int threshmax = st->threshmax;
  for (unsigned int t = 0; t < 1017; t++) {
    const unsigned long long pr = st->hist[t];
    const unsigned long long nn = pr * t;
    bl += nn; nbl += pr;
    bh -= nn; nbh -= pr;
    const double abh = __ddiv_rn((double)bh, (double)nbh);
    const double abl = __ddiv_rn((double)bl, (double)nbl);
    const double d = __dsub_rn(abh, abl);
    const double dif = __dmul_rn(__dmul_rn((double)(nbl * nbh), d), d);
    if (t > 0 && dif > difmax) { difmax = dif; threshmax = t; }
  }
  st->bh = bh; st->nbh = nbh; st->bl = bl; st->nbl = nbl;
  st->difmax = difmax; st->threshmax = threshmax; st->thresh = threshmax;
  st->sum = 0;
  for (int i = 0; i < 1024; i++) st->hist[i] = 0;
}

// copywalpha (edge.c:93-125) on every pixel: codes per colour byte 0 = black, 1 = source, 2 = white, 3 = leave
template <int PS>
__global__ __launch_bounds__(kBlock) void k_edge_paint(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height,
                                                         int pass, int mode, int aoffs, int inplace, const uint16_t *map,
                                                         const EdgeState *st) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= width) return;
  const unsigned int thresh = st->thresh;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const uint8_t *s = src + (size_t)y * irow + (size_t)x * PS;
    uint8_t *d = dst + (size_t)y * orow + (size_t)x * PS;
    const bool edge = map[(size_t)y * width + x] >= thresh;
    int code[3];
    if (edge) {
      if (pass == 0) code[0] = code[1] = code[2] = (mode == 1 ? 2 : 1);
      else { code[0] = pass == 1 ? 2 : 3; code[1] = pass == 2 ? 2 : 3; code[2] = pass == 3 ? 2 : 3; }
    } else {
      if (pass != 0) continue;
      code[0] = code[1] = code[2] = 0;
    }
    int o = 0;
    if (aoffs == 1) { if (!inplace) d[0] = s[0]; o = 1; }
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (code[k] != 3) d[o + k] = code[k] == 1 ? s[o + k] : code[k] == 0 ? 0 : 255;
    if (aoffs == 0 && !inplace) d[3] = s[3];
  }
}

// per (device, stream) scratch: gradient map + state
struct EdgeScratch { uint16_t *map = nullptr; size_t cap = 0; EdgeState *st = nullptr; };
static std::mutex g_edge_mu;
static std::map<std::pair<int, void *>, EdgeScratch> g_edge;

}  // namespace lgpu

using namespace lgpu;

extern "C" int lgpu_softlight(const uint8_t *const src_d[4], const int irow[4], uint8_t *const dst_d[4], const int orow[4],
                              int width, int height, int palette, int unclamped, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && irow && orow, "null plane tables");
  LGPU_REQUIRE(palette == 544 || palette == 545 || palette == 522 || palette == 512 || palette == 513,
               "palette must be YUV444P, YUVA4444P, YUV422P, YUV420P or YVU420P (softlight.c:162-164)");
  LGPU_REQUIRE(width >= 3 && height >= 3, "softlight needs at least 3 x 3 samples");
  const int nplanes = palette == 545 ? 4 : 3;
  for (int i = 0; i < nplanes; i++) LGPU_REQUIRE(src_d[i] && dst_d[i] && src_d[i] != dst_d[i], "null plane, or in place (the filter is not CAN_DO_INPLACE)");
  LGPU_REQUIRE(irow[0] >= width && orow[0] >= width, "rowstride smaller than a row");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv((unsigned)width, kStW), cdiv((unsigned)height, kStH));
  hipLaunchKernelGGL(k_softlight, grid, dim3(kBlock), 0, st, src_d[0], irow[0], dst_d[0], orow[0], width, height, unclamped ? 0 : 16,
                     unclamped ? 255 : 235);
  LGPU_CHECK_LAUNCH();
  // the other planes are copied (softlight.c:143-151); alpha of YUVA4444P has the chroma geometry of 4:4:4
  const int cw = (palette == 512 || palette == 513 || palette == 522) ? width >> 1 : width;
  const int chh = (palette == 512 || palette == 513) ? height >> 1 : height;
  for (int i = 1; i < nplanes; i++)
    LGPU_HIP(hipMemcpy2DAsync(dst_d[i], (size_t)orow[i], src_d[i], (size_t)irow[i], (size_t)cw, (size_t)chh, hipMemcpyDeviceToDevice, st));
  return LGPU_OK;
}

extern "C" int lgpu_edge(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette, int mode,
                         void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && width > 0 && height > 0, "null frame or empty geometry");
  LGPU_REQUIRE(palette >= 1 && palette <= 5, "palette must be RGB24, BGR24, RGBA32, BGRA32 or ARGB32");
  LGPU_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (normal), 1 (monochrome) or 2 (supercolour)");
  const int psize = palette <= 2 ? 3 : 4;
  LGPU_REQUIRE(irow >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  const int order = (palette == 1 || palette == 3) ? 0 : (palette == 2 || palette == 4) ? 1 : 2;
  const int aoffs = palette == 5 ? 1 : psize == 3 ? -1 : 0;
  const int inplace = src_d == dst_d;
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  EdgeScratch sc;
  {
    std::lock_guard<std::mutex> lk(g_edge_mu);
    EdgeScratch &e = g_edge[std::make_pair(dev, stream)];
    const size_t need = (size_t)width * height * sizeof(uint16_t);
    if (e.cap < need) {
      if (e.map) LGPU_HIP(hipFree(e.map));
      e.map = nullptr; e.cap = 0;
      if (hipMalloc((void **)&e.map, need) != hipSuccess) { set_error("hipMalloc(%zu) for the edge map failed", need); return LGPU_E_NOMEM; }
      e.cap = need;
    }
    if (!e.st) LGPU_HIP(hipMalloc((void **)&e.st, sizeof(EdgeState)));
    sc = e;
  }
  const unsigned long long count = (width > 4 && height > 4) ? (unsigned long long)(width - 4) * (unsigned long long)(height - 4) : 0ull;
  const dim3 tgrid(cdiv((unsigned)width, kStW), cdiv((unsigned)height, kStH));
  const dim3 pgrid(cdiv((unsigned)width, kBlock), (unsigned)(height < 1024 ? height : 1024));
  hipLaunchKernelGGL(k_edge_reset, dim3(1), dim3(256), 0, st, sc.st);
  for (int pass = 0; pass < 4; pass++) {
    if (psize == 4) hipLaunchKernelGGL(k_edge_map<4>, tgrid, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, sc.st);
    else hipLaunchKernelGGL(k_edge_map<3>, tgrid, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, sc.st);
    hipLaunchKernelGGL(k_edge_otsu, dim3(1), dim3(64), 0, st, sc.st, count);
    if (psize == 4) hipLaunchKernelGGL(k_edge_paint<4>, pgrid, dim3(kBlock), 0, st, src_d, irow, dst_d, orow, width, height, pass, mode, aoffs, inplace, sc.map, sc.st);
    else hipLaunchKernelGGL(k_edge_paint<3>, pgrid, dim3(kBlock), 0, st, src_d, irow, dst_d, orow, width, height, pass, mode, aoffs, inplace, sc.map, sc.st);
    if (mode < 2) break;
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
