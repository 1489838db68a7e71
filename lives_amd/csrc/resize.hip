// resize.hip -- the separable FIR engine: K7 resize, B1 5x5 gaussian and the fused per-track chain
// (convert -> resize -> [blur] -> chroma blend -> gamma) of BASELINE config 5 / the north_star headline.
//
// Replaces the sws_scale() call of resize_layer_full (src/colourspace.c:14711; setup :14940-15259), the
// post-resize LUT pass (:14718-14720) and, in the chain, convert_swap3postalpha_frame (:9626),
// simple_blend.c:117-150 and gamma_convert_layer_thread (:14034-14060) fused around it.
// Numerics: spec "lgpu-polyphase-v1"  -- PARITY UNPINNED against libswscale.
//
// Roofline: HBM.  Per output tile (64 x TH pixels) a workgroup
//   1. streams the source window (rows x cols the taps reach, edge-replicated) into LDS, applying the
//      BGRA->RGBA byte swap on the fly,
//   2. runs the horizontal pass LDS->LDS (lane = output column, so filter position and taps are per-lane
//      registers; result kept as 4 x int16, 15-bit with 7 fractional bits),
//   3. runs the vertical pass from LDS (row taps are wave-uniform -> scalar loads), then blends with the
//      second layer, applies the gamma LUT from LDS and stores.
// Source bytes cross HBM once (tile halos are L2 hits); intermediates never leave the CU.
#include "lgpu_common.h"
#include "../../include/lives_gpu_weed_abi.h"
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace lgpu {

constexpr int kTileW = 64;          // output columns per workgroup == wavefront width
constexpr int kStageMax = 6;        // 16-byte loads a thread keeps in flight while staging a window

struct SepArgs {
  int sw, sh, irow;                 // source geometry
  int dw, dh, orow;                 // destination geometry
  const int32_t *hpos, *vpos;       // first tap per output column / row (device)
  const int16_t *hco, *vco;         // taps, [dst][ntaps] (device)
  int nth, ntv;
  int hround, hshift, vround, vshift;
  int swt, sht;                     // LDS window capacity: columns, rows
  int th;                           // output rows per tile
  uint32_t src_sel;                 // v_perm selector applied to every source pixel (0x03020100 = identity)
  int blend, irow2;                 // chroma blend with layer 2 (bf / nbf below)
  uint32_t bf, nbf;
  const int32_t *bf_d;              // optional device-resident blend amount (shared parameter block)
  int use_lut;
  int vec;                          // source rows are 16-byte aligned: stage with 16-byte loads
  int tiles_x, tiles_y;
};
struct SepTracks {
  const uint8_t *src[LGPU_CHAIN_MAX_TRACKS];
  const uint8_t *l2[LGPU_CHAIN_MAX_TRACKS];
  uint8_t *dst[LGPU_CHAIN_MAX_TRACKS];
};

__device__ __forceinline__ uint32_t mix_pairs2(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  return ((b * bf + a * nbf) >> 8) & 0x00FF00FFu;
}
__device__ __forceinline__ uint32_t mix4b(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  return mix_pairs2(a & 0x00FF00FFu, b & 0x00FF00FFu, bf, nbf) | (mix_pairs2((a >> 8) & 0x00FF00FFu, (b >> 8) & 0x00FF00FFu, bf, nbf) << 8);
}
__device__ __forceinline__ int clamp_i16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

// chroma blend of one RGBA pixel pair, dst alpha = track alpha (simple_blend.c:128-146, host-inplace channel)
__device__ __forceinline__ uint32_t chroma_rgba(uint32_t p1, uint32_t p2, uint32_t bf, uint32_t nbf) {
  const uint32_t al = p2 >> 24;
  uint32_t r;
  if (al == 255) r = mix4b(p1, p2, bf, nbf);
  else {
    const float alpha = (float)((double)(float)al / 255.), inv = (float)(1. - (double)alpha);
    uint32_t s2 = 0, s1 = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s2 |= ((uint32_t)(int)__fmul_rn((float)((p2 >> (8 * c)) & 0xFF), alpha) & 0xFF) << (8 * c);
      s1 |= ((uint32_t)(int)__fmul_rn((float)((p1 >> (8 * c)) & 0xFF), inv) & 0xFF) << (8 * c);
    }
    r = mix4b(s1, s2, bf, nbf);
  }
  return (r & 0x00FFFFFFu) | (p1 & 0xFF000000u);
}

// NTH / NTV: compile-time tap counts, 0 = runtime (a.nth / a.ntv)
template <int NTH, int NTV>
__global__ __launch_bounds__(kBlock) void k_separable(SepArgs a, SepTracks trk, Lut8 lut) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *s_src = reinterpret_cast<uint32_t *>(smem);                               // [sht][swt]
  uint2 *s_h = reinterpret_cast<uint2 *>(smem + (size_t)a.sht * a.swt * 4);          // [sht][64] 4 x int16
  uint8_t *s_lut = smem + (size_t)a.sht * a.swt * 4 + (size_t)a.sht * kTileW * 8;     // [256]

  const int nth = NTH ? NTH : a.nth, ntv = NTV ? NTV : a.ntv;
  // tile coordinates; blockIdx.x walks tiles row-major, blockIdx.y = track
  const int tile = blockIdx.x, track = blockIdx.y;
  const int tx0 = (tile % a.tiles_x) * kTileW, ty0 = (tile / a.tiles_x) * a.th;
  const int tw = min(kTileW, a.dw - tx0), thh = min(a.th, a.dh - ty0);
  const uint8_t *src = trk.src[track];

  // source window of this tile (unclamped coordinates)
  const int sx0 = a.hpos[tx0], sx1 = a.hpos[tx0 + tw - 1] + nth;
  const int sy0 = a.vpos[ty0], sy1 = a.vpos[ty0 + thh - 1] + ntv;
  const int wcols = sx1 - sx0, wrows = sy1 - sy0;     // host guarantees <= swt / sht

  if (a.use_lut) stage_lut(s_lut, lut);
  uint32_t bf = a.bf, nbf = a.nbf;
  if (a.blend && a.bf_d) { bf = (uint32_t)a.bf_d[0] & 0xFF; nbf = 0xFF - bf; }

  // ---- 1. stage the window (edge replicate), byte swap on the fly ----
  // The window is widened to 4-pixel (16 B) columns: sx0a = sx0 rounded down to a multiple of 4.  Every
  // thread first ISSUES all of its 16-byte loads (independent, so HBM latency is paid once per tile, not once
  // per load), then permutes and writes them to LDS.
  const int sx0a = sx0 & ~3;
  const int wchunks = (sx1 - sx0a + 3) >> 2;            // 16-byte chunks per window row
  const int nitems = wrows * wchunks;
  if (a.vec) {
    uint4 v[kStageMax];
#pragma unroll
    for (int k = 0; k < kStageMax; k++) {
      const int it = threadIdx.x + k * kBlock;
      if (it < nitems) {
        const int r = it / wchunks, ch = it - r * wchunks;
        int sy = sy0 + r;
        sy = sy < 0 ? 0 : sy >= a.sh ? a.sh - 1 : sy;
        const uint8_t *srow = src + (size_t)sy * a.irow;
        const int x = sx0a + ch * 4;
        if (x >= 0 && x + 4 <= a.sw) v[k] = *reinterpret_cast<const uint4 *>(srow + (size_t)x * 4);
        else {
          const uint32_t *sp = reinterpret_cast<const uint32_t *>(srow);
          const int m = a.sw - 1;
          v[k] = make_uint4(sp[min(max(x, 0), m)], sp[min(max(x + 1, 0), m)], sp[min(max(x + 2, 0), m)], sp[min(max(x + 3, 0), m)]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kStageMax; k++) {
      const int it = threadIdx.x + k * kBlock;
      if (it < nitems) {
        const int r = it / wchunks, ch = it - r * wchunks;
        uint4 o;
        o.x = __builtin_amdgcn_perm(0u, v[k].x, a.src_sel); o.y = __builtin_amdgcn_perm(0u, v[k].y, a.src_sel);
        o.z = __builtin_amdgcn_perm(0u, v[k].z, a.src_sel); o.w = __builtin_amdgcn_perm(0u, v[k].w, a.src_sel);
        *reinterpret_cast<uint4 *>(s_src + r * a.swt + ch * 4) = o;
      }
    }
    for (int it = threadIdx.x + kStageMax * kBlock; it < nitems; it += kBlock) {   // windows larger than the register batch
      const int r = it / wchunks, ch = it - r * wchunks;
      int sy = sy0 + r;
      sy = sy < 0 ? 0 : sy >= a.sh ? a.sh - 1 : sy;
      const uint32_t *sp = reinterpret_cast<const uint32_t *>(src + (size_t)sy * a.irow);
      const int m = a.sw - 1, x = sx0a + ch * 4;
      uint32_t *d = s_src + r * a.swt + ch * 4;
      for (int q = 0; q < 4; q++) d[q] = __builtin_amdgcn_perm(0u, sp[min(max(x + q, 0), m)], a.src_sel);
    }
  } else {
    for (int r = threadIdx.x >> 6; r < wrows; r += kBlock >> 6) {
      int sy = sy0 + r;
      sy = sy < 0 ? 0 : sy >= a.sh ? a.sh - 1 : sy;
      const uint32_t *srow = reinterpret_cast<const uint32_t *>(src + (size_t)sy * a.irow);
      for (int c = threadIdx.x & 63; c < wchunks * 4; c += 64) {
        int sx = sx0a + c;
        sx = sx < 0 ? 0 : sx >= a.sw ? a.sw - 1 : sx;
        s_src[r * a.swt + c] = __builtin_amdgcn_perm(0u, srow[sx], a.src_sel);
      }
    }
  }
  __syncthreads();

  // ---- 2. horizontal pass: lane = output column ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ox = tx0 + (lane < tw ? lane : tw - 1);
  const int hoff = a.hpos[ox] - sx0a;
  int hc[NTH ? NTH : 1];
  if (NTH) {
#pragma unroll
    for (int j = 0; j < NTH; j++) hc[j] = a.hco[(size_t)ox * NTH + j];
  }
  for (int r = wave; r < wrows; r += kBlock >> 6) {
    const uint32_t *row = s_src + r * a.swt + hoff;
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    if (NTH) {
#pragma unroll
      for (int j = 0; j < NTH; j++) {
        const uint32_t p = row[j];
        acc0 += hc[j] * (int)(p & 0xFF); acc1 += hc[j] * (int)((p >> 8) & 0xFF);
        acc2 += hc[j] * (int)((p >> 16) & 0xFF); acc3 += hc[j] * (int)(p >> 24);
      }
    } else {
      for (int j = 0; j < nth; j++) {
        const uint32_t p = row[j];
        const int cf = a.hco[(size_t)ox * nth + j];
        acc0 += cf * (int)(p & 0xFF); acc1 += cf * (int)((p >> 8) & 0xFF);
        acc2 += cf * (int)((p >> 16) & 0xFF); acc3 += cf * (int)(p >> 24);
      }
    }
    acc0 = clamp_i16((acc0 + a.hround) >> a.hshift); acc1 = clamp_i16((acc1 + a.hround) >> a.hshift);
    acc2 = clamp_i16((acc2 + a.hround) >> a.hshift); acc3 = clamp_i16((acc3 + a.hround) >> a.hshift);
    s_h[r * kTileW + lane] = make_uint2((uint32_t)(acc0 & 0xFFFF) | ((uint32_t)acc1 << 16), (uint32_t)(acc2 & 0xFFFF) | ((uint32_t)acc3 << 16));
  }
  __syncthreads();

  // ---- 3. vertical pass + epilogue: wave = output row, lane = output column ----
  uint8_t *dst = trk.dst[track];
  const uint8_t *l2 = trk.l2[track];
  for (int ly = wave; ly < thh; ly += kBlock >> 6) {
    const int oy = ty0 + ly;
    const int voff = a.vpos[oy] - sy0;                 // wave-uniform
    const int16_t *vc = a.vco + (size_t)oy * ntv;      // wave-uniform -> scalar loads
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const uint2 *col = s_h + voff * kTileW + lane;
    if (NTV) {
#pragma unroll
      for (int j = 0; j < NTV; j++) {
        const uint2 t = col[j * kTileW];
        const int cf = vc[j];
        acc0 += cf * (int)(short)(t.x & 0xFFFF); acc1 += cf * ((int)t.x >> 16);
        acc2 += cf * (int)(short)(t.y & 0xFFFF); acc3 += cf * ((int)t.y >> 16);
      }
    } else {
      for (int j = 0; j < ntv; j++) {
        const uint2 t = col[j * kTileW];
        const int cf = vc[j];
        acc0 += cf * (int)(short)(t.x & 0xFFFF); acc1 += cf * ((int)t.x >> 16);
        acc2 += cf * (int)(short)(t.y & 0xFFFF); acc3 += cf * ((int)t.y >> 16);
      }
    }
    if (lane < tw) {
      uint32_t p = (uint32_t)clamp255((acc0 + a.vround) >> a.vshift) | ((uint32_t)clamp255((acc1 + a.vround) >> a.vshift) << 8) |
                   ((uint32_t)clamp255((acc2 + a.vround) >> a.vshift) << 16) | ((uint32_t)clamp255((acc3 + a.vround) >> a.vshift) << 24);
      if (a.blend) {
        const uint32_t q = reinterpret_cast<const uint32_t *>(l2 + (size_t)oy * a.irow2)[tx0 + lane];
        p = chroma_rgba(p, q, bf, nbf);
      }
      if (a.use_lut) p = lut3_rgba(s_lut, p);
      reinterpret_cast<uint32_t *>(dst + (size_t)oy * a.orow)[tx0 + lane] = p;
    }
  }
}

// ---- generic two-launch path: any pixel size (bytes are independent channels), global int16 scratch ----
__global__ __launch_bounds__(kBlock) void k_hpass_generic(const uint8_t *src, int irow, int sw, int sh, int16_t *tmp, int dw,
                                                           int psize, const int32_t *pos, const int16_t *co, int nt, int round, int shift) {
  const int i = blockIdx.x * kBlock + threadIdx.x;      // output byte within the row
  if (i >= dw * psize) return;
  const int x = i / psize, c = i - x * psize;
  for (int y = blockIdx.y; y < sh; y += gridDim.y) {
    const uint8_t *s = src + (size_t)y * irow;
    int acc = 0;
    for (int j = 0; j < nt; j++) {
      int sx = pos[x] + j;
      sx = sx < 0 ? 0 : sx >= sw ? sw - 1 : sx;
      acc += (int)co[(size_t)x * nt + j] * s[sx * psize + c];
    }
    tmp[(size_t)y * dw * psize + i] = (int16_t)clamp_i16((acc + round) >> shift);
  }
}
__global__ __launch_bounds__(kBlock) void k_vpass_generic(const int16_t *tmp, int sh, uint8_t *dst, int orow, int dwb, int dh,
                                                           const int32_t *pos, const int16_t *co, int nt, int round, int shift,
                                                           int psize, int use_lut, Lut8 lut) {
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  stage_lut(s_lut, lut);
  __syncthreads();
  const int i = blockIdx.x * kBlock + threadIdx.x;      // output byte within the row
  if (i >= dwb) return;
  const bool colour = (psize != 4) || ((i & 3) != 3);
  for (int y = blockIdx.y; y < dh; y += gridDim.y) {
    int acc = 0;
    for (int j = 0; j < nt; j++) {
      int sy = pos[y] + j;
      sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
      acc += (int)co[(size_t)y * nt + j] * tmp[(size_t)sy * dwb + i];
    }
    int v = clamp255((acc + round) >> shift);
    if (use_lut && colour) v = s_lut[v];
    dst[(size_t)y * orow + i] = (uint8_t)v;
  }
}

// ---- host side: filter-bank cache (per device) -----------------------------------------------------------
struct Bank {
  int32_t *pos = nullptr;
  int16_t *co = nullptr;
  int nt = 0;
  int max_span = 0;      // max over 64-column (or th-row) tiles is derived by the caller from host copies
  std::vector<int32_t> hpos;
};
static std::mutex g_bank_mu;
static std::map<std::tuple<int, int, int, int>, Bank> g_banks;   // (device, srcn, dstn, kernel)  kernel 100 = gauss5

static int get_bank(int srcn, int dstn, int kernel, const Bank **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_bank_mu);
  auto key = std::make_tuple(dev, srcn, dstn, kernel);
  auto it = g_banks.find(key);
  if (it == g_banks.end()) {
    Bank b;
    std::vector<int16_t> co;
    b.hpos.resize(dstn);
    if (kernel == 100) {
      b.nt = 5;
      co.resize((size_t)dstn * 5);
      static const int16_t g5[5] = {1, 4, 6, 4, 1};
      for (int i = 0; i < dstn; i++) { b.hpos[i] = i - 2; for (int j = 0; j < 5; j++) co[(size_t)i * 5 + j] = g5[j]; }
    } else {
      co.resize((size_t)dstn * 256);
      int rc = lgpu_make_filter(srcn, dstn, kernel, &b.nt, b.hpos.data(), co.data(), 256);
      if (rc) { set_error("resize %d -> %d needs more than 256 taps", srcn, dstn); return rc; }
      co.resize((size_t)dstn * b.nt);
    }
    LGPU_HIP(hipMalloc((void **)&b.pos, sizeof(int32_t) * dstn));
    LGPU_HIP(hipMalloc((void **)&b.co, sizeof(int16_t) * co.size()));
    LGPU_HIP(hipMemcpy(b.pos, b.hpos.data(), sizeof(int32_t) * dstn, hipMemcpyHostToDevice));
    LGPU_HIP(hipMemcpy(b.co, co.data(), sizeof(int16_t) * co.size(), hipMemcpyHostToDevice));
    it = g_banks.emplace(key, std::move(b)).first;
  }
  *out = &it->second;
  return LGPU_OK;
}

static int max_tile_span(const Bank &b, int dstn, int tile) {
  int m = 0;
  for (int t0 = 0; t0 < dstn; t0 += tile) {
    const int t1 = (t0 + tile < dstn ? t0 + tile : dstn) - 1;
    const int span = b.hpos[t1] + b.nt - b.hpos[t0];
    if (span > m) m = span;
  }
  return m;
}

static int kernel_for_interp(int interp, bool upscale) {
  // LIVES_INTERP_BEST: bicubic when shrinking, lanczos when enlarging; NORMAL / FAST: bilinear
  // (src/colourspace.c:14991-14997)
  if (interp == LIVES_INTERP_BEST) return upscale ? 2 : 1;
  return 0;
}

struct SepPlan {
  SepArgs a;
  size_t lds;
  dim3 grid;
  int variant;   // 0 generic taps, 1 = (8,8), 2 = (5,5), 3 = (4,4), 4 = (2,2), 5 = (6,6)
};

static int plan_sep(const Bank *hb, const Bank *vb, int sw, int sh, int irow, int dw, int dh, int orow, int ntracks,
                    int hround, int hshift, int vround, int vshift, SepPlan *p) {
  SepArgs &a = p->a;
  a.sw = sw; a.sh = sh; a.irow = irow; a.dw = dw; a.dh = dh; a.orow = orow;
  a.hpos = hb->pos; a.hco = hb->co; a.nth = hb->nt;
  a.vpos = vb->pos; a.vco = vb->co; a.ntv = vb->nt;
  a.hround = hround; a.hshift = hshift; a.vround = vround; a.vshift = vshift;
  a.swt = (max_tile_span(*hb, dw, kTileW) + 3 + 3) & ~3;   // +3: window start rounded down to a 4-pixel column
  int th = 16;
  for (;; th >>= 1) {
    a.th = th;
    a.sht = max_tile_span(*vb, dh, th);
    p->lds = (size_t)a.sht * a.swt * 4 + (size_t)a.sht * kTileW * 8 + 256;
    if (p->lds <= 64 * 1024 || th == 1) break;
  }
  if (p->lds > 160 * 1024) { set_error("resize window does not fit LDS (%zu bytes)", p->lds); return LGPU_E_UNSUPPORTED; }
  a.tiles_x = (dw + kTileW - 1) / kTileW;
  a.tiles_y = (dh + a.th - 1) / a.th;
  p->grid = dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)ntracks, 1);
  p->variant = (a.nth == 8 && a.ntv == 8) ? 1 : (a.nth == 5 && a.ntv == 5) ? 2 : (a.nth == 4 && a.ntv == 4) ? 3 :
               (a.nth == 2 && a.ntv == 2) ? 4 : (a.nth == 6 && a.ntv == 6) ? 5 : 0;
  return LGPU_OK;
}

static int launch_sep(const SepPlan &p, const SepTracks &t, const Lut8 &l, hipStream_t st) {
  const dim3 blk(kBlock);
#define SEP_LAUNCH(H, V)                                                                                     \
  do {                                                                                                       \
    if (p.lds > 48 * 1024)                                                                                   \
      LGPU_HIP(hipFuncSetAttribute((const void *)k_separable<H, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds)); \
    hipLaunchKernelGGL((k_separable<H, V>), p.grid, blk, p.lds, st, p.a, t, l);                              \
  } while (0)
  switch (p.variant) {
  case 1: SEP_LAUNCH(8, 8); break;
  case 2: SEP_LAUNCH(5, 5); break;
  case 3: SEP_LAUNCH(4, 4); break;
  case 4: SEP_LAUNCH(2, 2); break;
  case 5: SEP_LAUNCH(6, 6); break;
  default: SEP_LAUNCH(0, 0); break;
  }
#undef SEP_LAUNCH
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// scratch for the multi-launch paths (per device; grown on demand, never shrunk)
static std::mutex g_scratch_mu;
static std::map<int, std::pair<void *, size_t>> g_scratch;
static int get_scratch(size_t bytes, void **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto &s = g_scratch[dev];
  if (s.second < bytes) {
    if (s.first) { LGPU_HIP(hipDeviceSynchronize()); LGPU_HIP(hipFree(s.first)); s.first = nullptr; s.second = 0; }
    if (hipMalloc(&s.first, bytes) != hipSuccess) { set_error("hipMalloc(%zu) for scratch failed", bytes); return LGPU_E_NOMEM; }
    s.second = bytes;
  }
  *out = s.first;
  return LGPU_OK;
}

}  // namespace lgpu

using namespace lgpu;

extern "C" int lgpu_resize(const uint8_t *src_d, int irow, int sw, int sh, uint8_t *dst_d, int orow, int dw, int dh,
                           int psize, int interp, const uint8_t *lut8, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && sw > 0 && sh > 0 && dw > 0 && dh > 0, "null frame or empty geometry");
  LGPU_REQUIRE(psize == 1 || psize == 3 || psize == 4, "psize must be 1, 3 or 4");
  LGPU_REQUIRE(irow >= sw * psize && orow >= dw * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(src_d != dst_d, "resize cannot run in place");
  const int kernel = kernel_for_interp(interp, dw > sw || dh > sh);
  const Bank *hb, *vb;
  if ((rc = get_bank(sw, dw, kernel, &hb))) return rc;
  if ((rc = get_bank(sh, dh, kernel, &vb))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const Lut8 l = pack_lut(lut8);
  const bool al4 = (((uintptr_t)src_d | (uintptr_t)irow | (uintptr_t)dst_d | (uintptr_t)orow) & 3) == 0;
  if (psize == 4 && al4) {
    SepPlan p;
    if ((rc = plan_sep(hb, vb, sw, sh, irow, dw, dh, orow, 1, 64, 7, 1 << 20, 21, &p)) == LGPU_OK) {
      p.a.src_sel = 0x03020100u; p.a.blend = 0; p.a.irow2 = 0; p.a.bf = 0; p.a.nbf = 255; p.a.bf_d = nullptr; p.a.use_lut = lut8 ? 1 : 0;
      p.a.vec = (((uintptr_t)src_d | (uintptr_t)irow) & 15) == 0;
      SepTracks t;
      t.src[0] = src_d; t.l2[0] = nullptr; t.dst[0] = dst_d;
      return launch_sep(p, t, l, st);
    }
    if (rc != LGPU_E_UNSUPPORTED) return rc;
  }
  // generic: horizontal into an int16 scratch, then vertical
  void *scratch;
  const size_t need = sizeof(int16_t) * (size_t)sh * dw * psize;
  if ((rc = get_scratch(need, &scratch))) return rc;
  unsigned gy = (unsigned)(sh > 2048 ? 2048 : sh);
  hipLaunchKernelGGL(k_hpass_generic, dim3(cdiv((unsigned)(dw * psize), kBlock), gy), dim3(kBlock), 0, st, src_d, irow, sw, sh,
                     (int16_t *)scratch, dw, psize, hb->pos, hb->co, hb->nt, 64, 7);
  LGPU_CHECK_LAUNCH();
  gy = (unsigned)(dh > 2048 ? 2048 : dh);
  hipLaunchKernelGGL(k_vpass_generic, dim3(cdiv((unsigned)(dw * psize), kBlock), gy), dim3(kBlock), 0, st, (const int16_t *)scratch, sh,
                     dst_d, orow, dw * psize, dh, vb->pos, vb->co, vb->nt, 1 << 20, 21, psize, lut8 ? 1 : 0, l);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_gauss5(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && width > 0 && height > 0, "null frame or empty geometry");
  LGPU_REQUIRE(psize == 1 || psize == 3 || psize == 4, "psize must be 1, 3 or 4");
  LGPU_REQUIRE(irow >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  LGPU_REQUIRE(src_d != dst_d, "gauss5 cannot run in place");
  const Bank *hb, *vb;
  if ((rc = get_bank(width, width, 100, &hb))) return rc;
  if ((rc = get_bank(height, height, 100, &vb))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const Lut8 l = pack_lut(nullptr);
  const bool al4 = (((uintptr_t)src_d | (uintptr_t)irow | (uintptr_t)dst_d | (uintptr_t)orow) & 3) == 0;
  if (psize == 4 && al4) {
    SepPlan p;
    if ((rc = plan_sep(hb, vb, width, height, irow, width, height, orow, 1, 0, 0, 128, 8, &p))) return rc;
    p.a.src_sel = 0x03020100u; p.a.blend = 0; p.a.irow2 = 0; p.a.bf = 0; p.a.nbf = 255; p.a.bf_d = nullptr; p.a.use_lut = 0;
    p.a.vec = (((uintptr_t)src_d | (uintptr_t)irow) & 15) == 0;
    SepTracks t;
    t.src[0] = src_d; t.l2[0] = nullptr; t.dst[0] = dst_d;
    return launch_sep(p, t, l, st);
  }
  void *scratch;
  if ((rc = get_scratch(sizeof(int16_t) * (size_t)height * width * psize, &scratch))) return rc;
  unsigned gy = (unsigned)(height > 2048 ? 2048 : height);
  hipLaunchKernelGGL(k_hpass_generic, dim3(cdiv((unsigned)(width * psize), kBlock), gy), dim3(kBlock), 0, st, src_d, irow, width, height,
                     (int16_t *)scratch, width, psize, hb->pos, hb->co, 5, 0, 0);
  LGPU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_vpass_generic, dim3(cdiv((unsigned)(width * psize), kBlock), gy), dim3(kBlock), 0, st, (const int16_t *)scratch,
                     height, dst_d, orow, width * psize, height, vb->pos, vb->co, 5, 128, 8, psize, 0, l);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

static int chain_launch(const lgpu_chain_params *pr, const lgpu_chain_track *tracks, int ntracks, hipStream_t st) {
  int rc;
  LGPU_REQUIRE(pr && tracks && ntracks > 0 && ntracks <= LGPU_CHAIN_MAX_TRACKS, "1..64 tracks");
  LGPU_REQUIRE(pr->sw > 0 && pr->sh > 0 && pr->dw > 0 && pr->dh > 0, "empty geometry");
  LGPU_REQUIRE(pr->irow >= pr->sw * 4 && pr->orow >= pr->dw * 4 && pr->irow2 >= pr->dw * 4, "rowstride smaller than a row");
  LGPU_REQUIRE(((pr->irow | pr->orow | pr->irow2) & 3) == 0, "rowstrides must be multiples of 4");
  for (int i = 0; i < ntracks; i++) {
    LGPU_REQUIRE(tracks[i].src_d && tracks[i].layer2_d && tracks[i].dst_d, "null track pointer");
    LGPU_REQUIRE((((uintptr_t)tracks[i].src_d | (uintptr_t)tracks[i].layer2_d | (uintptr_t)tracks[i].dst_d) & 3) == 0, "frames must be 4-byte aligned");
  }
  const bool same = (pr->sw == pr->dw && pr->sh == pr->dh);
  const int kernel = kernel_for_interp(pr->interp, pr->dw > pr->sw || pr->dh > pr->sh);
  const Lut8 l = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
  const uint32_t sel = pr->swap_rb ? 0x03000102u : 0x03020100u;   // swap3postalpha: [in2 in1 in0 in3]
  const Bank *hb, *vb, *gh, *gv;
  SepTracks t;
  uintptr_t src_bits = (uintptr_t)pr->irow;
  for (int i = 0; i < ntracks; i++) src_bits |= (uintptr_t)tracks[i].src_d;
  const int src_vec = (src_bits & 15) == 0;
  if (!pr->do_blur) {
    LGPU_REQUIRE(!same, "chain without resize: use lgpu_swizzle + lgpu_blend_chroma + lgpu_gamma_apply");
    if ((rc = get_bank(pr->sw, pr->dw, kernel, &hb)) || (rc = get_bank(pr->sh, pr->dh, kernel, &vb))) return rc;
    SepPlan p;
    if ((rc = plan_sep(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->orow, ntracks, 64, 7, 1 << 20, 21, &p))) return rc;
    p.a.src_sel = sel; p.a.blend = 1; p.a.irow2 = pr->irow2; p.a.bf = (uint32_t)pr->bf & 0xFF; p.a.nbf = 0xFF - p.a.bf; p.a.bf_d = pr->param_block_d;
    p.a.use_lut = pr->use_lut ? 1 : 0; p.a.vec = src_vec;
    for (int i = 0; i < ntracks; i++) { t.src[i] = tracks[i].src_d; t.l2[i] = tracks[i].layer2_d; t.dst[i] = tracks[i].dst_d; }
    return launch_sep(p, t, l, st);
  }
  // with blur: resize into scratch (per track), then gaussian with the blend + gamma epilogue
  LGPU_REQUIRE(!same, "chain needs a resize stage");
  void *scratch;
  const size_t per = (size_t)pr->dw * 4 * pr->dh;
  if ((rc = get_scratch(per * ntracks, &scratch))) return rc;
  if ((rc = get_bank(pr->sw, pr->dw, kernel, &hb)) || (rc = get_bank(pr->sh, pr->dh, kernel, &vb))) return rc;
  if ((rc = get_bank(pr->dw, pr->dw, 100, &gh)) || (rc = get_bank(pr->dh, pr->dh, 100, &gv))) return rc;
  SepPlan p1, p2;
  if ((rc = plan_sep(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->dw * 4, ntracks, 64, 7, 1 << 20, 21, &p1))) return rc;
  p1.a.src_sel = sel; p1.a.blend = 0; p1.a.irow2 = 0; p1.a.bf = 0; p1.a.nbf = 255; p1.a.bf_d = nullptr; p1.a.use_lut = 0; p1.a.vec = src_vec;
  for (int i = 0; i < ntracks; i++) { t.src[i] = tracks[i].src_d; t.l2[i] = nullptr; t.dst[i] = (uint8_t *)scratch + per * i; }
  if ((rc = launch_sep(p1, t, pack_lut(nullptr), st))) return rc;
  if ((rc = plan_sep(gh, gv, pr->dw, pr->dh, pr->dw * 4, pr->dw, pr->dh, pr->orow, ntracks, 0, 0, 128, 8, &p2))) return rc;
  p2.a.src_sel = 0x03020100u; p2.a.blend = 1; p2.a.irow2 = pr->irow2; p2.a.bf = (uint32_t)pr->bf & 0xFF; p2.a.nbf = 0xFF - p2.a.bf; p2.a.bf_d = pr->param_block_d;
  p2.a.use_lut = pr->use_lut ? 1 : 0;
  p2.a.vec = ((((uintptr_t)scratch | per | (uintptr_t)(pr->dw * 4)) & 15) == 0);
  for (int i = 0; i < ntracks; i++) { t.src[i] = (uint8_t *)scratch + per * i; t.l2[i] = tracks[i].layer2_d; t.dst[i] = tracks[i].dst_d; }
  return launch_sep(p2, t, l, st);
}

extern "C" int lgpu_chain(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  return chain_launch(params, tracks, ntracks, (hipStream_t)stream);
}

extern "C" int lgpu_chain_timed(const lgpu_chain_params *params, const lgpu_chain_track *tracks, int ntracks, int reps,
                                float *ms_total, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(reps > 0 && ms_total, "reps > 0");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  LGPU_HIP(hipEventCreate(&e0));
  LGPU_HIP(hipEventCreate(&e1));
  LGPU_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; i++)
    if ((rc = chain_launch(params, tracks, ntracks, st))) break;
  LGPU_HIP(hipEventRecord(e1, st));
  LGPU_HIP(hipEventSynchronize(e1));
  LGPU_HIP(hipEventElapsedTime(ms_total, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}
