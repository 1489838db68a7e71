// yuv.hip -- K2: planar YUV 4:2:0 / 4:2:2 -> packed RGB(A) with the reference's chroma super-sampling and
// (optionally) the gamma LUT of the gamma_convert_layer() pass fused into the store.
//
// Replaces convert_yuv420p_to_rgb_frame (src/colourspace.c:3260-3904).  HBM-bound: 1.5 B read + 3/4 B
// written per pixel.  One lane owns a 2x2 output quad (one chroma column pair k of one luma row pair),
// consecutive lanes own consecutive k, so luma loads are 2 B/lane contiguous, chroma loads hit the same
// cache lines three times (k-1, k, k+1) and stores are 8 B/lane contiguous per row.  The five int32
// conversion tables (5 KB) are staged in LDS once per workgroup together with the LUT.
//
// Reference behaviours kept bit-for-bit (see DESIGN.md, quirk list K2-a..e):
//   - left pixel of a pair: second-row U sum rebuilt from the first row (:3461); V of row r paired with the
//     previous V of row r+1 (:3544); "last V" of row r+1 frozen at column 0
//   - right pixel of the last pair reads the chroma sample one past the row end
//   - last row (1-thread reference): luma from row 0, "next" chroma from chroma row 0, even x only
// Pixels whose reference value is undefined (row 0 odd x: out-of-bounds table index; last row odd x: never
// written) get the evident intent.
#include "lgpu_common.h"
#include <atomic>

namespace lgpu {

// frames of one batch (same geometry): one launch converts them all, blockIdx.z picks the frame (tracks of a multitrack timeline)
struct YuvBatch {
  const uint8_t *y[LGPU_CHAIN_MAX_TRACKS], *u[LGPU_CHAIN_MAX_TRACKS], *v[LGPU_CHAIN_MAX_TRACKS];
  uint8_t *dst[LGPU_CHAIN_MAX_TRACKS];
};
struct YuvArgs {
  const uint8_t *y, *u, *v;
  uint8_t *dst;
  const int32_t *tables;   // [5][256] RGB_Y R_Cr G_Cb G_Cr B_Cb (device)
  long usize, vsize;
  int ys, us, vs, orow;
  int width, height;
  int opsize, order;       // order: 0 RGB(A), 1 BGR(A), 2 ARGB
  int clamped, low_quality, fix_edges, use_lut;
  const uint16_t *lut16;   // device, 65536 entries: the fused LUT16 of xyuv2rgb_with_gamma (:2386-2390), or null
};

__device__ __forceinline__ int cuv_c(int n) {   // CLAMP16_240 (src/colourspace.h:19)
  if (n < 0) return 16;
  if (n > 255 || (n & 0xF0) == 0xF0) return 240;
  return (n & 0xF0) ? n : 16;
}

struct YuvCtx {
  const int32_t *ty, *rcr, *gcb, *gcr, *bcb;
  const uint8_t *lut;
  const uint16_t *lut16;
  bool clamped, lowq, use_lut;
  int opsize, order;
  __device__ __forceinline__ int cuv(int n) const { return clamped ? cuv_c(n) : (n < 0 ? 0 : n > 255 ? 255 : n); }
  // (2a + b) / 3 and (a + 2b) / 3 on doubled sums; (int)(s / 3. + .5) == (s + 1) / 3 for s >= 0 (:3464-3469)
  __device__ __forceinline__ void vblend(int s1, int s2, int &top, int &bot) const {
    if (!lowq) { top = cuv((s1 + (s2 >> 1) + 1) / 3); bot = cuv(((s1 >> 1) + s2 + 1) / 3); }
    else { top = cuv(s1 >> 1); bot = cuv(s2 >> 1); }
  }
  __device__ __forceinline__ uint32_t rgb(int y, int u, int v) const {   // xyuv2rgb (:2351-2356), >>16 (:832-835)
    const int yy = ty[y];
    uint32_t r, g, b;
    if (lut16) {          // lut[CLAMP16biti(sum >> 8)] >> 8: a 128 KB table, L2 resident
      const int ir = (yy + rcr[v]) >> 8, ig = (yy + gcb[u] + gcr[v]) >> 8, ib = (yy + bcb[u]) >> 8;
      r = lut16[ir > 65535 ? 65535 : ir < 0 ? 0 : ir] >> 8;
      g = lut16[ig > 65535 ? 65535 : ig < 0 ? 0 : ig] >> 8;
      b = lut16[ib > 65535 ? 65535 : ib < 0 ? 0 : ib] >> 8;
    } else {
      r = clamp255((yy + rcr[v]) >> 16); g = clamp255((yy + gcb[u] + gcr[v]) >> 16); b = clamp255((yy + bcb[u]) >> 16);
      if (use_lut) { r = lut[r]; g = lut[g]; b = lut[b]; }
    }
    if (order == 0) return r | (g << 8) | (b << 16) | 0xFF000000u;
    if (order == 1) return b | (g << 8) | (r << 16) | 0xFF000000u;
    return 0xFFu | (r << 8) | (g << 16) | (b << 24);
  }
  __device__ __forceinline__ void store2(uint8_t *d, uint32_t p0, uint32_t p1) const {
    if (opsize == 4) {
      if ((reinterpret_cast<uintptr_t>(d) & 7) == 0) *reinterpret_cast<uint2 *>(d) = make_uint2(p0, p1);
      else { reinterpret_cast<uint32_t *>(d)[0] = p0; reinterpret_cast<uint32_t *>(d)[1] = p1; }
    } else {
      d[0] = (uint8_t)p0; d[1] = (uint8_t)(p0 >> 8); d[2] = (uint8_t)(p0 >> 16);
      d[3] = (uint8_t)p1; d[4] = (uint8_t)(p1 >> 8); d[5] = (uint8_t)(p1 >> 16);
    }
  }
};

// one (unit, chroma column k) cell of the 4:2:0 walk: 2 pixels of row 0, a 2 x 2 quad of a row pair, or 2 pixels of the trailing row
// units: 0 = row 0; p >= 1 = rows (2p-1, 2p) while 2p <= H-1; then (even H) the trailing row H-1
__device__ __forceinline__ void yuv420_cell(const YuvArgs &a, const YuvCtx &c, int unit, int k, int hw, int npairs) {
  const int ops = a.opsize;
  const int H = a.height;
  auto PU = [&](int r, int kk) -> int { long i = (long)r * a.us + kk; return a.u[i < a.usize ? i : a.usize - 1]; };
  auto PV = [&](int r, int kk) -> int { long i = (long)r * a.vs + kk; return a.v[i < a.vsize ? i : a.vsize - 1]; };
  if (unit == 0) {
    // row 0 (:3399-3443)
    const int kp = k ? k - 1 : 0, kn = (k + 1 < hw) ? k + 1 : hw - 1;
    const uint32_t p0 = c.rgb(a.y[2 * k], c.cuv((PU(0, k) + PU(0, kp)) >> 1), c.cuv((PV(0, k) + PV(0, kp)) >> 1));
    const uint32_t p1 = c.rgb(a.y[2 * k + 1], c.cuv((PU(0, k) + PU(0, kn)) >> 1), c.cuv((PV(0, k) + PV(0, kn)) >> 1));
    c.store2(a.dst + (size_t)(2 * k) * ops, p0, p1);
  } else if (unit <= npairs) {
    // rows (i, i+1), chroma rows r and r+1 (:3445-3554)
    const int i = 2 * unit - 1, r = i >> 1;
    const uint8_t *y0 = a.y + (size_t)i * a.ys + 2 * k, *y1 = y0 + a.ys;
    uint8_t *d0 = a.dst + (size_t)i * a.orow + (size_t)(2 * k) * ops, *d1 = d0 + a.orow;
    const int u_rk = PU(r, k), v_rk = PV(r, k), v_r1k = PV(r + 1, k);
    int ut, ub, vt, vb;
    // left pixel
    const int lu1 = k ? PU(r, k - 1) : u_rk;
    const int lv1 = k ? PV(r + 1, k - 1) : v_rk;
    const int lv2 = PV(r + 1, 0);
    c.vblend(u_rk + lu1, u_rk + lu1, ut, ub);
    c.vblend(v_rk + lv1, v_r1k + lv2, vt, vb);
    const uint32_t a0 = c.rgb(y0[0], ut, vt), b0 = c.rgb(y1[0], ub, vb);
    // right pixel
    c.vblend(u_rk + PU(r, k + 1), PU(r + 1, k) + PU(r + 1, k + 1), ut, ub);
    c.vblend(v_rk + PV(r, k + 1), v_r1k + PV(r + 1, k + 1), vt, vb);
    const uint32_t a1 = c.rgb(y0[1], ut, vt), b1 = c.rgb(y1[1], ub, vb);
    c.store2(d0, a0, a1);
    c.store2(d1, b0, b1);
  } else {
    // trailing row H-1 (:3556-3592)
    const int i = H - 1, r = i >> 1;
    const int kp = k ? k - 1 : 0, kn = (k + 1 < hw) ? k + 1 : hw - 1;
    const uint8_t *yr = a.y + (size_t)i * a.ys;
    uint32_t p0;
    if (a.fix_edges) {
      p0 = c.rgb(yr[2 * k], c.cuv((PU(r, k) + PU(r, kp)) >> 1), c.cuv((PV(r, k) + PV(r, kp)) >> 1));
    } else {
      // 1-thread reference: this/last walk = {row r col 0, row r col 0, row 0 col 1, row 0 col 2, ...}
      const int tu = k ? PU(0, k) : PU(r, 0), tv = k ? PV(0, k) : PV(r, 0);
      const int lu = (k >= 2) ? PU(0, k - 1) : PU(r, 0), lv = (k >= 2) ? PV(0, k - 1) : PV(r, 0);
      p0 = c.rgb(a.y[2 * k], c.cuv((tu + lu) >> 1), c.cuv((tv + lv) >> 1));
    }
    const uint32_t p1 = c.rgb(yr[2 * k + 1], c.cuv((PU(r, k) + PU(r, kn)) >> 1), c.cuv((PV(r, k) + PV(r, kn)) >> 1));
    c.store2(a.dst + (size_t)i * a.orow + (size_t)(2 * k) * ops, p0, p1);
  }
}

__global__ __launch_bounds__(kBlock) void k_yuv420p_to_rgb(YuvArgs a, Lut8 lut, YuvBatch bt, int batched) {
  if (batched) { a.y = bt.y[blockIdx.z]; a.u = bt.u[blockIdx.z]; a.v = bt.v[blockIdx.z]; a.dst = bt.dst[blockIdx.z]; }
  __shared__ int32_t s_tab[5 * 256];
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  const int hw = a.width >> 1;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  const int npairs = (a.height - 1) / 2;                      // number of full row pairs starting at row 1
  const int nunits = 1 + npairs + (((a.height - 1) & 1) ? 1 : 0);
  // A single 1080p frame is one cell per lane: the workgroup's 5 KB of tables and the cell's fifteen samples are both a memory latency away.  The samples of the
  // first cell are requested BEFORE the tables are staged (interior cells of a row pair only -- everything else walks yuv420_cell as before), so the two
  // latencies overlap instead of adding up.
  const int unit0 = blockIdx.y;
  const int i0 = 2 * unit0 - 1, r0 = i0 >> 1;
  const bool pre = k >= 1 && k + 1 < hw && unit0 >= 1 && unit0 <= npairs && (long)(r0 + 1) * a.us + k + 2 <= a.usize && (long)(r0 + 1) * a.vs + k + 2 <= a.vsize;
  int py00 = 0, py01 = 0, py10 = 0, py11 = 0, u_l = 0, u_c = 0, u_n = 0, u1_c = 0, u1_n = 0, v_c = 0, v_n = 0, v1_l = 0, v1_c = 0, v1_n = 0, v1_0 = 0;
  if (pre) {
    const uint8_t *y0 = a.y + (size_t)i0 * a.ys + 2 * k, *y1 = y0 + a.ys;
    const uint8_t *ur = a.u + (size_t)r0 * a.us + k, *vr = a.v + (size_t)r0 * a.vs + k;
    py00 = y0[0]; py01 = y0[1]; py10 = y1[0]; py11 = y1[1];
    u_l = ur[-1]; u_c = ur[0]; u_n = ur[1]; u1_c = ur[a.us]; u1_n = ur[a.us + 1];
    v_c = vr[0]; v_n = vr[1]; v1_l = vr[a.vs - 1]; v1_c = vr[a.vs]; v1_n = vr[a.vs + 1]; v1_0 = a.v[(size_t)(r0 + 1) * a.vs];
  }
  for (int i = threadIdx.x; i < 5 * 256; i += kBlock) s_tab[i] = a.tables[i];
  stage_lut(s_lut, lut);
  __syncthreads();
  YuvCtx c;
  c.ty = s_tab; c.rcr = s_tab + 256; c.gcb = s_tab + 512; c.gcr = s_tab + 768; c.bcb = s_tab + 1024;
  c.lut = s_lut; c.lut16 = a.lut16; c.clamped = a.clamped; c.lowq = a.low_quality; c.use_lut = a.use_lut; c.opsize = a.opsize; c.order = a.order;
  if (k >= hw) return;
  int unit = unit0;
  if (pre) {
    // the row-pair branch of yuv420_cell on the samples already here (same expressions, :3445-3554)
    uint8_t *d0 = a.dst + (size_t)i0 * a.orow + (size_t)(2 * k) * a.opsize, *d1 = d0 + a.orow;
    int ut, ub, vt, vb;
    c.vblend(u_c + u_l, u_c + u_l, ut, ub);
    c.vblend(v_c + v1_l, v1_c + v1_0, vt, vb);
    const uint32_t a0 = c.rgb(py00, ut, vt), b0 = c.rgb(py10, ub, vb);
    c.vblend(u_c + u_n, u1_c + u1_n, ut, ub);
    c.vblend(v_c + v_n, v1_c + v1_n, vt, vb);
    const uint32_t a1 = c.rgb(py01, ut, vt), b1 = c.rgb(py11, ub, vb);
    c.store2(d0, a0, a1);
    c.store2(d1, b0, b1);
    unit += gridDim.y;
  }
  for (; unit < nunits; unit += gridDim.y) yuv420_cell(a, c, unit, k, hw, npairs);
}

// The same walk with one lane per FOUR chroma columns (8 x 2 pixels of a row pair): the quad's luma comes in as two 8-byte loads and
// every chroma row as three dwords (the columns left of, of and right of the group), 16 loads instead of 64 byte loads, and the two
// output rows leave as 16-byte stores.  Only cells whose loads stay inside the planes and whose rows are aligned take this form
// (`fast`, decided on the host per launch + per lane here); every other cell (row 0, the trailing row, the first column group, the
// plane ends) goes through yuv420_cell above, so the result is the same bytes.
__global__ __launch_bounds__(kBlock) void k_yuv420p_to_rgb4(YuvArgs a, Lut8 lut, YuvBatch bt, int batched) {
  if (batched) { a.y = bt.y[blockIdx.z]; a.u = bt.u[blockIdx.z]; a.v = bt.v[blockIdx.z]; a.dst = bt.dst[blockIdx.z]; }
  __shared__ int32_t s_tab[5 * 256];
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  for (int i = threadIdx.x; i < 5 * 256; i += kBlock) s_tab[i] = a.tables[i];
  stage_lut(s_lut, lut);
  __syncthreads();
  YuvCtx c;
  c.ty = s_tab; c.rcr = s_tab + 256; c.gcb = s_tab + 512; c.gcr = s_tab + 768; c.bcb = s_tab + 1024;
  c.lut = s_lut; c.lut16 = a.lut16; c.clamped = a.clamped; c.lowq = a.low_quality; c.use_lut = a.use_lut; c.opsize = a.opsize; c.order = a.order;
  const int hw = a.width >> 1;
  const int k0 = 4 * (blockIdx.x * kBlock + threadIdx.x);
  if (k0 >= hw) return;
  const int npairs = (a.height - 1) / 2;
  const int nunits = 1 + npairs + (((a.height - 1) & 1) ? 1 : 0);
  for (int unit = blockIdx.y; unit < nunits; unit += gridDim.y) {
    const int i = 2 * unit - 1, r = i >> 1;
    const bool fast = unit >= 1 && unit <= npairs && k0 >= 4 && k0 + 4 <= hw && (long)(r + 1) * a.us + k0 + 8 <= a.usize &&
                      (long)(r + 1) * a.vs + k0 + 8 <= a.vsize;
    if (!fast) {
      for (int k = k0; k < k0 + 4 && k < hw; k++) yuv420_cell(a, c, unit, k, hw, npairs);
      continue;
    }
    const uint2 ya = *reinterpret_cast<const uint2 *>(a.y + (size_t)i * a.ys + 2 * k0), yb = *reinterpret_cast<const uint2 *>(a.y + (size_t)(i + 1) * a.ys + 2 * k0);
    const uint8_t *ur = a.u + (size_t)r * a.us + k0, *vr = a.v + (size_t)r * a.vs + k0;
    const uint32_t u0m = *reinterpret_cast<const uint32_t *>(ur - 4), u0c = *reinterpret_cast<const uint32_t *>(ur), u0p = *reinterpret_cast<const uint32_t *>(ur + 4);
    const uint32_t u1c = *reinterpret_cast<const uint32_t *>(ur + a.us), u1p = *reinterpret_cast<const uint32_t *>(ur + a.us + 4);
    const uint32_t v0c = *reinterpret_cast<const uint32_t *>(vr), v0p = *reinterpret_cast<const uint32_t *>(vr + 4);
    const uint32_t v1m = *reinterpret_cast<const uint32_t *>(vr + a.vs - 4), v1c = *reinterpret_cast<const uint32_t *>(vr + a.vs), v1p = *reinterpret_cast<const uint32_t *>(vr + a.vs + 4);
    const int lv2 = a.v[(size_t)(r + 1) * a.vs];                            // PV(r + 1, 0): the reference's constant "last" sample
    // samples k0-1 .. k0+4 of each row as 6-element byte strings: [m.b3, c.b0..b3, p.b0]
    auto at6 = [](uint32_t m, uint32_t cc, uint32_t pp, int j) -> int {   // j = -1 .. 4
      return j < 0 ? (int)(m >> 24) : j < 4 ? (int)((cc >> (8 * j)) & 0xFF) : (int)(pp & 0xFF);
    };
    uint32_t top[8], bot[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int u_rk = at6(0, u0c, 0, j), v_rk = at6(0, v0c, 0, j), v_r1k = at6(0, v1c, 0, j);
      const int lu1 = at6(u0m, u0c, u0p, j - 1), lv1 = at6(v1m, v1c, v1p, j - 1);
      int ut, ub, vt, vb;
      c.vblend(u_rk + lu1, u_rk + lu1, ut, ub);
      c.vblend(v_rk + lv1, v_r1k + lv2, vt, vb);
      const int y00 = (int)(((j < 2 ? ya.x : ya.y) >> (16 * (j & 1))) & 0xFF), y01 = (int)(((j < 2 ? ya.x : ya.y) >> (16 * (j & 1) + 8)) & 0xFF);
      const int y10 = (int)(((j < 2 ? yb.x : yb.y) >> (16 * (j & 1))) & 0xFF), y11 = (int)(((j < 2 ? yb.x : yb.y) >> (16 * (j & 1) + 8)) & 0xFF);
      top[2 * j] = c.rgb(y00, ut, vt); bot[2 * j] = c.rgb(y10, ub, vb);
      c.vblend(u_rk + at6(u0m, u0c, u0p, j + 1), at6(0, u1c, u1p, j) + at6(0, u1c, u1p, j + 1), ut, ub);
      c.vblend(v_rk + at6(0, v0c, v0p, j + 1), v_r1k + at6(v1m, v1c, v1p, j + 1), vt, vb);
      top[2 * j + 1] = c.rgb(y01, ut, vt); bot[2 * j + 1] = c.rgb(y11, ub, vb);
    }
    uint4 *d0 = reinterpret_cast<uint4 *>(a.dst + (size_t)i * a.orow + (size_t)(2 * k0) * 4), *d1 = reinterpret_cast<uint4 *>(a.dst + (size_t)(i + 1) * a.orow + (size_t)(2 * k0) * 4);
    d0[0] = make_uint4(top[0], top[1], top[2], top[3]); d0[1] = make_uint4(top[4], top[5], top[6], top[7]);
    d1[0] = make_uint4(bot[0], bot[1], bot[2], bot[3]); d1[1] = make_uint4(bot[4], bot[5], bot[6], bot[7]);
  }
}

// ---- the same cells for launches that fill the device (batches of tracks, 4K frames): k_yuv420p_to_rgb16 ---------------------------
// profiles/r02/k2_pmc.md: k_yuv420p_to_rgb4 on the 16-frame batch spends 58 VALU operations and 9.3 LDS gathers per pixel, 53 % of its
// LDS cycles are bank conflicts (random indices into 256-entry tables: ~3.5 lanes of every 32 meet on a bank).  This kernel keeps the
// cell arithmetic (same bytes) and changes what is around it:
//   * tables in SIXTEEN interleaved copies (entry i of copy c at slot 16 i + c; lane uses copy lane & 15): a copy owns 2 of the 32
//     banks and is used by 2 lanes of a half-wave, so a gather costs at most 2 LDS cycles per half instead of ~3.5;
//   * {R_Cr, G_Cr}[v] and {G_Cb, B_Cb}[u] as 8-byte entries: 3 table gathers per pixel instead of 5 (the left pixels of a quad share
//     their U pair: 11 per 2 x 2 quad);
//   * CLAMP0255f + gamma LUT as ONE table indexed by (sum >> 16) + kY16Bias (RGB_Y carries the bias): no clamp instructions;
//   * CLAMP16_240 / the 0..255 clamp as one v_med3, / 3 as a multiply-shift, chroma bytes taken by SDWA operands.
// 139 KB of tables per workgroup, so a workgroup is 1024 threads, one per CU, persistent over the cells (dealt out round-robin over the whole grid).
// Cells near the frame edges go through yuv420_cell() as before.
constexpr int kY16Bias = 320, kY16Lut = 896;        // (sum >> 16) of every table set lies in [-320, 575] (checked on the host per launch)
constexpr int kY16OffLut = 0, kY16OffTy = kY16Lut * 64, kY16OffRG = kY16OffTy + 16384, kY16OffGB = kY16OffRG + 32768, kY16OffTab = kY16OffGB + 32768,
              kY16OffLut8 = kY16OffTab + 5 * 1024, kY16Lds = kY16OffLut8 + 256;     // the table bases travel in the per-lane copy offsets, the LUT sits at 0
template <int ORDER, bool LUT>
__global__ __launch_bounds__(1024) void k_yuv420p_to_rgb16(YuvArgs a, Lut8 lut, YuvBatch bt, int nframes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *s_ty = reinterpret_cast<uint32_t *>(smem + kY16OffTy);
  uint2 *s_rg = reinterpret_cast<uint2 *>(smem + kY16OffRG), *s_gb = reinterpret_cast<uint2 *>(smem + kY16OffGB);
  uint32_t *s_lx = reinterpret_cast<uint32_t *>(smem + kY16OffLut);
  int32_t *s_tab = reinterpret_cast<int32_t *>(smem + kY16OffTab);
  uint8_t *s_lut = smem + kY16OffLut8;
  const int tid = threadIdx.x;
  for (int i = tid; i < 5 * 256; i += 1024) s_tab[i] = a.tables[i];
  stage_lut(s_lut, lut);
  __syncthreads();
  const int clo = a.clamped ? 16 : 0, chi = a.clamped ? 240 : 255;
  for (int i = tid; i < 256 * 16; i += 1024) {
    const int e = i >> 4;
    s_ty[i] = (uint32_t)(s_tab[e] + (kY16Bias << 16));
    const int ec = e < clo ? clo : e > chi ? chi : e;                                   // CLAMP16_240 / 0..255 of the blended chroma, folded into the index
    s_rg[i] = make_uint2((uint32_t)s_tab[256 + ec], (uint32_t)s_tab[768 + ec]);        // R_Cr, G_Cr (indexed by V)
    s_gb[i] = make_uint2((uint32_t)s_tab[512 + ec], (uint32_t)s_tab[1024 + ec]);       // G_Cb, B_Cb (indexed by U)
  }
  for (int i = tid; i < kY16Lut * 16; i += 1024) {
    const int e = (i >> 4) - kY16Bias, cl = e < 0 ? 0 : e > 255 ? 255 : e;
    s_lx[i] = a.use_lut ? s_lut[cl] : (uint32_t)cl;
  }
  __syncthreads();
  YuvCtx c;
  c.ty = s_tab; c.rcr = s_tab + 256; c.gcb = s_tab + 512; c.gcr = s_tab + 768; c.bcb = s_tab + 1024;
  c.lut = s_lut; c.lut16 = nullptr; c.clamped = a.clamped; c.lowq = false; c.use_lut = a.use_lut; c.opsize = 4; c.order = ORDER;
  const int hw = a.width >> 1, ncg = (hw + 3) >> 2;
  const int npairs = (a.height - 1) / 2;
  const int nunits = 1 + npairs + (((a.height - 1) & 1) ? 1 : 0);
  const int total = nunits * nframes;
  const uint32_t zmagic = (uint32_t)(((1ull << 32) + (uint32_t)nunits - 1u) / (uint32_t)nunits);   // ul / nunits == umulhi(ul, zmagic) for ul < 64 * nunits, 2 <= nunits < 8192
  const uint32_t c4 = (uint32_t)(tid & 15) * 4u, c4y = c4 + kY16OffTy, c8v = c4 * 2u + kY16OffRG, c8u = c4 * 2u + kY16OffGB;
  typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) u32x2v *lds_u64;
  // A thread walks its cells with the loads of the NEXT cell issued before the arithmetic of the current one (16 waves per CU hide little latency by themselves).
  // Cells (unit, column group) are numbered linearly over the whole launch and dealt out round-robin to ALL threads of the grid: with a static
  // (unit -> workgroup quarter, column group -> lane) split 16 of every 256 lanes had no column group at 1080p (96 of 256 at 1280 wide) and the workgroups
  // got 8 or 9 units.  At 16 x 1080p both splits measure the same (53.2 us, A / B on one box): the launch is bound by the LDS gathers, not by lane-time.
  struct Cell { uint2 ya, yb; uint32_t u0m, u0c, u0p, u1c, u1p, v0c, v0p, v1m, v1c, v1p, lv2; int z, unit, k0; bool valid, fast; };
  const uint32_t cstep = gridDim.x * 1024u;
  const int dul = (int)(cstep / (uint32_t)ncg), dcg = (int)(cstep - (uint32_t)dul * (uint32_t)ncg);      // one step of a thread in (unit, column group) terms
  auto fetch = [&](int ul, int cg) -> Cell {
    Cell q;
    q.valid = ul < total;
    q.fast = false;
    if (!q.valid) return q;
    q.z = (int)__umulhi((uint32_t)ul, zmagic); q.unit = ul - q.z * nunits; q.k0 = 4 * cg;
    const int i = 2 * q.unit - 1, r = i >> 1;
    q.fast = q.unit >= 1 && q.unit <= npairs && q.k0 + 4 <= hw && (long)(r + 1) * a.us + q.k0 + 8 <= a.usize &&
             (long)(r + 1) * a.vs + q.k0 + 8 <= a.vsize;
    if (!q.fast) return q;
    const uint8_t *py = bt.y[q.z], *pu = bt.u[q.z], *pv = bt.v[q.z];
    q.ya = *reinterpret_cast<const uint2 *>(py + (size_t)i * a.ys + 2 * q.k0); q.yb = *reinterpret_cast<const uint2 *>(py + (size_t)(i + 1) * a.ys + 2 * q.k0);
    const uint8_t *ur = pu + (size_t)r * a.us + q.k0, *vr = pv + (size_t)r * a.vs + q.k0;
    q.u0c = *reinterpret_cast<const uint32_t *>(ur); q.u0p = *reinterpret_cast<const uint32_t *>(ur + 4);
    q.u1c = *reinterpret_cast<const uint32_t *>(ur + a.us); q.u1p = *reinterpret_cast<const uint32_t *>(ur + a.us + 4);
    q.v0c = *reinterpret_cast<const uint32_t *>(vr); q.v0p = *reinterpret_cast<const uint32_t *>(vr + 4);
    q.v1c = *reinterpret_cast<const uint32_t *>(vr + a.vs); q.v1p = *reinterpret_cast<const uint32_t *>(vr + a.vs + 4);
    // the sample left of the group: column k0 - 1, or for the first group the reference's k == 0 case: lu1 = U[r][0], lv1 = V[r][0] (:3447-3452)
    if (q.k0) { q.u0m = *reinterpret_cast<const uint32_t *>(ur - 4); q.v1m = *reinterpret_cast<const uint32_t *>(vr + a.vs - 4); }
    else { q.u0m = q.u0c << 24; q.v1m = q.v0c << 24; }
    q.lv2 = pv[(size_t)(r + 1) * a.vs];                                     // PV(r + 1, 0): the reference's constant "last" sample
    return q;
  };
  // sample j = -1 .. 4 of a chroma row held as [m.b3 | c.b0..b3 | p.b0]
  auto at6 = [](uint32_t m, uint32_t cc, uint32_t pp, int j) -> uint32_t { return j < 0 ? (m >> 24) : j < 4 ? ((cc >> (8 * j)) & 0xFF) : (pp & 0xFF); };
  // (2a + b) / 3 resp. (a + 2b) / 3 on doubled sums, clamped, as the byte offset of the 8-byte table entry: (int)(s / 3. + .5) == (s + 1) / 3
  // -> LDS byte address of the 8-byte table entry (the clamp lives in the table, the copy and table offsets in cbase)
  auto blend = [&](uint32_t s1, uint32_t s2, uint32_t cbase) -> uint32_t { return ((__umul24(s1 + (s2 >> 1) + 1u, 43691u) >> 17) << 7) + cbase; };
  auto pixel = [&](uint32_t yv, uint32_t ua, uint32_t va) -> uint32_t {
    const uint32_t yy = *(lds_u32)(uintptr_t)((yv << 6) + c4y);
    const u32x2v rg = *(lds_u64)(uintptr_t)va, gb = *(lds_u64)(uintptr_t)ua;
    const uint32_t sr = yy + rg.x, sg = yy + gb.x + rg.y, sb = yy + gb.y;
    uint32_t r_, g_, b_;
    if (LUT) {
      r_ = *(lds_u32)(uintptr_t)(kY16OffLut + (((sr >> 10) & 0xFFC0u) | c4));
      g_ = *(lds_u32)(uintptr_t)(kY16OffLut + (((sg >> 10) & 0xFFC0u) | c4));
      b_ = *(lds_u32)(uintptr_t)(kY16OffLut + (((sb >> 10) & 0xFFC0u) | c4));
    } else {            // no gamma LUT: CLAMP0255f alone is one v_med3 per channel instead of an LDS gather (half of the kernel's gathers)
      r_ = (uint32_t)min(max((int)(sr >> 16) - kY16Bias, 0), 255);
      g_ = (uint32_t)min(max((int)(sg >> 16) - kY16Bias, 0), 255);
      b_ = (uint32_t)min(max((int)(sb >> 16) - kY16Bias, 0), 255);
    }
    // two byte permutes per pixel (selector 0x0C = 0x00, 0x0D = 0xFF: the alpha byte costs nothing)
    if (ORDER == 0) return __builtin_amdgcn_perm(b_, __builtin_amdgcn_perm(g_, r_, 0x0C0C0400u), 0x0D040100u);
    if (ORDER == 1) return __builtin_amdgcn_perm(r_, __builtin_amdgcn_perm(g_, b_, 0x0C0C0400u), 0x0D040100u);
    return __builtin_amdgcn_perm(b_, __builtin_amdgcn_perm(g_, r_, 0x0C04000Du), 0x04020100u);
  };
  const uint32_t idx0 = blockIdx.x * 1024u + (uint32_t)tid;
  int ul = (int)(idx0 / (uint32_t)ncg), cg = (int)(idx0 - (uint32_t)ul * (uint32_t)ncg);
  Cell cur = fetch(ul, cg);
  while (cur.valid) {
    ul += dul; cg += dcg;
    if (cg >= ncg) { cg -= ncg; ul++; }
    const Cell nxt = fetch(ul, cg);
    if (!cur.fast) {
      YuvArgs f = a;
      f.y = bt.y[cur.z]; f.u = bt.u[cur.z]; f.v = bt.v[cur.z]; f.dst = bt.dst[cur.z];
      for (int k = cur.k0; k < cur.k0 + 4 && k < hw; k++) yuv420_cell(f, c, cur.unit, k, hw, npairs);
    } else {
      const int i = 2 * cur.unit - 1;
      uint32_t top[8], bot[8];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t u_rk = at6(0, cur.u0c, 0, j), v_rk = at6(0, cur.v0c, 0, j), v_r1k = at6(0, cur.v1c, 0, j);
        const uint32_t yw0 = j < 2 ? cur.ya.x : cur.ya.y, yw1 = j < 2 ? cur.yb.x : cur.yb.y;
        const uint32_t y00 = (yw0 >> (16 * (j & 1))) & 0xFF, y01 = (yw0 >> (16 * (j & 1) + 8)) & 0xFF;
        const uint32_t y10 = (yw1 >> (16 * (j & 1))) & 0xFF, y11 = (yw1 >> (16 * (j & 1) + 8)) & 0xFF;
        // left pixel: U row pair (s, s) -> top == bottom (:3461); V of row r with the previous V of row r + 1, the "last V" frozen at column 0 (:3544)
        const uint32_t su = u_rk + at6(cur.u0m, cur.u0c, cur.u0p, j - 1);
        const uint32_t uleft = blend(su, su, c8u);
        const uint32_t s1v = v_rk + at6(cur.v1m, cur.v1c, cur.v1p, j - 1), s2v = v_r1k + cur.lv2;
        top[2 * j] = pixel(y00, uleft, blend(s1v, s2v, c8v));
        bot[2 * j] = pixel(y10, uleft, blend(s2v, s1v, c8v));
        // right pixel
        const uint32_t s1u = u_rk + at6(cur.u0m, cur.u0c, cur.u0p, j + 1), s2u = at6(0, cur.u1c, cur.u1p, j) + at6(0, cur.u1c, cur.u1p, j + 1);
        const uint32_t s1w = v_rk + at6(0, cur.v0c, cur.v0p, j + 1), s2w = v_r1k + at6(cur.v1m, cur.v1c, cur.v1p, j + 1);
        top[2 * j + 1] = pixel(y01, blend(s1u, s2u, c8u), blend(s1w, s2w, c8v));
        bot[2 * j + 1] = pixel(y11, blend(s2u, s1u, c8u), blend(s2w, s1w, c8v));
      }
      uint8_t *dst = bt.dst[cur.z];
      uint4 *d0 = reinterpret_cast<uint4 *>(dst + (size_t)i * a.orow + (size_t)(2 * cur.k0) * 4), *d1 = reinterpret_cast<uint4 *>(dst + (size_t)(i + 1) * a.orow + (size_t)(2 * cur.k0) * 4);
      // non-temporal: written once, not read back by this launch (the same change bought 1.3 % on the chain kernel, profiles/r02/ring_experiment.md)
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      u32x4s *e0 = reinterpret_cast<u32x4s *>(d0), *e1 = reinterpret_cast<u32x4s *>(d1);
      const u32x4s t0 = {top[0], top[1], top[2], top[3]}, t1 = {top[4], top[5], top[6], top[7]}, b0 = {bot[0], bot[1], bot[2], bot[3]}, b1 = {bot[4], bot[5], bot[6], bot[7]};
      __builtin_nontemporal_store(t0, e0); __builtin_nontemporal_store(t1, e0 + 1);
      __builtin_nontemporal_store(b0, e1); __builtin_nontemporal_store(b1, e1 + 1);
    }
    cur = nxt;
  }
}

// 4:2:2 (:3593-3640 / :3858-3901): row i, pair k; "last/this" are seeded from chroma row i>>1 (reference)
__global__ __launch_bounds__(kBlock) void k_yuv422p_to_rgb(YuvArgs a, Lut8 lut, YuvBatch bt, int batched) {
  if (batched) { a.y = bt.y[blockIdx.z]; a.u = bt.u[blockIdx.z]; a.v = bt.v[blockIdx.z]; a.dst = bt.dst[blockIdx.z]; }
  __shared__ int32_t s_tab[5 * 256];
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  for (int i = threadIdx.x; i < 5 * 256; i += kBlock) s_tab[i] = a.tables[i];
  stage_lut(s_lut, lut);
  __syncthreads();
  YuvCtx c;
  c.ty = s_tab; c.rcr = s_tab + 256; c.gcb = s_tab + 512; c.gcr = s_tab + 768; c.bcb = s_tab + 1024;
  c.lut = s_lut; c.lut16 = a.lut16; c.clamped = a.clamped; c.lowq = a.low_quality; c.use_lut = a.use_lut; c.opsize = a.opsize; c.order = a.order;
  const int hw = a.width >> 1;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= hw) return;
  auto PU = [&](int r, int kk) -> int { long i = (long)r * a.us + kk; return a.u[i < a.usize ? i : a.usize - 1]; };
  auto PV = [&](int r, int kk) -> int { long i = (long)r * a.vs + kk; return a.v[i < a.vsize ? i : a.vsize - 1]; };
  for (int i = blockIdx.y; i < a.height; i += gridDim.y) {
    const int tu = k ? PU(i, k) : PU(i >> 1, 0), tv = k ? PV(i, k) : PV(i >> 1, 0);
    const int lu = (k >= 2) ? PU(i, k - 1) : PU(i >> 1, 0), lv = (k >= 2) ? PV(i, k - 1) : PV(i >> 1, 0);
    const int nu = PU(i, k + 1), nv = PV(i, k + 1);
    const uint8_t *yr = a.y + (size_t)i * a.ys + 2 * k;
    const uint32_t p0 = c.rgb(yr[0], c.cuv((tu + lu) >> 1), c.cuv((tv + lv) >> 1));
    const uint32_t p1 = c.rgb(yr[1], c.cuv((tu + nu) >> 1), c.cuv((tv + nv) >> 1));
    c.store2(a.dst + (size_t)i * a.orow + (size_t)(2 * k) * a.opsize, p0, p1);
  }
}

}  // namespace lgpu

using namespace lgpu;

static int yuv420p_to_rgb_impl(const uint8_t *y_d, const uint8_t *u_d, const uint8_t *v_d, const int istrides[3],
                               long u_size, long v_size, uint8_t *dst_d, int orow, int width, int height,
                               int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                               const uint8_t *lut8, const uint16_t *lut16_d, int flags, void *stream,
                               const YuvBatch *batch = nullptr, int nbatch = 1) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(y_d && u_d && v_d && dst_d && istrides, "null plane");
  LGPU_REQUIRE(width >= 2 && !(width & 1) && height >= 1, "width must be even and >= 2");
  LGPU_REQUIRE(opsize == 3 || opsize == 4, "opsize must be 3 or 4");
  LGPU_REQUIRE(out_order >= 0 && out_order <= 2, "bad out_order");
  if (out_order == 2) opsize = 4;
  LGPU_REQUIRE(orow >= width * opsize, "output rowstride smaller than a row");
  LGPU_REQUIRE(opsize == 3 || (((uintptr_t)dst_d | (uintptr_t)orow) & 3) == 0, "4-byte output must be 4-byte aligned");
  LGPU_REQUIRE(u_size > 0 && v_size > 0, "chroma plane sizes required");
  // pb_quality HIGH (3, what init_conversions() selects for rendering / transcoding, :2104-2108) only changes _spc_rnd (:832-835):
  // (int)((float)val / 65536.) instead of val >> 16.  (float)val is exact below 2^24 and every larger sum clamps to 255 either way;
  // for negative sums truncation and floor both end below the lower clamp: bit-identical to MED after CLAMP0255f / the Y, UV clamps
  // (tests/test_oracle_cpu.py checks the reference slices with both settings, and the scalar identity exhaustively).
  LGPU_REQUIRE(pb_quality >= 1 && pb_quality <= 3, "pb_quality is 1 (LOW), 2 (MED) or 3 (HIGH)");
  const DeviceTables *t = device_tables();
  YuvArgs a;
  a.y = y_d; a.u = u_d; a.v = v_d; a.dst = dst_d; a.tables = t->yuv2rgb[which_tables & 3];
  a.usize = u_size; a.vsize = v_size;
  a.ys = istrides[0]; a.us = istrides[1]; a.vs = istrides[2]; a.orow = orow;
  a.width = width; a.height = height; a.opsize = opsize; a.order = out_order;
  a.clamped = !(which_tables & 1); a.low_quality = (pb_quality == 1); a.fix_edges = (flags & LGPU_YUV_FIX_EDGES) ? 1 : 0;
  a.use_lut = lut8 ? 1 : 0;
  a.lut16 = lut16_d;
  const Lut8 l = pack_lut(lut8);
  const int units = is_422 ? height : height / 2 + 1;
  // every workgroup stages 5 KB of tables first: a few hundred workgroups that each walk several row pairs, not one per row pair
  static const int gy_cap = getenv("LGPU_YUV_GY") ? atoi(getenv("LGPU_YUV_GY")) : 2048;
  dim3 grid(cdiv((unsigned)(width >> 1), kBlock), (unsigned)(units > gy_cap ? gy_cap : units), (unsigned)nbatch);
  YuvBatch none = {};
  const YuvBatch &bt = batch ? *batch : none;
  static const bool classic = getenv("LGPU_YUV_CLASSIC") != nullptr;
  bool wide = !is_422 && !classic && opsize == 4 && (a.ys & 7) == 0 && (a.us & 3) == 0 && (a.vs & 3) == 0 && (orow & 15) == 0;
  for (int f = 0; f < nbatch && wide; f++) {
    const uintptr_t py = (uintptr_t)(batch ? batch->y[f] : y_d), pu = (uintptr_t)(batch ? batch->u[f] : u_d), pv = (uintptr_t)(batch ? batch->v[f] : v_d),
                    pd = (uintptr_t)(batch ? batch->dst[f] : dst_d);
    wide = (py & 7) == 0 && (pu & 3) == 0 && (pv & 3) == 0 && (pd & 15) == 0;
  }
  // one lane per four columns means a quarter of the lanes: only worth it when the launch still fills the device (a batch of tracks, or a
  // frame of 4K and up); a single 1080p frame keeps the one-column form (measured 13 vs 18 us)
  const dim3 g4x(cdiv((unsigned)((width >> 1) + 3) / 4, kBlock), grid.y, grid.z);
  if (wide && (unsigned long long)g4x.x * g4x.y * g4x.z < 2048ull && !getenv("LGPU_YUV_WIDE") && !getenv("LGPU_YUV_FORCE16")) wide = false;
  // launches that fill the device take the 16-copy-table kernel (one 1024-thread workgroup per CU, 139 KB of tables each)
  static const bool no16 = getenv("LGPU_YUV_NO16") != nullptr;
  const bool force16 = getenv("LGPU_YUV_FORCE16") != nullptr;            // tests: the 16-copy kernel at any size
  if (wide && !no16 && !lut16_d && !a.low_quality && units >= 2 && units < 8192 && nbatch <= 64 && (force16 || (unsigned long long)g4x.x * g4x.y * g4x.z * kBlock >= 256ull * 1024ull)) {
    // the clamp + LUT table covers (sum >> 16) in [-kY16Bias, kY16Lut - kY16Bias): true for the reference's four table sets, checked here
    static std::atomic<int> range_ok[4];            // 0 unknown, 1 ok, -1 no (host threads race to the same answer: a pure function of the table set)
    const int w4 = which_tables & 3;
    if (!range_ok[w4]) {
      int32_t rgb2yuv[9 * 256], t5[5 * 256];
      lgpu_conversion_tables(w4, rgb2yuv, t5);
      long lo = 0, hi = 0;
      for (int y = 0; y < 256; y++)
        for (int cidx = 0; cidx < 256; cidx++) {
          const long yy = t5[y];
          const long s[4] = {yy + t5[256 + cidx], yy + t5[1024 + cidx], yy + t5[512 + cidx] + t5[768], yy + t5[512 + cidx] + t5[768 + 255]};
          for (long v : s) { if ((v >> 16) < lo) lo = v >> 16; if ((v >> 16) > hi) hi = v >> 16; }
        }
      long gmin = 0, gmax = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
      for (int cidx = 0; cidx < 256; cidx++) {
        if (t5[512 + cidx] < a2) a2 = t5[512 + cidx];
        if (t5[512 + cidx] > b2) b2 = t5[512 + cidx];
        if (t5[768 + cidx] < a3) a3 = t5[768 + cidx];
        if (t5[768 + cidx] > b3) b3 = t5[768 + cidx];
      }
      gmin = (a2 + a3) >> 16; gmax = ((long)t5[255] + b2 + b3) >> 16;
      if (gmin < lo) lo = gmin;
      if (gmax > hi) hi = gmax;
      range_ok[w4] = (lo >= -kY16Bias && hi < kY16Lut - kY16Bias) ? 1 : -1;
    }
    if (range_ok[w4] == 1) {
      static std::atomic<int> g_cus_dev[16];          // CU count per device ordinal
      int dev = 0;
      LGPU_HIP(hipGetDevice(&dev));
      int g_cus = g_cus_dev[dev & 15].load();
      if (!g_cus) { hipDeviceProp_t prop; LGPU_HIP(hipGetDeviceProperties(&prop, dev)); g_cus = prop.multiProcessorCount; g_cus_dev[dev & 15].store(g_cus); }
      YuvBatch one = {};
      if (!batch) { one.y[0] = y_d; one.u[0] = u_d; one.v[0] = v_d; one.dst[0] = dst_d; }
      const YuvBatch &b16 = batch ? *batch : one;
      const int total = units * nbatch;
      int g16 = (total + 3) / 4;
      if (g16 > g_cus) g16 = g_cus;
#define Y16_LAUNCH(ORDER_)                                                                                                             \
      do {                                                                                                                             \
        if (a.use_lut) {                                                                                                             \
          LGPU_HIP(hipFuncSetAttribute((const void *)k_yuv420p_to_rgb16<ORDER_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kY16Lds));  \
          hipLaunchKernelGGL((k_yuv420p_to_rgb16<ORDER_, true>), dim3((unsigned)g16), dim3(1024), kY16Lds, (hipStream_t)stream, a, l, b16, nbatch); \
        } else {                                                                                                                     \
          LGPU_HIP(hipFuncSetAttribute((const void *)k_yuv420p_to_rgb16<ORDER_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kY16Lds));  \
          hipLaunchKernelGGL((k_yuv420p_to_rgb16<ORDER_, false>), dim3((unsigned)g16), dim3(1024), kY16Lds, (hipStream_t)stream, a, l, b16, nbatch); \
        }                                                                                                                            \
      } while (0)
      if (out_order == 0) Y16_LAUNCH(0); else if (out_order == 1) Y16_LAUNCH(1); else Y16_LAUNCH(2);
#undef Y16_LAUNCH
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if (is_422) hipLaunchKernelGGL(k_yuv422p_to_rgb, grid, dim3(kBlock), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0);
  else if (wide) {
    const dim3 g4(cdiv((unsigned)((width >> 1) + 3) / 4, kBlock), grid.y, grid.z);
    hipLaunchKernelGGL(k_yuv420p_to_rgb4, g4, dim3(kBlock), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0);
  } else hipLaunchKernelGGL(k_yuv420p_to_rgb, grid, dim3(kBlock), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_yuv420p_to_rgb(const uint8_t *y_d, const uint8_t *u_d, const uint8_t *v_d, const int istrides[3],
                                   long u_size, long v_size, uint8_t *dst_d, int orow, int width, int height,
                                   int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                                   const uint8_t *lut8, int flags, void *stream) {
  return yuv420p_to_rgb_impl(y_d, u_d, v_d, istrides, u_size, v_size, dst_d, orow, width, height, opsize, out_order, is_422, which_tables,
                             pb_quality, lut8, nullptr, flags, stream);
}

extern "C" int lgpu_yuv420p_to_rgb_lut16(const uint8_t *y_d, const uint8_t *u_d, const uint8_t *v_d, const int istrides[3],
                                         long u_size, long v_size, uint8_t *dst_d, int orow, int width, int height,
                                         int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                                         const uint16_t *lut16_d, int flags, void *stream) {
  if (!lut16_d) { set_error("lgpu_yuv420p_to_rgb_lut16: null LUT"); return LGPU_E_BADARG; }
  return yuv420p_to_rgb_impl(y_d, u_d, v_d, istrides, u_size, v_size, dst_d, orow, width, height, opsize, out_order, is_422, which_tables,
                             pb_quality, nullptr, lut16_d, flags, stream);
}


extern "C" int lgpu_yuv420p_to_rgb_batch(int nframes, const lgpu_yuv_frame *frames, const int istrides[3], long u_size, long v_size, int orow,
                                         int width, int height, int opsize, int out_order, int is_422, int which_tables, int pb_quality,
                                         const uint8_t *lut8, int flags, void *stream) {
  if (!frames || nframes < 1 || nframes > LGPU_CHAIN_MAX_TRACKS) { set_error("lgpu_yuv420p_to_rgb_batch: 1..%d frames", LGPU_CHAIN_MAX_TRACKS); return LGPU_E_BADARG; }
  YuvBatch b = {};
  for (int i = 0; i < nframes; i++) {
    if (!frames[i].y_d || !frames[i].u_d || !frames[i].v_d || !frames[i].dst_d) { set_error("lgpu_yuv420p_to_rgb_batch: null plane in frame %d", i); return LGPU_E_BADARG; }
    b.y[i] = frames[i].y_d; b.u[i] = frames[i].u_d; b.v[i] = frames[i].v_d; b.dst[i] = frames[i].dst_d;
    if (opsize == 4 && ((uintptr_t)frames[i].dst_d & 3)) { set_error("lgpu_yuv420p_to_rgb_batch: 4-byte output must be 4-byte aligned"); return LGPU_E_BADARG; }
  }
  return yuv420p_to_rgb_impl(frames[0].y_d, frames[0].u_d, frames[0].v_d, istrides, u_size, v_size, frames[0].dst_d, orow, width, height, opsize, out_order,
                             is_422, which_tables, pb_quality, lut8, nullptr, flags, stream, &b, nframes);
}
