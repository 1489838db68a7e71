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
#include <algorithm>
#include <atomic>
#include <type_traits>

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
  int cs = 1;              // stride of the four chroma tables (2 where they are interleaved in pairs)
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
      const int ir = (yy + rcr[v * cs]) >> 8, ig = (yy + gcb[u * cs] + gcr[v * cs]) >> 8, ib = (yy + bcb[u * cs]) >> 8;
      r = lut16[ir > 65535 ? 65535 : ir < 0 ? 0 : ir] >> 8;
      g = lut16[ig > 65535 ? 65535 : ig < 0 ? 0 : ig] >> 8;
      b = lut16[ib > 65535 ? 65535 : ib < 0 ? 0 : ib] >> 8;
    } else {
      r = clamp255((yy + rcr[v * cs]) >> 16); g = clamp255((yy + gcb[u * cs] + gcr[v * cs]) >> 16); b = clamp255((yy + bcb[u * cs]) >> 16);
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

// one (row i, chroma column k) cell of the 4:2:2 walk (:3593-3640 / :3858-3901): "last / this" are seeded from chroma row i >> 1 (reference), so the first pair of a
// row takes its left samples from there
__device__ __forceinline__ void yuv422_cell(const YuvArgs &a, const YuvCtx &c, int i, int k) {
  auto PU = [&](int r, int kk) -> int { long q = (long)r * a.us + kk; return a.u[q < a.usize ? q : a.usize - 1]; };
  auto PV = [&](int r, int kk) -> int { long q = (long)r * a.vs + kk; return a.v[q < a.vsize ? q : a.vsize - 1]; };
  const int tu = k ? PU(i, k) : PU(i >> 1, 0), tv = k ? PV(i, k) : PV(i >> 1, 0);
  const int lu = (k >= 2) ? PU(i, k - 1) : PU(i >> 1, 0), lv = (k >= 2) ? PV(i, k - 1) : PV(i >> 1, 0);
  const int nu = PU(i, k + 1), nv = PV(i, k + 1);
  const uint8_t *yr = a.y + (size_t)i * a.ys + 2 * k;
  const uint32_t p0 = c.rgb(yr[0], c.cuv((tu + lu) >> 1), c.cuv((tv + lv) >> 1));
  const uint32_t p1 = c.rgb(yr[1], c.cuv((tu + nu) >> 1), c.cuv((tv + nv) >> 1));
  c.store2(a.dst + (size_t)i * a.orow + (size_t)(2 * k) * a.opsize, p0, p1);
}

// ---- every launch whose rows are aligned: k_yuv420p_to_rgb_s --------------------------------------------------------------------------
// profiles/r03/k2_single_pmc.md: the one-cell-per-lane kernel above spends 269 VALU + 179 SALU + 34 LDS instructions per wave on 256 pixels, 8,656 waves in
// two generations of 2,164 workgroups that each stage 5.4 KB of tables (one 1080p frame: 11.6 us).  This form:
//   * {R_Cr, G_Cr}[v] and {G_Cb, B_Cb}[u] as 8-byte LDS entries with CLAMP16_240 / the 0..255 clamp folded into the index: 3 table gathers per pixel, no
//     chroma clamp instructions; / 3 as a multiply-shift; CLAMP0255f as one v_med3; two byte permutes assemble the pixel (26.6 VALU per pixel);
//   * cells of NC chroma columns (2 NC x 2 pixels) numbered linearly over the frame (no idle lanes at 960 chroma columns); every chroma row of a cell is ONE
//     unaligned load of the NC + 2 samples the cell needs;
//   * the samples of a thread's first cell are requested before the tables are staged, those of its next cell before the arithmetic of the current one
//     (launches with more cells than resident threads walk them with a grid stride: the tables are staged once per workgroup, not once per cell);
//   * edge cells (row 0, the trailing row, plane ends) walk yuv420_cell() on the same LDS tables.
// One 1080p frame: 11.6 -> 6.5 us; 16 x 1080p: 47.9 us (the 16-copy-table kernel of round 2, removed) -> 40 us (profiles/r03/k2_forms.txt).
constexpr int kYsOffTy = 0, kYsOffRG = 1024, kYsOffGB = 3072, kYsOffLut = 5120, kYsLds = 5376;
// V422: planar 4:2:2 (one luma row per cell, chroma averaged along the row only; the first column group of a row goes through yuv422_cell())
template <int NC, int ORDER, bool LUT, bool V422 = false>
__global__ __launch_bounds__(1024) void k_yuv420p_to_rgb_s(YuvArgs a, Lut8 lut, YuvBatch bt, int batched, uint32_t cgmagic) {
  if (batched) { a.y = bt.y[blockIdx.y]; a.u = bt.u[blockIdx.y]; a.v = bt.v[blockIdx.y]; a.dst = bt.dst[blockIdx.y]; }
  __shared__ __attribute__((aligned(16))) uint8_t smem[kYsLds];
  typedef typename std::conditional<NC == 4, uint64_t, uint32_t>::type win_t;      // NC + 2 chroma samples of a row (NC = 1 reads one byte more than it uses)
  typedef typename std::conditional<NC == 4, uint64_t, typename std::conditional<NC == 2, uint32_t, uint16_t>::type>::type ywin_t;      // 2 NC luma samples
  const int tid = threadIdx.x, nth = blockDim.x;
  const int hw = a.width >> 1, ncg = (hw + NC - 1) / NC;
  const int npairs = (a.height - 1) / 2;
  const int nunits = V422 ? a.height : 1 + npairs + (((a.height - 1) & 1) ? 1 : 0);
  struct Cell { ywin_t ya, yb; win_t u0, u1, v0, v1; uint32_t lv2; int unit, k0; bool valid, fast; };
  auto fetch = [&](uint32_t idx) -> Cell {
    Cell q;
    uint32_t unit = __umulhi(idx, cgmagic);                  // floor magic: the quotient or one less
    uint32_t cg = idx - unit * (uint32_t)ncg;
    if (cg >= (uint32_t)ncg) { cg -= ncg; unit++; }
    q.unit = (int)unit; q.k0 = NC * (int)cg;
    q.valid = unit < (uint32_t)nunits;
    q.ya = q.yb = 0; q.u0 = q.u1 = q.v0 = q.v1 = 0; q.lv2 = 0;
    if (V422) {
      // row q.unit; columns k0 >= 2 only (the first pairs of a row take their left samples from chroma row i >> 1), the window k0 - 1 .. inside the plane
      const int i2 = q.unit;
      // plane offsets are 32-bit products of 24-bit operands (the host sends larger planes to the general kernel): v_mul_u32_u24 instead of 64-bit multiply-adds at a quarter of the rate
      const uint32_t ou = __umul24((uint32_t)i2, (uint32_t)a.us) + (uint32_t)q.k0 - 1u, ov = __umul24((uint32_t)i2, (uint32_t)a.vs) + (uint32_t)q.k0 - 1u;
      q.fast = q.valid && q.k0 >= 2 && q.k0 + NC <= hw && (long)ou + (long)sizeof(win_t) <= a.usize && (long)ov + (long)sizeof(win_t) <= a.vsize;
      if (q.fast) {
        auto ld = [](const uint8_t *p) -> win_t { win_t w; __builtin_memcpy(&w, p, sizeof(win_t)); return w; };
        q.ya = *reinterpret_cast<const ywin_t *>(a.y + (__umul24((uint32_t)i2, (uint32_t)a.ys) + 2u * (uint32_t)q.k0));
        q.u0 = ld(a.u + ou); q.v0 = ld(a.v + ov);
      }
      return q;
    }
    const int i = 2 * q.unit - 1, r = i >> 1;
    // 32-bit plane offsets from 24-bit operands (see above); the edge units (unit 0, the trailing row) never take this path, so i >= 1 and r >= 0 here
    const uint32_t ru = __umul24((uint32_t)(r + 1), (uint32_t)a.us), rv = __umul24((uint32_t)(r + 1), (uint32_t)a.vs);      // row r + 1 of the chroma planes
    q.fast = q.valid && q.unit >= 1 && q.unit <= npairs && q.k0 + NC <= hw && (long)ru + q.k0 + (long)sizeof(win_t) <= a.usize &&
             (long)rv + q.k0 + (long)sizeof(win_t) <= a.vsize;
    if (q.fast) {
      auto ld = [](const uint8_t *p) -> win_t { win_t w; __builtin_memcpy(&w, p, sizeof(win_t)); return w; };
      const uint32_t oy = __umul24((uint32_t)i, (uint32_t)a.ys) + 2u * (uint32_t)q.k0;
      q.ya = *reinterpret_cast<const ywin_t *>(a.y + oy); q.yb = *reinterpret_cast<const ywin_t *>(a.y + (oy + (uint32_t)a.ys));
      const int o = q.k0 ? 1 : 0;                             // the first group has no sample on its left: fixed up where the cell is computed
      const uint32_t cu = ru + (uint32_t)q.k0 - (uint32_t)o, cv = rv + (uint32_t)q.k0 - (uint32_t)o;      // the window in chroma row r + 1; row r is one stride back
      q.u0 = ld(a.u + (cu - (uint32_t)a.us)); q.u1 = ld(a.u + cu); q.v0 = ld(a.v + (cv - (uint32_t)a.vs)); q.v1 = ld(a.v + cv);
      q.lv2 = a.v[rv];                                        // PV(r + 1, 0): the reference's constant "last" sample
    }
    return q;
  };
  const uint32_t stride = gridDim.x * (uint32_t)nth;
  uint32_t idx = blockIdx.x * (uint32_t)nth + (uint32_t)tid;
  Cell cur = fetch(idx);
  // tables: RGB_Y as is, {R_Cr, G_Cr}[v] and {G_Cb, B_Cb}[u] as 8-byte entries indexed by the UNCLAMPED blended chroma
  {
    uint32_t *s_ty = reinterpret_cast<uint32_t *>(smem + kYsOffTy);
    uint2 *s_rg = reinterpret_cast<uint2 *>(smem + kYsOffRG), *s_gb = reinterpret_cast<uint2 *>(smem + kYsOffGB);
    const int clo = a.clamped ? 16 : 0, chi = a.clamped ? 240 : 255;
    // 256 entries x 3 tables over the workgroup's groups of 256 threads; every load of a thread is issued before its first LDS write
    const int e = tid & 255, grp = tid >> 8, ngr = nth >> 8;
    const int ec = e < clo ? clo : e > chi ? chi : e;       // CLAMP16_240 (16 below 16, 240 from 0xF0 up) / 0..255; the blend itself never leaves 0..255
    const bool d0 = grp == 0, d1 = grp == 1 % ngr, d2 = grp == 2 % ngr;
    uint32_t t_y = 0, t_a = 0, t_b = 0, t_c = 0, t_d = 0;
    if (d0) t_y = (uint32_t)a.tables[e];
    if (d1) { t_a = (uint32_t)a.tables[256 + ec]; t_b = (uint32_t)a.tables[768 + ec]; }
    if (d2) { t_c = (uint32_t)a.tables[512 + ec]; t_d = (uint32_t)a.tables[1024 + ec]; }
    if (d0) s_ty[e] = t_y;
    if (d1) s_rg[e] = make_uint2(t_a, t_b);
    if (d2) s_gb[e] = make_uint2(t_c, t_d);
    if (LUT) { const int t2 = nth - 1 - tid; if (t2 < 64) reinterpret_cast<uint32_t *>(smem + kYsOffLut)[t2] = lut.w[t2]; }
  }
  __syncthreads();
  typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
  typedef const __attribute__((address_space(3))) uint8_t *lds_u8;
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) u32x2v *lds_u64;
  const uint32_t sbase = (uint32_t)(uintptr_t)smem;
  // sample j = -1 .. NC of a chroma row window, luma byte j = 0 .. 2 NC - 1
  auto at = [](win_t w, int j) -> uint32_t { return (uint32_t)(w >> (8 * (j + 1))) & 0xFFu; };
  auto yat = [](ywin_t w, int j) -> uint32_t { return (uint32_t)(w >> (8 * j)) & 0xFFu; };
  // (2a + b) / 3 on doubled sums, (int)(s / 3. + .5) == (s + 1) / 3 == (s + 1) * 43691 >> 17 for s < 2^15, as the LDS address of the 8-byte table entry
  auto blend = [&](uint32_t s1, uint32_t s2, uint32_t base) -> uint32_t { return ((__umul24(s1 + (s2 >> 1) + 1u, 43691u) >> 17) << 3) + base; };
  const uint32_t bu = sbase + kYsOffGB, bv = sbase + kYsOffRG;
  auto pixel = [&](uint32_t yv, uint32_t ua, uint32_t va) -> uint32_t {
    const uint32_t yy = *(lds_u32)(uintptr_t)(sbase + kYsOffTy + (yv << 2));
    const u32x2v rg = *(lds_u64)(uintptr_t)va, gb = *(lds_u64)(uintptr_t)ua;
    const int sr = (int)(yy + rg.x), sg = (int)(yy + gb.x + rg.y), sb = (int)(yy + gb.y);
    uint32_t r_ = (uint32_t)min(max(sr >> 16, 0), 255), g_ = (uint32_t)min(max(sg >> 16, 0), 255), b_ = (uint32_t)min(max(sb >> 16, 0), 255);
    if (LUT) {
      r_ = *(lds_u8)(uintptr_t)(sbase + kYsOffLut + r_); g_ = *(lds_u8)(uintptr_t)(sbase + kYsOffLut + g_); b_ = *(lds_u8)(uintptr_t)(sbase + kYsOffLut + b_);
    }
    // two byte permutes per pixel (selector 0x0C = 0x00, 0x0D = 0xFF: the alpha byte costs nothing)
    if (ORDER == 0) return __builtin_amdgcn_perm(b_, __builtin_amdgcn_perm(g_, r_, 0x0C0C0400u), 0x0D040100u);
    if (ORDER == 1) return __builtin_amdgcn_perm(r_, __builtin_amdgcn_perm(g_, b_, 0x0C0C0400u), 0x0D040100u);
    return __builtin_amdgcn_perm(b_, __builtin_amdgcn_perm(g_, r_, 0x0C04000Du), 0x04020100u);
  };
  while (cur.valid) {
    idx += stride;
    const Cell nxt = fetch(idx);
    if (!cur.fast) {
      // the paired tables hold tables[clamp(e)]: the cell clamps its index before the lookup, and the clamp is idempotent
      YuvCtx c;
      const int32_t *t32 = reinterpret_cast<const int32_t *>(smem);
      c.ty = t32 + kYsOffTy / 4; c.rcr = t32 + kYsOffRG / 4; c.gcr = t32 + kYsOffRG / 4 + 1; c.gcb = t32 + kYsOffGB / 4; c.bcb = t32 + kYsOffGB / 4 + 1; c.cs = 2;
      c.lut = smem + kYsOffLut; c.lut16 = nullptr; c.clamped = a.clamped; c.lowq = false; c.use_lut = LUT; c.opsize = 4; c.order = ORDER;
      for (int k = cur.k0; k < cur.k0 + NC && k < hw; k++) { if (V422) yuv422_cell(a, c, cur.unit, k); else yuv420_cell(a, c, cur.unit, k, hw, npairs); }
    } else if (V422) {
      uint32_t px[2 * NC];
#pragma unroll
      for (int j = 0; j < NC; j++) {
        const uint32_t tu = at(cur.u0, j), tv = at(cur.v0, j);
        px[2 * j] = pixel(yat(cur.ya, 2 * j), (((tu + at(cur.u0, j - 1)) >> 1) << 3) + bu, (((tv + at(cur.v0, j - 1)) >> 1) << 3) + bv);
        px[2 * j + 1] = pixel(yat(cur.ya, 2 * j + 1), (((tu + at(cur.u0, j + 1)) >> 1) << 3) + bu, (((tv + at(cur.v0, j + 1)) >> 1) << 3) + bv);
      }
      uint8_t *d0 = a.dst + (__umul24((uint32_t)cur.unit, (uint32_t)a.orow) + 8u * (uint32_t)cur.k0);
      if (NC == 1) *reinterpret_cast<uint2 *>(d0) = make_uint2(px[0], px[1]);
      else {
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < NC / 2; q++) { const u32x4s t = {px[4 * q], px[4 * q + 1], px[4 * q + 2], px[4 * q + 3]}; reinterpret_cast<u32x4s *>(d0)[q] = t; }
      }
    } else {
      win_t u0 = cur.u0, u1 = cur.u1, v0 = cur.v0, v1 = cur.v1;
      if (!cur.k0) {
        // first group: the reference's k == 0 case takes U[r][0] / V[r][0] as the samples left of the group (:3447-3452)
        u0 = (u0 << 8) | (u0 & 0xFF); u1 = u1 << 8; v1 = (v1 << 8) | (v0 & 0xFF); v0 = v0 << 8;
      }
      uint32_t top[2 * NC], bot[2 * NC];
#pragma unroll
      for (int j = 0; j < NC; j++) {
        const uint32_t u_rk = at(u0, j), v_rk = at(v0, j), v_r1k = at(v1, j);
        // left pixel: U row pair (s, s) -> top == bottom (:3461); V of row r with the previous V of row r + 1, the "last V" frozen at column 0 (:3544)
        const uint32_t su = u_rk + at(u0, j - 1);
        const uint32_t uleft = blend(su, su, bu);
        const uint32_t s1v = v_rk + at(v1, j - 1), s2v = v_r1k + cur.lv2;
        top[2 * j] = pixel(yat(cur.ya, 2 * j), uleft, blend(s1v, s2v, bv));
        bot[2 * j] = pixel(yat(cur.yb, 2 * j), uleft, blend(s2v, s1v, bv));
        // right pixel
        const uint32_t s1u = u_rk + at(u0, j + 1), s2u = at(u1, j) + at(u1, j + 1);
        const uint32_t s1w = v_rk + at(v0, j + 1), s2w = v_r1k + at(v1, j + 1);
        top[2 * j + 1] = pixel(yat(cur.ya, 2 * j + 1), blend(s1u, s2u, bu), blend(s1w, s2w, bv));
        bot[2 * j + 1] = pixel(yat(cur.yb, 2 * j + 1), blend(s2u, s1u, bu), blend(s2w, s1w, bv));
      }
      uint8_t *d0 = a.dst + (__umul24((uint32_t)(2 * cur.unit - 1), (uint32_t)a.orow) + 8u * (uint32_t)cur.k0), *d1 = d0 + a.orow;
      if (NC == 1) {
        *reinterpret_cast<uint2 *>(d0) = make_uint2(top[0], top[1]); *reinterpret_cast<uint2 *>(d1) = make_uint2(bot[0], bot[1]);
      } else {
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
        u32x4s *e0 = reinterpret_cast<u32x4s *>(d0), *e1 = reinterpret_cast<u32x4s *>(d1);
#pragma unroll
        for (int q = 0; q < NC / 2; q++) {
          const u32x4s t = {top[4 * q], top[4 * q + 1], top[4 * q + 2], top[4 * q + 3]}, b = {bot[4 * q], bot[4 * q + 1], bot[4 * q + 2], bot[4 * q + 3]};
          e0[q] = t; e1[q] = b;
        }
      }
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
  for (int i = blockIdx.y; i < a.height; i += gridDim.y) yuv422_cell(a, c, i, k);
}

}  // namespace lgpu

using namespace lgpu;

// launch shape of k_yuv420p_to_rgb_s: chroma columns per cell (1, 2, 4; 0 = never take that form), threads per workgroup, resident 256-thread groups per CU.
// Defaults from profiles/r03/k2_forms.txt; LGPU_YUV_S_NC / _BLOCK / _WGS at first use or lgpu_yuv420_tuning() (tests walk every cell width) override them.
struct YuvTuning { std::atomic<int> nc{2}, block{512}, wgs{8}; };
static YuvTuning &yuv_tuning() {
  static YuvTuning t;
  static std::atomic<bool> init{false};
  if (!init.load()) {
    const char *e;
    if ((e = getenv("LGPU_YUV_S_NC"))) { const int v = atoi(e); if (v == 0 || v == 1 || v == 2 || v == 4) t.nc = v; }
    if ((e = getenv("LGPU_YUV_S_BLOCK"))) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) t.block = v; }
    if ((e = getenv("LGPU_YUV_S_WGS"))) { const int v = atoi(e); if (v >= 1) t.wgs = v; }
    init = true;
  }
  return t;
}
extern "C" int lgpu_yuv420_tuning(int cell_columns, int block, int groups_per_cu) {
  YuvTuning &t = yuv_tuning();
  LGPU_REQUIRE(cell_columns == -1 || cell_columns == 0 || cell_columns == 1 || cell_columns == 2 || cell_columns == 4, "cell_columns is 0 (off), 1, 2 or 4 (-1 keeps it)");
  LGPU_REQUIRE(block == -1 || block == 256 || block == 512 || block == 1024, "block is 256, 512 or 1024 (-1 keeps it)");
  LGPU_REQUIRE(groups_per_cu == -1 || groups_per_cu >= 1, "groups_per_cu >= 1 (-1 keeps it)");
  if (cell_columns >= 0) t.nc = cell_columns;
  if (block > 0) t.block = block;
  if (groups_per_cu > 0) t.wgs = groups_per_cu;
  return LGPU_OK;
}

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
  dim3 grid(cdiv((unsigned)(width >> 1), kBlock), (unsigned)(units > 2048 ? 2048 : units), (unsigned)nbatch);
  YuvBatch none = {};
  const YuvBatch &bt = batch ? *batch : none;
  // aligned rows, 4-byte pixels, no LUT16, not the LOW quality setting: the paired-table form; everything else the one-cell-per-lane kernels
  const YuvTuning &tn = yuv_tuning();
  const int s_nc = tn.nc.load(), s_block = tn.block.load(), s_wgs = tn.wgs.load();
  // ... and planes whose byte offsets fit 31 bits with strides and row counts below 2^24 (the kernel multiplies them as 24-bit operands)
  const long long lim31 = 1ll << 31;
  bool form_s = s_nc && opsize == 4 && !lut16_d && !a.low_quality && (a.ys & (2 * s_nc - 1)) == 0 && (orow & (s_nc == 1 ? 7 : 15)) == 0 && nbatch <= 65535 &&
                height < (1 << 23) && a.ys < (1 << 24) && a.us < (1 << 24) && a.vs < (1 << 24) && orow < (1 << 24) && a.us > 0 && a.vs > 0 &&
                (long long)a.ys * (height + 1) < lim31 && (long long)a.us * (height + 2) < lim31 && (long long)a.vs * (height + 2) < lim31 && (long long)orow * (height + 1) < lim31;
  for (int f = 0; f < nbatch && form_s; f++) {
    const uintptr_t py = (uintptr_t)(batch ? batch->y[f] : y_d), pd = (uintptr_t)(batch ? batch->dst[f] : dst_d);
    form_s = (py & (2 * s_nc - 1)) == 0 && (pd & (s_nc == 1 ? 7 : 15)) == 0;
  }
  if (form_s) {
    const int hw = width >> 1, ncg = (hw + s_nc - 1) / s_nc;
    const int npairs_h = (height - 1) / 2, nunits_h = is_422 ? height : 1 + npairs_h + (((height - 1) & 1) ? 1 : 0);
    const unsigned long long cells = (unsigned long long)ncg * nunits_h;
    if (cells < (1ull << 31)) {
      const uint32_t magic = (uint32_t)((1ull << 32) / (unsigned)ncg - (ncg == 1 ? 1 : 0));
      // at most s_wgs x 256 threads per CU over the whole launch: a thread walks its cells with a grid stride
      const int cus = device_cus();
      unsigned gx = (unsigned)((cells + s_block - 1) / s_block);
      const unsigned cap = (unsigned)std::max(1, (int)((long)cus * s_wgs * 256 / s_block / nbatch));
      if (gx > cap) gx = cap;
      const dim3 gs(gx, (unsigned)nbatch);
#define YS_LAUNCH(NC_, ORDER_)                                                                                                          \
      do {                                                                                                                             \
        if (is_422) {                                                                                                                  \
          if (a.use_lut) hipLaunchKernelGGL((k_yuv420p_to_rgb_s<NC_, ORDER_, true, true>), gs, dim3(s_block), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0, magic);  \
          else hipLaunchKernelGGL((k_yuv420p_to_rgb_s<NC_, ORDER_, false, true>), gs, dim3(s_block), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0, magic);        \
        } else if (a.use_lut) hipLaunchKernelGGL((k_yuv420p_to_rgb_s<NC_, ORDER_, true>), gs, dim3(s_block), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0, magic);  \
        else hipLaunchKernelGGL((k_yuv420p_to_rgb_s<NC_, ORDER_, false>), gs, dim3(s_block), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0, magic);        \
      } while (0)
      if (s_nc == 4) { if (out_order == 0) YS_LAUNCH(4, 0); else if (out_order == 1) YS_LAUNCH(4, 1); else YS_LAUNCH(4, 2); }
      else if (s_nc == 1) { if (out_order == 0) YS_LAUNCH(1, 0); else if (out_order == 1) YS_LAUNCH(1, 1); else YS_LAUNCH(1, 2); }
      else { if (out_order == 0) YS_LAUNCH(2, 0); else if (out_order == 1) YS_LAUNCH(2, 1); else YS_LAUNCH(2, 2); }
#undef YS_LAUNCH
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if (is_422) hipLaunchKernelGGL(k_yuv422p_to_rgb, grid, dim3(kBlock), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0);
  else hipLaunchKernelGGL(k_yuv420p_to_rgb, grid, dim3(kBlock), 0, (hipStream_t)stream, a, l, bt, batch ? 1 : 0);
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
