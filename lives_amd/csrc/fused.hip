// fused.hip -- BASELINE config 4 as one launch: 5x5 separable gaussian -> colour-key composite against a second frame.
//
//   gaussian   : no reference loop exists (BASELINE names the op only); the repo's spec, identical to lgpu_gauss5: [1 4 6 4 1] / 16 per axis, border
//                replicated, 16-bit intermediate, one rounding (sum + 128) >> 8, every byte of the pixel alike
//   colour key : lives-plugins/weed-plugins/scripts/colorkey.script <process> -- box test on R, G, B against [min, max] per channel (green's box is built from
//                2 x delta), matching pixels become (uint8_t)(a * (1 - opac) + b * opac) per colour byte in double, others keep the first frame's pixel.
//                RGB24 / BGR24 as in the reference; RGBA32 / BGRA32 is the extension SURVEY 8d names for the headline size (the alpha byte stays the blurred
//                frame's).
//
// k_gauss5_colorkey: no LDS.  A wave owns a strip of 248 columns (62 lanes x 4 pixels; lanes 0 and 63 only feed their neighbours) and walks a band of rows.
// Per source row a lane loads its 4 pixels (16 or 12 bytes), takes the two pixels it needs on each side from the adjacent lanes (DPP wave shifts), blurs
// horizontally with the bytes spread to 16-bit lanes (one 32-bit operation = two channels) and keeps the last five blurred rows in registers; every new row
// completes one output row, which is keyed against the second frame's pixels (loaded a row ahead) and stored.  The key's box test runs on two 16-bit lanes per
// operation; its two double products come from 256-entry tables the workgroup computes once (the same IEEE products, one f64 add + one conversion per byte).  Odd bands walk upwards, so that the four source
// rows two neighbouring bands share are read at the same time and the second read hits L2 (same scheme as k_pb_half).  HBM-bound: 3 frames of traffic.
#include "lgpu_common.h"
#include <cmath>

namespace lgpu {

struct GckArgs {
  const uint8_t *src0, *src1;
  uint8_t *dst;
  int irow0, irow1, orow, width, height;
  int strips, cgroups, bands, th;
  uint32_t mn_e, mx_e, mn_o, mx_o;                    // the key's box on (byte 0, byte 2) and (byte 1, byte 3) as 16-bit lanes: mn = 0x8000 - min, mx = max | 0x8000 per lane
  double opac, opacx;
  int key;                                            // 0: blur only
};
typedef unsigned gk_u4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) gk_u3 { uint32_t x, y, z; };          // 12 bytes at a 4-byte aligned address: global_load / store_dwordx3

template <int PS>
__global__ __launch_bounds__(256) void k_gauss5_colorkey(const GckArgs A_, const FxFrames F) {
  GckArgs A = A_;                                       // blockIdx.y: the frame of a batched launch (lgpu_fx_batch)
  A.src0 = F.in0[blockIdx.y][0]; A.src1 = F.in1[blockIdx.y][0]; A.dst = F.out[blockIdx.y][0];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ double s_ta[256], s_tb[256];               // a * (1 - opac), b * opac for every byte value: the script's two products
  if (A.key) { s_ta[threadIdx.x] = (double)(int)threadIdx.x * A.opacx; s_tb[threadIdx.x] = (double)(int)threadIdx.x * A.opac; }
  __syncthreads();
  // XCD-contiguous order: (column group, band) runs, band-minor (see k_pb_half)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nseq = A.cgroups * A.bands, per_xcd = (nseq + 7) >> 3;
  const int seq = xcd * per_xcd + slot;
  if (seq >= nseq || slot >= per_xcd) return;
  const int cg = seq / A.bands, band = seq - cg * A.bands, strip = cg * 4 + wave;
  if (strip >= A.strips) return;
  const int k = strip * 62 - 1 + lane, kmax = (A.width >> 2) - 1;        // this lane's pixels: 4k .. 4k + 3
  const int kc = k < 0 ? 0 : k > kmax ? kmax : k;
  const bool out_lane = lane >= 1 && lane <= 62 && k <= kmax;
  const bool edge_strip = strip == 0 || (strip + 1) * 62 + 1 >= kmax;
  const int y0 = band * A.th, rows = min(A.th, A.height - y0);
  const uint32_t off = (uint32_t)(4 * PS) * (uint32_t)kc;

  // buffer loads / stores: one descriptor per frame in SGPRs, the row as the scalar offset, the lane's place in the row as a constant VGPR offset -- no 64-bit
  // address arithmetic on the vector unit (it was a v_mad_i64_i32 per access); lanes that must not store carry an offset beyond the descriptor's range
  auto srd = [](const void *p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    void *u = (void *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_src0 = srd(A.src0, (uint32_t)A.height * (uint32_t)A.irow0);
  const __amdgpu_buffer_rsrc_t r_src1 = srd(A.key ? A.src1 : A.src0, A.key ? (uint32_t)A.height * (uint32_t)A.irow1 : 16u);
  const __amdgpu_buffer_rsrc_t r_dst = srd(A.dst, (uint32_t)A.height * (uint32_t)A.orow);
  auto load4 = [&](const __amdgpu_buffer_rsrc_t &r, int irow, int y) -> gk_u4 {
    y = __builtin_amdgcn_readfirstlane(y < 0 ? 0 : y > A.height - 1 ? A.height - 1 : y);
    if (PS == 4) return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, y * irow, 0);
    typedef unsigned gk_v3 __attribute__((ext_vector_type(3)));
    const gk_v3 t = __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, y * irow, 0);
    uint32_t q[4];
    unpack3(t.x, t.y, t.z, q);
    gk_u4 r4;
    r4.x = q[0]; r4.y = q[1]; r4.z = q[2]; r4.w = q[3];
    return r4;
  };
  const uint32_t st_off = out_lane ? (uint32_t)(4 * PS) * (uint32_t)k : 0xFFFFFFF0u;
  auto fix = [&](gk_u4 q) -> gk_u4 {          // the gaussian replicates the frame's first / last column
    if (edge_strip) {
      if (k < 0) { q.y = q.x; q.z = q.x; q.w = q.x; }
      if (k > kmax) { q.x = q.w; q.y = q.w; q.z = q.w; }
    }
    return q;
  };
  // horizontal pass of one row: h[2 j] = (byte 0, byte 2), h[2 j + 1] = (byte 1, byte 3) sums of pixel j, in 16-bit lanes
  auto hrow = [&](gk_u4 q, uint32_t h[8]) {
    // the lane's own four pixels are spread to 16-bit lanes first (one AND for bytes 0 / 2, one byte permute for bytes 1 / 3), and the SPREAD values travel to the
    // neighbours: 8 + 8 operations per row instead of 4 lane moves + 24 to spread eight raw pixels
    uint32_t e[8], o[8];
    const uint32_t own[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { e[2 + i] = own[i] & 0x00FF00FFu; o[2 + i] = __builtin_amdgcn_perm(0u, own[i], 0x0C030C01u); }
    e[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[4], 0x138, 0xF, 0xF, true); e[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[5], 0x138, 0xF, 0xF, true);   // wave_shr:1: the left lane's pixels 2, 3
    o[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[4], 0x138, 0xF, 0xF, true); o[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[5], 0x138, 0xF, 0xF, true);
    e[6] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[2], 0x130, 0xF, 0xF, true); e[7] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[3], 0x130, 0xF, 0xF, true);   // wave_shl:1: the right lane's pixels 0, 1
    o[6] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[2], 0x130, 0xF, 0xF, true); o[7] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[3], 0x130, 0xF, 0xF, true);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      h[2 * j] = gauss5_taps(e[j], e[j + 1], e[j + 2], e[j + 3], e[j + 4]);
      h[2 * j + 1] = gauss5_taps(o[j], o[j + 1], o[j + 2], o[j + 3], o[j + 4]);
    }
  };

  const int vr0 = y0 - 2, vr1 = y0 + rows + 1;
  const int ylo = vr0 < 0 ? 0 : vr0, yhi = vr1 > A.height - 1 ? A.height - 1 : vr1;
  const int d = (band & 1) ? -1 : 1;
  const int ystart = d > 0 ? ylo : yhi, vstart = d > 0 ? vr0 : vr1;
  gk_u4 qn = load4(r_src0, A.irow0, ystart);
  gk_u4 b4 = {0, 0, 0, 0};      // the second frame's pixels of the current output row: taken over from `nb` at the end of every step (step 3 loads the first output row's)
  uint32_t ring[5][8];
  // one output row from the five ring rows around it (oldest first), keyed against the second frame's pixels, stored
  auto finish_row = [&](int y, const uint32_t *r0, const uint32_t *r1, const uint32_t *r2, const uint32_t *r3, const uint32_t *r4, const gk_u4 &bb) __attribute__((always_inline)) {
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t ve = gauss5_taps(r0[2 * j], r1[2 * j], r2[2 * j], r3[2 * j], r4[2 * j], 0x00800080u);
      const uint32_t vo = gauss5_taps(r0[2 * j + 1], r1[2 * j + 1], r2[2 * j + 1], r3[2 * j + 1], r4[2 * j + 1], 0x00800080u);
      uint32_t a = __builtin_amdgcn_perm(vo, ve, 0x07030501u);         // the high byte of every 16-bit lane is the blurred value: (ve.1, vo.1, ve.3, vo.3) in one byte permute
      if (A.key) {
        // the box test on both 16-bit lanes at once: x >= min <=> bit 15 of x + (0x8000 - min), x <= max <=> bit 15 of (max | 0x8000) - x (the alpha lane's box is
        // 0 .. 255); the constants come ready from the host, the adds / subtracts / ANDs are the vector unit's cheap class
        const uint32_t e = a & 0x00FF00FFu, o = __builtin_amdgcn_perm(0u, a, 0x0C030C01u);
        const uint32_t t = (e + A.mn_e) & (A.mx_e - e) & (o + A.mn_o) & (A.mx_o - o) & 0x80008000u;
        if (t == 0x80008000u) {
          // (uint8_t)(a * (1 - opac) + b * opac) in double, the products from the workgroup's two 256-entry tables: one f64 add and one conversion per byte
          const uint32_t b = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;
          const uint32_t m0 = (uint32_t)(uint8_t)(s_ta[a & 0xFF] + s_tb[b & 0xFF]);
          const uint32_t m1 = (uint32_t)(uint8_t)(s_ta[(a >> 8) & 0xFF] + s_tb[(b >> 8) & 0xFF]);
          const uint32_t m2 = (uint32_t)(uint8_t)(s_ta[(a >> 16) & 0xFF] + s_tb[(b >> 16) & 0xFF]);
          a = m0 | (m1 << 8) | (m2 << 16) | (a & 0xFF000000u);
        }
      }
      px[j] = a;
    }
    const int so = __builtin_amdgcn_readfirstlane(y * A.orow);
    if (PS == 4) {
      gk_u4 o4;
      o4.x = px[0]; o4.y = px[1]; o4.z = px[2]; o4.w = px[3];
      __builtin_amdgcn_raw_buffer_store_b128(o4, r_dst, (int)st_off, so, 2);       // not read again by this launch: non-temporal
    } else {
      typedef unsigned gk_v3 __attribute__((ext_vector_type(3)));
      gk_v3 o3;
      uint32_t w0, w1, w2;
      pack3(px, w0, w1, w2);
      o3.x = w0; o3.y = w1; o3.z = w2;
      __builtin_amdgcn_raw_buffer_store_b96(o3, r_dst, (int)st_off, so, 0);
    }
  };
  const int nsteps = vr1 - vr0 + 1;
  if (vr0 >= 0 && vr1 <= A.height - 1) {
    // Bands clear of the frame's first / last rows: every step blurs a NEW row, so the walk is straight-line code.  The next source row is requested INTO the
    // registers of the row the horizontal pass has just read, the second frame's pixels of a step's output row at the top of that step; both are read most of a
    // step later and no wait sits behind a store.  (The general loop below took the new row over at the top of the next step, behind the store: s_waitcnt vmcnt(0)
    // there made every step wait for its own store and for the load it had just issued -- the same finding as in k_pb_half, profiles/r05/blur_investigation.md.)
#pragma unroll
    for (int s = 0; s < 4; s++) {
      hrow(fix(qn), ring[s]);
      qn = load4(r_src0, A.irow0, vstart + d * (s + 1));             // nsteps >= 5: the row exists
    }
    constexpr int kSlot[5] = {4, 0, 1, 2, 3};
    for (int s0 = 4; s0 < nsteps; s0 += 5) {
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int s = s0 + u, t = kSlot[u];
        if (s >= nsteps) break;
        if (A.key) b4 = load4(r_src1, A.irow1, vstart + d * (s - 2));      // this step's output row
        hrow(fix(qn), ring[t]);
        if (s + 1 < nsteps) qn = load4(r_src0, A.irow0, vstart + d * (s + 1));
        finish_row(vstart + d * (s - 2), ring[(t + 1) % 5], ring[(t + 2) % 5], ring[(t + 3) % 5], ring[(t + 4) % 5], ring[t], b4);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) ring[i][j] = 0;
  int produced = ystart - d;
  // the ring rotates by slot index, five steps per trip of the outer loop: slot u takes the new row, (u + 1) % 5 is the oldest -- no register moves
  for (int step0 = 0; step0 < nsteps; step0 += 5) {
#pragma unroll
    for (int u = 0; u < 5; u++) {
      const int step = step0 + u;
      if (step >= nsteps) break;
      const int vr = vstart + d * step;
      const int yy = vr < 0 ? 0 : vr > A.height - 1 ? A.height - 1 : vr;
      gk_u4 nb = {0, 0, 0, 0};
      if (A.key) nb = load4(r_src1, A.irow1, vr - d);              // the second frame's pixels of the NEXT output row
      if (yy != produced) {
        const gk_u4 q = qn;
        qn = load4(r_src0, A.irow0, yy + d);                         // the next source row, in flight during this row's arithmetic (two rows ahead: 26.4 -> 28.8 us, the registers cost a wave per SIMD)
        hrow(fix(q), ring[u]);
        produced = yy;
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) ring[u][j] = ring[(u + 4) % 5][j];   // a row beyond the frame: the border row again
      }
      if (step >= 4) finish_row(vr - 2 * d, ring[(u + 1) % 5], ring[(u + 2) % 5], ring[(u + 3) % 5], ring[(u + 4) % 5], ring[u], b4);
      b4 = nb;                    // unconditionally: as a select on (step >= 3) it was four v_cndmask per step
    }
  }
}

}  // namespace lgpu

using namespace lgpu;

// the gaussian alone through the same kernel (key off): what lgpu_gauss5 takes for aligned 3- and 4-byte frames.  LGPU_E_UNSUPPORTED: the caller's other kernels.
namespace lgpu {
int gauss5_rows(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)src_d | (uintptr_t)dst_d | (uintptr_t)irow | (uintptr_t)orow;
  if ((psize != 3 && psize != 4) || (width & 3) || (bits & (psize == 4 ? 15 : 3))) return LGPU_E_UNSUPPORTED;
  if ((long long)height * irow >= (1ll << 31) || (long long)height * orow >= (1ll << 31)) return LGPU_E_UNSUPPORTED;      // 32-bit buffer offsets in the kernel
  GckArgs a;
  __builtin_memset(&a, 0, sizeof a);
  a.src0 = src_d; a.src1 = nullptr; a.dst = dst_d; a.irow0 = irow; a.irow1 = 0; a.orow = orow; a.width = width; a.height = height;
  a.strips = (int)cdiv((unsigned)width, 248); a.cgroups = (a.strips + 3) / 4;
  a.th = 8;
  { const int v = tune(TUNE_GCK_TH); if (v >= 1 && v <= 1024) a.th = v; }
  a.bands = (int)cdiv((unsigned)height, (unsigned)a.th);
  a.key = 0;
  const dim3 grid(8u * cdiv((unsigned)(a.cgroups * a.bands), 8u));
  FxFrames F = {};
  F.in0[0][0] = src_d; F.out[0][0] = dst_d;
  if (psize == 4) hipLaunchKernelGGL(k_gauss5_colorkey<4>, grid, dim3(256), 0, st, a, F);
  else hipLaunchKernelGGL(k_gauss5_colorkey<3>, grid, dim3(256), 0, st, a, F);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}
}  // namespace lgpu

// 5x5 gaussian of frame 0, then the colour key of the blurred frame against frame 1, in one launch.  psize 3 (RGB24 / BGR24: the reference's palettes) or 4
// (RGBA32 / BGRA32: extension, alpha = the blurred frame's).  LGPU_E_UNSUPPORTED when the width is not a multiple of 4 or a frame is not 4- (psize 3) / 16-byte
// (psize 4) aligned: the caller then runs lgpu_gauss5 and lgpu_colorkey one after the other.
int lgpu::gauss5_colorkey_n(const FxFrames &F, int nframes, int irow0, int irow1, int orow, int width, int height, int psize,
                            int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, hipStream_t stream) {
  LGPU_REQUIRE(width > 0 && height > 0, "empty geometry");
  LGPU_REQUIRE(psize == 3 || psize == 4, "psize must be 3 or 4");
  LGPU_REQUIRE(irow0 >= width * psize && irow1 >= width * psize && orow >= width * psize, "rowstride smaller than a row");
  uintptr_t bits = (uintptr_t)irow0 | (uintptr_t)irow1 | (uintptr_t)orow;
  for (int f = 0; f < nframes; f++) {
    LGPU_REQUIRE(F.in0[f][0] && F.in1[f][0] && F.out[f][0], "null frame");
    LGPU_REQUIRE(F.in0[f][0] != F.out[f][0], "the blur cannot run in place");
    bits |= (uintptr_t)F.in0[f][0] | (uintptr_t)F.in1[f][0] | (uintptr_t)F.out[f][0];
  }
  if ((width & 3) || (bits & (psize == 4 ? 15 : 3))) { set_error("lgpu_gauss5_colorkey: width %% 4 or alignment outside the fused kernel's range"); return LGPU_E_UNSUPPORTED; }
  if ((long long)height * irow0 >= (1ll << 31) || (long long)height * irow1 >= (1ll << 31) || (long long)height * orow >= (1ll << 31)) {
    set_error("lgpu_gauss5_colorkey: a plane of 2 GiB or more is outside the fused kernel's 32-bit buffer offsets"); return LGPU_E_UNSUPPORTED; }
  GckArgs a;
  a.src0 = nullptr; a.src1 = nullptr; a.dst = nullptr; a.irow0 = irow0; a.irow1 = irow1; a.orow = orow; a.width = width; a.height = height;
  a.strips = (int)cdiv((unsigned)width, 248); a.cgroups = (a.strips + 3) / 4;
  // short bands: the launch is bound by the time a wave needs for its rows, not by the rows the bands share (profiles/r03/c4_band_sweep.txt: 4K RGBA32 27 us at 6 rows,
  // 28 at 8, 32 at 16, 45 at 32)
  a.th = 8;            // one frame: 4,320 waves, one generation at five workgroups per CU (82 / 86 VGPRs); 6-row bands would need a second generation for their last 640
  // a launch of more than one generation of workgroups (several frames: lgpu_fx_batch) is no longer a matter of one wave's latency: taller bands, fewer rows
  // blurred twice (profiles/r05: 8 x 4K RGBA32 164 us at 6 rows, 155 at 8, 162 at 12; RGB24 173 at 8, 168 at 12, 167 at 16)
  if ((long long)a.cgroups * cdiv((unsigned)height, (unsigned)a.th) * nframes > (long long)device_cus() * 5) {
    a.th = psize == 4 ? 8 : 12;
    if (psize == 3) {
      // 3-byte pixels: of 12 / 15 / 18 rows the height whose LAST generation of workgroups (five per CU) is fullest -- 8 x 4K: 18 rows = three whole generations,
      // 157.6 -> 152.8 us (4-byte pixels are flat from 8 to 9 rows and lose above: profiles/r05/late/c4_generations.txt)
      const long long slots = (long long)device_cus() * 5;
      double best = 0.;
      for (int th = 12; th <= 18; th += 3) {
        const long long wgs = (long long)a.cgroups * cdiv((unsigned)height, (unsigned)th) * nframes;
        const double fill = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
        if (fill > best + 1e-9) { best = fill; a.th = th; }
      }
    }
  }
  { const int v = tune(TUNE_GCK_TH); if (v >= 1 && v <= 1024) a.th = v; }      // tuning probe
  a.bands = (int)cdiv((unsigned)height, (unsigned)a.th);
  // parameter preparation exactly as the script does it (host side, double)
  double xdelta = delta * 2.;
  delta /= 2.;
  const int rmin = col_r - (int)(col_r * delta + .5);
  const int gmin = col_g - (int)(col_g * xdelta + .5);
  const int bmin = col_b - (int)(col_b * delta + .5);
  xdelta *= 2.;
  delta *= 2.;
  const int rmax = col_r + (int)((255 - col_r) * delta + .5);
  const int gmax = col_g + (int)((255 - col_g) * xdelta + .5);
  const int bmax = col_b + (int)((255 - col_b) * delta + .5);
  // the box as packed 16-bit lanes in memory byte order (byte 0 / byte 2 = red / blue or blue / red); bounds outside 0 .. 255 are clamped where that keeps the
  // test's meaning, a box no byte can enter switches the key off (the blurred frame alone is the result then)
  auto lo = [](int v) { return (uint32_t)(v < 0 ? 0 : v); };
  auto hi = [](int v) { return (uint32_t)(v > 255 ? 255 : v); };
  const bool empty = rmin > 255 || gmin > 255 || bmin > 255 || rmax < 0 || gmax < 0 || bmax < 0 || rmin > rmax || gmin > gmax || bmin > bmax;
  const int c0min = is_bgr ? bmin : rmin, c0max = is_bgr ? bmax : rmax, c2min = is_bgr ? rmin : bmin, c2max = is_bgr ? rmax : bmax;
  if (!empty) {
    a.mn_e = (0x8000u - lo(c0min)) | ((0x8000u - lo(c2min)) << 16); a.mx_e = (hi(c0max) | (hi(c2max) << 16)) | 0x80008000u;       // mn_*: 0x8000 - min per lane (the kernel ADDS it)
    a.mn_o = (0x8000u - lo(gmin)) | (0x8000u << 16); a.mx_o = (hi(gmax) | (255u << 16)) | 0x80008000u;
  } else { a.mn_e = a.mx_e = a.mn_o = a.mx_o = 0; }
  a.opac = opac; a.opacx = 1. - opac; a.key = empty ? 0 : 1;
  const dim3 grid(8u * cdiv((unsigned)(a.cgroups * a.bands), 8u), (unsigned)nframes);
  if (psize == 4) hipLaunchKernelGGL(k_gauss5_colorkey<4>, grid, dim3(256), 0, stream, a, F);
  else hipLaunchKernelGGL(k_gauss5_colorkey<3>, grid, dim3(256), 0, stream, a, F);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_gauss5_colorkey(const uint8_t *src0_d, int irow0, const uint8_t *src1_d, int irow1, uint8_t *dst_d, int orow, int width, int height, int psize,
                                    int is_bgr, double delta, double opac, int col_r, int col_g, int col_b, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  FxFrames F = {};
  F.in0[0][0] = src0_d; F.in1[0][0] = src1_d; F.out[0][0] = dst_d;
  return gauss5_colorkey_n(F, 1, irow0, irow1, orow, width, height, psize, is_bgr, delta, opac, col_r, col_g, col_b, (hipStream_t)stream);
}
