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
#include <stdlib.h>
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
  // k_sep2: taps as packed int16 pairs.  hco2[dst][nph] = (tap 2j, tap 2j+1); vco2[dst][npv] = the row's taps laid over EVEN-aligned
  // source row pairs: (v0, v1), (v2, v3) ... when vpos is even, (0, v0), (v1, v2) ... when it is odd
  const uint32_t *hco2, *vco2;
  int nph, npv;
  int ntracks;                      // k_sep2p: tracks of the launch (the tile list is tiles_x * tiles_y * ntracks long)
  unsigned long long *dbg;          // k_sep2p, LGPU_S2P_DEBUG=1: per-wave phase cycle sums [grid][6][8] (nullptr otherwise)
  // k_sep2p with the horizontal pass on the matrix cores (uniform integer ratio r: every output column has the same taps and starts r pixels after
  // its neighbour): B fragments [KB][hi, lo][64 lanes] of v_mfma_i32_16x16x64_i8, see sep2p_bfrag()
  const void *bfrag;
  int mh_r;
};
struct SepTracks {
  const uint8_t *src[LGPU_CHAIN_MAX_TRACKS];
  const uint8_t *l2[LGPU_CHAIN_MAX_TRACKS];
  uint8_t *dst[LGPU_CHAIN_MAX_TRACKS];
};

__device__ __forceinline__ uint32_t mix_pairs2(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  // operands are < 2^24 (two bytes in 16-bit lanes times a byte): v_mul_u32_u24 / v_mad_u32_u24, not 32/64-bit multiplies
  return ((__umul24(b, bf) + __umul24(a, nbf)) >> 8) & 0x00FF00FFu;
}
__device__ __forceinline__ uint32_t mix4b(uint32_t a, uint32_t b, uint32_t bf, uint32_t nbf) {
  return mix_pairs2(a & 0x00FF00FFu, b & 0x00FF00FFu, bf, nbf) | (mix_pairs2((a >> 8) & 0x00FF00FFu, (b >> 8) & 0x00FF00FFu, bf, nbf) << 8);
}
typedef short short2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
// clamp_u8(v >> 21) of four accumulators packed into one pixel: high halves (>> 16) paired by a byte permute, packed
// arithmetic >> 5, v_sat_pk_u8_i16 (signed 16 -> unsigned 8 saturation of both halves), one permute to join
__device__ __forceinline__ uint32_t pack_sat_shr21(int a0, int a1, int a2, int a3) {
  const short2v h01 = __builtin_bit_cast(short2v, __builtin_amdgcn_perm((uint32_t)a1, (uint32_t)a0, 0x07060302u)) >> 5;
  const short2v h23 = __builtin_bit_cast(short2v, __builtin_amdgcn_perm((uint32_t)a3, (uint32_t)a2, 0x07060302u)) >> 5;
  uint32_t b01, b23;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(h01));
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(h23));
  return __builtin_amdgcn_perm(b23, b01, 0x05040100u);
}
__device__ __forceinline__ int clamp_i16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

// (bf * s2_c + nbf * s1_c) >> 8 for the three colour bytes with v_dot4_u32_u8: interleave the two pixels' bytes so one
// dot4 against (bf, nbf, 0, 0) / (0, 0, bf, nbf) yields one channel; returns [r0 r1 r2 0]
__device__ __forceinline__ uint32_t mix3_dot4(uint32_t s1, uint32_t s2, uint32_t w_lo, uint32_t w_hi) {
  const uint32_t x01 = __builtin_amdgcn_perm(s1, s2, 0x05010400u);     // [s2.b0 s1.b0 s2.b1 s1.b1]
  const uint32_t x2 = __builtin_amdgcn_perm(s1, s2, 0x0C0C0602u);      // [s2.b2 s1.b2 0 0]
  const uint32_t r0 = __builtin_amdgcn_udot4(x01, w_lo, 0u, false), r1 = __builtin_amdgcn_udot4(x01, w_hi, 0u, false);
  const uint32_t r2 = __builtin_amdgcn_udot4(x2, w_lo, 0u, false);
  // each r < 2^16; wanted byte = bits 8..15
  return __builtin_amdgcn_perm(r1, r0, 0x0C0C0501u) | ((r2 << 8) & 0x00FF0000u);
}

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
  const int wrows = sy1 - sy0;                         // host guarantees <= sht (and sx1 - sx0 <= swt)

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
      __builtin_nontemporal_store(p, reinterpret_cast<uint32_t *>(dst + (size_t)oy * a.orow) + tx0 + lane);      // written once, not read back
    }
  }
}


// k_sep2: the same separable filter for RGBA32 with both passes on v_dot2_i32_i16 (two multiply-adds per operation):
//   horizontal: per window row pair and tap pair, one v_perm per channel builds (pixel j, pixel j+1) as two 16-bit lanes, one dot2 against the
//               packed tap pair accumulates them: one VALU operation per multiply-add (k_separable: byte extract + v_mad = two), and the rounding
//               rides in the accumulator seed; v_cvt_pk_i16_i32 clamps and packs two rows' results;
//   vertical  : the intermediate is stored as EVEN-aligned row pairs (one uint4 = 4 channels x 2 rows per column), the taps of an output row
//               arrive packed over such pairs (vco2), so a row pair costs four dot2 = half an operation per multiply-add.
// Same arithmetic as the oracle's orc_resize (bit-identical; the tests run every path against it).  Measured on the ratios k_half8s does not
// take: profiles/r02/resize_ratios.md.
template <int NPH>
__global__ __launch_bounds__(kBlock) void k_sep2(SepArgs a, SepTracks trk, Lut8 lut) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *s_src = reinterpret_cast<uint32_t *>(smem);                               // [sht][swt]
  uint4 *s_p = reinterpret_cast<uint4 *>(smem + (size_t)a.sht * a.swt * 4);           // [sht / 2][64]: 4 channels x (row 2k, row 2k+1) int16
  uint8_t *s_lut = smem + (size_t)a.sht * a.swt * 4 + (size_t)(a.sht >> 1) * kTileW * 16;
  uint32_t *s_vc = reinterpret_cast<uint32_t *>(s_lut + 256);                           // [th][npv + 1]: first row pair, then the packed taps of the row

  const int tile = blockIdx.x, track = blockIdx.y;
  const int tx0 = (tile % a.tiles_x) * kTileW, ty0 = (tile / a.tiles_x) * a.th;
  const int tw = min(kTileW, a.dw - tx0), thh = min(a.th, a.dh - ty0);
  const uint8_t *src = trk.src[track];
  // the vertical taps of this tile's rows go to LDS once (a scalar load per tap inside the row loop costs its latency per row)
  for (int i = threadIdx.x; i < thh * (a.npv + 1); i += kBlock) {
    const int ly = i / (a.npv + 1), j = i - ly * (a.npv + 1);
    s_vc[i] = j ? a.vco2[(size_t)(ty0 + ly) * a.npv + (j - 1)] : (uint32_t)(a.vpos[ty0 + ly] >> 1);
  }
  const int sx0 = a.hpos[tx0], sx1 = a.hpos[tx0 + tw - 1] + 2 * NPH;       // padded taps read one pixel further
  const int sy0 = a.vpos[ty0] & ~1, sy1 = a.vpos[ty0 + thh - 1] + a.ntv;   // window rows start on an even source row
  const int wrows = (sy1 - sy0 + 1) & ~1, npairs = wrows >> 1;
  if (a.use_lut) stage_lut(s_lut, lut);
  uint32_t bf = a.bf, nbf = a.nbf;
  if (a.blend && a.bf_d) { bf = (uint32_t)a.bf_d[0] & 0xFF; nbf = 0xFF - bf; }

  // ---- 1. stage the window (edge replicate), byte swap on the fly: as k_separable ----
  const int sx0a = sx0 & ~3;
  const int wchunks = (sx1 - sx0a + 3) >> 2;
  const int nitems = wrows * wchunks;
  if (a.vec) {
    for (int base = 0; base < nitems; base += kStageMax * kBlock) {
      uint4 v[kStageMax];
#pragma unroll
      for (int k = 0; k < kStageMax; k++) {
        const int it = base + threadIdx.x + k * kBlock;
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
        const int it = base + threadIdx.x + k * kBlock;
        if (it < nitems) {
          const int r = it / wchunks, ch = it - r * wchunks;
          uint4 o;
          o.x = __builtin_amdgcn_perm(0u, v[k].x, a.src_sel); o.y = __builtin_amdgcn_perm(0u, v[k].y, a.src_sel);
          o.z = __builtin_amdgcn_perm(0u, v[k].z, a.src_sel); o.w = __builtin_amdgcn_perm(0u, v[k].w, a.src_sel);
          *reinterpret_cast<uint4 *>(s_src + r * a.swt + ch * 4) = o;
        }
      }
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

  // ---- 2. horizontal pass: lane = output column, a wave takes window row pairs ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ox = tx0 + (lane < tw ? lane : tw - 1);
  const int hoff = a.hpos[ox] - sx0a;
  short2v hc2[NPH];
#pragma unroll
  for (int j = 0; j < NPH; j++) hc2[j] = __builtin_bit_cast(short2v, a.hco2[(size_t)ox * NPH + j]);
  for (int k = wave; k < npairs; k += kBlock >> 6) {
    const uint32_t *r0 = s_src + (2 * k) * a.swt + hoff, *r1 = r0 + a.swt;
    int e0 = a.hround, e1 = a.hround, e2 = a.hround, e3 = a.hround, o0 = a.hround, o1 = a.hround, o2 = a.hround, o3 = a.hround;
#pragma unroll
    for (int j = 0; j < NPH; j++) {
      const uint32_t p0 = r0[2 * j], p1 = r0[2 * j + 1], q0 = r1[2 * j], q1 = r1[2 * j + 1];
      // (pixel 2j, pixel 2j+1) of one channel as two 16-bit lanes: selector bytes [c, 0, 4 + c, 0]
      e0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, 0x0C040C00u)), hc2[j], e0, false);
      e1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, 0x0C050C01u)), hc2[j], e1, false);
      e2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, 0x0C060C02u)), hc2[j], e2, false);
      e3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, 0x0C070C03u)), hc2[j], e3, false);
      o0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, 0x0C040C00u)), hc2[j], o0, false);
      o1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, 0x0C050C01u)), hc2[j], o1, false);
      o2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, 0x0C060C02u)), hc2[j], o2, false);
      o3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, 0x0C070C03u)), hc2[j], o3, false);
    }
    // clamp_i16(acc >> hshift) of (even row, odd row) packed by one v_cvt_pk_i16_i32 per channel
    uint4 pk;
    pk.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e0 >> a.hshift, o0 >> a.hshift));
    pk.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e1 >> a.hshift, o1 >> a.hshift));
    pk.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e2 >> a.hshift, o2 >> a.hshift));
    pk.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e3 >> a.hshift, o3 >> a.hshift));
    s_p[k * kTileW + lane] = pk;
  }
  __syncthreads();

  // ---- 3. vertical pass + epilogue: wave = output row, lane = output column ----
  uint8_t *dst = trk.dst[track];
  const uint8_t *l2 = trk.l2[track];
  for (int ly = wave; ly < thh; ly += kBlock >> 6) {
    const int oy = ty0 + ly;
    const uint32_t *vc2 = s_vc + ly * (a.npv + 1);                  // LDS, the same address for every lane (broadcast)
    const int kp = (int)vc2[0] - (sy0 >> 1);                        // first row pair
    int acc0 = a.vround, acc1 = a.vround, acc2 = a.vround, acc3 = a.vround;
    const uint4 *col = s_p + kp * kTileW + lane;
    for (int j = 0; j < a.npv; j++) {
      const uint4 t = col[j * kTileW];
      const short2v cf = __builtin_bit_cast(short2v, vc2[1 + j]);
      acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, t.x), cf, acc0, false);
      acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, t.y), cf, acc1, false);
      acc2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, t.z), cf, acc2, false);
      acc3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, t.w), cf, acc3, false);
    }
    if (lane < tw) {
      uint32_t p = a.vshift == 21 ? pack_sat_shr21(acc0, acc1, acc2, acc3)
                                  : (uint32_t)clamp255(acc0 >> a.vshift) | ((uint32_t)clamp255(acc1 >> a.vshift) << 8) |
                                        ((uint32_t)clamp255(acc2 >> a.vshift) << 16) | ((uint32_t)clamp255(acc3 >> a.vshift) << 24);
      if (a.blend) {
        const uint32_t q = reinterpret_cast<const uint32_t *>(l2 + (size_t)oy * a.irow2)[tx0 + lane];
        p = chroma_rgba(p, q, bf, nbf);
      }
      if (a.use_lut) p = lut3_rgba(s_lut, p);
      __builtin_nontemporal_store(p, reinterpret_cast<uint32_t *>(dst + (size_t)oy * a.orow) + tx0 + lane);      // written once, not read back
    }
  }
}

// =====================================================================================================================
// k_gauss5x -- 5x5 binomial blur of RGBA32 frames (the chain's blur stage, BASELINE config 4/5, and lgpu_gauss5).
// Same arithmetic as k_separable<5,5> with the [1 4 6 4 1] bank (bit-identical; rows that are not 8-byte aligned still take that one): exact row sums,
// one rounding (sum + 128) >> 8, edge pixels replicated.  Because the taps are 1/4/6 the whole thing fits SWAR: a
// pixel is split once into its even bytes and its odd bytes (two dwords holding two 16-bit lanes each); a row sum is
// <= 16 * 255 = 4080 and a column sum of row sums <= 65280, + 128 = 65408 < 2^16, so neither pass can carry across a
// 16-bit lane and both passes are plain 32-bit adds / shifts on four channels at a time (~8 VALU per channel-quad per
// pass instead of 5 multiply-adds per channel).  Tile = 64 x 16 output pixels, window 68 x 20 staged with 8-byte loads.
// =====================================================================================================================
struct G5Args {
  int w, h, irow, orow, irow2;
  int blend, use_lut;
  uint32_t bf, nbf;
  const int32_t *bf_d;
  int tiles_x;
  const uint2 *kscale;      // device [256] {K2, K1}: lgpu_alpha_scalers (the chroma blend's translucent scaling as integers)
};
constexpr int kG5W = 64, kG5H = 16, kG5WW = kG5W + 4, kG5WH = kG5H + 4;
constexpr int kG5Sub = 4;          // sub-tiles (of kG5H rows) a workgroup walks down, prefetching the next window in registers

__device__ __forceinline__ uint32_t g5_sum(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e) {
  return a + e + ((b + d) << 2) + (c << 2) + (c << 1);   // 32-bit SWAR operands: no 24-bit multiply
}

// MH (measurement variant, LGPU_G5_MFMA=1): the horizontal pass as a banded-Toeplitz product on the matrix cores, the form BASELINE config 4
// names ("MFMA row/col").  The window is kept as packed pixels biased to int8; A = 16 window rows x 16 pixels (one ds_read_b128 per lane),
// B[k = (pixel, byte)][n = (column, channel)] = {1 4 6 4 1}[pixel - column] on matching channels; 16 source pixels give 12 output columns
// = three v_mfma_i32_16x16x64_i8 per A fragment, six fragments per 64 columns, two row blocks (rows 0..15 and 4..19) per 20-row window:
// 36 MFMAs per tile.  Every D register is one channel of one column: it is un-biased (+ 16 * 128) and written as a 16-bit lane of the
// SWAR layout the vertical pass reads.  Same bytes as the SWAR pass; the timing is in profiles/r02/gauss5_mfma.md.
constexpr int kG5MPitch = 76 * 4;        // bytes per window row of the MFMA variant: 6 fragments x 12 pixels + 4
template <bool EPI, bool MH = false>   // EPI: chroma blend with layer 2 and / or the gamma LUT after the blur (the chain); false = plain blur
__global__ __launch_bounds__(256) void k_gauss5x(G5Args a, SepTracks t, Lut8 l) {
  __shared__ uint4 s_win4[MH ? (kG5WH * kG5MPitch + 64) / 16 : kG5WH * kG5WW / 2];
  __shared__ uint2 s_h[kG5WH * kG5W];
  __shared__ uint8_t s_lut[256];
  __shared__ uint2 s_k[EPI ? 256 : 1];
  uint2 *s_win = reinterpret_cast<uint2 *>(s_win4);
  const int tid = threadIdx.x, trk = blockIdx.y;
  if (EPI && a.blend) s_k[tid] = a.kscale[tid];          // simple_blend.c:137-145 as (c * K) >> 16 (the first barrier below publishes it)
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int tx0 = tx * kG5W, ty00 = ty * (kG5H * kG5Sub);
  const int nsub = min(kG5Sub, (a.h - ty00 + kG5H - 1) / kG5H);
  const uint8_t *src = t.src[trk];
  if (EPI && a.use_lut) stage_lut(s_lut, l);
  constexpr uint32_t M = 0x00FF00FFu;
  constexpr int kPairs = kG5WH * (kG5WW / 2), kIter = (kPairs + 255) / 256;
  // per-lane window slots: row / pair-of-pixels index, column offsets (clamped) are the same for every sub-tile
  int wrow[kIter], off0[kIter], off1[kIter];
#pragma unroll
  for (int k = 0; k < kIter; k++) {
    const int i = tid + k * 256;
    const int row = i / (kG5WW / 2), pr = i - row * (kG5WW / 2);
    const int x = tx0 - 2 + 2 * pr;
    const int x0 = x < 0 ? 0 : x >= a.w ? a.w - 1 : x, x1 = x + 1 < 0 ? 0 : x + 1 >= a.w ? a.w - 1 : x + 1;
    wrow[k] = row; off0[k] = 4 * x0; off1[k] = 4 * x1;
  }
  uint2 cur[kIter];
  auto fetch = [&](int ty0) {
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      if (tid + k * 256 < kPairs) {
        int y = ty0 - 2 + wrow[k];
        y = y < 0 ? 0 : y >= a.h ? a.h - 1 : y;
        const uint8_t *rp = src + (size_t)y * a.irow;
        if (off1[k] == off0[k] + 4) cur[k] = *reinterpret_cast<const uint2 *>(rp + off0[k]);
        else cur[k] = make_uint2(*reinterpret_cast<const uint32_t *>(rp + off0[k]), *reinterpret_cast<const uint32_t *>(rp + off1[k]));
      }
    }
  };
  fetch(ty00);
  uint32_t bf = a.bf, nbf = a.nbf;
  if (EPI && a.blend && a.bf_d) { bf = (uint32_t)*a.bf_d & 0xFFu; nbf = 0xFFu - bf; }
  const uint8_t *l2 = t.l2[trk];
  uint8_t *dst = t.dst[trk];
  const int col = tid & 63, rg = tid >> 6;
  const int ox = tx0 + col;
  for (int sub = 0; sub < nsub; sub++) {
    const int ty0 = ty00 + sub * kG5H;
    if (!MH) {
#pragma unroll
      for (int k = 0; k < kIter; k++)
        if (tid + k * 256 < kPairs) s_win4[tid + k * 256] = make_uint4(cur[k].x & M, (cur[k].x >> 8) & M, cur[k].y & M, (cur[k].y >> 8) & M);
    } else {
      uint8_t *s_raw = reinterpret_cast<uint8_t *>(s_win4);
#pragma unroll
      for (int k = 0; k < kIter; k++) {
        const int i = tid + k * 256;
        if (i < kPairs) {
          const int row = i / (kG5WW / 2), pr = i - row * (kG5WW / 2);
          *reinterpret_cast<uint2 *>(s_raw + row * kG5MPitch + pr * 8) = make_uint2(cur[k].x ^ 0x80808080u, cur[k].y ^ 0x80808080u);
        }
      }
    }
    __syncthreads();
    if (sub + 1 < nsub) fetch(ty0 + kG5H);       // next window's loads fly under this sub-tile's two passes
    if (!MH) {
      for (int i = tid; i < kG5WH * kG5W; i += 256) {
        const int row = i >> 6, c = i & 63;
        const uint2 *p = s_win + row * kG5WW + c;
        const uint2 A = p[0], B = p[1], C = p[2], D = p[3], E = p[4];
        s_h[i] = make_uint2(g5_sum(A.x, B.x, C.x, D.x, E.x), g5_sum(A.y, B.y, C.y, D.y, E.y));
      }
    } else {
      const uint8_t *s_raw = reinterpret_cast<const uint8_t *>(s_win4);
      const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
      // B fragments of the three 4-column groups of a 16-pixel span: lane (g, n) supplies k = 16 g + e, e = 0..15
      int4v bfr[3];
#pragma unroll
      for (int nb = 0; nb < 3; nb++) {
        uint32_t w4[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t v = 0;
#pragma unroll
          for (int e4 = 0; e4 < 4; e4++) {
            const int k = 16 * g + 4 * d + e4, px = k >> 2, ch = k & 3, col = 4 * nb + (m >> 2), och = m & 3, j = px - col;
            const int tap = (ch == och && j >= 0 && j <= 4) ? (j == 0 || j == 4 ? 1 : j == 2 ? 6 : 4) : 0;
            v |= (uint32_t)tap << (8 * e4);
          }
          w4[d] = v;
        }
        bfr[nb] = int4v{(int)w4[0], (int)w4[1], (int)w4[2], (int)w4[3]};
      }
      uint16_t *s_h16 = reinterpret_cast<uint16_t *>(s_h);
      const int slot = (m & 1) * 2 + ((m >> 1) & 1);                 // 16-bit lane of channel m & 3 in the uint2 {ch0 | ch2 << 16, ch1 | ch3 << 16}
      // 12 (row block, fragment) jobs over the four waves
      for (int job = wave; job < 12; job += 4) {
        const int blk = job / 6, fj = job - blk * 6;
        const int row0 = blk * 4;                                      // block 1 covers rows 4..19 and delivers rows 16..19
        const int4v afr = *reinterpret_cast<const int4v *>(s_raw + (row0 + m) * kG5MPitch + fj * 48 + g * 16);
        const int4v zero = {0, 0, 0, 0};
#pragma unroll
        for (int nb = 0; nb < 3; nb++) {
          const int4v d = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr, bfr[nb], zero, 0, 0, 0);
          const int col = fj * 12 + nb * 4 + (m >> 2);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = row0 + 4 * g + r;
            if (col < kG5W && (blk == 0 || row >= 16)) s_h16[(row * kG5W + col) * 4 + slot] = (uint16_t)(d[r] + 2048);
          }
        }
      }
    }
    __syncthreads();
    if (ox < a.w) {
      uint2 r[8];
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = s_h[(rg * 4 + k) * kG5W + col];
      uint32_t px[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t ev = g5_sum(r[j].x, r[j + 1].x, r[j + 2].x, r[j + 3].x, r[j + 4].x) + 0x00800080u;
        const uint32_t od = g5_sum(r[j].y, r[j + 1].y, r[j + 2].y, r[j + 3].y, r[j + 4].y) + 0x00800080u;
        px[j] = __builtin_amdgcn_perm(od, ev, 0x07030501u);     // >> 8 of the four 16-bit lanes, re-interleaved
      }
      const int oy0 = ty0 + rg * 4;
      if (EPI) {
        if (a.blend) {
          // chroma blend + gamma LUT as in k_half8s: mix sums r_c = bf * s2_c + nbf * s1_c from v_dot4, translucent layer-2 pixels scale both
          // sources first by the integer scalers of lgpu_alpha_scalers (alpha = 255 -> identity), the LUT is gathered from r_c >> 8
          uint32_t q[4];
#pragma unroll
          for (int j = 0; j < 4; j++) q[j] = (oy0 + j < a.h) ? reinterpret_cast<const uint32_t *>(l2 + (size_t)(oy0 + j) * a.irow2)[ox] : 0xFF000000u;
          const uint32_t w_lo = bf | (nbf << 8), w_hi = w_lo << 16;
          uint32_t r0[4], r1[4], r2[4];
          bool opaque = true;
#pragma unroll
          for (int j = 0; j < 4; j++) opaque = opaque && (q[j] >= 0xFF000000u);
          if (__all(opaque)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const uint32_t x01 = __builtin_amdgcn_perm(px[j], q[j], 0x05010400u), x2 = __builtin_amdgcn_perm(px[j], q[j], 0x0C0C0602u);
              r0[j] = __builtin_amdgcn_udot4(x01, w_lo, 0u, false); r1[j] = __builtin_amdgcn_udot4(x01, w_hi, 0u, false);
              r2[j] = __builtin_amdgcn_udot4(x2, w_lo, 0u, false);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const uint2 kk = s_k[q[j] >> 24];
              const uint32_t qq = q[j], p = px[j];
              const uint32_t qa = __umul24(qq & 0xFF, kk.x), qb = __umul24((qq >> 8) & 0xFF, kk.x), qc = __umul24((qq >> 16) & 0xFF, kk.x);
              const uint32_t pa = __umul24(p & 0xFF, kk.y), pb = __umul24((p >> 8) & 0xFF, kk.y), pc = __umul24((p >> 16) & 0xFF, kk.y);
              r0[j] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pa, qa, 0x0C0C0602u), w_lo, 0u, false);
              r1[j] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pb, qb, 0x0C0C0602u), w_lo, 0u, false);
              r2[j] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pc, qc, 0x0C0C0602u), w_lo, 0u, false);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (a.use_lut) {
              uint32_t tq = ((uint32_t)s_lut[r2[j] >> 8] << 8) | s_lut[r1[j] >> 8];
              tq = (tq << 8) | s_lut[r0[j] >> 8];
              px[j] = (px[j] & 0xFF000000u) | tq;
            } else px[j] = __builtin_amdgcn_perm(r1[j], r0[j], 0x0C0C0501u) | ((r2[j] << 8) & 0x00FF0000u) | (px[j] & 0xFF000000u);
          }
        } else if (a.use_lut) {
#pragma unroll
          for (int j = 0; j < 4; j++) px[j] = lut3_rgba(s_lut, px[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (oy0 + j < a.h) __builtin_nontemporal_store(px[j], reinterpret_cast<uint32_t *>(dst + (size_t)(oy0 + j) * a.orow) + ox);      // written once, not read back
    }
  }
}

// =====================================================================================================================
// k_half8s -- fast path for an exact 2:1 reduction with a uniform 8-tap filter (the headline 3840x2160 -> 1920x1080
// bicubic case of lgpu_chain / lgpu_resize).  Same arithmetic as k_separable<8,8> (bit-identical output; the tests
// run both), restructured because the generic kernel is VALU-bound on gfx950 (profiles/r01/step1_separable_v1.md:
// integer VOP3 ops issue at ~4.7 clk each) and, as a one-role kernel, parked its waves on VMEM issue
// (profiles/r01/step3_half8.md).  A workgroup is 4 compute waves + 2 memory waves, two workgroups per CU, persistent
// and XCD-aware, walking 64 x 16 output tiles:
//   memory waves : source windows (38 x 136 packed pixels, raw bytes) by global_load_lds into a 2-slot ring, issued a
//                  tile and a half ahead; once a window has landed its bytes are biased to int8 IN LDS with
//                  ds_xor_b64 (no VGPR round trip, 41 instructions per window); layer-2 tiles by global_load_lds into a
//                  2-slot ring; finished tiles read back from their ring slot and stored with 16-byte stores;
//                  edge-replicate fix-up of frame-border windows
//   compute waves: LDS, MFMA and VALU only
//     horizontal : banded-Toeplitz product on the matrix cores over packed RGBA pixels: A = 16 window rows x 64 bytes
//                  (one ds_read_b128 per lane), v_mfma_i32_16x16x64_i8 twice (Q14 tap = 64 * hi + lo, lo stored
//                  doubled), B[k][n] = tap[pixel - 2 * column] on matching channels (a BGRA source permutes B).  The
//                  third 16-row block holds only window rows 32..37: its A rows are ordered so that they come out as
//                  registers 0 / 1 of lane groups 0..2 and only those are post-processed.
//     vertical   : lane = column, 4 consecutive output rows per wave, v_dot2c_i32_i16 on row pairs
//     epilogue   : chroma blend with layer 2 from the ring slot, gamma LUT from LDS, result into the ring slot.  The
//                  reference's float scaling of translucent pixels, (uint8_t)((float)c * alpha) and
//                  (uint8_t)((float)c * (1 - alpha)) (simple_blend.c:137-145), is evaluated as (c * K[alpha]) >> 16 with
//                  two 17-bit constants per alpha from a 2 KB LDS table (host_tables.cpp: lgpu_alpha_scalers proves
//                  the equality for all 2 x 65,536 operand pairs when it builds the table; alpha = 255 maps to the
//                  identity, so opaque and translucent pixels share one instruction stream).
// Two workgroup barriers per tile (A(n): window n / layer-2 n landed and biased, the row-pair buffer free; B(n): row
// pairs complete and window slot n & 1 free), raw s_barrier so the memory waves' DMA stays in flight across them; they
// wait with a counted vmcnt that leaves exactly the newest window outstanding.
// =====================================================================================================================
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address

constexpr int kH8Pitch = 544;                   // window row pitch in bytes (136 px; 136 dwords = 8 mod 16): conflict-free for the lane
                                                //   groups ds_read_b128 really uses ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32)
constexpr int kH8Chunks = 34;                   // 16-byte column chunks per window row (134 px -> 33.5)

struct Half8Args {
  int sw, sh, irow, dw, dh, orow;
  const int4v *bfrag;               // device: [2][64] B fragments (hi, lo)
  const uint2 *kscale;              // device: [256] {K2, K1}: layer-2 / track colour scalers per layer-2 alpha (lgpu_alpha_scalers)
  uint32_t vc[4];                   // vertical tap pairs (c0,c1) (c2,c3) (c4,c5) (c6,c7), 2 x int16 each
  int swap_rb;
  int xoff;                         // 1: windows start one pixel further left (source x = 2 * tx0 - 4) so that every 16-byte request is 16-byte aligned
  int blend, irow2;
  uint32_t bf, nbf;
  const int32_t *bf_d;
  int use_lut;
  int tiles_x, tiles_y, ntracks;
  unsigned long long *dbg;          // profiling build (LGPU_PROFILING): per-wave phase cycle sums [grid][6][8]
  int nt_out;                       // result stores non-temporal (the result is not read back by the next launch)
};

constexpr int kH8sCW = 4;                        // compute waves per workgroup (+ 2 memory waves)
constexpr int kH8sThreads = (kH8sCW + 2) * 64;
struct H8S {
  static constexpr int kRows = 38, kPairs = 19, kMBlocks = 3, kTileH = 16;
  static constexpr int kRPW = 16 / kH8sCW;             // vertical pass: output rows per compute wave
  static constexpr int kNQ = 16 / kH8sCW;              // horizontal pass: 4-column blocks per compute wave
  static constexpr int kWinBytes = kRows * 544;        // 20672 = 20 x 1024 + 192: 21 DMA instructions, the last with 12 lanes
  static constexpr int kRounds = 21, kTailLanes = 12;
  static constexpr int kHPitch = 264;                  // dwords per row pair of the intermediate: 256 + 8, so the four row groups of an
                                                       //   MFMA result (pairs 2g, g = 0..3) land on different LDS banks (2 * 264 * g mod 64 = 16 g)
  // LDS map; the two small tables come first so that their addresses fit the 16-bit offset field of the ds instructions
  static constexpr int kOffLut = 0;                                  // gamma LUT, 256 bytes
  static constexpr int kOffK = 256;                                  // uint2[256] alpha scalers
  static constexpr int kOffWin = kOffK + 2048;                       // window slots
};
struct H8SL {
  static constexpr int kOffH = H8S::kOffWin + 2 * H8S::kWinBytes;              // [19][264] dwords of 2 x int16
  static constexpr int kOffQ = kOffH + H8S::kPairs * H8S::kHPitch * 4;         // 2 x [16][64] pixels: layer 2 in, finished tile out
  static constexpr size_t kLds = kOffQ + 2 * 4096;                             // 71904: two workgroups per CU (one window slot with three
};                                                                             //   workgroups per CU measured 223-245 us against 169)

#define H8S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
typedef const __attribute__((address_space(1))) void *h8s_gptr;
typedef __attribute__((address_space(3))) void *h8s_lptr;

// bytes [b0, b1) of a landed window ^= 0x80 (uint8 -> int8 for the matrix cores), 8 bytes per lane and instruction,
// executed by the LDS itself.  b0, b1 multiples of 8.  Ordered behind the caller's counted vmcnt (the DMA data is
// there) and in front of barrier A by its lgkmcnt(0).
template <int B0, int B1>
__device__ __forceinline__ void h8s_bias_window(uint8_t *win, int lane) {
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  const u32x2v m = {0x80808080u, 0x80808080u};
  const uint32_t addr = (uint32_t)(uintptr_t)(h8s_lptr)win + (uint32_t)lane * 8u;
#pragma unroll
  for (int off = B0; off < B1; off += 512) {
    if (off + 512 <= B1) asm volatile("ds_xor_b64 %0, %1 offset:%2" :: "v"(addr), "v"(m), "n"(off) : "memory");
    else if (off + lane * 8 < B1) asm volatile("ds_xor_b64 %0, %1 offset:%2" :: "v"(addr), "v"(m), "n"(off) : "memory");
  }
}


// window chunk c = 64 k + lane (16 bytes, LDS offset 16 c: a row is exactly 34 chunks) <- source row sy0 + c / 34,
// pixels sx0 + 4 (c % 34) ..+3; rows are clamped to the frame here, columns are fetched from the nearest in-frame
// 4-pixel position and re-selected by h8s_fix_edges()
struct H8sLaneOff { uint32_t v[H8S::kRounds]; };     // per-lane byte offset of chunk 64 k + lane inside an unclamped window
__device__ __forceinline__ void h8s_lane_offsets(const Half8Args &a, int lane, H8sLaneOff &lo) {
#pragma unroll
  for (int k = 0; k < H8S::kRounds; k++) {
    const int c = k * 64 + lane, r = (c * 241) >> 13, ch = c - r * kH8Chunks;
    lo.v[k] = (uint32_t)r * (uint32_t)a.irow + (uint32_t)ch * 16u;
  }
}
// frame-border windows (rare): rows clamped to the frame, columns fetched from the nearest in-frame 4-pixel position
// (re-selected by h8s_fix_edges()).  Out of line and rolled: the common path must stay small in the instruction cache.
__device__ __noinline__ void h8s_issue_window_border(int sw, int sh, int irow, const uint8_t *src, int sx0, int sy0, int lane, uint8_t *win,
                                                     int k0, int k1) {
  for (int k = k0; k < k1; k++) {
    const int c = k * 64 + lane, r = (c * 241) >> 13, ch = c - r * kH8Chunks;     // c / 34 for c < 1344
    int sy = sy0 + r;
    sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
    int x = sx0 + ch * 4;
    x = x < 0 ? 0 : x > sw - 4 ? sw - 4 : x;
    const uint8_t *g = src + ((uint32_t)sy * (uint32_t)irow + (uint32_t)x * 4u);
    if (c < H8S::kRows * kH8Chunks)
      __builtin_amdgcn_global_load_lds((h8s_gptr)g, (h8s_lptr)(win + k * 1024), 16, 0, 0);
  }
}
template <int K0, int K1, int AUX = 0>
__device__ __forceinline__ void h8s_issue_window(const Half8Args &a, const uint8_t *src, int tx0, int ty0, int lane, uint8_t *win,
                                                 const H8sLaneOff &lo) {
  const int sx0 = 2 * tx0 - 3 - a.xoff, sy0 = 2 * ty0 - 3;
  if (sx0 >= 0 && sx0 + 4 * kH8Chunks <= a.sw && sy0 >= 0 && sy0 + H8S::kRows <= a.sh) {
    // window inside the frame (tile-uniform): scalar base + the precomputed lane offsets, no per-request address math
    const uint8_t *base = src + ((uint32_t)sy0 * (uint32_t)a.irow + (uint32_t)sx0 * 4u);
#pragma unroll
    for (int k = K0; k < K1; k++)
      if (k < H8S::kRounds - 1 || lane < H8S::kTailLanes)
        __builtin_amdgcn_global_load_lds((h8s_gptr)(base + lo.v[k]), (h8s_lptr)(win + k * 1024), 16, 0, AUX);
  } else h8s_issue_window_border(a.sw, a.sh, a.irow, src, sx0, sy0, lane, win, K0, K1);
}
__device__ __forceinline__ uint32_t h8s_pick(const uint4 &v, int j) { return j <= 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
// frame-border windows only (tile-uniform test by the caller): re-select the pixels of the chunks that were fetched
// from a clamped position, i.e. edge replicate.  Only the chunks that can be affected are visited (the first chunk of
// a row at the left frame edge, the last few at the right edge), one per lane; each memory wave fixes the chunks it
// fetched itself (c0 <= c < c1).  The LDS accesses are asm so that the compiler does not drain the DMA queue in front
// of them: the caller's counted vmcnt has already covered this window, newer requests go to the other slot.
__device__ __noinline__ void h8s_fix_edges(uint8_t *win, int sx0, int sw, int lane, int c0, int c1) {
  const int nlo = sx0 < 0 ? (-sx0 + 3) >> 2 : 0;                               // chunks starting left of the frame
  int chh = sw - 4 - sx0 < 0 ? 0 : ((sw - 4 - sx0) >> 2) + 1;                  // first chunk reaching past the right edge
  if (chh < nlo) chh = nlo;
  const int nhi = chh < kH8Chunks ? kH8Chunks - chh : 0, na = nlo + nhi;
  for (int idx = lane; idx < H8S::kRows * na; idx += 64) {
    const int r = idx / na, j = idx - r * na, ch = j < nlo ? j : chh + (j - nlo), c = r * kH8Chunks + ch;
    if (c >= c0 && c < c1) {
      const int x = sx0 + ch * 4, xl = x < 0 ? 0 : x > sw - 4 ? sw - 4 : x;
      const uint32_t addr = (uint32_t)(uintptr_t)(h8s_lptr)(win + c * 16);
      typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
      u32x4v fv;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fv) : "v"(addr) : "memory");
      const uint4 f = make_uint4(fv.x, fv.y, fv.z, fv.w);
      const int lim = sw - 1 - xl;
      int j0 = x - xl, j1 = j0 + 1, j2 = j0 + 2, j3 = j0 + 3;
      j0 = j0 > lim ? lim : j0; j1 = j1 > lim ? lim : j1; j2 = j2 > lim ? lim : j2; j3 = j3 > lim ? lim : j3;
      const u32x4v o = {h8s_pick(f, j0), h8s_pick(f, j1), h8s_pick(f, j2), h8s_pick(f, j3)};
      asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(o) : "memory");
    }
  }
}
__device__ __forceinline__ bool h8s_border(int sx0, int sw) { return sx0 < 0 || sx0 + 4 * kH8Chunks > sw; }

__device__ __noinline__ void h8s_issue_q2_partial(const uint8_t *l2, int irow2, int dw, int dh, int tx0, int ty0, int lane, uint8_t *slot) {
  const int x = min(tx0 + lane, dw - 1);              // partial-width tile: one row per request, columns clamped
  for (int r = 0; r < 16; r++) {
    const int oy = min(ty0 + r, dh - 1);
    const uint8_t *g = l2 + ((uint32_t)oy * (uint32_t)irow2 + (uint32_t)x * 4u);
    __builtin_amdgcn_global_load_lds((h8s_gptr)g, (h8s_lptr)(slot + r * 256), 4, 0, 0);
  }
}
__device__ __forceinline__ void h8s_issue_q2(const Half8Args &a, const uint8_t *l2, int tx0, int ty0, int lane, uint8_t *slot) {
  if (tx0 + kTileW <= a.dw) {          // 4 rows x 256 B per request, LDS image lane-linear
    const int row = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int oy = min(ty0 + k * 4 + row, a.dh - 1);
      const uint8_t *g = l2 + ((uint32_t)oy * (uint32_t)a.irow2 + (uint32_t)(tx0 + chunk * 4) * 4u);
      __builtin_amdgcn_global_load_lds((h8s_gptr)g, (h8s_lptr)(slot + k * 1024), 16, 0, 2);      // nt: layer 2 is read once (no halo, unlike the source windows)
    }
  } else h8s_issue_q2_partial(l2, a.irow2, a.dw, a.dh, tx0, ty0, lane, slot);
}
__device__ __forceinline__ void h8s_read_tile(int lane, const uint8_t *slot, uint4 (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const uint4 *>(slot + (k * 64 + lane) * 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot may be overwritten (layer-2 DMA) from here on
}
__device__ __noinline__ void h8s_store_tile_partial(uint8_t *dst, int orow, int tw, int thh, int tx0, int ty0, int lane, uint4 v0, uint4 v1, uint4 v2, uint4 v3) {
  const int row = lane >> 4, chunk = lane & 15;
  const uint4 v[4] = {v0, v1, v2, v3};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int ly = k * 4 + row;
    uint32_t *d = reinterpret_cast<uint32_t *>(dst + (size_t)(ty0 + ly) * orow) + tx0 + chunk * 4;
    if (ly < thh) {
      if (chunk * 4 + 0 < tw) d[0] = v[k].x;
      if (chunk * 4 + 1 < tw) d[1] = v[k].y;
      if (chunk * 4 + 2 < tw) d[2] = v[k].z;
      if (chunk * 4 + 3 < tw) d[3] = v[k].w;
    }
  }
}
__device__ __forceinline__ void h8s_store_tile(const Half8Args &a, uint8_t *dst, int tx0, int ty0, int lane, const uint4 (&v)[4]) {
  const int tw = min(kTileW, a.dw - tx0), thh = min(H8S::kTileH, a.dh - ty0);
  const int row = lane >> 4, chunk = lane & 15;
  if (tw == kTileW && !(a.orow & 15)) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ly = k * 4 + row;
      // non-temporal: the result stream is written once and not read back by this launch; with the default policy its 133 MB cost ~40 us next to 98 us
      // of source reads in a free-running probe, with `nt` ~27 us (tools/dma_ceiling.hip, profiles/r02/ring_experiment.md)
      if (ly < thh) {
        const u32x4v o = {v[k].x, v[k].y, v[k].z, v[k].w};
        u32x4v *op = reinterpret_cast<u32x4v *>(dst + ((size_t)(ty0 + ly) * a.orow + (size_t)(tx0 + chunk * 4) * 4));
        if (a.nt_out) __builtin_nontemporal_store(o, op); else *op = o;
      }
    }
  } else h8s_store_tile_partial(dst, a.orow, tw, thh, tx0, ty0, lane, v[0], v[1], v[2], v[3]);
}

// walks a workgroup's (track, tile) list with stride wstride without per-step divisions (all wave-uniform, SALU)
struct H8sTile {
  int track, tx, ty;                 // tile column / row index
  int dq, dr, tiles_x, tiles_y;      // wstride = dq * tiles_x + dr
  __device__ __forceinline__ void init(const Half8Args &a, int work, int wstride) {
    tiles_x = a.tiles_x; tiles_y = a.tiles_y;
    const int tiles = tiles_x * tiles_y;
    track = work / tiles;
    const int tile = work - track * tiles;
    ty = tile / tiles_x; tx = tile - ty * tiles_x;
    dq = wstride / tiles_x; dr = wstride - dq * tiles_x;
  }
  __device__ __forceinline__ void step() {
    tx += dr; ty += dq;
    if (tx >= tiles_x) { tx -= tiles_x; ty += 1; }
    while (ty >= tiles_y) { ty -= tiles_y; track += 1; }
  }
  __device__ __forceinline__ int tx0() const { return tx * kTileW; }
  __device__ __forceinline__ int ty0() const { return ty * H8S::kTileH; }
};

// ---- compute waves (shared by both loader designs below) -------------------------------------------------------------
// ABL (A / B builds only): 16 = barriers only (no compute), 32 = no source window loads
template <int DBG, int ABL, int NWAVES>
__device__ __forceinline__ void h8s_compute(const Half8Args &a, uint8_t *smem, int wave, int lane, int work, int wend, int wstride) {
  using C = H8S;
  using L = H8SL;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define H8S_T(i)                                                                                     \
  if (DBG) {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                     \
    tacc[i] += now_ - tprev; tprev = now_;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  uint8_t *s_win = smem + C::kOffWin;
  uint8_t *s_h = smem + L::kOffH;
  uint8_t *s_q = smem + L::kOffQ;
  uint8_t *s_lut = smem + C::kOffLut;
  uint32_t bf = a.bf, nbf = a.nbf;
  if (a.blend && a.bf_d) { bf = (uint32_t)a.bf_d[0] & 0xFF; nbf = 0xFF - bf; }
  const int4v b_hi = a.bfrag[lane], b_lo = a.bfrag[64 + lane];
  // The window holds px - 128 and the taps sum to 16384, so the matrix product is 128 * (t - 16384) for the spec's
  // t = clamp_i16((h + 64) >> 7): the intermediate is kept as t' = t - 16384 (fits int16 without wrapping for the
  // filters try_half8 admits), the clamp becomes min(t', 16383), and the vertical pass adds 16384 * sum(vc) = 2^28 back.
  // The low-part taps are stored doubled so that t' sits in bits 8..23 of (dh << 7) + dl: a byte permute extracts it.
  const int kb = 128;                                     // 2 * 64: the rounding of >> 7, doubled
  const int4v cbias = {kb, kb, kb, kb};
  const short2v tmax = {16383, 16383};
  const short2v vc0 = __builtin_bit_cast(short2v, a.vc[0]), vc1 = __builtin_bit_cast(short2v, a.vc[1]);
  const short2v vc2 = __builtin_bit_cast(short2v, a.vc[2]), vc3 = __builtin_bit_cast(short2v, a.vc[3]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  constexpr int RPW = C::kRPW, NQ = C::kNQ;
  const int ly0 = wave * RPW;
  const int m = lane & 15, g = lane >> 4;
  // A fragments: lane (g, m) supplies row m, bytes 16 g .. 16 g + 15 of the block's 64-byte span.  Blocks 0 / 1 are window
  // rows 16 mb + m.  Block 2: D register r of lane group g is row 4 g + r of the block, so with the block's row
  // 4 g' + r' := window row 32 + 2 g' + (r' & 1) the six live rows 32..37 come out as registers 0 / 1 of groups 0..2,
  // already paired the way the vertical pass wants them (row pair 16 + g); group 3 / registers 2, 3 are not looked at
  // (rows 38 / 39 read whatever follows the slot: it is multiplied and dropped).
  const uint32_t aoff = (uint32_t)(m * kH8Pitch + wave * (NQ * 32) + g * 16);
  const uint32_t aoff2 = (uint32_t)((32 + 2 * (m >> 2) + (m & 1)) * kH8Pitch + wave * (NQ * 32) + g * 16);
  int par = 0;
  if (DBG) tprev = __builtin_amdgcn_s_memtime();
  for (; work < wend; work += wstride) {
    H8S_T(0)
    H8S_BARRIER();                                                                     // A(n)
    H8S_T(1)
    if (ABL & 16) { H8S_BARRIER(); par ^= 1; continue; }
    // ---- horizontal pass on the matrix cores ----
    // Software pipelined over the three 16-row blocks: the four A fragments of block mb + 1 are read while block mb is
    // on the matrix pipe, and a block's eight MFMAs are issued back to back before any result is consumed.
    const uint8_t *s_pl = s_win + par * C::kWinBytes;
    {
      uint32_t *hbase = reinterpret_cast<uint32_t *>(s_h) + wave * (NQ * 16) + m;   // + pr * kHPitch + q * 16
      int4v av[NQ], an[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) av[q] = *reinterpret_cast<const int4v *>(s_pl + aoff + q * 32);
#pragma unroll
      for (int mb = 0; mb < C::kMBlocks; mb++) {
        if (mb + 1 < C::kMBlocks) {
#pragma unroll
          for (int q = 0; q < NQ; q++)
            an[q] = *reinterpret_cast<const int4v *>(s_pl + (mb + 1 == 2 ? aoff2 : aoff + (mb + 1) * 16 * kH8Pitch) + q * 32);
        }
        int4v dh[NQ], dl[NQ];
        const int4v zero = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          dh[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[q], b_hi, zero, 0, 0, 0);
          dl[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[q], b_lo, cbias, 0, 0, 0);
        }
        if (mb == 2) {
          const int pr = 16 + g;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const uint32_t x0 = (uint32_t)((dh[q][0] << 7) + dl[q][0]), x1 = (uint32_t)((dh[q][1] << 7) + dl[q][1]);
            const short2v q0 = __builtin_elementwise_min(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(x1, x0, 0x06050201u)), tmax);
            if (g < 3) hbase[pr * C::kHPitch + q * 16] = __builtin_bit_cast(uint32_t, q0);
          }
        } else {
          const int pr = mb * 8 + 2 * g;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const uint32_t x0 = (uint32_t)((dh[q][0] << 7) + dl[q][0]), x1 = (uint32_t)((dh[q][1] << 7) + dl[q][1]);
            const uint32_t x2 = (uint32_t)((dh[q][2] << 7) + dl[q][2]), x3 = (uint32_t)((dh[q][3] << 7) + dl[q][3]);
            const short2v q0 = __builtin_elementwise_min(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(x1, x0, 0x06050201u)), tmax);
            const short2v q1 = __builtin_elementwise_min(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(x3, x2, 0x06050201u)), tmax);
            uint32_t *hp = hbase + pr * C::kHPitch + q * 16;
            hp[0] = __builtin_bit_cast(uint32_t, q0);
            hp[C::kHPitch] = __builtin_bit_cast(uint32_t, q1);
          }
        }
        if (mb + 1 < C::kMBlocks) {
#pragma unroll
          for (int q = 0; q < NQ; q++) av[q] = an[q];
        }
      }
    }
    H8S_T(2)
    H8S_BARRIER();                                                                     // B(n)
    H8S_T(3)
    // ---- vertical pass + epilogue; layer 2 comes from / the result goes to the ring slot ----
    {
      uint32_t *qs = reinterpret_cast<uint32_t *>(s_q + par * 4096) + ly0 * kTileW + lane;
      const uint4 *col = reinterpret_cast<const uint4 *>(s_h) + lane;
      uint4 w[RPW + 3];
#pragma unroll
      for (int i = 0; i < RPW + 3; i++) w[i] = col[(ly0 + i) * (C::kHPitch / 4)];
      uint32_t q2[RPW];
#pragma unroll
      for (int i = 0; i < RPW; i++) q2[i] = a.blend ? qs[i * kTileW] : 0xFF000000u;
      uint32_t px[RPW];
#pragma unroll
      for (int i = 0; i < RPW; i++) {
        const int vr = (1 << 20) + (1 << 28);
        int a0 = vr, a1 = vr, a2 = vr, a3 = vr;
#define H8_DOT(acc, fld)                                                                        \
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, w[i].fld), vc0, acc, false);     \
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, w[i + 1].fld), vc1, acc, false); \
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, w[i + 2].fld), vc2, acc, false); \
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, w[i + 3].fld), vc3, acc, false);
        H8_DOT(a0, x) H8_DOT(a1, y) H8_DOT(a2, z) H8_DOT(a3, w)
#undef H8_DOT
        px[i] = pack_sat_shr21(a0, a1, a2, a3);
      }
      // The workgroup has no static LDS, so the dynamic segment starts at LDS address 0 and a table's address is its offset:
      // the table base folds into the ds_read offset field.
      typedef const __attribute__((address_space(3))) uint8_t *lds_u8;
      if (a.blend) {
        // mix sums r_c = bf * s2_c + nbf * s1_c (< 2^16, the blended byte is r_c >> 8) for the three colour channels
        const uint32_t w_lo = bf | (nbf << 8), w_hi = w_lo << 16;
        uint32_t r0[RPW], r1[RPW], r2[RPW];
        bool opaque = true;
#pragma unroll
        for (int i = 0; i < RPW; i++) opaque = opaque && (q2[i] >= 0xFF000000u);
        if (__all(opaque)) {           // what decoded video is: no scaling (simple_blend.c:128-131)
#pragma unroll
          for (int i = 0; i < RPW; i++) {
            const uint32_t x01 = __builtin_amdgcn_perm(px[i], q2[i], 0x05010400u);     // [q.b0 p.b0 q.b1 p.b1]
            const uint32_t x2 = __builtin_amdgcn_perm(px[i], q2[i], 0x0C0C0602u);      // [q.b2 p.b2 0 0]
            r0[i] = __builtin_amdgcn_udot4(x01, w_lo, 0u, false); r1[i] = __builtin_amdgcn_udot4(x01, w_hi, 0u, false);
            r2[i] = __builtin_amdgcn_udot4(x2, w_lo, 0u, false);
          }
        } else {
          // s2_c = (q_c * K2[alpha]) >> 16, s1_c = (p_c * K1[alpha]) >> 16: byte 2 of a 24-bit product each (simple_blend.c:137-145
          // as integers, lgpu_alpha_scalers); alpha = 255 maps to the identity, so opaque pixels need no select
          typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
          u32x2v kk[RPW];
#pragma unroll
          for (int i = 0; i < RPW; i++) kk[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x2v *>((uintptr_t)(C::kOffK + (q2[i] >> 24) * 8));
#pragma unroll
          for (int i = 0; i < RPW; i++) {
            const uint32_t q = q2[i], p = px[i];
            const uint32_t qa = __umul24(q & 0xFF, kk[i].x), qb = __umul24((q >> 8) & 0xFF, kk[i].x), qc = __umul24((q >> 16) & 0xFF, kk[i].x);
            const uint32_t pa = __umul24(p & 0xFF, kk[i].y), pb = __umul24((p >> 8) & 0xFF, kk[i].y), pc = __umul24((p >> 16) & 0xFF, kk[i].y);
            r0[i] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pa, qa, 0x0C0C0602u), w_lo, 0u, false);   // [s2 s1 0 0] . [bf nbf 0 0]
            r1[i] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pb, qb, 0x0C0C0602u), w_lo, 0u, false);
            r2[i] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pc, qc, 0x0C0C0602u), w_lo, 0u, false);
          }
        }
        if (a.use_lut) {               // the LUT is gathered straight from the mix sums: index = r_c >> 8
          uint32_t o0[RPW], o1[RPW], o2[RPW];
#pragma unroll
          for (int i = 0; i < RPW; i++) {
            o0[i] = *(lds_u8)(uintptr_t)(C::kOffLut + (r0[i] >> 8)); o1[i] = *(lds_u8)(uintptr_t)(C::kOffLut + (r1[i] >> 8));
            o2[i] = *(lds_u8)(uintptr_t)(C::kOffLut + (r2[i] >> 8));
          }
#pragma unroll
          for (int i = 0; i < RPW; i++) {
            uint32_t t = (o2[i] << 8) | o1[i];
            t = (t << 8) | o0[i];
            px[i] = (px[i] & 0xFF000000u) | t;
          }
        } else {
#pragma unroll
          for (int i = 0; i < RPW; i++)
            px[i] = __builtin_amdgcn_perm(r1[i], r0[i], 0x0C0C0501u) | ((r2[i] << 8) & 0x00FF0000u) | (px[i] & 0xFF000000u);
        }
      } else if (a.use_lut) {
#pragma unroll
        for (int i = 0; i < RPW; i++) px[i] = lut3_rgba(s_lut, px[i]);
      }
#pragma unroll
      for (int i = 0; i < RPW; i++) qs[i * kTileW] = px[i];
    }
    par ^= 1;
  }
  H8S_BARRIER();                                                                       // A(last + 1)
  if (DBG && lane == 0)
    for (int i = 0; i < 8; i++) a.dbg[((size_t)blockIdx.x * NWAVES + wave) * 8 + i] = tacc[i];
#undef H8S_T
}

// shared prologue: tables into LDS, this workgroup's share of the XCD-aware persistent work list (each XCD owns a contiguous
// eighth of the (track, tile) list, so that the window halos of neighbouring tiles hit that XCD's L2)
__device__ __forceinline__ bool h8s_prologue(const Half8Args &a, const Lut8 &lut, uint8_t *smem, int &work, int &wend, int &wstride) {
  const int tid = threadIdx.x;
  if (a.use_lut) stage_lut(smem + H8S::kOffLut, lut);
  if (a.blend && tid < 256) reinterpret_cast<uint2 *>(smem + H8S::kOffK)[tid] = a.kscale[tid];
  const int nwork = a.tiles_x * a.tiles_y * a.ntracks;
  const int xcd = blockIdx.x & 7;
  wstride = (int)(gridDim.x >> 3);
  const int chunk = (nwork + 7) >> 3;
  wend = min((xcd + 1) * chunk, nwork);
  work = xcd * chunk + (int)(blockIdx.x >> 3);
  return work < wend;                              // workgroup-uniform
}

// ---- the kernel: compute waves above + two memory waves feeding the window ring by LDS-DMA (global_load_lds) ----------
// Measured alternatives (profiles/r02/step3_*): one window slot with three workgroups per CU 223-245 us against 169; a non-temporal
// policy on the source stream 196 against 176 (the halos live in L2); windows staged through registers by two loader waves and
// a mover wave (two more windows in flight per workgroup, hand-counted vmcnt) 173.3 against 173.3, memory side alone 163 against
// 161: the bytes in flight are not the limit -- with the compute waves ablated the launch moves its 818 MB at 5.0 TB/s, the
// rate a device-to-device hipMemcpy of the same size reaches on the same box (5.17 TB/s).
template <int DBG, int ABL>
__global__ __launch_bounds__(kH8sThreads, 3) void k_half8s(Half8Args a, SepTracks trk, Lut8 lut) {
  using C = H8S;
  using L = H8SL;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *s_win = smem + C::kOffWin;            // 2 x [38][136] packed source pixels
  uint8_t *s_q = smem + L::kOffQ;                // 2 x [16][64] pixels: layer 2 in, finished tile out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int work, wend, wstride;
  if (!h8s_prologue(a, lut, smem, work, wend, wstride)) return;
  if (wave < kH8sCW) { h8s_compute<DBG, ABL, kH8sCW + 2>(a, smem, wave, lane, work, wend, wstride); return; }
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define H8S_T(i)                                                                                     \
  if (DBG) {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                     \
    tacc[i] += now_ - tprev; tprev = now_;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  {

    // ------------------------------------------------ memory waves ------------------------------------------------
    __builtin_amdgcn_s_setprio(3);     // few instructions, all of them feeding the DMA queues: let them issue ahead of the compute waves (-0.8 % per launch)
    // Wave 4 fetches window DMA instructions 0..15, wave 5 fetches 16..20 and also moves the layer-2 / result tiles.
    // Window n+2 goes to slot n & 1, free after B(n) and needed at A(n+2); each wave issues the first part of its
    // share between B(n) and A(n+1) and the rest between A(n+1) and B(n+1), so both barrier intervals carry traffic:
    //   A(n)   | rest of window n+1; wave 5: read tile n-1 from its ring slot, layer-2 DMA n+1 into it, store tile n-1
    //   B(n)   | first part of window n+2 | wait for all but that part: window n+1 (and layer-2 n+1) have landed |
    //          | edge fix-up and int8 bias of window n+1 (own chunks)
    constexpr int WAUX = 0;                                      // cache policy of the source stream (non-temporal measured 196 us against 176: the halos live in L2)
    constexpr int K2 = (ABL & 64) ? 11 : 16;                     // A / B: balanced split of a window between the two waves
    constexpr int K0 = 0, K1 = (ABL & 128) ? K2 : 6, K3 = 21, K4 = 21;     // wave 4: [K0,K1) after B + [K1,K2) after A; wave 5: [K2,K3) after B (+ [K3,K4) after A)
    const bool w5 = wave == kH8sCW + 1;
    H8sLaneOff lo;
    h8s_lane_offsets(a, lane, lo);
    H8sTile t0, t1, t2;               // tiles n, n+1, n+2
    t0.init(a, work, wstride); t1 = t0; t1.step(); t2 = t1; t2.step();
    const bool has1 = work + wstride < wend;
    if (w5) {
      if (a.blend) h8s_issue_q2(a, trk.l2[t0.track], t0.tx0(), t0.ty0(), lane, s_q);
      if (!(ABL & 32)) h8s_issue_window<K2, K4, WAUX>(a, trk.src[t0.track], t0.tx0(), t0.ty0(), lane, s_win, lo);
      if (has1) { if (!(ABL & 32)) h8s_issue_window<K2, K3, WAUX>(a, trk.src[t1.track], t1.tx0(), t1.ty0(), lane, s_win + C::kWinBytes, lo); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K3 - K2) : "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (h8s_border(2 * t0.tx0() - 3 - a.xoff, a.sw)) h8s_fix_edges(s_win, 2 * t0.tx0() - 3 - a.xoff, a.sw, lane, K2 * 64, K4 * 64);
      h8s_bias_window<K2 * 1024, C::kWinBytes>(s_win, lane);
    } else {
      if (!(ABL & 32)) h8s_issue_window<K0, K2, WAUX>(a, trk.src[t0.track], t0.tx0(), t0.ty0(), lane, s_win, lo);
      if (has1) { if (!(ABL & 32)) h8s_issue_window<K0, K1, WAUX>(a, trk.src[t1.track], t1.tx0(), t1.ty0(), lane, s_win + C::kWinBytes, lo); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K1 - K0) : "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (h8s_border(2 * t0.tx0() - 3 - a.xoff, a.sw)) h8s_fix_edges(s_win, 2 * t0.tx0() - 3 - a.xoff, a.sw, lane, K0 * 64, K2 * 64);
      h8s_bias_window<K0 * 1024, K2 * 1024>(s_win, lane);
    }
    H8sTile tp = t0;                  // tile n-1
    bool has_prev = false;
    int par = 0;
    if (DBG) tprev = __builtin_amdgcn_s_memtime();
    for (; work < wend; work += wstride) {
      H8S_T(0)
      H8S_BARRIER();                                                                   // A(n)
      H8S_T(1)
      const bool has_next = work + wstride < wend, has_nn = work + 2 * wstride < wend;
      uint8_t *w1 = s_win + (par ^ 1) * C::kWinBytes;
      if (w5) {
        if (has_next) if (!(ABL & 32)) h8s_issue_window<K3, K4, WAUX>(a, trk.src[t1.track], t1.tx0(), t1.ty0(), lane, w1, lo);
        H8S_T(2)
        uint4 ov[4];
        if (has_prev) h8s_read_tile(lane, s_q + (par ^ 1) * 4096, ov);
        if (has_next && a.blend) h8s_issue_q2(a, trk.l2[t1.track], t1.tx0(), t1.ty0(), lane, s_q + (par ^ 1) * 4096);
        if (has_prev) h8s_store_tile(a, trk.dst[tp.track], tp.tx0(), tp.ty0(), lane, ov);
      } else {
        if (has_next) if (!(ABL & 32)) h8s_issue_window<K1, K2, WAUX>(a, trk.src[t1.track], t1.tx0(), t1.ty0(), lane, w1, lo);
        H8S_T(2)
      }
      H8S_T(3)
      H8S_BARRIER();                                                                   // B(n): window slot n & 1 is free
      H8S_T(4)
      if (has_nn) {
        uint8_t *w2 = s_win + par * C::kWinBytes;
        if (w5) { if (!(ABL & 32)) h8s_issue_window<K2, K3, WAUX>(a, trk.src[t2.track], t2.tx0(), t2.ty0(), lane, w2, lo); H8S_T(5) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K3 - K2) : "memory"); }
        else { if (!(ABL & 32)) h8s_issue_window<K0, K1, WAUX>(a, trk.src[t2.track], t2.tx0(), t2.ty0(), lane, w2, lo); H8S_T(5) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K1 - K0) : "memory"); }
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      H8S_T(6)
      if (has_next) {
        const int sx1 = 2 * t1.tx0() - 3 - a.xoff;
        if (h8s_border(sx1, a.sw)) {
          if (w5) h8s_fix_edges(w1, sx1, a.sw, lane, K2 * 64, K4 * 64); else h8s_fix_edges(w1, sx1, a.sw, lane, K0 * 64, K2 * 64);
        }
        {
          if (w5) h8s_bias_window<K2 * 1024, C::kWinBytes>(w1, lane); else h8s_bias_window<K0 * 1024, K2 * 1024>(w1, lane);
        }
      }
      tp = t0; has_prev = true; t0 = t1; t1 = t2; t2.step();
      par ^= 1;
    }
    H8S_BARRIER();                                                                     // A(last + 1): the last tile is in its slot
    if (w5) {
      uint4 ov[4];
      h8s_read_tile(lane, s_q + (par ^ 1) * 4096, ov);
      h8s_store_tile(a, trk.dst[tp.track], tp.tx0(), tp.ty0(), lane, ov);
    }
    if (DBG && lane == 0)
      for (int i = 0; i < 8; i++) a.dbg[((size_t)blockIdx.x * (kH8sCW + 2) + wave) * 8 + i] = tacc[i];
    return;
  
  }
#undef H8S_T
}

// =====================================================================================================================
// k_sep2p -- k_sep2's arithmetic (bit-identical) in the shape that made the 2:1 case fast: persistent, XCD-aware workgroups of
// 4 compute waves + 2 memory waves.  k_sep2 runs its three phases (stage window - horizontal - vertical) back to back with nothing in flight
// across them; here the source window of tile n+1 / n+2 travels by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a
// two-slot ring while the compute waves work on tile n, and the compute waves never touch global memory except for the result
// stores: the per-column taps / offsets and the per-row vertical taps of the next tile are fetched by the memory waves too and
// handed over through LDS.  Memory wave m owns the tiles i = m (mod 2) of the workgroup's list, so "my window has landed" is a
// plain vmcnt(0).  Per tile i:
//     A(i)  | compute: horizontal pass window i -> row pairs | wave (i+1)&1: wait for window i+1, edge fix-up, tables of tile i+1
//     B(i)  | compute: vertical pass + epilogue + stores     | wave i&1: issue window i+2 into slot i&1
// Every window has the plan's maximal shape [sht][swt]; rows are clamped to the frame by address, column chunks outside the frame
// are fetched from the nearest in-frame chunk and overwritten with the edge pixel after landing (source width % 4 == 0 required, so a
// 16-byte chunk is inside or outside as a whole).  The BGRA byte swap moves from staging into the horizontal pass's selectors.
// =====================================================================================================================
constexpr int kS2pMhCW = 4;       // 8 compute waves per workgroup measured 36 us against 24 for 4K -> 720p: two 10-wave workgroups do not run concurrently on a CU
constexpr int kS2pMaxReq = 48, kS2pMaxNpv = 12;       // compute waves per workgroup: template parameter CW (4 for the dot2 pass, 8 for the matrix-core pass), + 2 memory waves
struct S2pTile {
  int track, tx0, ty0;
  __device__ __forceinline__ void set(const SepArgs &a, int work) {
    const int tiles = a.tiles_x * a.tiles_y;
    track = work / tiles;
    const int tile = work - track * tiles, ty = tile / a.tiles_x;
    tx0 = (tile - ty * a.tiles_x) * kTileW; ty0 = ty * a.th;
  }
};
constexpr int kS2pPitch = kTileW + 2;   // row-pair buffer of k_sep2p: uint4 per column, two spare columns per row pair so that the matrix-core pass's four
                                        // row-pair groups (2 g pairs apart) land in different banks (a pitch of 64 uint4 put all four on the same 16)
struct S2pLds {       // byte offsets into the dynamic LDS block
  int win, p, lut, vc, hc, hdr, vcslot, hcslot;
  size_t total;
  __host__ __device__ S2pLds(int sht, int swt, int th, int npv, int nph) {
    win = sht * swt * 4;
    p = 2 * win;
    lut = p + (sht >> 1) * kS2pPitch * 16;
    vc = lut + 256;
    vcslot = th * (npv + 1) * 4;
    hc = vc + 2 * vcslot;
    hcslot = (nph + 1) * 256;
    hdr = hc + 2 * hcslot;
    total = (size_t)hdr;
  }
};
__device__ __noinline__ void s2p_issue_border(int sw, int sh, int irow, int swt, const uint8_t *src, int sx0a, int sy0, int lane, uint8_t *win, int nreq,
                                              int nchunks, uint32_t mcpr) {     // scalars, not the argument struct: a reference would force a stack copy of it
  const int cpr = swt >> 2;
  for (int k = 0; k < nreq; k++) {
    const int c = k * 64 + lane, r = (int)(((uint32_t)c * mcpr) >> 20), ch = c - r * cpr;
    int sy = sy0 + r;
    sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
    int x = sx0a + ch * 4;
    x = x < 0 ? 0 : x > sw - 4 ? sw - 4 : x;
    const uint8_t *g = src + ((uint32_t)sy * (uint32_t)irow + (uint32_t)x * 4u);
    if (c < nchunks) __builtin_amdgcn_global_load_lds((h8s_gptr)g, (h8s_lptr)(win + k * 1024), 16, 0, 0);
  }
}
__device__ __noinline__ void s2p_fix_edges(uint8_t *win, int sx0a, int sw, int swt, int sht, int lane) {
  const int cpr = swt >> 2;
  const int nlo = sx0a < 0 ? (-sx0a) >> 2 : 0;                 // chunks left of the frame (sx0a is a multiple of 4)
  int chh = (sw - sx0a) >> 2;                                  // first chunk right of the frame
  if (chh > cpr) chh = cpr;
  const int na = nlo + (cpr - chh);
  uint32_t *w = reinterpret_cast<uint32_t *>(win);
  for (int idx = lane; idx < sht * na; idx += 64) {
    const int r = idx / na, j = idx - r * na;
    const int ch = j < nlo ? j : chh + (j - nlo);
    const uint32_t v = w[r * swt + (j < nlo ? -sx0a : sw - 1 - sx0a)];
    *reinterpret_cast<uint4 *>(w + r * swt + ch * 4) = make_uint4(v, v, v, v);
  }
}

// The horizontal pass of k_sep2p on the matrix cores, for launches whose horizontal filter is the same for every output column at an integer ratio r
// (4K -> 720p, 4K -> 960x540, 2:1 with more than 8 taps): a banded-Toeplitz product as in k_half8s.  The landed window is flipped to int8 in LDS by
// its memory wave (^ 0x80).  Work item = (16 window rows, 4 output columns): A = the rows' 64 bytes starting at pixel 4 r q (16-byte aligned) per K block,
// B[k][n] = tap[pixel - (c0 + r * column)] where the source byte feeds the output channel (Q14 tap = 64 hi + lo, two fragments), KB K blocks of 16
// pixels cover c0 + 3 r + ntaps pixels.  D = 64 Dh + Dl = sum - 2^21 (the -128 bias times the tap sum 16384), so the accumulator starts at 2^21 + 64 and
// t = D >> 7 is the spec's (sum + 64) >> 7; v_cvt_pk_i16_i32 clamps and packs two rows.  Lane (g = l >> 4, n = l & 15) holds rows 4 g .. 4 g + 3 of
// (column n >> 2, channel n & 3): two 4-byte stores into the row-pair buffer the vertical pass reads.  Same bytes as the dot2 pass.
template <int KB, int CW>
__device__ __forceinline__ void s2p_hpass_mfma(const SepArgs &a, const uint8_t *win, uint32_t *s_p32, int npairs, int wave, int lane, const int4v (&bh)[KB], const int4v (&bl)[KB]) {
  const int pitch = a.swt * 4, nrb = (2 * npairs + 15) >> 4, cap = a.sht >> 1;
  const int4v zero = {0, 0, 0, 0}, cinit = {(1 << 21) + 64, (1 << 21) + 64, (1 << 21) + 64, (1 << 21) + 64};
  const int g = lane >> 4, n = lane & 15;
  // a wave owns NQ = 16 / CW neighbouring column blocks and walks the row blocks; the NQ products of a row block are independent chains (loads first,
  // then the matrix instructions interleaved): one item at a time measured no faster than the dot2 pass -- LDS and MFMA latency
  constexpr int NQ = 16 / CW;
  for (int R = 0; R < nrb; R++) {
    const int row = min(R * 16 + n, a.sht - 1);                          // A: lane supplies row (l & 15), bytes 16 (l >> 4) .. + 15 of the K block
    const uint8_t *ap = win + row * pitch + (NQ * wave) * (16 * a.mh_r) + g * 16;
    int4v av[NQ][KB], dh[NQ], dl[NQ];
#pragma unroll
    for (int u = 0; u < NQ; u++)
#pragma unroll
      for (int kb = 0; kb < KB; kb++) av[u][kb] = *reinterpret_cast<const int4v *>(ap + u * (16 * a.mh_r) + kb * 64);
#pragma unroll
    for (int u = 0; u < NQ; u++) { dh[u] = zero; dl[u] = cinit; }
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
      for (int u = 0; u < NQ; u++) {
        dh[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[u][kb], bh[kb], dh[u], 0, 0, 0);
        dl[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[u][kb], bl[kb], dl[u], 0, 0, 0);
      }
    const int pr = R * 8 + 2 * g;                                        // row pairs (4 g, 4 g + 1) and (4 g + 2, 4 g + 3) of this block
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int q = NQ * wave + u;
      const int t0 = ((dh[u].x << 6) + dl[u].x) >> 7, t1 = ((dh[u].y << 6) + dl[u].y) >> 7, t2 = ((dh[u].z << 6) + dl[u].z) >> 7, t3 = ((dh[u].w << 6) + dl[u].w) >> 7;
      uint32_t *d = s_p32 + ((size_t)pr * kS2pPitch + 4 * q + (n >> 2)) * 4 + (n & 3);
      if (pr < cap) d[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(t0, t1));
      if (pr + 1 < cap) d[kS2pPitch * 4] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(t2, t3));
    }
  }
}

template <int NPH, int KB, int CW>
__device__ __forceinline__ void s2p_compute(const SepArgs &a, const SepTracks &trk, uint8_t *smem, const S2pLds &L, int wave, int lane,
                                            int work, int wend, int wstride) {
  uint4 *s_p = reinterpret_cast<uint4 *>(smem + L.p);
  const uint8_t *s_lut = smem + L.lut;
  uint32_t bf = a.bf, nbf = a.nbf;
  if (a.blend && a.bf_d) { bf = (uint32_t)a.bf_d[0] & 0xFF; nbf = 0xFF - bf; }
  // selector of channel c: bytes [b, 0, 4 + b, 0] with b = the source byte that feeds output channel c (identity or the R <-> B swap)
  uint32_t sel[4];
#pragma unroll
  for (int c = 0; c < 4; c++) sel[c] = 0x0C040C00u + ((a.src_sel >> (8 * c)) & 3u) * 0x00010001u;
  int par = 0;
  int4v bh[KB ? KB : 1], bl[KB ? KB : 1];
  if (KB) {
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      bh[kb] = reinterpret_cast<const int4v *>(a.bfrag)[(kb * 2 + 0) * 64 + lane];
      bl[kb] = reinterpret_cast<const int4v *>(a.bfrag)[(kb * 2 + 1) * 64 + lane];
    }
  }
  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = a.dbg ? __builtin_amdgcn_s_memtime() : 0;
#define S2P_T(i) if (a.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
  for (; work < wend; work += wstride, par ^= 1) {
    S2pTile t;
    t.set(a, work);
    const int tw = min(kTileW, a.dw - t.tx0), thh = min(a.th, a.dh - t.ty0);
    uint8_t *dst = trk.dst[t.track];
    const uint8_t *l2 = trk.l2[t.track];
    // the second layer's pixels of the wave's first batch of output rows are requested here, a whole horizontal pass before they are blended in
    // (requested next to their use they cost a full memory latency per tile: 5,800 cycles of vertical pass against 2,650 without the blend)
    uint32_t qpre[4] = {0, 0, 0, 0};
    if (a.blend) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ly = wave + r * CW;
        if (ly < thh && lane < tw) qpre[r] = reinterpret_cast<const uint32_t *>(l2 + (size_t)(t.ty0 + ly) * a.irow2)[t.tx0 + lane];
      }
    }
    S2P_T(3)
    H8S_BARRIER();                                                                   // A(i): window i, tables i in place; row pairs free
    S2P_T(0)
    const uint32_t *s_src = reinterpret_cast<const uint32_t *>(smem + par * L.win);
    const uint32_t *s_hc = reinterpret_cast<const uint32_t *>(smem + L.hc + par * L.hcslot);
    const uint32_t *s_vc = reinterpret_cast<const uint32_t *>(smem + L.vc + par * L.vcslot);
    // the tables arrive raw: hpos of the lane's column, vpos of the rows; the window origin is (hpos[first column] & ~3, vpos[first row] & ~1)
    const int sx0a = (int)s_hc[0] & ~3, sy0 = (int)s_vc[0] & ~1;
    const int npairs = (((int)s_vc[(thh - 1) * (a.npv + 1)] + a.ntv - sy0 + 1) & ~1) >> 1;
    if (KB) {
      s2p_hpass_mfma<(KB ? KB : 1), CW>(a, reinterpret_cast<const uint8_t *>(s_src), reinterpret_cast<uint32_t *>(s_p), npairs, wave, lane, bh, bl);
    } else {
    // ---- horizontal pass: lane = output column, a wave takes window row pairs ----
    const int hoff = (int)s_hc[lane] - sx0a;
    short2v hc2[NPH];
#pragma unroll
    for (int j = 0; j < NPH; j++) hc2[j] = __builtin_bit_cast(short2v, s_hc[(1 + j) * 64 + lane]);
    for (int k = wave; k < npairs; k += CW) {
      const uint32_t *r0 = s_src + (2 * k) * a.swt + hoff, *r1 = r0 + a.swt;
      int e0 = a.hround, e1 = a.hround, e2 = a.hround, e3 = a.hround, o0 = a.hround, o1 = a.hround, o2 = a.hround, o3 = a.hround;
#pragma unroll
      for (int j = 0; j < NPH; j++) {
        const uint32_t p0 = r0[2 * j], p1 = r0[2 * j + 1], q0 = r1[2 * j], q1 = r1[2 * j + 1];
        e0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, sel[0])), hc2[j], e0, false);
        e1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, sel[1])), hc2[j], e1, false);
        e2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, sel[2])), hc2[j], e2, false);
        e3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(p1, p0, sel[3])), hc2[j], e3, false);
        o0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, sel[0])), hc2[j], o0, false);
        o1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, sel[1])), hc2[j], o1, false);
        o2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, sel[2])), hc2[j], o2, false);
        o3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(q1, q0, sel[3])), hc2[j], o3, false);
      }
      uint4 pk;
      pk.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e0 >> a.hshift, o0 >> a.hshift));
      pk.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e1 >> a.hshift, o1 >> a.hshift));
      pk.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e2 >> a.hshift, o2 >> a.hshift));
      pk.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(e3 >> a.hshift, o3 >> a.hshift));
      s_p[k * kS2pPitch + lane] = pk;
    }
    }
    S2P_T(1)
    H8S_BARRIER();                                                                   // B(i): row pairs complete, window slot i & 1 free
    S2P_T(2)
    // ---- vertical pass + epilogue: wave = output row, lane = output column ----
    // NB output rows of the wave at a time (4 when the tile gives a wave more than two rows, else 2): their row-pair / tap-pair reads are requested
    // together, so the LDS latency of the (run-time long) pair loop is paid once per batch (one row at a time measured 900 cycles per row, mostly waiting)
    auto vrows = [&](auto nb_tag) {
      constexpr int NB = decltype(nb_tag)::value;
      for (int lyb = wave; lyb < thh; lyb += NB * CW) {
        const uint32_t *vc2[NB];
        const uint4 *col[NB];
        int acc[NB][4];
        if (a.blend && lyb != wave) {                                 // later batches of a tall tile: requested before the taps are read
#pragma unroll
          for (int r = 0; r < NB; r++) {
            const int ly = lyb + r * CW;
            if (ly < thh && lane < tw) qpre[r] = reinterpret_cast<const uint32_t *>(l2 + (size_t)(t.ty0 + ly) * a.irow2)[t.tx0 + lane];
          }
        }
#pragma unroll
        for (int r = 0; r < NB; r++) {
          const int ly = min(lyb + r * CW, thh - 1);                  // rows past the tile repeat the last one (not stored)
          vc2[r] = s_vc + ly * (a.npv + 1);                           // the same address for every lane (broadcast)
          acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = a.vround;
        }
#pragma unroll
        for (int r = 0; r < NB; r++) col[r] = s_p + (((int)vc2[r][0] >> 1) - (sy0 >> 1)) * kS2pPitch + lane;       // first row pair, window relative
        for (int j = 0; j < a.npv; j++) {
          uint4 tt[NB];
          uint32_t cf[NB];
#pragma unroll
          for (int r = 0; r < NB; r++) { tt[r] = col[r][j * kS2pPitch]; cf[r] = vc2[r][1 + j]; }
#pragma unroll
          for (int r = 0; r < NB; r++) {
            const short2v c2 = __builtin_bit_cast(short2v, cf[r]);
            acc[r][0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, tt[r].x), c2, acc[r][0], false);
            acc[r][1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, tt[r].y), c2, acc[r][1], false);
            acc[r][2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, tt[r].z), c2, acc[r][2], false);
            acc[r][3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, tt[r].w), c2, acc[r][3], false);
          }
        }
#pragma unroll
        for (int r = 0; r < NB; r++) {
          const int ly = lyb + r * CW, oy = t.ty0 + ly;
          if (ly < thh && lane < tw) {
            uint32_t p = a.vshift == 21 ? pack_sat_shr21(acc[r][0], acc[r][1], acc[r][2], acc[r][3])
                                        : (uint32_t)clamp255(acc[r][0] >> a.vshift) | ((uint32_t)clamp255(acc[r][1] >> a.vshift) << 8) |
                                              ((uint32_t)clamp255(acc[r][2] >> a.vshift) << 16) | ((uint32_t)clamp255(acc[r][3] >> a.vshift) << 24);
            if (a.blend) p = chroma_rgba(p, qpre[r], bf, nbf);
            if (a.use_lut) p = lut3_rgba(s_lut, p);
            __builtin_nontemporal_store(p, reinterpret_cast<uint32_t *>(dst + (size_t)oy * a.orow) + t.tx0 + lane);      // written once, not read back
          }
        }
      }
    };
    if (a.th > 2 * CW) vrows(std::integral_constant<int, 4>()); else vrows(std::integral_constant<int, 2>());
  }
  S2P_T(3)
  if (a.dbg && lane == 0)          // [0] waiting at A, [1] horizontal pass, [2] waiting at B, [3] vertical pass + stores
    for (int i = 0; i < 4; i++) a.dbg[((size_t)blockIdx.x * (CW + 2) + wave) * 8 + i] = tacc[i];
#undef S2P_T
}

template <int NPH, int KB, int CW>
__global__ __launch_bounds__((CW + 2) * 64, (CW + 2) / 2) void k_sep2p(SepArgs a, SepTracks trk, Lut8 lut) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const S2pLds L(a.sht, a.swt, a.th, a.npv, NPH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.use_lut) stage_lut(smem + L.lut, lut);
  const int nwork = a.tiles_x * a.tiles_y * a.ntracks;
  const int xcd = blockIdx.x & 7, wstride = (int)(gridDim.x >> 3), chunk = (nwork + 7) >> 3;
  const int wend = min((xcd + 1) * chunk, nwork);
  int work = xcd * chunk + (int)(blockIdx.x >> 3);
  if (work >= wend) return;                                     // workgroup-uniform
  if (wave < CW) { s2p_compute<NPH, KB, CW>(a, trk, smem, L, wave, lane, work, wend, wstride); return; }

  // ------------------------------------------------ memory waves ------------------------------------------------
  __builtin_amdgcn_s_setprio(3);
  const int m = wave - CW;
  const int cpr = a.swt >> 2, nchunks = a.sht * cpr, nreq = (nchunks + 63) >> 6;
  const uint32_t mcpr = ((1u << 20) + (uint32_t)cpr - 1u) / (uint32_t)cpr;          // c / cpr == (c * mcpr) >> 20 for c < 3072, cpr >= 4
  const uint32_t mnpv = ((1u << 16) + (uint32_t)a.npv) / (uint32_t)(a.npv + 1);     // i / (npv + 1) likewise for i < 256
  // chunk 64 k + lane of an unclamped window sits at byte offset r * irow + 16 ch; stepping k by one moves (r, ch) by (64 / cpr, 64 % cpr)
  const uint32_t r_l = ((uint32_t)lane * mcpr) >> 20, ch_l = (uint32_t)lane - r_l * (uint32_t)cpr;
  const uint32_t off_l = r_l * (uint32_t)a.irow + ch_l * 16u;
  const uint32_t dr = 64u / (uint32_t)cpr, dch = 64u - dr * (uint32_t)cpr;
  const uint32_t doff = dr * (uint32_t)a.irow + dch * 16u, dwrap = (uint32_t)a.irow - (uint32_t)cpr * 16u;
  // Everything a tile needs travels by LDS-DMA -- the window (16 bytes per lane and request) and its tables as RAW words (4 bytes per lane and
  // request: hpos / tap pairs of the lane's column, vpos / tap pairs of the rows) -- so no load result lives in a register between the issue and
  // the barrier that follows and the compiler has nothing to wait for there; the compute waves derive offsets from the raw words.  The window
  // origin of the NEXT tile this wave will issue is fetched one step ahead (gx, gy).
  int sx0a_f = 0;
  auto geom = [&](int wk, int &gx, int &gy) {
    S2pTile t;
    t.set(a, wk);
    gx = a.hpos[t.tx0]; gy = a.vpos[t.ty0];
  };
  auto issue = [&](int wk, int slot, int gx, int gy) {
    S2pTile t;
    t.set(a, wk);
    const int tw = min(kTileW, a.dw - t.tx0), thh = min(a.th, a.dh - t.ty0);
    const int sx0a = gx & ~3, sy0 = gy & ~1;
    uint8_t *win = smem + slot * L.win;
    const uint8_t *src = trk.src[t.track];
    const bool inside = sx0a >= 0 && sx0a + a.swt <= a.sw && sy0 >= 0 && sy0 + a.sht <= a.sh;
    if (inside) {
      const uint8_t *base = src + ((uint32_t)sy0 * (uint32_t)a.irow + (uint32_t)sx0a * 4u);
      uint32_t off = off_l, ch = ch_l;
      for (int k = 0; k < nreq; k++) {
        if (k * 64 + lane < nchunks) __builtin_amdgcn_global_load_lds((h8s_gptr)(base + off), (h8s_lptr)(win + k * 1024), 16, 0, 0);
        ch += dch; off += doff;
        if (ch >= (uint32_t)cpr) { ch -= (uint32_t)cpr; off += dwrap; }
      }
    } else s2p_issue_border(a.sw, a.sh, a.irow, a.swt, src, sx0a, sy0, lane, win, nreq, nchunks, mcpr);
    const int ox = t.tx0 + (lane < tw ? lane : tw - 1);
    uint8_t *hc = smem + L.hc + slot * L.hcslot, *vc = smem + L.vc + slot * L.vcslot;
    __builtin_amdgcn_global_load_lds((h8s_gptr)(a.hpos + ox), (h8s_lptr)hc, 4, 0, 0);
#pragma unroll
    for (int j = 0; j < NPH; j++) __builtin_amdgcn_global_load_lds((h8s_gptr)(a.hco2 + ((size_t)ox * NPH + j)), (h8s_lptr)(hc + (1 + j) * 256), 4, 0, 0);
    const int nvc = thh * (a.npv + 1);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q * 64 < nvc) {                                        // wave-uniform
        const int i = q * 64 + lane;
        const int ly = (int)(((uint32_t)i * mnpv) >> 16), j = i - ly * (a.npv + 1);
        const uint32_t *g = j ? a.vco2 + ((size_t)(t.ty0 + ly) * a.npv + (j - 1)) : reinterpret_cast<const uint32_t *>(a.vpos + (t.ty0 + ly));
        if (i < nvc) __builtin_amdgcn_global_load_lds((h8s_gptr)g, (h8s_lptr)(vc + q * 256), 4, 0, 0);
      }
    }
    sx0a_f = sx0a;
  };
  auto land = [&](int slot) {                                    // my window in flight has to be complete before the barrier that publishes it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sx0a_f < 0 || sx0a_f + a.swt > a.sw) s2p_fix_edges(smem + slot * L.win, sx0a_f, a.sw, a.swt, a.sht, lane);
    if (KB) {                                                    // uint8 -> int8 for the matrix cores
      uint4 *w = reinterpret_cast<uint4 *>(smem + slot * L.win);
      for (int i = lane; i < (L.win >> 4); i += 64) {
        uint4 v = w[i];
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
        w[i] = v;
      }
    }
  };

  // wave m starts on tile m of the list; wave 0 publishes tile 0 before A(0)
  const int first = work + m * wstride;
  int gx = 0, gy = 0;
  if (first < wend) {
    geom(first, gx, gy);
    issue(first, m, gx, gy);
    if (first + 2 * wstride < wend) geom(first + 2 * wstride, gx, gy);
  }
  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = a.dbg ? __builtin_amdgcn_s_memtime() : 0;
#define S2P_T(i) if (a.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
  if (m == 0) land(0);
  S2P_T(1)
  int i = 0;
  for (; work < wend; work += wstride, i++) {
    H8S_BARRIER();                                                                   // A(i)
    S2P_T(0)
    if (((i + 1) & 1) == m && work + wstride < wend) land((i + 1) & 1);
    S2P_T(1)
    H8S_BARRIER();                                                                   // B(i)
    S2P_T(2)
    if ((i & 1) == m && work + 2 * wstride < wend) {
      issue(work + 2 * wstride, i & 1, gx, gy);
      if (work + 4 * wstride < wend) geom(work + 4 * wstride, gx, gy);
    }
    S2P_T(3)
  }
  if (a.dbg && lane == 0)          // [0] waiting at A, [1] waiting for the window + fix-up, [2] waiting at B, [3] issuing
    for (int k = 0; k < 4; k++) a.dbg[((size_t)blockIdx.x * (CW + 2) + wave) * 8 + k] = tacc[k];
#undef S2P_T
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
  std::vector<int16_t> hco;         // host copy of the taps
  int uniform2 = 0;                 // exact 2:1, the same <= 8 taps for every output, centred like the 8-tap case (taps8 = zero padded to pos[i] = 2i - 3), taps fit 64*int8 + 6 bits
  int16_t taps8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t *co2h = nullptr, *co2v = nullptr;   // k_sep2: taps as packed pairs, [dst][nph] resp. [dst][npv] (see SepArgs)
  int nph = 0, npv = 0;
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
    b.hco = co;
    if (b.nt <= 8 && !(b.nt & 1) && srcn == 2 * dstn && kernel != 100) {
      // k_half8s takes any uniform 2:1 filter that embeds into its 8-tap window: bicubic (8 taps) and, zero padded on both sides, bilinear (4 taps)
      const int pad = (8 - b.nt) / 2;
      b.uniform2 = 1;
      for (int i = 0; i < dstn && b.uniform2; i++) {
        if (b.hpos[i] != 2 * i - 3 + pad) b.uniform2 = 0;
        for (int j = 0; j < b.nt; j++) {
          const int c = co[(size_t)i * b.nt + j];
          if (c != co[j] || (c >> 6) < -128 || (c >> 6) > 127) b.uniform2 = 0;
        }
      }
      if (b.uniform2) for (int j = 0; j < b.nt; j++) b.taps8[pad + j] = co[j];
    }
    {
      b.nph = (b.nt + 1) / 2; b.npv = b.nt / 2 + 1;
      std::vector<uint32_t> h2((size_t)dstn * b.nph), v2((size_t)dstn * b.npv);
      auto pk = [](int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); };
      for (int i = 0; i < dstn; i++) {
        const int16_t *c = co.data() + (size_t)i * b.nt;
        for (int j = 0; j < b.nph; j++) h2[(size_t)i * b.nph + j] = pk(c[2 * j], 2 * j + 1 < b.nt ? c[2 * j + 1] : 0);
        const int odd = b.hpos[i] & 1;                                  // first tap on the second row of its (even-aligned) pair
        for (int j = 0; j < b.npv; j++) {
          const int t0 = 2 * j - odd, t1 = t0 + 1;
          v2[(size_t)i * b.npv + j] = pk(t0 >= 0 && t0 < b.nt ? c[t0] : 0, t1 >= 0 && t1 < b.nt ? c[t1] : 0);
        }
      }
      LGPU_HIP(hipMalloc((void **)&b.co2h, sizeof(uint32_t) * h2.size()));
      LGPU_HIP(hipMalloc((void **)&b.co2v, sizeof(uint32_t) * v2.size()));
      LGPU_HIP(hipMemcpy(b.co2h, h2.data(), sizeof(uint32_t) * h2.size(), hipMemcpyHostToDevice));
      LGPU_HIP(hipMemcpy(b.co2v, v2.data(), sizeof(uint32_t) * v2.size(), hipMemcpyHostToDevice));
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

// the chroma blend's alpha scalers (lgpu_alpha_scalers, proven while built) as one device table per device: {K2, K1} per layer-2 alpha
static std::mutex g_ks_mu;
static std::map<int, uint2 *> g_ks;
int get_kscale(const uint2 **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ks_mu);
  auto it = g_ks.find(dev);
  if (it == g_ks.end()) {
    uint32_t k2[256], k1[256];
    int rc = lgpu_alpha_scalers(k2, k1);           // simple_blend.c:137-145 as (c * K) >> 16
    if (rc) return rc;
    uint2 ks[256];
    for (int al = 0; al < 256; al++) ks[al] = make_uint2(k2[al], k1[al]);
    uint2 *d = nullptr;
    LGPU_HIP(hipMalloc((void **)&d, sizeof ks));
    LGPU_HIP(hipMemcpy(d, ks, sizeof ks, hipMemcpyHostToDevice));
    it = g_ks.emplace(dev, d).first;
  }
  *out = it->second;
  return LGPU_OK;
}

// ---- k_half8s host side --------------------------------------------------------------------------------------
struct Half8Const {
  int4v *bfrag = nullptr;     // device [2][64]
};
static std::mutex g_h8_mu;
static std::map<std::pair<int, std::vector<int16_t>>, Half8Const> g_h8;   // (device, 8 taps + swap flag)

static int get_half8_const(const int16_t taps[8], int swap_rb, int xoff, const Half8Const **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_h8_mu);
  std::vector<int16_t> kv(taps, taps + 8);
  kv.push_back((int16_t)(swap_rb | (xoff << 1)));
  auto key = std::make_pair(dev, kv);
  auto it = g_h8.find(key);
  if (it == g_h8.end()) {
    Half8Const c;
    // B fragment of v_mfma_i32_16x16x64_i8: lane l supplies B[k = 16 * (l >> 4) + e][n = l & 15], e = 0..15
    // (A uses the same k numbering; C/D: row = 4 * (l >> 4) + reg, col = l & 15 -- tools/mfma_probe.hip).
    // k = (window pixel 0..15, source byte 0..3), n = (output column 0..3, output channel 0..3):
    // B = tap[pixel - 2 * column] where the source byte feeds that output channel (bytes 0 / 2 trade places for BGRA sources)
    int8_t frag[2][64][16];
    for (int l = 0; l < 64; l++)
      for (int e = 0; e < 16; e++) {
        const int k = 16 * (l >> 4) + e, n = l & 15;
        const int px = k >> 2, sbyte = k & 3, col = n >> 2, och = n & 3;
        const int want = swap_rb ? (och == 0 ? 2 : och == 2 ? 0 : och) : och;
        const int j = px - 2 * col - xoff;
        const int tap = (sbyte == want && j >= 0 && j < 8) ? taps[j] : 0;
        frag[0][l][e] = (int8_t)(tap >> 6);        // tap = 64 * hi + lo, lo in [0, 63]
        frag[1][l][e] = (int8_t)(2 * (tap & 63));  // stored doubled (see k_half8s)
      }
    LGPU_HIP(hipMalloc((void **)&c.bfrag, sizeof frag));
    LGPU_HIP(hipMemcpy(c.bfrag, frag, sizeof frag, hipMemcpyHostToDevice));
    it = g_h8.emplace(key, c).first;
  }
  *out = &it->second;
  return LGPU_OK;
}

static int g_h8s_opt = 0;
#ifdef LGPU_H8S_AB
extern "C" int lgpu_h8s_set_opt(int opt) { g_h8s_opt = opt; return LGPU_OK; }   // A / B builds only (tools/ab_h8s.py), not part of the ABI
#endif
// returns LGPU_OK and launches when the fast path applies; LGPU_E_UNSUPPORTED when it does not
static int try_half8(const Bank *hb, const Bank *vb, int sw, int sh, int irow, int dw, int dh, int orow, int swap_rb, int blend,
                     int irow2, uint32_t bf, const int32_t *bf_d, int use_lut, const SepTracks &t, int ntracks, const Lut8 &l,
                     hipStream_t st, int nt_out = 1) {
  const bool disabled = tune_on(TUNE_DISABLE_HALF8);
  if (disabled || !hb->uniform2 || !vb->uniform2) return LGPU_E_UNSUPPORTED;
  if ((irow & 3) || (orow & 3)) return LGPU_E_UNSUPPORTED;
  {   // operand ranges of the int8 / int16 forms used by the kernel
    int hsum = 0, vsum = 0, hneg = 0;
    for (int j = 0; j < 8; j++) {
      const int th_ = hb->taps8[j], tv = vb->taps8[j];
      if (th_ < -8192 || th_ >= 8192) return LGPU_E_UNSUPPORTED;       // tap >> 6 must fit int8
      hsum += th_; vsum += tv; if (th_ < 0) hneg -= th_;
    }
    if (hsum != 16384 || vsum != 16384 || hneg > 4096) return LGPU_E_UNSUPPORTED;   // |t - 16384| <= 16384 + 2 * hneg + 1 < 2^15
  }
  // 16-byte aligned requests when every source row starts 16-byte aligned: the window then starts at source x = 2 * tx0 - 4
  // (a multiple of four pixels) and the tap matrix is shifted by one pixel instead
  int xoff = ((irow & 15) == 0) ? 1 : 0;
  for (int i = 0; i < ntracks; i++) if ((uintptr_t)t.src[i] & 15) xoff = 0;
  if (g_h8s_opt & 256) xoff = 0;                   // A / B builds
  const Half8Const *hc;
  int rc = get_half8_const(hb->taps8, swap_rb, xoff, &hc);
  if (rc) return rc;
  Half8Args a;
  if ((rc = get_kscale(&a.kscale))) return rc;
  a.xoff = xoff;
  a.sw = sw; a.sh = sh; a.irow = irow; a.dw = dw; a.dh = dh; a.orow = orow;
  a.bfrag = hc->bfrag;
  for (int k = 0; k < 4; k++) a.vc[k] = (uint32_t)(uint16_t)vb->taps8[2 * k] | ((uint32_t)(uint16_t)vb->taps8[2 * k + 1] << 16);
  a.swap_rb = swap_rb; a.blend = blend; a.irow2 = irow2; a.bf = bf; a.nbf = 0xFF - bf; a.bf_d = bf_d; a.use_lut = use_lut;
  a.ntracks = ntracks;
  a.dbg = nullptr;
  a.nt_out = nt_out;
  // persistent grid: as many workgroups as stay resident (two per CU by LDS), each walks its XCD's share of the work list
  const int g_cus = device_cus();
  a.tiles_x = (dw + kTileW - 1) / kTileW; a.tiles_y = (dh + H8S::kTileH - 1) / H8S::kTileH;
  const int nwork = a.tiles_x * a.tiles_y * ntracks;
  const int mode = g_h8s_opt & 255;                // A / B builds (tools/ab_h8s.py): ablations of k_half8s
  const size_t lds = H8SL::kLds;
  int grid = g_cus * (int)(160 * 1024 / lds);
  // multi-GPU hosts: leave a few workgroup slots (one per XCD) free, so that a kernel with a large LDS footprint enqueued on another stream -- RCCL's
  // broadcast of the next parameter block -- finds a CU while this persistent kernel runs (tools/corun_probe.py, profiles/r02/corun_probe.txt)
  { const int n = tune(TUNE_CHAIN_SPARE_WGS); if (n > 0 && n < grid / 2) grid -= n; }
  if (grid > nwork) grid = nwork;
  grid = (grid + 7) & ~7;                      // whole workgroups per XCD
  // An XCD's workgroups walk its share of the tile list with a stride of grid / 8 tiles.  When that stride shares a large factor with the number of tiles per
  // row the workgroups march down fixed tile columns in lockstep and the launch takes up to 1.7x as long (measured: 4096 x 2160 sources, 32 tiles per row,
  // stride 64: 283 us against 180 with stride 62; 24 tiles per row: 185 against 161; profiles/r02/chain_stride.md): take the largest stride whose gcd with
  // the row length is at most 2.
  if (grid >= 64) {
    auto gcd = [](int x, int y) { while (y) { const int t_ = x % y; x = y; y = t_; } return x; };
    int w = grid >> 3;
    for (int k = 0; k < 6 && w - k >= 8; k++)
      if (gcd(w - k, a.tiles_x) <= 2) { w -= k; break; }
    grid = w << 3;
  }
#define H8S_LAUNCH(DBG_, ABL_)                                                                                          \
  do {                                                                                                                  \
    LGPU_HIP(hipFuncSetAttribute((const void *)k_half8s<DBG_, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_half8s<DBG_, ABL_>), dim3((unsigned)grid), dim3(kH8sThreads), lds, st, a, t, l);              \
  } while (0)
#ifdef LGPU_PROFILING
  const bool dbg_s = tune_on(TUNE_PHASE_PROFILE);      // LGPU_PLAN_DEBUG only prints: the run being diagnosed keeps its timing and captures into graphs
  if (dbg_s) {     // in-kernel phase profile: s_memtime ticks between fixed points of the tile loop, per wave
    static unsigned long long *g_dbg_s = nullptr;
    constexpr int nw = kH8sCW + 2;
    if (!g_dbg_s) LGPU_HIP(hipMalloc((void **)&g_dbg_s, sizeof(unsigned long long) * 8 * nw * 4096));
    a.dbg = g_dbg_s;
    H8S_LAUNCH(1, 0);
    static int dumps = 0;
    if (dumps++ < 3) {
      LGPU_HIP(hipStreamSynchronize(st));
      std::vector<unsigned long long> h((size_t)grid * 8 * nw);
      LGPU_HIP(hipMemcpy(h.data(), g_dbg_s, h.size() * 8, hipMemcpyDeviceToHost));
      double c[8] = {0}, m[8] = {0}, m5[8] = {0};
      for (int b = 0; b < grid; b++)
        for (int i = 0; i < 8; i++) {
          for (int w = 0; w < kH8sCW; w++) c[i] += (double)h[((size_t)b * nw + w) * 8 + i] / kH8sCW;
          m[i] += (double)h[((size_t)b * nw + kH8sCW) * 8 + i]; m5[i] += (double)h[((size_t)b * nw + kH8sCW + 1) * 8 + i];
        }
      const double it = (double)nwork;
      fprintf(stderr, "[h8s compute wave, ticks/tile] V+epilogue %.0f | wait A %.0f | H %.0f | wait B %.0f\n", c[0] / it, c[1] / it, c[2] / it, c[3] / it);
      fprintf(stderr, "[h8s memory wave 4, ticks/tile] fix+bias %.0f | wait A %.0f | dma window b %.0f | - %.0f | wait B %.0f | dma window a %.0f | land %.0f\n",
              m[0] / it, m[1] / it, m[2] / it, m[3] / it, m[4] / it, m[5] / it, m[6] / it);
      fprintf(stderr, "[h8s memory wave 5, ticks/tile] fix+bias %.0f | wait A %.0f | dma window b %.0f | read + dma l2 + store %.0f | wait B %.0f | dma window a %.0f | land %.0f\n",
              m5[0] / it, m5[1] / it, m5[2] / it, m5[3] / it, m5[4] / it, m5[5] / it, m5[6] / it);
    }
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
#endif
  switch (mode) {
#ifdef LGPU_H8S_AB
    case 16: H8S_LAUNCH(0, 16); break;      // no compute: memory waves and barriers only
    case 32: H8S_LAUNCH(0, 32); break;      // no source window DMA: compute, layer 2 and stores only
    case 2: H8S_LAUNCH(0, 128); break;      // wave 4 issues its whole share of a window right after B
    case 3: H8S_LAUNCH(0, 192); break;      // the same with an 11 / 10 split of the DMA rounds
#endif
    default: H8S_LAUNCH(0, 0); break;
  }
#undef H8S_LAUNCH
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}


struct SepPlan {
  SepArgs a;
  size_t lds;
  dim3 grid;
  int variant;   // 0 generic taps, 1 = (8,8), 2 = (5,5), 3 = (4,4), 4 = (2,2), 5 = (6,6); 100 + nph: k_sep2
  // k_sep2p (persistent workgroups, prefetched windows) for the same taps: its own tile height / window rows; taken at launch when the source rows
  // are 16-byte aligned
  bool pers = false;
  int p_th = 0, p_sht = 0, p_tiles_y = 0;
  size_t p_lds = 0;
  // ... and its horizontal pass on the matrix cores when every output column has the same taps at an integer ratio (mh_r > 0)
  int mh_r = 0, mh_c0 = 0, mh_kb = 0, mh_nt = 0;
  int16_t mh_taps[32];
};

static int plan_sep(const Bank *hb, const Bank *vb, int sw, int sh, int irow, int dw, int dh, int orow, int ntracks,
                    int hround, int hshift, int vround, int vshift, SepPlan *p) {
  SepArgs &a = p->a;
  a.dbg = nullptr; a.ntracks = ntracks; a.hco2 = a.vco2 = nullptr; a.nph = a.npv = 0; a.bfrag = nullptr; a.mh_r = 0;
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
  // k_sep2 (dot2 on both passes): tap pair counts with an instantiation, windows that leave two workgroups per CU
  const bool no_sep2 = tune_on(TUNE_NO_SEP2);
  const int nph = hb->nph;
  if (!no_sep2 && (nph == 1 || nph == 2 || nph == 3 || nph == 4 || nph == 5 || nph == 6 || nph == 7 || nph == 8 || nph == 10 || nph == 12) && vb->nt <= 64) {
    SepArgs b = a;
    b.hco2 = hb->co2h; b.vco2 = vb->co2v; b.nph = nph; b.npv = vb->npv;
    // the padded last tap pair reads one pixel further, rows come in even-aligned pairs
    int swt2 = 0;
    for (int t0 = 0; t0 < dw; t0 += kTileW) {
      const int t1 = (t0 + kTileW < dw ? t0 + kTileW : dw) - 1;
      const int span = hb->hpos[t1] + 2 * nph - (hb->hpos[t0] & ~3);
      if (span > swt2) swt2 = span;
    }
    b.swt = (swt2 + 3) & ~3;
    auto window_rows = [&](int th2) {
      int sht2 = 0;
      for (int t0 = 0; t0 < dh; t0 += th2) {
        const int t1 = (t0 + th2 < dh ? t0 + th2 : dh) - 1;
        const int span = ((vb->hpos[t1] + vb->nt - (vb->hpos[t0] & ~1)) + 1) & ~1;
        if (span > sht2) sht2 = span;
      }
      return sht2 + 2;       // the parity-shifted tap layout may touch one row pair past the last tap
    };
    // tile height: 16 rows, 32 when enlarging (small windows: taller tiles amortise a workgroup's three phases; measured 29.8 against 33.9 us for
    // 1080p -> 4K, profiles/r02/resize_ratios.md)
    size_t sep2_lds_cap = 80 * 1024;
    { const int v = tune(TUNE_SEP2_LDS_KB); if (v >= 8 && v <= 160) sep2_lds_cap = (size_t)v * 1024; }      // tuning probe
    for (int th2 = dh > sh ? 32 : 16;; th2 >>= 1) {
      const int sht2 = window_rows(th2);
      const size_t lds2 = (size_t)sht2 * b.swt * 4 + (size_t)(sht2 >> 1) * kTileW * 16 + 256 + (size_t)th2 * (vb->npv + 1) * 4;
      if (lds2 <= sep2_lds_cap || th2 == 1) {
        if (lds2 <= 160 * 1024) {
          b.th = th2; b.sht = sht2;
          b.tiles_y = (dh + th2 - 1) / th2;
          a = b;
          p->lds = lds2;
          p->grid = dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)ntracks, 1);
          p->variant = 100 + nph;
        }
        break;
      }
    }
    // the horizontal pass as a matrix product: same taps for every column, integer ratio, taps that split into int8 hi / 6-bit lo, at most two K blocks
    // (decided before the tile height: such a launch carries no per-column tap tables in LDS)
    p->mh_r = 0;
    const bool no_sep2p = tune_on(TUNE_NO_SEP2P);
    const bool no_mh = tune_on(TUNE_NO_SEP2P_MFMA);
    const bool pers_ok = p->variant >= 100 && !no_sep2p && (sw & 3) == 0;
    if (pers_ok && !no_mh && dw >= 1 && sw % dw == 0 && sw / dw >= 2 && hb->nt <= 32 && hround == 64 && hshift == 7) {
      const int r = sw / dw, c0 = hb->hpos[0] & 3, nt = hb->nt;
      bool ok = (c0 + 3 * r + nt) <= 32;
      for (int c = 0; c < dw && ok; c++) {
        if (hb->hpos[c] != hb->hpos[0] + r * c) ok = false;
        for (int j = 0; j < nt && ok; j++) {
          const int tp = hb->hco[(size_t)c * nt + j];
          if (tp != hb->hco[j] || tp < -8192 || tp >= 8192) ok = false;
        }
      }
      if (ok) {
        p->mh_r = r; p->mh_c0 = c0; p->mh_nt = nt; p->mh_kb = (c0 + 3 * r + nt + 15) / 16;
        for (int j = 0; j < 32; j++) p->mh_taps[j] = j < nt ? hb->hco[j] : 0;
      }
    }
    // k_sep2p: two window slots + double-buffered tables must leave two workgroups per CU; a window is at most kS2pMaxReq DMA requests
    const int th_force = tune(TUNE_SEP2P_TH) > 0 ? tune(TUNE_SEP2P_TH) : 0;
    if (pers_ok) {
      for (int th2 = th_force ? th_force : dh > sh ? 32 : 16; th2 >= 1; th2 >>= 1) {
        const int sht2 = window_rows(th2);
        const S2pLds L(sht2, a.swt, th2, vb->npv, p->mh_r ? 1 : nph);
        if (L.total <= 80 * 1024 && sht2 * (a.swt >> 2) <= kS2pMaxReq * 64 && th2 * (vb->npv + 1) <= 256 && vb->npv <= kS2pMaxNpv) {
          p->pers = true; p->p_th = th2; p->p_sht = sht2; p->p_tiles_y = (dh + th2 - 1) / th2; p->p_lds = L.total;
          break;
        }
      }
    }
    if (!p->pers) p->mh_r = 0;
  }
  return LGPU_OK;
}

// B fragments of the matrix-core horizontal pass (s2p_hpass_mfma), cached per (device, taps, ratio, first-column offset, byte order).
// Lane l supplies B[k = 16 (l >> 4) + e][n = l & 15], e = 0..15, of K block kb: k = (window pixel 16 kb + (k >> 2), source byte k & 3),
// n = (output column n >> 2, output channel n & 3); value = tap[pixel - (c0 + r column)] where the source byte feeds the channel (src_sel), as 64 hi + lo.
static std::mutex g_s2pb_mu;
static std::map<std::vector<int>, void *> g_s2pb;
static int sep2p_bfrag(const SepPlan &p, const void **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::vector<int> key = {dev, p.mh_r, p.mh_c0, p.mh_nt, p.mh_kb, (int)p.a.src_sel};
  for (int j = 0; j < p.mh_nt; j++) key.push_back(p.mh_taps[j]);
  std::lock_guard<std::mutex> lk(g_s2pb_mu);
  auto it = g_s2pb.find(key);
  if (it == g_s2pb.end()) {
    std::vector<int8_t> frag((size_t)p.mh_kb * 2 * 64 * 16);
    for (int kb = 0; kb < p.mh_kb; kb++)
      for (int l = 0; l < 64; l++)
        for (int e = 0; e < 16; e++) {
          const int k = 16 * (l >> 4) + e, n = l & 15;
          const int px = 16 * kb + (k >> 2), sbyte = k & 3, col = n >> 2, och = n & 3;
          const int want = (int)((p.a.src_sel >> (8 * och)) & 3u);
          const int j = px - (p.mh_c0 + p.mh_r * col);
          const int tap = (sbyte == want && j >= 0 && j < p.mh_nt) ? p.mh_taps[j] : 0;
          frag[(((size_t)kb * 2 + 0) * 64 + l) * 16 + e] = (int8_t)(tap >> 6);
          frag[(((size_t)kb * 2 + 1) * 64 + l) * 16 + e] = (int8_t)(tap & 63);
        }
    void *d = nullptr;
    LGPU_HIP(hipMalloc(&d, frag.size()));
    LGPU_HIP(hipMemcpy(d, frag.data(), frag.size(), hipMemcpyHostToDevice));
    it = g_s2pb.emplace(key, d).first;
  }
  *out = it->second;
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
#define SEP2_LAUNCH(N)                                                                                       \
  do {                                                                                                       \
    if (p.lds > 48 * 1024)                                                                                   \
      LGPU_HIP(hipFuncSetAttribute((const void *)k_sep2<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds)); \
    hipLaunchKernelGGL((k_sep2<N>), p.grid, blk, p.lds, st, p.a, t, l);                                      \
  } while (0)
  if (tune_on(TUNE_PLAN_DEBUG)) fprintf(stderr, "plan: %dx%d -> %dx%d variant %d lds %zu th %d sht %d swt %d | pers %d vec %d mh_r %d p_th %d p_sht %d p_lds %zu npv %d nth %d\n", p.a.sw, p.a.sh, p.a.dw, p.a.dh, p.variant, p.lds, p.a.th, p.a.sht, p.a.swt, (int)p.pers, p.a.vec, p.mh_r, p.p_th, p.p_sht, p.p_lds, p.a.npv, p.a.nth);
  const bool s2p_force = tune_on(TUNE_SEP2P_FORCE);          // tests: the persistent kernel on small frames
  // k_sep2p pays when a workgroup gets a few tiles to pipeline and the windows are the heavy part (shrinking); measured in profiles/r02/resize_ratios.md
  if (p.pers && p.a.vec && p.variant >= 100 &&
      (s2p_force || (p.p_th >= 4 && (p.mh_r || (p.a.dh < p.a.sh && (long)p.a.tiles_x * p.p_tiles_y * (long)p.grid.y >= 3L * 512))))) {      // 2-row tiles (4:1: 498 against 470 us for 16 x 4K -> 960 x 540) lose to k_sep2
    SepArgs ap = p.a;
    ap.th = p.p_th; ap.sht = p.p_sht; ap.tiles_y = p.p_tiles_y; ap.ntracks = (int)p.grid.y;
    ap.bfrag = nullptr; ap.mh_r = p.mh_r;
    size_t lds_launch = p.p_lds;
    if (p.mh_r) {
      int rc = sep2p_bfrag(p, &ap.bfrag);
      if (rc) return rc;
      lds_launch = S2pLds(ap.sht, ap.swt, ap.th, ap.npv, 1).total;
    }
    const int g_cus = device_cus();
    const int nwork = ap.tiles_x * ap.tiles_y * ap.ntracks;
    int g = (nwork + 7) & ~7;
    if (g > 2 * g_cus) g = (2 * g_cus) & ~7;          // two workgroups per CU by LDS (three or four smaller ones measured slower: 38 / 48 us against 24 for 4K -> 720p)
    if (g < 8) g = 8;
    if (g >= 64) {                                     // the tile-list stride (g / 8) must not share a large factor with the tiles per row (see try_half8)
      auto gcd = [](int x, int y) { while (y) { const int t_ = x % y; x = y; y = t_; } return x; };
      int w = g >> 3;
      for (int k = 0; k < 6 && w - k >= 8; k++)
        if (gcd(w - k, ap.tiles_x) <= 2) { w -= k; break; }
      g = w << 3;
    }
    const bool s2p_dbg = tune_on(TUNE_PHASE_PROFILE);      // the instrumented kernel path (allocates, synchronises): its own switch, not LGPU_PLAN_DEBUG
    ap.dbg = nullptr;
    const int nwv = (p.mh_r ? kS2pMhCW : 4) + 2;
    if (s2p_dbg) { LGPU_HIP(hipMalloc((void **)&ap.dbg, (size_t)g * nwv * 8 * 8)); LGPU_HIP(hipMemsetAsync(ap.dbg, 0, (size_t)g * nwv * 8 * 8, st)); }
#define SEP2P_LAUNCH(N)                                                                                      \
  do {                                                                                                       \
    if (p.p_lds > 48 * 1024)                                                                                 \
      LGPU_HIP(hipFuncSetAttribute((const void *)k_sep2p<N, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.p_lds)); \
    hipLaunchKernelGGL((k_sep2p<N, 0, 4>), dim3((unsigned)g), dim3(6 * 64), p.p_lds, st, ap, t, l);           \
  } while (0)
    if (p.mh_r) {
      if (lds_launch > 48 * 1024) {
        LGPU_HIP(hipFuncSetAttribute((const void *)k_sep2p<1, 1, kS2pMhCW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
        LGPU_HIP(hipFuncSetAttribute((const void *)k_sep2p<1, 2, kS2pMhCW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
      }
      if (p.mh_kb == 1) hipLaunchKernelGGL((k_sep2p<1, 1, kS2pMhCW>), dim3((unsigned)g), dim3((kS2pMhCW + 2) * 64), lds_launch, st, ap, t, l);
      else hipLaunchKernelGGL((k_sep2p<1, 2, kS2pMhCW>), dim3((unsigned)g), dim3((kS2pMhCW + 2) * 64), lds_launch, st, ap, t, l);
    } else
    switch (p.variant) {
    case 101: SEP2P_LAUNCH(1); break;
    case 102: SEP2P_LAUNCH(2); break;
    case 103: SEP2P_LAUNCH(3); break;
    case 104: SEP2P_LAUNCH(4); break;
    case 105: SEP2P_LAUNCH(5); break;
    case 106: SEP2P_LAUNCH(6); break;
    case 107: SEP2P_LAUNCH(7); break;
    case 108: SEP2P_LAUNCH(8); break;
    case 110: SEP2P_LAUNCH(10); break;
    default: SEP2P_LAUNCH(12); break;
    }
#undef SEP2P_LAUNCH
    LGPU_CHECK_LAUNCH();
    if (s2p_dbg) {           // debugging aid: mean cycles (100 MHz s_memtime ticks) per phase over the workgroups, compute wave 0 and the two memory waves
      std::vector<unsigned long long> h((size_t)g * nwv * 8);
      LGPU_HIP(hipStreamSynchronize(st));
      LGPU_HIP(hipMemcpy(h.data(), ap.dbg, h.size() * 8, hipMemcpyDeviceToHost));
      LGPU_HIP(hipFree(ap.dbg));
      double acc[3][4] = {{0}};
      for (int b = 0; b < g; b++)
        for (int k = 0; k < 4; k++) { acc[0][k] += (double)h[((size_t)b * nwv + 0) * 8 + k]; acc[1][k] += (double)h[((size_t)b * nwv + nwv - 2) * 8 + k]; acc[2][k] += (double)h[((size_t)b * nwv + nwv - 1) * 8 + k]; }
      int occ = -1;
      if (p.mh_r) { if (p.mh_kb == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sep2p<1, 1, kS2pMhCW>, (kS2pMhCW + 2) * 64, lds_launch); else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sep2p<1, 2, kS2pMhCW>, (kS2pMhCW + 2) * 64, lds_launch); }
      fprintf(stderr, "occupancy (workgroups per CU by the runtime's calculator): %d\n", occ);
      fprintf(stderr, "k_sep2p<%d>%s grid %d th %d sht %d swt %d tiles %d lds %zu | compute w0: A %.0f H %.0f B %.0f V %.0f | mem w4: A %.0f land %.0f B %.0f issue %.0f | mem w5: A %.0f land %.0f B %.0f issue %.0f (ticks of 10 ns, mean per workgroup)\n",
              p.variant - 100, p.mh_r ? " (matrix-core horizontal pass)" : "", g, ap.th, ap.sht, ap.swt, nwork, lds_launch, acc[0][0] / g, acc[0][1] / g, acc[0][2] / g, acc[0][3] / g, acc[1][0] / g, acc[1][1] / g, acc[1][2] / g,
              acc[1][3] / g, acc[2][0] / g, acc[2][1] / g, acc[2][2] / g, acc[2][3] / g);
    }
    return LGPU_OK;
  }
  switch (p.variant) {
  case 101: SEP2_LAUNCH(1); break;
  case 102: SEP2_LAUNCH(2); break;
  case 103: SEP2_LAUNCH(3); break;
  case 104: SEP2_LAUNCH(4); break;
  case 105: SEP2_LAUNCH(5); break;
  case 106: SEP2_LAUNCH(6); break;
  case 107: SEP2_LAUNCH(7); break;
  case 108: SEP2_LAUNCH(8); break;
  case 110: SEP2_LAUNCH(10); break;
  case 112: SEP2_LAUNCH(12); break;
  case 1: SEP_LAUNCH(8, 8); break;
  case 2: SEP_LAUNCH(5, 5); break;
  case 3: SEP_LAUNCH(4, 4); break;
  case 4: SEP_LAUNCH(2, 2); break;
  case 5: SEP_LAUNCH(6, 6); break;
  default: SEP_LAUNCH(0, 0); break;
  }
#undef SEP_LAUNCH
#undef SEP2_LAUNCH
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// scratch for the multi-launch paths: per (device, stream), so that host threads working on their own streams never share
// an intermediate; grown on demand, never shrunk
// held across the launches of one multi-launch sequence: two host threads on the same stream must not interleave
// (writer A, writer B, reader A) on the shared intermediate
static std::mutex g_multi_mu;
static std::mutex g_scratch_mu;
static std::map<std::pair<int, hipStream_t>, std::pair<void *, size_t>> g_scratch;
static int get_scratch(size_t bytes, hipStream_t st, void **out) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto &s = g_scratch[std::make_pair(dev, st)];
  if (s.second < bytes) {
    if (s.first) { LGPU_HIP(hipStreamSynchronize(st)); LGPU_HIP(hipFree(s.first)); s.first = nullptr; s.second = 0; }
    if (hipMalloc(&s.first, bytes) != hipSuccess) { set_error("hipMalloc(%zu) for scratch failed", bytes); return LGPU_E_NOMEM; }
    s.second = bytes;
  }
  *out = s.first;
  return LGPU_OK;
}

// 5x5 binomial blur through k_gauss5x; LGPU_E_UNSUPPORTED when the frames are not 8-byte aligned (or LGPU_G5_CLASSIC
// asks for the generic separable kernel, for A/B runs)
static int try_gauss5x(int w, int h, int irow, int orow, int blend, int irow2, uint32_t bf, const int32_t *bf_d, int use_lut,
                       const SepTracks &t, int ntracks, const Lut8 &l, hipStream_t st) {
  const bool classic = tune_on(TUNE_G5_CLASSIC);
  if (classic) return LGPU_E_UNSUPPORTED;
  if (irow & 7) return LGPU_E_UNSUPPORTED;
  for (int i = 0; i < ntracks; i++)
    if (((uintptr_t)t.src[i] & 7) || ((uintptr_t)t.dst[i] & 3) || (blend && ((uintptr_t)t.l2[i] & 3))) return LGPU_E_UNSUPPORTED;
  if ((orow & 3) || (blend && (irow2 & 3))) return LGPU_E_UNSUPPORTED;
  G5Args a;
  a.w = w; a.h = h; a.irow = irow; a.orow = orow; a.irow2 = irow2; a.blend = blend; a.use_lut = use_lut;
  a.bf = bf & 0xFF; a.nbf = 0xFF - a.bf; a.bf_d = bf_d;
  a.kscale = nullptr;
  if (blend) { int rc = get_kscale(&a.kscale); if (rc) return rc; }
  a.tiles_x = (w + kG5W - 1) / kG5W;
  const int tiles_y = (h + kG5H * kG5Sub - 1) / (kG5H * kG5Sub);
  const dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)ntracks);
  const bool mfma_h = tune_on(TUNE_G5_MFMA);       // the matrix-core form of the horizontal pass (north_star names it; measured slower: profiles/r02/gauss5_mfma.md), LGPU_G5_MFMA / lgpu_tuning_set
  if (mfma_h) {
    if (blend || use_lut) hipLaunchKernelGGL((k_gauss5x<true, true>), grid, dim3(256), 0, st, a, t, l);
    else hipLaunchKernelGGL((k_gauss5x<false, true>), grid, dim3(256), 0, st, a, t, l);
  } else if (blend || use_lut) hipLaunchKernelGGL(k_gauss5x<true>, grid, dim3(256), 0, st, a, t, l);
  else hipLaunchKernelGGL(k_gauss5x<false>, grid, dim3(256), 0, st, a, t, l);
  LGPU_CHECK_LAUNCH();
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
    {
      SepTracks t1;
      t1.src[0] = src_d; t1.l2[0] = nullptr; t1.dst[0] = dst_d;
      rc = try_half8(hb, vb, sw, sh, irow, dw, dh, orow, 0, 0, 0, 0, nullptr, lut8 ? 1 : 0, t1, 1, l, st);
      if (rc != LGPU_E_UNSUPPORTED) return rc;
    }
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
  std::lock_guard<std::mutex> seq(g_multi_mu);
  if ((rc = get_scratch(need, st, &scratch))) return rc;
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

namespace lgpu { int gauss5_rows(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int psize, hipStream_t st); }
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
  // aligned 3- / 4-byte frames: the register-pipelined row walk of fused.hip (the same arithmetic; profiles/r03/ops_roofline.md)
  const bool no_rows = tune_on(TUNE_GAUSS5_NO_ROWS);
  if (!no_rows && (psize == 3 || psize == 4)) {
    rc = lgpu::gauss5_rows(src_d, irow, dst_d, orow, width, height, psize, st);
    if (rc != LGPU_E_UNSUPPORTED) return rc;
  }
  const bool al4 = (((uintptr_t)src_d | (uintptr_t)irow | (uintptr_t)dst_d | (uintptr_t)orow) & 3) == 0;
  if (psize == 4 && al4) {
    SepTracks tg;
    tg.src[0] = src_d; tg.l2[0] = nullptr; tg.dst[0] = dst_d;
    rc = try_gauss5x(width, height, irow, orow, 0, 0, 0, nullptr, 0, tg, 1, l, st);
    if (rc != LGPU_E_UNSUPPORTED) return rc;
    SepPlan p;
    if ((rc = plan_sep(hb, vb, width, height, irow, width, height, orow, 1, 0, 0, 128, 8, &p))) return rc;
    p.a.src_sel = 0x03020100u; p.a.blend = 0; p.a.irow2 = 0; p.a.bf = 0; p.a.nbf = 255; p.a.bf_d = nullptr; p.a.use_lut = 0;
    p.a.vec = (((uintptr_t)src_d | (uintptr_t)irow) & 15) == 0;
    SepTracks t;
    t.src[0] = src_d; t.l2[0] = nullptr; t.dst[0] = dst_d;
    return launch_sep(p, t, l, st);
  }
  void *scratch;
  std::lock_guard<std::mutex> seq(g_multi_mu);
  if ((rc = get_scratch(sizeof(int16_t) * (size_t)height * width * psize, st, &scratch))) return rc;
  unsigned gy = (unsigned)(height > 2048 ? 2048 : height);
  hipLaunchKernelGGL(k_hpass_generic, dim3(cdiv((unsigned)(width * psize), kBlock), gy), dim3(kBlock), 0, st, src_d, irow, width, height,
                     (int16_t *)scratch, width, psize, hb->pos, hb->co, 5, 0, 0);
  LGPU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_vpass_generic, dim3(cdiv((unsigned)(width * psize), kBlock), gy), dim3(kBlock), 0, st, (const int16_t *)scratch,
                     height, dst_d, orow, width * psize, height, vb->pos, vb->co, 5, 128, 8, psize, 0, l);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

namespace lgpu { int pb_chain(const lgpu_chain_params *pr, const lgpu_canvas *cv, const lgpu_chain_track *tracks, int ntracks, hipStream_t st, const uint8_t *amounts); }
// every LGPU_E_BADARG lgpu_chain can answer, and nothing else: pure argument checks, no device work.  lgpu_chain_step runs them before it feeds or enqueues anything.
extern "C" int lgpu_chain_check(const lgpu_chain_params *pr, const lgpu_chain_track *tracks, int ntracks) {
  LGPU_REQUIRE(pr && tracks && ntracks > 0 && ntracks <= LGPU_CHAIN_MAX_TRACKS, "1..64 tracks");
  LGPU_REQUIRE(pr->sw > 0 && pr->sh > 0 && pr->dw > 0 && pr->dh > 0, "empty geometry");
  LGPU_REQUIRE(pr->irow >= pr->sw * 4 && pr->orow >= pr->dw * 4 && pr->irow2 >= pr->dw * 4, "rowstride smaller than a row");
  LGPU_REQUIRE(((pr->irow | pr->orow | pr->irow2) & 3) == 0, "rowstrides must be multiples of 4");
  for (int i = 0; i < ntracks; i++) {
    LGPU_REQUIRE(tracks[i].src_d && tracks[i].layer2_d && tracks[i].dst_d, "null track pointer");
    LGPU_REQUIRE((((uintptr_t)tracks[i].src_d | (uintptr_t)tracks[i].layer2_d | (uintptr_t)tracks[i].dst_d) & 3) == 0, "frames must be 4-byte aligned");
    LGPU_REQUIRE(tracks[i].src_d != tracks[i].dst_d, "the chain cannot run in place");
  }
  LGPU_REQUIRE(!(pr->sw == pr->dw && pr->sh == pr->dh), "the chain needs a resize stage (same size: lgpu_swizzle + lgpu_blend_chroma + lgpu_gamma_apply)");
  if (pr->interp & LGPU_INTERP_PIXBUF) {
    const int ip = pr->interp & 0xFF;
    LGPU_REQUIRE(ip == 0 || ip == 2 || ip == 3, "interp must be 0 (NEAREST), 2 (BILINEAR) or 3 (HYPER) on the gdk-pixbuf arithmetic");
    LGPU_REQUIRE(pr->sw < 32768 && pr->sh < 32768 && pr->dw < 32768 && pr->dh < 32768, "frame sides must stay below 32768 (16.16 positions)");
  }
  return LGPU_OK;
}

static int chain_launch(const lgpu_chain_params *pr, const lgpu_chain_track *tracks, int ntracks, hipStream_t st) {
  int rc;
  if ((rc = lgpu_chain_check(pr, tracks, ntracks))) return rc;
  if (pr->interp & LGPU_INTERP_PIXBUF) {                 // the resize stage on the reference's gdk-pixbuf arithmetic (pixbuf.hip)
    return pb_chain(pr, nullptr, tracks, ntracks, st, nullptr);
  }
  const int kernel = kernel_for_interp(pr->interp & 0xFF, pr->dw > pr->sw || pr->dh > pr->sh);      // (flag bits such as LGPU_INTERP_OPAQUE mean nothing to the polyphase stage)
  const Lut8 l = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
  const uint32_t sel = pr->swap_rb ? 0x03000102u : 0x03020100u;   // swap3postalpha: [in2 in1 in0 in3]
  const Bank *hb, *vb, *gh, *gv;
  SepTracks t;
  uintptr_t src_bits = (uintptr_t)pr->irow;
  for (int i = 0; i < ntracks; i++) src_bits |= (uintptr_t)tracks[i].src_d;
  const int src_vec = (src_bits & 15) == 0;
  if (!pr->do_blur) {
    if ((rc = get_bank(pr->sw, pr->dw, kernel, &hb)) || (rc = get_bank(pr->sh, pr->dh, kernel, &vb))) return rc;
    for (int i = 0; i < ntracks; i++) { t.src[i] = tracks[i].src_d; t.l2[i] = tracks[i].layer2_d; t.dst[i] = tracks[i].dst_d; }
    rc = try_half8(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->orow, pr->swap_rb ? 1 : 0, 1, pr->irow2, (uint32_t)pr->bf & 0xFF,
                   pr->param_block_d, pr->use_lut ? 1 : 0, t, ntracks, l, st);
    if (rc != LGPU_E_UNSUPPORTED) return rc;
    SepPlan p;
    if ((rc = plan_sep(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->orow, ntracks, 64, 7, 1 << 20, 21, &p))) return rc;
    p.a.src_sel = sel; p.a.blend = 1; p.a.irow2 = pr->irow2; p.a.bf = (uint32_t)pr->bf & 0xFF; p.a.nbf = 0xFF - p.a.bf; p.a.bf_d = pr->param_block_d;
    p.a.use_lut = pr->use_lut ? 1 : 0; p.a.vec = src_vec;
    for (int i = 0; i < ntracks; i++) { t.src[i] = tracks[i].src_d; t.l2[i] = tracks[i].layer2_d; t.dst[i] = tracks[i].dst_d; }
    return launch_sep(p, t, l, st);
  }
  // with blur: resize into scratch (per track), then gaussian with the blend + gamma epilogue
  void *scratch;
  const size_t per = (size_t)pr->dw * 4 * pr->dh;
  std::lock_guard<std::mutex> seq(g_multi_mu);
  if ((rc = get_scratch(per * ntracks, st, &scratch))) return rc;
  if ((rc = get_bank(pr->sw, pr->dw, kernel, &hb)) || (rc = get_bank(pr->sh, pr->dh, kernel, &vb))) return rc;
  if ((rc = get_bank(pr->dw, pr->dw, 100, &gh)) || (rc = get_bank(pr->dh, pr->dh, 100, &gv))) return rc;
  SepPlan p1, p2;
  if ((rc = plan_sep(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->dw * 4, ntracks, 64, 7, 1 << 20, 21, &p1))) return rc;
  p1.a.src_sel = sel; p1.a.blend = 0; p1.a.irow2 = 0; p1.a.bf = 0; p1.a.nbf = 255; p1.a.bf_d = nullptr; p1.a.use_lut = 0; p1.a.vec = src_vec;
  for (int i = 0; i < ntracks; i++) { t.src[i] = tracks[i].src_d; t.l2[i] = nullptr; t.dst[i] = (uint8_t *)scratch + per * i; }
  rc = try_half8(hb, vb, pr->sw, pr->sh, pr->irow, pr->dw, pr->dh, pr->dw * 4, pr->swap_rb ? 1 : 0, 0, 0, 0, nullptr, 0, t, ntracks, pack_lut(nullptr), st,
                 0);      // the scratch frames are read back by the gaussian launch right behind: ordinary stores (non-temporal measured the same, 250.4 against 250.1 us)
  if (rc == LGPU_E_UNSUPPORTED) rc = launch_sep(p1, t, pack_lut(nullptr), st);
  if (rc) return rc;
  for (int i = 0; i < ntracks; i++) { t.src[i] = (uint8_t *)scratch + per * i; t.l2[i] = tracks[i].layer2_d; t.dst[i] = tracks[i].dst_d; }
  rc = try_gauss5x(pr->dw, pr->dh, pr->dw * 4, pr->orow, 1, pr->irow2, (uint32_t)pr->bf, pr->param_block_d, pr->use_lut ? 1 : 0, t, ntracks, l, st);
  if (rc != LGPU_E_UNSUPPORTED) return rc;
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
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}
