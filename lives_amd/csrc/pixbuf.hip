// pixbuf.hip -- the "pixbuf" resize arithmetic: what the reference's non-swscale resize body computes.
//
//   resize_layer_full -> layer_to_pixbuf -> lives_pixbuf_scale_simple(pixbuf, width, height, interp)     src/colourspace.c:15262-15322 (call :15295)
//   compositor: gdk_pixbuf_scale_simple(in_pixbuf, owidth, oheight, up_interp / down_interp)            lives-plugins/weed-plugins/gdk/compositor.c:263-265
//   LIVES_INTERP_BEST / NORMAL / FAST = GDK_INTERP_HYPER / BILINEAR / NEAREST                           src/widget-helper-gtk.h:1136-1138
//
// The scaler itself is gdk-pixbuf's (third party, not in the LiVES tree).  Its arithmetic, as pinned byte for byte against the runtime library
// 2.42.8 by tests/test_pixbuf_scale.py and tests/golden/pixbuf_scale.npz:
//   * source position of destination pixel j: x = j * x_step + floor(offset * 65536) in 16.16, x_step = (int)(65536 / scale); first tap at x >> 16,
//     phase (x >> 12) & 15; the same vertically
//   * one two-dimensional integer weight table per (y phase, x phase), n_y x n_x taps summing to exactly 65536:
//       BILINEAR enlarging: 2 taps, centre aligned; BILINEAR reducing: box over the source span, n = ceil(1 + 1 / scale);
//       HYPER: the bilinear kernel integrated over the destination pixel, n = ceil(1 / scale + 3), offset -1
//   * no alpha (3 bytes / pixel): c = (sum w * q + 0xffff) >> 16 (0x8000 when the filter is 2 x 2); pixels whose taps leave the row take the
//     library's per-pixel path, c = (sum 255 * w * q + 0xffffff) >> 24
//   * alpha (4 bytes / pixel): ta = alpha * w; c = (uint8_t)((double)(sum ta * q) * (1.0 / (double)(sum ta))), alpha' = (sum ta) >> 16, all 0 when sum ta == 0
//   * NEAREST: source pixel ((j * step + step / 2) >> 16), clamped
//   * reductions so strong that n_x * n_y > 1000 go through the library's two-step scaler: not covered, LGPU_E_UNSUPPORTED
//
// Kernels: k_pb_window (a 64 x tile_h output tile per workgroup; the clamped source window staged in LDS one dword per pixel, weights by scalar
// loads when the x phase is the same for every pixel -- integer ratios -- and per-lane loads otherwise), k_pb_direct (any ratio, source through
// the caches), k_pb_nearest.  HBM-bound work with an integer MAC per tap and channel; no matrix-core formulation: the per-phase tables are
// two-dimensional (rounded and corrected per phase), not separable.
#include <atomic>
#include "lgpu_common.h"
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace lgpu {

int get_kscale(const uint2 **out);      // resize.hip: the chroma blend's alpha scalers as a device table

struct PbArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, sw, sh, dw, dh;
  int x_step, y_step, xoff, yoff;
  int n_x, n_y;
  int tx0, tx1, ty0, ty1;   // the taps that are non-zero for some destination pixel of this call: [tx0, tx1) x [ty0, ty1)
  const int *table;     // device: [16][16][n_y][n_x]
  unsigned rnd;         // 3-byte interior rounding term
  int tile_h, win_w, win_h;
};

__device__ __forceinline__ int pb_clamp(int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; }

template <int CH>
__device__ __forceinline__ uint32_t pb_load_px(const uint8_t *row, int x) {
  if (CH == 4) return *reinterpret_cast<const uint32_t *>(row + 4 * (size_t)x);
  const uint8_t *p = row + 3 * (size_t)x;
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
}

// fl(1.0 / (double)a) for an integer 1 <= a < 2^24 -- the library's `1.0 / (double)a`, correctly rounded -- in five operations instead of the eleven of the compiler's
// IEEE division (scale, fix-up and the denormal / overflow handling are not needed for this range): the hardware estimate (2^-26 or better) and two Newton steps
// in fused multiply-adds.  Why the last step rounds correctly: e1 = 1 - a y1 is exact (a y1 has at most 77 bits and differs from 1 by ~2^-50), so the fma rounds
// the REAL number (1 / a)(1 - e1^2) once; and 1 / a for a 24-bit integer a that is not a power of two lies at least 2^-25 ulp away from every rounding boundary,
// far more than e1^2 ~ 2^-100.  Checked against the division for every a of the range on the device (lgpu_debug_recip_check, tests/test_pixbuf_scale.py).
__device__ __forceinline__ double pb_recip(uint32_t a) {
  const double x = (double)a;
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}
// the accumulators of one destination pixel -> its bytes, packed c0 | c1 << 8 | c2 << 16 | alpha << 24
template <int CH>
__device__ __forceinline__ uint32_t pb_finish_px(unsigned r, unsigned g, unsigned b, unsigned a, bool edge, unsigned rnd) {
  if (CH == 4) {
    // every tap of every lane opaque (the common frame): fl(1 / a) is a constant -- a WAVE-UNIFORM test, so the choice is a scalar branch; otherwise every lane runs the
    // five-operation reciprocal (a <= 0xFF0000; a == 0 computes on 1 and is zeroed below).  Per-lane tests of both cases made this a web of exec-mask branches.
    const bool opaque = __builtin_amdgcn_ballot_w64(a != 0xFF0000u) == 0;
    const double ia = opaque ? (1.0 / 16711680.0) : pb_recip(a ? a : 1u);
    const uint32_t o = (uint32_t)(uint8_t)((double)r * ia) | ((uint32_t)(uint8_t)((double)g * ia) << 8) | ((uint32_t)(uint8_t)((double)b * ia) << 16) | ((a >> 16) << 24);
    return a ? o : 0u;
  }
  if (edge) return ((r * 255u + 0xffffffu) >> 24) | (((g * 255u + 0xffffffu) >> 24) << 8) | (((b * 255u + 0xffffffu) >> 24) << 16);
  return (((r + rnd) >> 16) & 0xFFu) | ((((g + rnd) >> 16) & 0xFFu) << 8) | ((((b + rnd) >> 16) & 0xFFu) << 16);
}
template <int CH>
__device__ __forceinline__ void pb_finish(uint8_t *d, unsigned r, unsigned g, unsigned b, unsigned a, bool edge, unsigned rnd) {
  if (CH == 4) {
    *reinterpret_cast<uint32_t *>(d) = pb_finish_px<4>(r, g, b, a, edge, rnd);
  } else if (edge) {
    d[0] = (uint8_t)((r * 255u + 0xffffffu) >> 24); d[1] = (uint8_t)((g * 255u + 0xffffffu) >> 24); d[2] = (uint8_t)((b * 255u + 0xffffffu) >> 24);
  } else {
    d[0] = (uint8_t)((r + rnd) >> 16); d[1] = (uint8_t)((g + rnd) >> 16); d[2] = (uint8_t)((b + rnd) >> 16);
  }
}

template <int CH>
__device__ __forceinline__ void pb_tap(uint32_t q, unsigned w, unsigned &r, unsigned &g, unsigned &b, unsigned &a) {
  if (CH == 4) {
    const unsigned ta = __umul24(q >> 24, w);            // alpha * weight < 2^24
    r = __umul24(ta, q & 0xFF) + r; g = __umul24(ta, (q >> 8) & 0xFF) + g; b = __umul24(ta, (q >> 16) & 0xFF) + b; a += ta;
  } else {
    r = __umul24(w, q & 0xFF) + r; g = __umul24(w, (q >> 8) & 0xFF) + g; b = __umul24(w, (q >> 16) & 0xFF) + b;
  }
}

// 64 x tile_h output pixels per workgroup (4 waves, a wave per output row), clamped source window in LDS
// The frames of one launch: every scaler kernel below serves up to LGPU_CHAIN_MAX_TRACKS frames of ONE geometry (lgpu_pixbuf_scale_batch: the frames of the live
// tracks of one plan step) -- the grid's z index is the frame, the tables, the weight vectors and the launch are paid once.  The pointers in the argument
// structs are frame 0's (what a single-frame call passes).
struct PbFrames {
  const uint8_t *src[LGPU_CHAIN_MAX_TRACKS];
  uint8_t *dst[LGPU_CHAIN_MAX_TRACKS];
};
#define PB_FRAME_ARGS(TYPE) TYPE A = A_; A.src = F.src[blockIdx.z]; A.dst = F.dst[blockIdx.z]

template <int CH, int UNIFORM_X>
__global__ __launch_bounds__(256) void k_pb_window(const PbArgs A_, const PbFrames F) {
  PB_FRAME_ARGS(PbArgs);
  extern __shared__ uint32_t win[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // uniform, and the compiler is told so: row arithmetic on the scalar unit
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * A.tile_h;
  const int x0 = (int)(((long long)j0 * A.x_step + A.xoff) >> 16), y0 = (int)(((long long)i0 * A.y_step + A.yoff) >> 16);
  for (int wy = wave; wy < A.win_h; wy += 4) {
    const uint8_t *row = A.src + (size_t)pb_clamp(y0 + wy, A.sh - 1) * A.irow;
    uint32_t *wr = win + wy * A.win_w;
    for (int wx = lane; wx < A.win_w; wx += 64) wr[wx] = pb_load_px<CH>(row, pb_clamp(x0 + wx, A.sw - 1));
  }
  __syncthreads();
  const int j = j0 + lane;
  const long long x = (long long)j * A.x_step + A.xoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15;
  const bool edge = xs < 0 || xs + A.n_x > A.sw;
  const int nn = A.n_x * A.n_y;
  for (int r_ = wave; r_ < A.tile_h; r_ += 4) {
    const int i = i0 + r_;
    if (i >= A.dh) break;
    const long long y = (long long)i * A.y_step + A.yoff;
    const int ys = (int)(y >> 16), yph = (int)(y >> 12) & 15;
    const uint32_t *wp = win + (ys - y0) * A.win_w + (xs - x0);
    unsigned r = 0, g = 0, b = 0, a = 0;
    if (j < A.dw) {
      if (UNIFORM_X) {
        const int *wt = A.table + (size_t)__builtin_amdgcn_readfirstlane((yph * 16 + xph) * nn);
        wp += A.ty0 * A.win_w; wt += A.ty0 * A.n_x;
        for (int ty = A.ty0; ty < A.ty1; ty++, wp += A.win_w, wt += A.n_x)
          for (int tx = A.tx0; tx < A.tx1; tx++) pb_tap<CH>(wp[tx], (unsigned)wt[tx], r, g, b, a);
      } else {
        const int *wt = A.table + (size_t)(yph * 16 + xph) * nn + A.ty0 * A.n_x;
        wp += A.ty0 * A.win_w;
        for (int ty = A.ty0; ty < A.ty1; ty++, wp += A.win_w, wt += A.n_x)
          for (int tx = A.tx0; tx < A.tx1; tx++) pb_tap<CH>(wp[tx], (unsigned)wt[tx], r, g, b, a);
      }
      pb_finish<CH>(A.dst + (size_t)i * A.orow + (size_t)j * CH, r, g, b, a, edge, A.rnd);
    }
  }
}

// any ratio: a thread per destination pixel, taps straight from memory
template <int CH>
__global__ __launch_bounds__(256) void k_pb_direct(const PbArgs A_, const PbFrames F) {
  PB_FRAME_ARGS(PbArgs);
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= A.dw || i >= A.dh) return;
  const long long x = (long long)j * A.x_step + A.xoff, y = (long long)i * A.y_step + A.yoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15, ys = (int)(y >> 16), yph = (int)(y >> 12) & 15;
  const bool edge = xs < 0 || xs + A.n_x > A.sw;
  const int *wt = A.table + (size_t)(yph * 16 + xph) * A.n_x * A.n_y;
  unsigned r = 0, g = 0, b = 0, a = 0;
  wt += A.ty0 * A.n_x;
  for (int ty = A.ty0; ty < A.ty1; ty++, wt += A.n_x) {
    const uint8_t *row = A.src + (size_t)pb_clamp(ys + ty, A.sh - 1) * A.irow;
    for (int tx = A.tx0; tx < A.tx1; tx++) pb_tap<CH>(pb_load_px<CH>(row, pb_clamp(xs + tx, A.sw - 1)), (unsigned)wt[tx], r, g, b, a);
  }
  pb_finish<CH>(A.dst + (size_t)i * A.orow + (size_t)j * CH, r, g, b, a, edge, A.rnd);
}

template <int CH>
__global__ __launch_bounds__(256) void k_pb_nearest(const PbFrames F, int irow, int sw, int sh, int orow, int dw, int dh, int x_step, int y_step) {
  const uint8_t *src = F.src[blockIdx.z];
  uint8_t *dst = F.dst[blockIdx.z];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= dw || i >= dh) return;
  const int ys = pb_clamp((int)(((long long)i * y_step + y_step / 2) >> 16), sh - 1);
  const int xs = pb_clamp((int)(((long long)j * x_step + x_step / 2) >> 16), sw - 1);
  const uint8_t *s = src + (size_t)ys * irow + (size_t)xs * CH;
  uint8_t *d = dst + (size_t)i * orow + (size_t)j * CH;
  if (CH == 4) *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(s);
  else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
}

// =====================================================================================================================================================
// k_pb_half -- the exact 2:1 reduction of 4-byte pixels (the headline 3840x2160 -> 1920x1080 case), HYPER or BILINEAR, optionally with the rest of the
// chain fused behind it (chroma blend with layer 2, gamma LUT).  For this ratio every destination pixel has phase (0, 0) and the library's table is an
// exact outer product -- HYPER: 256 * [1 7 7 1]^T [1 7 7 1] on source pixels 2X-1 .. 2X+2 (the 5th row / column is zero), BILINEAR: 16384 * ones(2, 2)
// on 2X, 2X+1 (the host verifies that against the table it built) -- so the alpha-weighted sums separate exactly:
//     P_c = alpha * q_c (< 2^16),  H_c = vo * (P[2X-1] + P[2X+2]) + vi * (P[2X] + P[2X+1]) (< 2^20),  V_c = the same down the rows (< 2^24),
//     colour = (uint8_t)((double)V_c * (1.0 / (double)V_alpha)),  alpha' = (scale * V_alpha) >> 16      (the common power of two drops out of the quotient)
// No LDS, no matrix cores: a wave owns a strip of 124 output columns and walks down `th` output rows.  A lane loads 4 source pixels per row with one
// 16-byte load, premultiplies them into 16-bit pairs, takes the one neighbour pixel it needs on each side from the adjacent lanes (DPP wave shifts; lanes
// 0 and 63 only feed their neighbours), forms its two H columns with v_dot2_u32_u16 and keeps the last two H rows in registers; every second source row
// one output row (two pixels per lane, 8-byte stores) leaves.  HBM-bound: source read once (+ 2 of 2 th + 2 rows re-read at band seams, + 4 of 252 columns).
// =====================================================================================================================================================
struct PbHalfArgs {
  int sw, sh, irow, dw, dh, orow;
  int hyper;                     // 1: [1 7 7 1] (GDK_INTERP_HYPER), 0: [0 1 1 0] (GDK_INTERP_BILINEAR)
  const uint2 *kscale;           // device [256] {K2, K1}: the chroma blend's translucent scalers (lgpu_alpha_scalers)
  int ashift;                    // alpha' = V_alpha >> ashift   (HYPER 8, BILINEAR 2)
  int swap_rb, blend, irow2, use_lut;
  uint32_t bf;
  const int32_t *bf_d;
  int strips, cgroups, bands, th, ntracks;     // cgroups = ceil(strips / 4): workgroups per band
  int rem;                       // the first `rem` bands are th + 1 rows tall, the others th (bands * th + rem == dh: band heights differ by one row at most)
  int cw, ch, ox, oy;            // letterbox canvas (cw == 0: none): dst / layer 2 are cw x ch, the scaled frame sits at (ox, oy), the rest is opaque black under the blend
  int main_blocks, bar_blocks;   // workgroups of the frame proper / per track of the bars (1024 canvas pixels each)
  int bar_first;                 // the bars' workgroups come FIRST in the grid (a multiple of 8, so the frame's workgroups keep their XCD): they run while the frame's first loads are in flight
  int nt_out;
  int bgroup;                    // order 2: neighbouring bands per XCD turn (PBH_GROUP)
  int row_major;                 // work order (PBH_ORDER): 0 bands fastest, 1 column groups fastest, 2 that with the bands dealt round robin to the XCDs
  int aligned;                   // host side: strips of 64 quads (k_pb_half<.., ALIGNED>)
  int bf_tracks;                 // 1: the blend amount of track t is PbTracks.bf[t] (bf / bf_d unused)
};
struct PbTracks {
  const uint8_t *src[LGPU_CHAIN_MAX_TRACKS];
  const uint8_t *l2[LGPU_CHAIN_MAX_TRACKS];
  uint8_t *dst[LGPU_CHAIN_MAX_TRACKS];
  uint8_t bf[LGPU_CHAIN_MAX_TRACKS];       // PbHalfArgs.bf_tracks: a blend amount per track (lgpu_chain_amounts: the tracks of a tick need not share one)
};
typedef unsigned short pb_us2 __attribute__((ext_vector_type(2)));
typedef unsigned pb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned pb_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pb_dot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(pb_us2, a), __builtin_bit_cast(pb_us2, b), c, false);
}
// (alpha * byte C) of two pixels as a 16-bit pair, straight from the packed pixels: SDWA multiplies select the bytes and place the 16-bit product
template <int C>
__device__ __forceinline__ uint32_t pb_premul_pair(uint32_t q0, uint32_t q1) {
  uint32_t d;
  if (C == 0) {
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0" : "=v"(d) : "v"(q0));
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_0" : "+v"(d) : "v"(q1));
  } else if (C == 1) {
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_1" : "=v"(d) : "v"(q0));
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_1" : "+v"(d) : "v"(q1));
  } else {
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2" : "=v"(d) : "v"(q0));
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_2" : "+v"(d) : "v"(q1));
  }
  return d;
}
__device__ __forceinline__ uint32_t pb_add_hi_lo(uint32_t x, uint32_t y) {      // x.hi16 + y.lo16
  uint32_t d;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
// a * 7 + c as ONE 24-bit multiply-add whatever the compiler has learnt about the operands: with the all-opaque form's known-zero high bytes it otherwise turns
// __umul24 into a plain multiply and selects v_mad_u64_u32 / v_mul_lo_u32 (quarter rate) for it
__device__ __forceinline__ uint32_t pb_mad7(uint32_t a, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, 7, %2" : "=v"(d) : "v"(a), "v"(c));
  return d;
}
// one source row of a lane: 4 pixels -> the two H columns of its 4 channels (h[c] = column 2k, h[4 + c] = column 2k + 1)
// e (ALIGNED strips only, otherwise 0): lane 0 holds pixel P[4k-1] there, lane 63 pixel P[4k+4] (clamped into the row), every other lane 0 -- the two taps the
// wave shifts cannot deliver.  SWAP: channel 0 is fed from byte 2 and channel 2 from byte 0 (the R <-> B conversion of the chain costs nothing: the three colours
// are treated alike until they are stored).
// OPAQUE (the caller states that every source pixel has alpha 255 -- decoded video, a frame that has just been given its alpha channel): alpha * colour is 255 * colour
// for every tap, the 255 and the library's reciprocal cancel EXACTLY (trunc(255 T * fl(1 / 65280)) == T >> 8 for every T the 256 weight units can make of bytes, and
// the bilinear twin with 1020 and >> 2: checked for all of them when the kernel's tables are built, pb_opaque_check), so the colours go through as plain bytes: one
// v_perm per channel pair instead of two SDWA multiplies, three channels instead of four, a shift instead of the reciprocal and three double-precision products.
template <int HYPER, int ALIGNED = 0, int SWAP = 0, int OPAQUE = 0>
__device__ __forceinline__ void pb_half_hrow(pb_u4 q, uint32_t h[8], uint32_t e = 0u, uint32_t em = 0u) {
  static_assert(!(OPAQUE && ALIGNED), "the all-opaque form exists for the strips with feeder lanes (the gaussian chain)");
  uint32_t A[4], B[4];
  if (OPAQUE) {
    constexpr uint32_t s0 = SWAP ? 0x0C060C02u : 0x0C040C00u, s2 = SWAP ? 0x0C040C00u : 0x0C060C02u;
    A[0] = __builtin_amdgcn_perm(q.y, q.x, s0); B[0] = __builtin_amdgcn_perm(q.w, q.z, s0);
    A[1] = __builtin_amdgcn_perm(q.y, q.x, 0x0C050C01u); B[1] = __builtin_amdgcn_perm(q.w, q.z, 0x0C050C01u);
    A[2] = __builtin_amdgcn_perm(q.y, q.x, s2); B[2] = __builtin_amdgcn_perm(q.w, q.z, s2);
    A[3] = 0u; B[3] = 0u;
  } else {
  A[0] = pb_premul_pair<SWAP ? 2 : 0>(q.x, q.y); B[0] = pb_premul_pair<SWAP ? 2 : 0>(q.z, q.w);
  A[1] = pb_premul_pair<1>(q.x, q.y); B[1] = pb_premul_pair<1>(q.z, q.w);
  A[2] = pb_premul_pair<SWAP ? 0 : 2>(q.x, q.y); B[2] = pb_premul_pair<SWAP ? 0 : 2>(q.z, q.w);
  A[3] = __builtin_amdgcn_perm(q.y, q.x, 0x0C070C03u); B[3] = __builtin_amdgcn_perm(q.w, q.z, 0x0C070C03u);       // the alpha pairs
  }
  // ALIGNED: the one pixel beyond the strip, premultiplied and already in the half of the dword where the wave shift would have delivered it -- em = 65536 in
  // lane 0 (P[4k-1] belongs in the high half of the left neighbour's pair), 1 in lane 63 (P[4k+4] in the low half of the right neighbour's), 0 elsewhere; it then
  // rides into the lane exchange as the value the shift leaves in lanes that have no source lane (DPP without bound_ctrl keeps the destination): 5 operations per row
  uint32_t xe[4] = {0u, 0u, 0u, 0u};
  if (HYPER && ALIGNED) {
    const uint32_t am = __umul24(e >> 24, em);                           // alpha * {65536, 1, 0} <= 0xFF0000: a 24-bit operand
    if (SWAP) {
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(xe[0]) : "v"(am), "v"(e));
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(xe[2]) : "v"(am), "v"(e));
    } else {
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(xe[0]) : "v"(am), "v"(e));
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(xe[2]) : "v"(am), "v"(e));
    }
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(xe[1]) : "v"(am), "v"(e));
    xe[3] = am;
  }
#pragma unroll
  for (int c = 0; c < (OPAQUE ? 3 : 4); c++) {
    if (HYPER) {
      // wave_shr:1 -- the left lane's (P[4k-2], P[4k-1]);  wave_shl:1 -- the right lane's (P[4k+4], P[4k+5]);  lanes 0 / 63 keep xe (0 in strips with feeder lanes)
      const uint32_t bl = ALIGNED ? (uint32_t)__builtin_amdgcn_update_dpp((int)xe[c], (int)B[c], 0x138, 0xF, 0xF, false) : (uint32_t)__builtin_amdgcn_mov_dpp((int)B[c], 0x138, 0xF, 0xF, true);
      const uint32_t ar = ALIGNED ? (uint32_t)__builtin_amdgcn_update_dpp((int)xe[c], (int)A[c], 0x130, 0xF, 0xF, false) : (uint32_t)__builtin_amdgcn_mov_dpp((int)A[c], 0x130, 0xF, 0xF, true);
      h[c] = pb_dot2(A[c], 0x00070007u, pb_add_hi_lo(bl, B[c]));          // P[4k-1] + 7 P[4k] + 7 P[4k+1] + P[4k+2]
      h[4 + c] = pb_dot2(B[c], 0x00070007u, pb_add_hi_lo(A[c], ar));      // P[4k+1] + 7 P[4k+2] + 7 P[4k+3] + P[4k+4]
    } else {
      h[c] = pb_dot2(A[c], 0x00010001u, 0u);
      h[4 + c] = pb_dot2(B[c], 0x00010001u, 0u);
    }
  }
}

// the all-opaque forms replace the library's (uint8_t)((double)V_c * (1.0 / (double)V_alpha)) by a shift: V_c = 255 T, V_alpha = 255 * 256 (HYPER: T = the 256 weight
// units on bytes, T <= 65280) or 255 * 4 (BILINEAR: T <= 1020).  Every T is compared on the device in the library's own double arithmetic before the first such launch.
__global__ void k_pb_opaque_check(unsigned int *bad) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  {      // the general ratios: T = the 65536 weight units on bytes, T <= 65536 * 255 < 2^24; a thread takes 256 of them
    const double ic = pb_recip(255u * 65536u);
    for (uint32_t k = 0; k < 256u; k++) {
      const uint32_t T = t * 256u + k;
      if (T <= 65536u * 255u && (uint32_t)(int)((double)(255ull * T) * ic) != (T >> 16)) atomicAdd(bad, 1u);
    }
  }
  if (t <= 65280u) {
    const double ia = pb_recip(65280u);
    if ((uint32_t)(int)((double)(255u * t) * ia) != (t >> 8)) atomicAdd(bad, 1u);
  }
  if (t <= 1020u) {
    const double ib = pb_recip(1020u);
    if ((uint32_t)(int)((double)(255u * t) * ib) != (t >> 2)) atomicAdd(bad, 1u);
  }
}
static int pb_opaque_check() {
  static std::atomic<int> state{0};          // 0 not run, 1 good, -1 bad
  int st = state.load();
  if (st == 0) {
    unsigned int *d = nullptr, h = 1;
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) { set_error("pb_opaque_check: hipMalloc failed"); return LGPU_E_NOMEM; }
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(k_pb_opaque_check, dim3(256), dim3(256), 0, (hipStream_t)0, d);
    const bool ok = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    st = (ok && h == 0) ? 1 : -1;
    state.store(st);
  }
  if (st < 0) { set_error("the all-opaque shift differs from the library's double-precision un-premultiply on this build: LGPU_INTERP_OPAQUE is refused"); return LGPU_E_HIP; }
  return LGPU_OK;
}
__global__ void k_pb_recip_check(uint32_t lo, uint32_t hi, unsigned long long *bad) {
  const uint32_t a = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (a < lo || a >= hi || a == 0) return;
  const double want = 1.0 / (double)a;
  if (__builtin_bit_cast(unsigned long long, pb_recip(a)) != __builtin_bit_cast(unsigned long long, want)) atomicAdd(bad, 1ull);
}
// V_c * fl(1 / V_alpha), truncated, for the three colours of one pixel
__device__ __forceinline__ void pb_half_colours(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t va, uint32_t c[3]) {
  const double ia = pb_recip(va);
  c[0] = (uint32_t)(int)((double)v0 * ia); c[1] = (uint32_t)(int)((double)v1 * ia); c[2] = (uint32_t)(int)((double)v2 * ia);
}

// BLUR (chain only): BASELINE config 5's 5x5 gaussian ([1 4 6 4 1] / 16 per axis, edge replicate, one rounding: (sum + 128) >> 8) between the scaler and the
// blend, in the same launch.  The scaled row of a lane (two RGBA pixels) is blurred horizontally with its neighbours' pixels (four more DPP moves; bytes in
// 16-bit lanes, so one 32-bit operation serves two channels), the last five blurred rows stay in registers and every new one completes an output row.  A strip
// then yields 120 columns (lanes 2 .. 61) and a band computes 4 more scaled rows than it stores.
//
// Memory operations are buffer loads / stores: one 128-bit descriptor per frame in SGPRs (built once per wave from uniform values), the row as the scalar offset,
// the lane's place in the row as a constant 32-bit VGPR offset -- no address arithmetic on the vector unit at all; lanes that must not store carry an offset beyond
// the descriptor's range and the hardware drops their store (no exec-mask branch per row).
// CHAIN: 0 the scaler alone; 1 [R <-> B] -> scale -> chroma blend with layer 2 -> gamma LUT; 2 the same without a layer 2 (LGPU_INTERP_NOBLEND: a track that is
// not blended with anything -- no layer-2 loads, no blend arithmetic)
template <int CHAIN, int HYPER, int BLUR, int ALIGNED = 0, int SWAP = 0, int OPAQUE = 0>
__global__ __launch_bounds__(256) void k_pb_half(const PbHalfArgs A, const PbTracks T, const Lut8 lut) {
  // the gamma LUT and the blend's alpha scalers.  ONE copy per workgroup, but no workgroup barrier on the frame path: every wave writes the whole of both tables
  // itself (the same bytes) and reads them after its own writes have landed; a slower wave writing the same bytes again changes nothing
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
  __shared__ pb_u2 s_k[256];
  // ALIGNED (no blur): strips of 64 quads, no feeder lanes -- a wave's row is 1024 source bytes and 512 result bytes on 128-byte lines; the two taps beyond the
  // strip come from one extra 4-byte load per source row in lanes 0 and 63
  constexpr int kHalo = BLUR ? 2 : ALIGNED ? 0 : 1, kCols = 64 - 2 * kHalo;      // lanes that only feed their neighbours on each side / lanes that store
  if (CHAIN && blockIdx.x < (unsigned)A.bar_first) {
    // letterbox bars (letterbox_layer's black canvas, src/colourspace.c:15417-15503, under the rest of the chain): opaque black -> chroma blend with layer 2 -> LUT
    stage_lut(s_lut, lut);
    {
      const uint2 kk = A.kscale[threadIdx.x];
      pb_u2 kv; kv.x = kk.x; kv.y = kk.y;
      s_k[threadIdx.x] = kv;
    }
    __syncthreads();
    const int b = blockIdx.x, track = b / A.bar_blocks, chunk = b - track * A.bar_blocks;
    if (track >= A.ntracks) return;                       // padding up to a multiple of 8
    uint32_t bf = A.bf;
    if (A.bf_d) bf = (uint32_t)A.bf_d[0] & 0xFF;
    if (A.bf_tracks) bf = T.bf[track];
    const uint32_t w_lo = bf | ((255u - bf) << 8);
    const int top = A.oy * A.cw, bottom = (A.ch - A.oy - A.dh) * A.cw, sw_ = A.cw - A.dw, total = top + bottom + A.dh * sw_;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int p = chunk * 1024 + i * 256 + threadIdx.x, x, y;
      if (p >= total) break;
      if (p < top) { y = p / A.cw; x = p - y * A.cw; }
      else if (p < top + bottom) { p -= top; y = p / A.cw; x = p - y * A.cw; y += A.oy + A.dh; }
      else { p -= top + bottom; y = p / sw_; x = p - y * sw_; y += A.oy; if (x >= A.ox) x += A.dw; }
      if (CHAIN == 2) { const uint32_t c = s_lut[0]; reinterpret_cast<uint32_t *>(T.dst[track] + (size_t)y * A.orow)[x] = c * 0x010101u | 0xFF000000u; continue; }
      const uint32_t q = reinterpret_cast<const uint32_t *>(T.l2[track] + (size_t)y * A.irow2)[x];
      const pb_u2 kk = s_k[q >> 24];
      const uint32_t qa_ = __umul24(q & 0xFF, kk.x), qb_ = __umul24((q >> 8) & 0xFF, kk.x), qc_ = __umul24((q >> 16) & 0xFF, kk.x);
      uint32_t c0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(0u, qa_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
      uint32_t c1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(0u, qb_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
      uint32_t c2 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(0u, qc_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
      c0 = s_lut[c0]; c1 = s_lut[c1]; c2 = s_lut[c2];
      reinterpret_cast<uint32_t *>(T.dst[track] + (size_t)y * A.orow)[x] = c0 | (c1 << 8) | (c2 << 16) | 0xFF000000u;
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform, and the compiler is told so: scalar row / track arithmetic
  // Work order.  A workgroup = the 4 adjacent strips of one band of one track (a "column group").  Workgroups reach the 8 XCDs round robin, each XCD with its own L2;
  // two bands that follow each other vertically share two source rows (ten with the gaussian), and odd bands walk upwards so that the pair reaches them together.
  // A.row_major picks how (track, band, column group) map to (XCD, slot): 0 -- every XCD a contiguous run of the sequence, a column group's bands one after the other
  // (rounds 3 / 4: the shared rows hit that XCD's L2, 999 -> 883 MB per 16-track launch); 1 -- the same runs, the column groups of a band one after the other
  // (consecutive workgroups read a band across the whole row); 2 -- the bands of a track dealt round robin to the XCDs, column groups fastest: the whole device
  // sweeps one frame at a time, as a linear stream of the same bytes would.  Measured in pb_chain_half()'s comment.
  const unsigned bid = blockIdx.x - (CHAIN ? (unsigned)A.bar_first : 0u);
  const int xcd = bid & 7, slot = bid >> 3;
  const int nseq = A.cgroups * A.bands * A.ntracks, per_xcd = (nseq + 7) >> 3;
  int seq = xcd * per_xcd + slot;
  int strip = 0, band = 0, track = 0;
  bool spare = (seq >= nseq || slot >= per_xcd) && A.row_major != 2;
  if (!spare) {
    if (A.row_major == 2) {                                 // groups of A.bgroup neighbouring bands dealt round robin to the XCDs, column groups fastest: all XCDs sweep ONE frame together
      const int G = A.bgroup, ngroups = (A.bands + G - 1) / G, ng_x = (ngroups + 7 - xcd) >> 3, per_track = ng_x * G * A.cgroups;
      spare = per_track == 0 || slot >= per_track * A.ntracks;
      if (!spare) {
        track = slot / per_track;
        const int idx = slot - track * per_track, bi = idx / A.cgroups, gi = bi / G;
        band = G * (xcd + 8 * gi) + (bi - gi * G);
        strip = (idx - bi * A.cgroups) * 4 + wave;
        spare = band >= A.bands;
      }
    } else {                                      // column groups fastest: consecutive workgroups of an XCD read one band across the whole row (contiguous 15 KB per source row)
      const int per_track = A.cgroups * A.bands;
      track = seq / per_track;
      const int idx = seq - track * per_track;
      band = idx / A.cgroups;
      strip = (idx - band * A.cgroups) * 4 + wave;
    }
    spare = spare || strip >= A.strips;
  }
  if (spare) return;                                        // no workgroup barrier on this path: a wave without work simply ends
  const int k = strip * kCols - kHalo + lane;             // this lane's source quad: pixels 4k .. 4k + 3 -> output columns 2k, 2k + 1
  const int kmax = (A.sw >> 2) - 1;
  const int kc = k < 0 ? 0 : k > kmax ? kmax : k;
  const int y0 = band * A.th + min(band, A.rem), rows = A.th + (band < A.rem ? 1 : 0);
  const bool out_lane = lane >= kHalo && lane < 64 - kHalo && k <= kmax;
  const bool edge_strip = strip == 0 || (strip + 1) * kCols + kHalo >= kmax;        // wave-uniform: some lanes of this strip lie outside the frame
  // BLUR: a band whose scaled rows (its own and the two above / below) keep clear of the frame's first and last row walks in straight-line code (further down).  There
  // every source row index lies in [1, sh - 2], and the lanes outside the frame read the ONE pixel their neighbour wants from them where it lies -- the lane left of
  // the frame gets P[0] as its fourth pixel (12 bytes before the row), the lane right of it P[sw - 1] as its first -- so no pixel has to be fixed up after it arrived
  const bool fastp = BLUR && y0 - 2 >= 1 && y0 + rows + 1 <= A.dh - 2;
  uint32_t bf = A.bf;
  if (CHAIN && A.bf_d) bf = (uint32_t)A.bf_d[0] & 0xFF;
  if (CHAIN && A.bf_tracks) bf = T.bf[track];
  const uint32_t w_lo = bf | ((255u - bf) << 8);

  // descriptors: base pointers made provably uniform (readfirstlane of both halves), range = the whole frame
  auto srd = [](const void *p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    void *u = (void *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  const int out_rows = A.cw ? A.ch : A.dh;
  const __amdgpu_buffer_rsrc_t r_src = srd(T.src[track], (uint32_t)A.sh * (uint32_t)A.irow);
  const __amdgpu_buffer_rsrc_t r_dst = srd(T.dst[track], (uint32_t)out_rows * (uint32_t)A.orow);
  const __amdgpu_buffer_rsrc_t r_l2 = srd(CHAIN == 1 ? (const void *)T.l2[track] : (const void *)T.src[track], CHAIN == 1 ? (uint32_t)out_rows * (uint32_t)A.irow2 : 16u);
  const uint32_t lane_off = fastp ? (k < 0 ? 4u : k > kmax ? 16u * (uint32_t)(kmax + 1) + 12u : 16u * (uint32_t)(k + 1)) : 16u * (uint32_t)kc;
  const int row_adj = __builtin_amdgcn_readfirstlane(fastp ? -16 : 0);      // fastp: the lane offsets are written against 16 bytes before the row
  auto load_row = [&](int sy) -> pb_u4 {
    sy = __builtin_amdgcn_readfirstlane(sy < 0 ? 0 : sy > A.sh - 1 ? A.sh - 1 : sy);      // uniform by construction; said so, the row offset stays scalar
    // plain loads: measured faster than non-temporal ones (band seams and strip halos are re-read through L2)
    return __builtin_amdgcn_raw_buffer_load_b128(r_src, (int)lane_off, sy * A.irow + row_adj, 0);
  };
  // ALIGNED: the pixel left of the strip (lane 0) / right of it (lane 63), clamped into the row -- which is the library's edge rule at the frame's two ends
  const bool e_lane = HYPER && ALIGNED && (lane == 0 || lane == 63);
  const int e_x = lane == 0 ? 4 * k - 1 : 4 * k + 4;
  const uint32_t e_off = 4u * (uint32_t)(e_x < 0 ? 0 : e_x > A.sw - 1 ? A.sw - 1 : e_x);
  auto load_e = [&](int sy) -> uint32_t {
    uint32_t e = 0u;
    if (e_lane) {
      sy = __builtin_amdgcn_readfirstlane(sy < 0 ? 0 : sy > A.sh - 1 ? A.sh - 1 : sy);
      e = __builtin_amdgcn_raw_buffer_load_b32(r_src, (int)e_off, sy * A.irow, 0);
    }
    return e;
  };
  // lanes outside the frame (edge strips only, a wave-uniform test) repeat the border pixel; applied when a row is consumed, so that no load is waited for early
  auto fix = [&](pb_u4 q) -> pb_u4 {
    if (edge_strip && !fastp) {
      asm volatile("" ::: "memory");                            // keeps this a (wave-uniform) branch: as selects it costs every strip six operations per source row
      if (!ALIGNED && k < 0) { q.y = q.x; q.z = q.x; q.w = q.x; }           // left of the frame: pixel 0 repeated (only P[-1] is ever used)
      if (k > kmax) { q.x = q.w; q.y = q.w; q.z = q.w; }       // right of the frame: the last pixel repeated
    }
    return q;
  };
  const uint32_t l2_off = 8u * (uint32_t)kc + 4u * (uint32_t)A.ox;
  auto load_l2 = [&](int y) -> pb_u2 {
    y = __builtin_amdgcn_readfirstlane(y < 0 ? 0 : y > A.dh - 1 ? A.dh - 1 : y);
    return __builtin_amdgcn_raw_buffer_load_b64(r_l2, (int)l2_off, (y + A.oy) * A.irow2, 2);      // read once: non-temporal
  };
  // the rest of the chain on one pixel whose colours are still apart: chroma blend (simple_blend.c:117-146): s2 = (layer-2 colour * K2[alpha2]) >> 16,
  // s1 = (track colour * K1[alpha2]) >> 16 (the reference's float scaling of translucent pixels as integers, lgpu_alpha_scalers; alpha 255 = identity), then
  // (bf * s2 + (255 - bf) * s1) >> 8, then the gamma LUT (the identity table when the chain has none)
  auto finish = [&](uint32_t c0, uint32_t c1, uint32_t c2, uint32_t al, uint32_t q) -> uint32_t {
    if (CHAIN == 1) {
      const pb_u2 kk = s_k[q >> 24];
      const uint32_t qa_ = __umul24(q & 0xFF, kk.x), qb_ = __umul24((q >> 8) & 0xFF, kk.x), qc_ = __umul24((q >> 16) & 0xFF, kk.x);
      const uint32_t pa = __umul24(c0, kk.y), pb = __umul24(c1, kk.y), pc = __umul24(c2, kk.y);
      c0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pa, qa_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
      c1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pb, qb_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
      c2 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(pc, qc_, 0x0C0C0602u), w_lo, 0u, false) >> 8;
    }
    if (CHAIN) { c0 = s_lut[c0]; c1 = s_lut[c1]; c2 = s_lut[c2]; }
    return c0 | (c1 << 8) | (c2 << 16) | al;
  };
  const uint32_t st_off = out_lane ? 8u * (uint32_t)k + 4u * (uint32_t)A.ox : 0xFFFFFFF0u;      // beyond the descriptor's range: the hardware drops the store
  auto store_row = [&](int y, uint32_t p0, uint32_t p1) {
    pb_u2 o;
    o.x = p0; o.y = p1;
    const int so = __builtin_amdgcn_readfirstlane((y + A.oy) * A.orow);
    if (CHAIN) __builtin_amdgcn_raw_buffer_store_b64(o, r_dst, (int)st_off, so, 2);       // results are not read again by this launch: non-temporal
    else __builtin_amdgcn_raw_buffer_store_b64(o, r_dst, (int)st_off, so, 0);
  };

  // scaled rows this band has to produce: its own, plus two above and below for the blur (clamped to the frame: the gaussian replicates the border rows).
  // Odd bands walk UPWARDS (both filters are symmetric, so only the row addresses change): a band and its lower neighbour then reach the two source rows
  // they share at the same time -- both at their end, or both at their start -- and the second read hits the XCD's L2 instead of HBM.
  const int vr0 = BLUR ? y0 - 2 : y0, vr1 = BLUR ? y0 + rows + 1 : y0 + rows - 1;
  const int ylo = vr0 < 0 ? 0 : vr0, yhi = vr1 > A.dh - 1 ? A.dh - 1 : vr1;
  const int d = (band & 1) ? -1 : 1;
  const int ystart = d > 0 ? ylo : yhi, vstart = d > 0 ? vr0 : vr1;
  const int S0 = d > 0 ? 2 * ylo - 1 : 2 * yhi + 2;           // source rows are consumed in the order S0, S0 + d, S0 + 2 d, ...
  // carry[i] = (outer tap) * H[first row] + (inner tap) * H[second row] of a scaled row: the half that is known before its last two source rows arrive
  uint32_t carry[8], hr[8], hs[8];
  pb_u4 q0 = load_row(S0), q1 = load_row(S0 + d), qa = load_row(S0 + 2 * d), qb = load_row(S0 + 3 * d);
  uint32_t e0 = load_e(S0), e1 = load_e(S0 + d), ea = load_e(S0 + 2 * d), eb = load_e(S0 + 3 * d);
  pb_u2 l2;
  l2.x = 0; l2.y = 0;
  if (CHAIN == 1 && !fastp) l2 = load_l2(d > 0 ? y0 : y0 + rows - 1);
  if (CHAIN) {        // the two small tables, requested while the first source rows are in flight; first read an output row later
    reinterpret_cast<uint32_t *>(s_lut)[lane] = lut.w[lane];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint2 kk = A.kscale[lane + 64 * i];
      pb_u2 kv; kv.x = kk.x; kv.y = kk.y;
      s_k[lane + 64 * i] = kv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t e_m = lane == 0 ? 65536u : lane == 63 ? 1u : 0u;
  pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(fix(q0), hr, e0, e_m);
  pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(fix(q1), hs, e1, e_m);
#pragma unroll
  for (int i = 0; i < 8; i++) if (!OPAQUE || (i & 3) != 3) carry[i] = (HYPER && OPAQUE) ? pb_mad7(hs[i], hr[i]) : HYPER ? __umul24(hs[i], 7u) + hr[i] : hs[i];

  // one scaled row from its last two source rows (the first two are in `carry`): colours apart in cc, alpha in place (<< 24) in al
  auto scale_row = [&](const pb_u4 &ra, const pb_u4 &rb, uint32_t xa, uint32_t xb, uint32_t cc[2][3], uint32_t al[2]) __attribute__((always_inline)) {
    pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(fix(ra), hr, xa, e_m);
    pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(fix(rb), hs, xb, e_m);
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OPAQUE && (i & 3) == 3) { v[i] = 0u; continue; }          // no alpha sums: the alpha is 255 by the caller's word
      if (HYPER && OPAQUE) { v[i] = pb_mad7(hr[i], carry[i] + hs[i]); carry[i] = pb_mad7(hs[i], hr[i]); continue; }
      if (HYPER) { v[i] = carry[i] + __umul24(hr[i], 7u) + hs[i]; carry[i] = __umul24(hs[i], 7u) + hr[i]; }
      else { v[i] = carry[i] + hr[i]; carry[i] = hs[i]; }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (OPAQUE) { cc[j][0] = v[4 * j] >> (HYPER ? 8 : 2); cc[j][1] = v[4 * j + 1] >> (HYPER ? 8 : 2); cc[j][2] = v[4 * j + 2] >> (HYPER ? 8 : 2); al[j] = 0xFF000000u; continue; }
      const uint32_t va = v[4 * j + 3];
      pb_half_colours(v[4 * j], v[4 * j + 1], v[4 * j + 2], va ? va : 1u, cc[j]);       // V_alpha == 0 makes every V_c 0 too: 0 * fl(1 / 1) = 0, the library's all-zero pixel
      al[j] = (va >> A.ashift) << 24;
    }
  };

  if (!BLUR) {
    // two scaled rows per trip, the source rows of one in (qa, qb), of the other in (na, nb): a row's loads land in the registers its arithmetic reads, issued a whole
    // row of arithmetic earlier, and nothing is moved between registers
    pb_u4 na = qa, nb = qb;
    uint32_t nea = 0, neb = 0;
    pb_u2 nl2;
    nl2.x = 0; nl2.y = 0;
    auto one = [&](int r, pb_u4 &ca, pb_u4 &cb, uint32_t &cea, uint32_t &ceb, pb_u2 &cl2, pb_u4 &xa, pb_u4 &xb, uint32_t &xea, uint32_t &xeb, pb_u2 &xl2) __attribute__((always_inline)) {
      const int yy = d > 0 ? ystart + r : ystart - r;
      if (r + 1 < rows) {       // the next scaled row's two new source rows and the layer-2 pixels of the next output row: in flight during this row's arithmetic
        xa = load_row(S0 + d * (2 * r + 4)); xb = load_row(S0 + d * (2 * r + 5));
        xea = load_e(S0 + d * (2 * r + 4)); xeb = load_e(S0 + d * (2 * r + 5));
        if (CHAIN == 1) xl2 = load_l2(yy + d);
      }
      uint32_t cc[2][3], al[2];
      scale_row(ca, cb, cea, ceb, cc, al);
      store_row(yy, finish(cc[0][0], cc[0][1], cc[0][2], al[0], cl2.x), finish(cc[1][0], cc[1][1], cc[1][2], al[1], cl2.y));
    };
    int r = 0;
    for (; r + 1 < rows; r += 2) {
      one(r, qa, qb, ea, eb, l2, na, nb, nea, neb, nl2);
      one(r + 1, na, nb, nea, neb, nl2, qa, qb, ea, eb, l2);
    }
    if (r < rows) one(r, qa, qb, ea, eb, l2, na, nb, nea, neb, nl2);
    return;
  }

  // horizontal gaussian of one scaled row on bytes in 16-bit lanes: e = (byte 0, byte 2), o = (byte 1, byte 3) of a pixel; columns 2k-2 .. 2k+3 around this lane's two
  auto hblur = [&](const uint32_t cc[2][3], const uint32_t al[2], uint32_t slot[4]) __attribute__((always_inline)) {
    uint32_t e[6], o[6];
    e[2] = cc[0][0] | (cc[0][2] << 16); o[2] = cc[0][1] | (al[0] >> 8); e[3] = cc[1][0] | (cc[1][2] << 16); o[3] = cc[1][1] | (al[1] >> 8);
    e[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[2], 0x138, 0xF, 0xF, true); o[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[2], 0x138, 0xF, 0xF, true);
    e[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[3], 0x138, 0xF, 0xF, true); o[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[3], 0x138, 0xF, 0xF, true);
    e[4] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[2], 0x130, 0xF, 0xF, true); o[4] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[2], 0x130, 0xF, 0xF, true);
    e[5] = (uint32_t)__builtin_amdgcn_mov_dpp((int)e[3], 0x130, 0xF, 0xF, true); o[5] = (uint32_t)__builtin_amdgcn_mov_dpp((int)o[3], 0x130, 0xF, 0xF, true);
    if (edge_strip) {       // the gaussian replicates the frame's first / last column
      if (k == 0) { e[0] = e[2]; e[1] = e[2]; o[0] = o[2]; o[1] = o[2]; }
      if (k == kmax) { e[4] = e[3]; e[5] = e[3]; o[4] = o[3]; o[5] = o[3]; }
    }
    slot[0] = gauss5_taps(e[0], e[1], e[2], e[3], e[4]); slot[1] = gauss5_taps(o[0], o[1], o[2], o[3], o[4]);
    slot[2] = gauss5_taps(e[1], e[2], e[3], e[4], e[5]); slot[3] = gauss5_taps(o[1], o[2], o[3], o[4], o[5]);
  };
  // vertical gaussian over five ring rows (oldest first), blend with the layer-2 pixels, LUT, store
  auto vblur_store = [&](int y, const uint32_t *r0, const uint32_t *r1, const uint32_t *r2, const uint32_t *r3, const uint32_t *r4, const pb_u2 &lp) __attribute__((always_inline)) {
    uint32_t pxo[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t ve = gauss5_taps(r0[2 * j], r1[2 * j], r2[2 * j], r3[2 * j], r4[2 * j], 0x00800080u);
      const uint32_t vo = gauss5_taps(r0[2 * j + 1], r1[2 * j + 1], r2[2 * j + 1], r3[2 * j + 1], r4[2 * j + 1], 0x00800080u);
      // the high byte of each 16-bit lane is the blurred value: ve -> (c0, c2), vo -> (c1, alpha)
      pxo[j] = finish((ve >> 8) & 0xFF, (vo >> 8) & 0xFF, ve >> 24, vo & 0xFF000000u, j ? lp.y : lp.x);
    }
    store_row(y, pxo[0], pxo[1]);
  };
  uint32_t ring[5][4];                     // horizontally blurred rows; [row][column * 2 + (0: bytes 0 and 2, 1: bytes 1 and 3)] in 16-bit lanes
  const int nsteps = vr1 - vr0 + 1;

  if (fastp) {
    // Every step scales a NEW row; the two source rows of step s + 1 are requested during step s INTO the registers that held step s's rows, each as soon as its
    // horizontal pass has read it (no second pair of row registers, no copies), the layer-2 pixels of a step's own output row at its top: each is read almost a step
    // of arithmetic later, and no wait sits between a store and the next loads.  (The general loop below copies the new rows right behind the scaler and the
    // layer-2 pixels behind the store: a step then waits for the loads it has just issued and for its own store -- profiles/r05/blur_investigation.md.)
    uint32_t cc[2][3], al[2];
    auto scale_refill = [&](int s, bool more) __attribute__((always_inline)) {
      pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(qa, hr, 0u, e_m);
      if (more) qa = load_row(S0 + d * (2 * s + 4));
      pb_half_hrow<HYPER, ALIGNED, SWAP, OPAQUE>(qb, hs, 0u, e_m);
      if (more) qb = load_row(S0 + d * (2 * s + 5));
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OPAQUE && (i & 3) == 3) { v[i] = 0u; continue; }
        if (HYPER && OPAQUE) { v[i] = pb_mad7(hr[i], carry[i] + hs[i]); carry[i] = pb_mad7(hs[i], hr[i]); continue; }
        if (HYPER) { v[i] = carry[i] + __umul24(hr[i], 7u) + hs[i]; carry[i] = __umul24(hs[i], 7u) + hr[i]; }
        else { v[i] = carry[i] + hr[i]; carry[i] = hs[i]; }
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (OPAQUE) { cc[j][0] = v[4 * j] >> (HYPER ? 8 : 2); cc[j][1] = v[4 * j + 1] >> (HYPER ? 8 : 2); cc[j][2] = v[4 * j + 2] >> (HYPER ? 8 : 2); al[j] = 0xFF000000u; continue; }
        const uint32_t va = v[4 * j + 3];
        pb_half_colours(v[4 * j], v[4 * j + 1], v[4 * j + 2], va ? va : 1u, cc[j]);
        al[j] = (va >> A.ashift) << 24;
      }
    };
#pragma unroll
    for (int s = 0; s < 4; s++) {            // the four rows above (below) the band's first output row: no output yet (nsteps >= 5: a next step exists)
      scale_refill(s, true);
      hblur(cc, al, ring[s]);
    }
    constexpr int kSlot[5] = {4, 0, 1, 2, 3};          // ring slot of step 4 + u (= step % 5); the oldest of the five rows is the slot after it
    int s0 = 4;
    for (; s0 + 5 < nsteps; s0 += 5) {       // whole groups of five steps, each with a step behind it: no conditions inside
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int s = s0 + u, t = kSlot[u];
        l2 = load_l2(vstart + d * (s - 2));                // this step's output row
        scale_refill(s, true);
        hblur(cc, al, ring[t]);
        vblur_store(vstart + d * (s - 2), ring[(t + 1) % 5], ring[(t + 2) % 5], ring[(t + 3) % 5], ring[(t + 4) % 5], ring[t], l2);
      }
    }
#pragma unroll
    for (int u = 0; u < 5; u++) {            // the last one to five steps
      const int s = s0 + u, t = kSlot[u];
      if (s >= nsteps) break;
      l2 = load_l2(vstart + d * (s - 2));
      scale_refill(s, s + 1 < nsteps);
      hblur(cc, al, ring[t]);
      vblur_store(vstart + d * (s - 2), ring[(t + 1) % 5], ring[(t + 2) % 5], ring[(t + 3) % 5], ring[(t + 4) % 5], ring[t], l2);
    }
    return;
  }

  int produced = ystart - d;              // the last scaled row that exists
  uint32_t cc[2][3] = {{0, 0, 0}, {0, 0, 0}}, al[2] = {0, 0};      // the current scaled row of this lane
#pragma unroll
  for (int i = 0; i < 5; i++) { ring[i][0] = 0; ring[i][1] = 0; ring[i][2] = 0; ring[i][3] = 0; }
  // the first and the last band of a frame: rows beyond the frame repeat the border row.  The ring rotates by slot index, five steps per trip of the outer loop:
  // slot u takes the new row, (u + 1) % 5 is the oldest -- no register moves
  for (int step0 = 0; step0 < nsteps; step0 += 5) {
#pragma unroll
    for (int u = 0; u < 5; u++) {
      const int step = step0 + u;
      if (step >= nsteps) break;
      const int vr = vstart + d * step;
      const int yy = vr < 0 ? 0 : vr > A.dh - 1 ? A.dh - 1 : vr;
      pb_u2 nl2;
      nl2.x = 0; nl2.y = 0;
      if (yy != produced) {
        // the next scaled row's two new source rows and the layer-2 pixels of the next output row: in flight during this row's arithmetic
        const int r = d > 0 ? yy - ystart : ystart - yy;
        if (CHAIN == 1) nl2 = load_l2(vr - d);
        scale_row(qa, qb, 0u, 0u, cc, al);
        qa = load_row(S0 + d * (2 * r + 4)); qb = load_row(S0 + d * (2 * r + 5));       // into the registers just read (these two bands per track are not where the time goes)
        produced = yy;
        hblur(cc, al, ring[u]);
      } else {          // a row beyond the frame's first / last: the border row again
        if (CHAIN == 1) nl2 = load_l2(vr - d);
#pragma unroll
        for (int i = 0; i < 4; i++) ring[u][i] = ring[(u + 4) % 5][i];
      }
      if (step >= 4)                      // the ring holds the five rows around output row vr - 2 d: oldest (u + 1) % 5 ... newest u
        vblur_store(vr - 2 * d, ring[(u + 1) % 5], ring[(u + 2) % 5], ring[(u + 3) % 5], ring[(u + 4) % 5], ring[u], l2);
      if (step >= 3) l2 = nl2;
    }
  }
}

// =====================================================================================================================================================
// k_pb_half3 -- the same exact 2:1 reduction for 3-byte pixels (RGB24 / BGR24 / YUV888: a pixbuf WITHOUT alpha), standalone form only.  No alpha weighting:
// P = the byte, colour = (scale * V + 0xffff) >> 16 inside the row, (255 * scale * V + 0xffffff) >> 24 on the columns whose taps leave it (the library's
// per-pixel path: column 0 for HYPER, the last column for both).  A lane loads 12 bytes (4 pixels) per row and makes 2 output pixels; lanes pair up (DPP inside
// the quad) so that two lanes' 12 output bytes leave as three dword stores.  Strips of 120 columns (lanes 2 .. 61 store; the pairing needs even lanes on even quads).
struct __attribute__((aligned(4))) pb_u3 { uint32_t x, y, z; };
template <int HYPER>
__global__ __launch_bounds__(256) void k_pb_half3(const PbHalfArgs A, const PbFrames F) {
  const uint8_t *src_ = F.src[blockIdx.z];
  uint8_t *dst_ = F.dst[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nseq = A.cgroups * A.bands, per_xcd = (nseq + 7) >> 3;
  const int seq = xcd * per_xcd + slot;
  if (seq >= nseq || slot >= per_xcd) return;
  const int cg = seq / A.bands, band = seq - cg * A.bands, strip = cg * 4 + wave;
  if (strip >= A.strips) return;
  const int k = strip * 60 - 2 + lane, kmax = (A.sw >> 2) - 1;
  const int kc = k < 0 ? 0 : k > kmax ? kmax : k;
  const bool out_lane = lane >= 2 && lane <= 61 && k <= kmax;
  const bool edge_strip = strip == 0 || (strip + 1) * 60 + 2 >= kmax;
  const int y0 = band * A.th, rows = min(A.th, A.dh - y0);
  const uint32_t lane_off = 12u * (uint32_t)kc;
  auto load_row = [&](int sy) -> pb_u4 {
    sy = sy < 0 ? 0 : sy > A.sh - 1 ? A.sh - 1 : sy;
    const pb_u3 t = *reinterpret_cast<const pb_u3 *>(src_ + (size_t)sy * A.irow + lane_off);
    uint32_t q[4];
    unpack3(t.x, t.y, t.z, q);
    pb_u4 r;
    r.x = q[0]; r.y = q[1]; r.z = q[2]; r.w = q[3];
    return r;
  };
  auto fix = [&](pb_u4 q) -> pb_u4 {
    if (edge_strip) {
      if (k < 0) { q.y = q.x; q.z = q.x; q.w = q.x; }
      if (k > kmax) { q.x = q.w; q.y = q.w; q.z = q.w; }
    }
    return q;
  };
  auto hrow = [&](pb_u4 q, uint32_t h[6]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint32_t sel = 0x0C040C00u + 0x00010001u * c;
      const uint32_t Ap = __builtin_amdgcn_perm(q.y, q.x, sel), Bp = __builtin_amdgcn_perm(q.w, q.z, sel);      // (byte c of pixel 0 | of pixel 1 << 16), (pixel 2 | pixel 3)
      if (HYPER) {
        const uint32_t bl = (uint32_t)__builtin_amdgcn_mov_dpp((int)Bp, 0x138, 0xF, 0xF, true), ar = (uint32_t)__builtin_amdgcn_mov_dpp((int)Ap, 0x130, 0xF, 0xF, true);
        h[c] = pb_dot2(Ap, 0x00070007u, pb_add_hi_lo(bl, Bp));
        h[3 + c] = pb_dot2(Bp, 0x00070007u, pb_add_hi_lo(Ap, ar));
      } else {
        h[c] = pb_dot2(Ap, 0x00010001u, 0u);
        h[3 + c] = pb_dot2(Bp, 0x00010001u, 0u);
      }
    }
  };
  const int ylo = y0, yhi = y0 + rows - 1;
  const int d = (band & 1) ? -1 : 1;
  const int ystart = d > 0 ? ylo : yhi;
  const int S0 = d > 0 ? 2 * ylo - 1 : 2 * yhi + 2;
  uint32_t carry[6], hr[6], hs[6];
  pb_u4 q0 = load_row(S0), q1 = load_row(S0 + d), qa = load_row(S0 + 2 * d), qb = load_row(S0 + 3 * d);
  hrow(fix(q0), hr);
  hrow(fix(q1), hs);
#pragma unroll
  for (int i = 0; i < 6; i++) carry[i] = HYPER ? __umul24(hs[i], 7u) + hr[i] : hs[i];
  const uint32_t scale = HYPER ? 256u : 16384u;
  const int X0 = 2 * k;                                                   // this lane's output columns X0, X0 + 1
  const bool e0 = HYPER && X0 == 0, e1 = X0 + 1 == A.dw - 1;              // columns whose taps leave the row
  for (int r = 0; r < rows; r++) {
    const pb_u4 na = load_row(S0 + d * (2 * r + 4)), nb = load_row(S0 + d * (2 * r + 5));
    hrow(fix(qa), hr);
    hrow(fix(qb), hs);
    uint32_t px[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      uint32_t c[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int t = 3 * j + i;
        uint32_t v;
        if (HYPER) { v = carry[t] + __umul24(hr[t], 7u) + hs[t]; carry[t] = __umul24(hs[t], 7u) + hr[t]; }
        else { v = carry[t] + hr[t]; carry[t] = hs[t]; }
        const uint32_t rr = v * scale;                                      // the library's r: <= 255 * 65536
        c[i] = (j ? e1 : e0) ? (rr * 255u + 0xffffffu) >> 24 : (rr + 0xffffu) >> 16;
      }
      px[j] = c[0] | (c[1] << 8) | (c[2] << 16);
    }
    // two lanes' four pixels -> three dwords: lanes 2m (columns X, X + 1) and 2m + 1 (X + 2, X + 3) of a quad
    uint32_t q[4], w0, w1, w2;
    q[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px[0], 0xA0, 0xF, 0xF, true); q[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px[1], 0xA0, 0xF, 0xF, true);     // quad_perm [0, 0, 2, 2]
    q[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px[0], 0xF5, 0xF, 0xF, true); q[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px[1], 0xF5, 0xF, 0xF, true);     // quad_perm [1, 1, 3, 3]
    pack3(q, w0, w1, w2);
    const int y = d > 0 ? ystart + r : ystart - r;
    if (out_lane) {
      uint32_t *dp = reinterpret_cast<uint32_t *>(dst_ + (size_t)y * A.orow) + 3 * (k >> 1);
      if (lane & 1) dp[1] = w1; else { dp[0] = w0; dp[2] = w2; }
    }
    qa = na; qb = nb;
  }
}

// k_pb_double -- the exact 1:2 enlargement of 4-byte pixels (1080p -> 4K), HYPER or BILINEAR: both tables are exactly 4096 * [1 3] x [1 3] outer products on the two
// nearest source pixels per axis (host-checked), so with P = alpha * q:  out(2i) = P[i-1] + 3 P[i],  out(2i+1) = 3 P[i] + P[i+1]  along a row, the same down the
// rows, colour = (uint8_t)((double)V_c * (1.0 / (double)V_alpha)), alpha' = V_alpha >> 4 (the common 4096 drops out).  A lane owns two source pixels (one 8-byte load per
// source row, the neighbours by DPP), i.e. four output columns, and every new source row completes two output rows (16-byte stores).  The arithmetic that is left is
// the library's double division, once per output pixel; frames that are opaque where they are sampled take the constant reciprocal.
template <int DUMMY>
__global__ __launch_bounds__(256) void k_pb_double(const PbHalfArgs A, const PbFrames F) {
  const uint8_t *src_ = F.src[blockIdx.z];
  uint8_t *dst_ = F.dst[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nseq = A.cgroups * A.bands, per_xcd = (nseq + 7) >> 3;
  const int seq = xcd * per_xcd + slot;
  if (seq >= nseq || slot >= per_xcd) return;
  const int cg = seq / A.bands, band = seq - cg * A.bands, strip = cg * 4 + wave;
  if (strip >= A.strips) return;
  const int k = strip * 62 - 1 + lane, kmax = (A.sw >> 1) - 1;            // this lane's source pixels 2k, 2k + 1 -> output columns 4k .. 4k + 3
  const int kc = k < 0 ? 0 : k > kmax ? kmax : k;
  const bool out_lane = lane >= 1 && lane <= 62 && k <= kmax;
  const bool edge_strip = strip == 0 || (strip + 1) * 62 + 1 >= kmax;
  const int r0 = band * A.th, nrows = min(A.th, A.sh - r0);               // source rows r0 .. r0 + nrows - 1 -> output rows 2 r0 .. 2 (r0 + nrows) - 1
  const uint32_t lane_off = 8u * (uint32_t)kc;
  auto load_row = [&](int sy) -> pb_u2 {
    sy = sy < 0 ? 0 : sy > A.sh - 1 ? A.sh - 1 : sy;
    return *reinterpret_cast<const pb_u2 *>(src_ + (size_t)sy * A.irow + lane_off);
  };
  // one source row of a lane -> its four H columns per channel: h[4 * col + c]
  auto hrow = [&](pb_u2 q, uint32_t h[16]) {
    if (edge_strip) {
      if (k < 0) q.y = q.x;              // left of the frame: pixel 0 again (only its right pixel is ever read by lane k = 0)
      if (k > kmax) q.x = q.y;           // right of the frame: the last pixel again
    }
    uint32_t pr[4];
    pr[0] = pb_premul_pair<0>(q.x, q.y); pr[1] = pb_premul_pair<1>(q.x, q.y); pr[2] = pb_premul_pair<2>(q.x, q.y); pr[3] = __builtin_amdgcn_perm(q.y, q.x, 0x0C070C03u);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t left = (uint32_t)__builtin_amdgcn_mov_dpp((int)pr[c], 0x138, 0xF, 0xF, true) >> 16;       // P[2k - 1]: the left lane's second pixel
      const uint32_t right = (uint32_t)__builtin_amdgcn_mov_dpp((int)pr[c], 0x130, 0xF, 0xF, true) & 0xFFFFu;   // P[2k + 2]: the right lane's first pixel
      const uint32_t p0 = pr[c] & 0xFFFFu, p1 = pr[c] >> 16;
      h[c] = left + 3u * p0; h[4 + c] = 3u * p0 + p1; h[8 + c] = p0 + 3u * p1; h[12 + c] = 3u * p1 + right;
    }
  };
  auto emit = [&](int oy, const uint32_t a3[16], const uint32_t b1[16]) {     // output row oy = 3 * a3 + b1
    pb_u4 o;
    uint32_t px[4];
#pragma unroll
    for (int col = 0; col < 4; col++) {
      const uint32_t va = 3u * a3[4 * col + 3] + b1[4 * col + 3];
      uint32_t p = 0;
      if (va) {
        const double ia = (va == 4080u) ? (1.0 / 4080.0) : pb_recip(va);
        const uint32_t c0 = (uint32_t)(int)((double)(3u * a3[4 * col] + b1[4 * col]) * ia), c1 = (uint32_t)(int)((double)(3u * a3[4 * col + 1] + b1[4 * col + 1]) * ia),
                       c2 = (uint32_t)(int)((double)(3u * a3[4 * col + 2] + b1[4 * col + 2]) * ia);
        p = c0 | (c1 << 8) | (c2 << 16) | ((va >> 4) << 24);
      }
      px[col] = p;
    }
    o.x = px[0]; o.y = px[1]; o.z = px[2]; o.w = px[3];
    if (out_lane) __builtin_nontemporal_store(o, reinterpret_cast<pb_u4 *>(dst_ + (size_t)oy * A.orow + 16 * (size_t)k));
  };
  uint32_t hp[16], hc[16];
  // a lane's row is 8 bytes: the next three source rows are always in flight (a band is short and its loads are a dependent chain otherwise)
  const pb_u2 qm = load_row(r0 - 1), q0 = load_row(r0);
  pb_u2 qn = load_row(r0 + 1), qn2 = load_row(r0 + 2), qn3 = load_row(r0 + 3);
  hrow(qm, hp);
  hrow(q0, hc);
  for (int r = 0; r < nrows; r++) {
    // rows r0 + r - 1 (hp), r0 + r (hc) are here; r0 + r + 1 arrives: output rows 2 (r0 + r) = hp + 3 hc and 2 (r0 + r) + 1 = 3 hc + hn
    const pb_u2 q = qn;
    qn = qn2; qn2 = qn3;
    qn3 = load_row(r0 + r + 4);
    uint32_t hn[16];
    hrow(q, hn);
    emit(2 * (r0 + r), hc, hp);
    emit(2 * (r0 + r) + 1, hc, hn);
#pragma unroll
    for (int i = 0; i < 16; i++) { hp[i] = hc[i]; hc[i] = hn[i]; }
  }
}

// chroma blend of simple_blend.c:117-146 on an RGBA pair (the staged path's form): opaque layer-2 pixels through the integer table expression, translucent ones
// through the reference's float scaling of both sources first; dst alpha = the track's alpha
__device__ __forceinline__ uint32_t pb_chroma_rgba(uint32_t p1, uint32_t p2, uint32_t bf, uint32_t nbf) {
  const uint32_t al = p2 >> 24;
  uint32_t s1 = p1, s2 = p2;
  if (al != 255) {
    const float alpha = (float)((double)(float)al / 255.), inv = (float)(1. - (double)alpha);
    s1 = 0; s2 = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s2 |= ((uint32_t)(int)__fmul_rn((float)((p2 >> (8 * c)) & 0xFF), alpha) & 0xFF) << (8 * c);
      s1 |= ((uint32_t)(int)__fmul_rn((float)((p1 >> (8 * c)) & 0xFF), inv) & 0xFF) << (8 * c);
    }
  }
  const uint32_t lo = ((__umul24(s2 & 0x00FF00FFu, bf) + __umul24(s1 & 0x00FF00FFu, nbf)) >> 8) & 0x00FF00FFu;
  const uint32_t hi = ((__umul24((s2 >> 8) & 0xFFu, bf) + __umul24((s1 >> 8) & 0xFFu, nbf))) & 0x0000FF00u;
  return lo | hi | (p1 & 0xFF000000u);
}

// The chain's last stages inside a scaler's store (lgpu_chain off the exact 2:1 case, no gaussian, no canvas): the destination pixel goes [R <-> B] -> chroma blend with
// layer 2 -> gamma LUT before it is written -- no scratch frame, no second launch.  Kernels take it as their last argument: PbNoEpi (nothing) or PbEpi.
#define PB_NOT_FUSED 0x7ff0
struct PbNoEpi {};
struct PbEpi {
  const uint8_t *l2[LGPU_CHAIN_MAX_TRACKS];    // layer 2 of frame z (the grid's z index)
  uint8_t bf[LGPU_CHAIN_MAX_TRACKS];           // bf_tracks: a blend amount per frame
  const int32_t *bf_d;                         // else, when set: the amount's low byte read on the device
  uint32_t bf0;
  int irow2, swap_rb, use_lut, bf_tracks, blend;        // blend 0: no layer 2 (LGPU_INTERP_NOBLEND)
  Lut8 lut;
};
template <typename EA> struct pb_has_epi { static constexpr bool value = true; };
template <> struct pb_has_epi<PbNoEpi> { static constexpr bool value = false; };
__device__ __forceinline__ void pb_epi_stage(uint8_t *, const PbNoEpi &) {}
__device__ __forceinline__ void pb_epi_stage(uint8_t *s_lut, const PbEpi &E) { stage_lut(s_lut, E.lut); }      // the caller's next barrier covers it
__device__ __forceinline__ uint32_t pb_epi_px(const PbNoEpi &, const uint8_t *, uint32_t px, int, int) { return px; }
__device__ __forceinline__ uint32_t pb_epi_px(const PbEpi &E, const uint8_t *s_lut, uint32_t px, int i, int j) {
  const int z = blockIdx.z;
  const uint32_t bf = E.bf_tracks ? (uint32_t)E.bf[z] : E.bf_d ? ((uint32_t)E.bf_d[0] & 0xFF) : E.bf0;
  if (E.swap_rb) px = __builtin_amdgcn_perm(px, px, 0x03000102u);
  if (E.blend) px = pb_chroma_rgba(px, reinterpret_cast<const uint32_t *>(E.l2[z] + (size_t)i * E.irow2)[j], bf, 255u - bf);
  if (E.use_lut) px = lut3_rgba(s_lut, px);
  return px;
}

// the bars of a letterbox canvas for the one-launch forms off 2:1: opaque black through the chain's last stages, every canvas pixel outside the inner rectangle
// (groups of four pixels along a row; frame = grid z)
__global__ __launch_bounds__(256) void k_pb_bars_epi(const PbTracks T, const PbEpi E, int orow, int cw, int ch, int ox, int oy, int iw, int ih) {
  __shared__ uint8_t s_lut[256];
  pb_epi_stage(s_lut, E);
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cw || y >= ch) return;
  if (y >= oy && y < oy + ih && x >= ox && x < ox + iw) return;
  reinterpret_cast<uint32_t *>(T.dst[blockIdx.z] + (size_t)y * orow)[x] = pb_epi_px(E, s_lut, 0xFF000000u, y, x);
}

// the chain on frames that are NOT resized (lgpu_chain_amounts with sw == dw, sh == dh: the plan steps of a track that already has the canvas's size, or is only
// letterboxed): canvas pixel <- source pixel (inside the inner rectangle) or opaque black, then the chain's last stages; frame = grid z
__global__ __launch_bounds__(256) void k_pb_flat_n(const PbTracks T, const PbEpi E, int irow, int orow, int cw, int ch, int ox, int oy, int iw, int ih) {
  __shared__ uint8_t s_lut[256];
  pb_epi_stage(s_lut, E);
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cw || y >= ch) return;
  const bool in = y >= oy && y < oy + ih && x >= ox && x < ox + iw;
  const uint32_t p = in ? reinterpret_cast<const uint32_t *>(T.src[blockIdx.z] + (size_t)(y - oy) * irow)[x - ox] : 0xFF000000u;
  reinterpret_cast<uint32_t *>(T.dst[blockIdx.z] + (size_t)y * orow)[x] = pb_epi_px(E, s_lut, p, y, x);
}

// the rest of the chain behind a resize that was not fused: [R <-> B] -> chroma blend with layer 2 -> gamma LUT, one RGBA pixel per thread;
// the tracks of one staged group: track = grid z, its scaled frame at scratch + z * per, layer 2 / destination / blend amount from the track table
__global__ __launch_bounds__(256) void k_pb_epilogue_n(const uint8_t *scratch, size_t per, int irow, const PbTracks T, int bf_tracks, int irow2, int orow, int width, int height,
                                                       int swap_rb, uint32_t bf, const int32_t *bf_d, int use_lut, const Lut8 lut) {
  __shared__ uint8_t s_lut[256];
  stage_lut(s_lut, lut);
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
  if (x >= width || y >= height) return;
  if (bf_d) bf = (uint32_t)bf_d[0] & 0xFF;
  if (bf_tracks & 1) bf = T.bf[z];
  uint32_t p = reinterpret_cast<const uint32_t *>(scratch + (size_t)z * per + (size_t)y * irow)[x];
  if (swap_rb) p = __builtin_amdgcn_perm(p, p, 0x03000102u);
  if (!(bf_tracks & 2)) p = pb_chroma_rgba(p, reinterpret_cast<const uint32_t *>(T.l2[z] + (size_t)y * irow2)[x], bf, 255u - bf);      // (bit 1: no layer 2)
  if (use_lut) p = lut3_rgba(s_lut, p);
  reinterpret_cast<uint32_t *>(T.dst[z] + (size_t)y * orow)[x] = p;
}

// =====================================================================================================================================================
// k_pb_pairs -- every ratio whose per-phase weights fit 16 bits (all but tables with a single 65536 tap): two taps per v_dot2_u32_u16.
// The window sits in LDS as premultiplied 16-bit values on ALIGNED pixel pairs: per pair 16 bytes = (P_c[2p] | P_c[2p+1] << 16) for c = 0, 1, 2 and the alpha pair
// (3-byte pixels: P = the byte itself, no alpha).  A destination pixel's taps start at an even or odd source pixel; the weight rows are stored for both parities
// as pairs aligned the same way (a zero weight pads the odd end), [y phase][tap row][x phase][parity][pair] (the 64 vectors a wave requests for one tap row then
// lie in 512 bytes instead of being spread over 3 KB: 24 cache lines per request became 4), so a tap row costs one ds_read_b128 per pair, one
// weight dword per pair (read four at a time) and four dot2 -- about 2 VALU per tap and channel instead of 9 in k_pb_window.
// =====================================================================================================================================================
struct PbPairArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, sw, sh, dw, dh;
  int x_step, y_step, xoff, yoff;
  int n_x, tx0, ty0, ny_eff, nq;       // nq: groups of four pairs per tap row
  const uint32_t *pairs;               // device: [16 y phases][ny_eff][16 x phases][2 parities][4 nq]
  unsigned rnd;
  int tile_h, wpairs, win_h;
  int gx, gy, per_xcd;                 // tiles per row / per column; per_xcd > 0: the grid is ONE-dimensional, 8 * per_xcd workgroups, and workgroup b works on tile
                                       // (b & 7) * per_xcd + (b >> 3) of the row-major tile sequence -- every XCD (workgroups reach them round robin) a contiguous run of
                                       // tiles, so that the window rows two vertically neighbouring tiles share are found in that XCD's L2
};

// NPC: pairs per tap row when there are at most four (compile time: no work on the padding of the weight row), 0: any count, four at a time.
// NY: tap rows when known at compile time (then a destination pixel's weight vectors are requested together, before the first tap, instead of one exposed load
// per tap row), 0: any count.  Measured (profiles/r03/pb_pairs_ab.txt): a gain for 2-3 tap rows (enlarging: 29.3 -> 27.3 us), a loss for 5-6 (4K -> 1706x960: 24.6 -> 37.2 us;
// the 24 weight registers cost more than the exposed loads), so only the short filters are instantiated that way.
// ONE: tiles of at most four rows -- a wave has ONE destination row, and the code that requests and takes over the next row's weight vectors (24 register moves per row that the
// compiler does not branch around) is not there
// OPQ (4-byte pixels; lgpu_pixbuf_scale with LGPU_INTERP_OPAQUE: the caller states that every source pixel has alpha 255): the window holds the colour BYTES as 16-bit
// pairs (one v_perm per pair and channel, no alpha product), the alpha sums are not formed (three dot products per pair instead of four), and the library's
// (uint8_t)((double)(255 T) * fl(1 / (255 * 65536))) is T >> 16 -- equal for every T < 2^24, checked on the device before the first such launch (pb_opaque_check)
template <int CH, int NPC, int NY, int ONE = 0, int OPQ = 0, typename EA = PbNoEpi>
__global__ __launch_bounds__(256) void k_pb_pairs(const PbPairArgs A_, const PbFrames F, const EA E) {
  static_assert(!OPQ || CH == 4, "the all-opaque form is a form of the 4-byte kernel");
  static_assert(!pb_has_epi<EA>::value || CH == 4, "the chain's stages ride on 4-byte pixels");
  PB_FRAME_ARGS(PbPairArgs);
  extern __shared__ pb_u4 winp[];                      // [win_h][wpairs]
  __shared__ uint8_t s_lut[pb_has_epi<EA>::value ? 256 : 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // uniform, and the compiler is told so: row arithmetic on the scalar unit
  int bx = blockIdx.x, by = blockIdx.y;
  pb_epi_stage(s_lut, E);                              // (the barrier behind the window covers it)
  if (A.per_xcd) {
    const int t = (int)(blockIdx.x & 7u) * A.per_xcd + (int)(blockIdx.x >> 3);
    if (t >= A.gx * A.gy) return;                       // padding of the last XCD's run (before any barrier: the whole workgroup leaves)
    by = t / A.gx; bx = t - by * A.gx;
  }
  const int j0 = bx * 64, i0 = by * A.tile_h;
  const int wx0 = ((int)(((long long)j0 * A.x_step + A.xoff) >> 16) + A.tx0) & ~3;          // the window starts on a source pixel that is a multiple of 4: aligned pairs, 16-byte loads
  const int ys0 = (int)(((long long)i0 * A.y_step + A.yoff) >> 16) + A.ty0;
  const int j = j0 + lane;
  const bool live = j < A.dw;                          // lanes past the row's end compute its last pixel and store nothing: every lane stays active for the bpermutes below
  const long long x = (long long)(live ? j : A.dw - 1) * A.x_step + A.xoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15;
  const bool edge = xs < 0 || xs + A.n_x > A.sw;
  const int pos = xs + A.tx0, par = pos & 1, pidx = (pos - par - wx0) >> 1;
  const pb_u4 *pairs4 = reinterpret_cast<const pb_u4 *>(A.pairs);      // uniform base, 32-bit per-lane index: scalar-base addressing
  // The weight vectors (rows of one vector, NPC > 0).  A (y phase, tap row) block holds one 16-byte vector per (x phase, parity): 32 vectors, 512 bytes.  Fetched per
  // lane -- lane -> vector xph * 2 + par -- that is a gather whose neighbouring lanes sit in different cache lines: the texture unit looks up 64 tags per
  // instruction (profiles/r06/pb_strip_diary.md).  Lane l requests vector l & 31 instead -- one contiguous 512-byte read -- and every lane takes its own vector from the lane
  // that holds it (ds_bpermute_b32, one per pair).
  const int lwsrc = (xph * 2 + par) * 4;
  auto take = [&](const pb_u4 w, uint32_t *o) {
    const uint32_t wq[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < (NPC ? NPC : 1); k++) o[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(lwsrc, (int)wq[k]);
  };
  // tap rows known at compile time: the weight vectors of the wave's first destination row are requested BEFORE the window (they land while it is staged: the
  // in-order vector-memory counter makes the window's wait cover them), the following rows' during the taps of the row before -- no exposed load per tap row
  pb_u4 wv[(NPC && NY) ? NY : 1];
  if (NPC && NY) {
    const int i = min(i0 + wave, A.dh - 1), jc = min(j, A.dw - 1);
    const long long xc = (long long)jc * A.x_step + A.xoff;
    const uint32_t wi0 = (uint32_t)(((int)(((long long)i * A.y_step + A.yoff) >> 12) & 15) * NY * 32 + (lane & 31));
    (void)xc;
#pragma unroll
    for (int ty = 0; ty < NY; ty++) wv[ty] = pairs4[wi0 + 32 * ty];
  }
  // ---- the window: premultiplied pairs.  4-byte pixels whose window lies inside the row (uniform per workgroup) come four at a time, one 16-byte load -> two pairs
  const bool inside = CH == 4 && wx0 >= 0 && wx0 + 2 * A.wpairs <= A.sw && (wx0 & 3) == 0 && ((uintptr_t)A.src & 15) == 0 && (A.irow & 15) == 0 && (A.wpairs & 1) == 0;
  if (inside) {
    const int wq = A.wpairs >> 1;                       // quads per window row
    for (int wy = wave; wy < A.win_h; wy += 4) {
      const pb_u4 *row = reinterpret_cast<const pb_u4 *>(A.src + (size_t)pb_clamp(ys0 + wy, A.sh - 1) * A.irow + 4 * (size_t)wx0);
      pb_u4 *wr = winp + wy * A.wpairs;
      for (int qd = lane; qd < wq; qd += 64) {
        const pb_u4 q = row[qd];
        pb_u4 v0, v1;
        if (OPQ) {
          v0.x = __builtin_amdgcn_perm(q.y, q.x, 0x0C040C00u); v0.y = __builtin_amdgcn_perm(q.y, q.x, 0x0C050C01u); v0.z = __builtin_amdgcn_perm(q.y, q.x, 0x0C060C02u); v0.w = 0u;
          v1.x = __builtin_amdgcn_perm(q.w, q.z, 0x0C040C00u); v1.y = __builtin_amdgcn_perm(q.w, q.z, 0x0C050C01u); v1.z = __builtin_amdgcn_perm(q.w, q.z, 0x0C060C02u); v1.w = 0u;
        } else {
        v0.x = pb_premul_pair<0>(q.x, q.y); v0.y = pb_premul_pair<1>(q.x, q.y); v0.z = pb_premul_pair<2>(q.x, q.y); v0.w = __builtin_amdgcn_perm(q.y, q.x, 0x0C070C03u);
        v1.x = pb_premul_pair<0>(q.z, q.w); v1.y = pb_premul_pair<1>(q.z, q.w); v1.z = pb_premul_pair<2>(q.z, q.w); v1.w = __builtin_amdgcn_perm(q.w, q.z, 0x0C070C03u);
        }
        wr[2 * qd] = v0; wr[2 * qd + 1] = v1;
      }
    }
  } else if (CH == 3 && wx0 >= 0 && wx0 + 2 * A.wpairs <= A.sw && 3 * (wx0 + 2 * A.wpairs) + 8 <= A.irow && ((uintptr_t)A.src & 3) == 0 && (A.irow & 3) == 0) {
    // 3-byte pixels, window inside the row with a few spare bytes behind it: a pair's 6 bytes out of three aligned dwords (v_alignbyte) instead of six byte loads
    for (int wy = wave; wy < A.win_h; wy += 4) {
      const uint8_t *row = A.src + (size_t)pb_clamp(ys0 + wy, A.sh - 1) * A.irow;
      pb_u4 *wr = winp + wy * A.wpairs;
      for (int p = lane; p < A.wpairs; p += 64) {
        const uint32_t bo = 3u * (uint32_t)(wx0 + 2 * p);
        const uint32_t *dp = reinterpret_cast<const uint32_t *>(row + (bo & ~3u));
        const uint32_t d0 = dp[0], d1 = dp[1], d2 = dp[2];
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, bo & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, bo & 3u);     // bytes bo .. bo + 7
        pb_u4 v;
        v.x = __builtin_amdgcn_perm(lo, lo, 0x0C030C00u); v.y = __builtin_amdgcn_perm(hi, lo, 0x0C040C01u); v.z = __builtin_amdgcn_perm(hi, lo, 0x0C050C02u); v.w = 0u;
        wr[p] = v;
      }
    }
  } else {
    for (int wy = wave; wy < A.win_h; wy += 4) {
      const uint8_t *row = A.src + (size_t)pb_clamp(ys0 + wy, A.sh - 1) * A.irow;
      pb_u4 *wr = winp + wy * A.wpairs;
      for (int p = lane; p < A.wpairs; p += 64) {
        const uint32_t q0 = pb_load_px<CH>(row, pb_clamp(wx0 + 2 * p, A.sw - 1)), q1 = pb_load_px<CH>(row, pb_clamp(wx0 + 2 * p + 1, A.sw - 1));
        pb_u4 v;
        if (CH == 4 && !OPQ) {
          v.x = pb_premul_pair<0>(q0, q1); v.y = pb_premul_pair<1>(q0, q1); v.z = pb_premul_pair<2>(q0, q1); v.w = __builtin_amdgcn_perm(q1, q0, 0x0C070C03u);
        } else {
          v.x = __builtin_amdgcn_perm(q1, q0, 0x0C040C00u); v.y = __builtin_amdgcn_perm(q1, q0, 0x0C050C01u); v.z = __builtin_amdgcn_perm(q1, q0, 0x0C060C02u); v.w = 0u;
        }
        wr[p] = v;
      }
    }
  }
  __syncthreads();
  const bool quads3 = CH == 3 && (((uintptr_t)A.dst | (uintptr_t)A.orow) & 3) == 0;
  // the accumulators of destination pixel (i, j) -> its bytes, stored
  auto emit = [&](int i, unsigned r, unsigned g, unsigned b, unsigned a) {
    const uint32_t px = OPQ ? ((r >> 16) | ((g >> 16) << 8) | ((b >> 16) << 16) | 0xFF000000u) : pb_finish_px<CH>(r, g, b, a, edge, A.rnd);
    uint8_t *drow = A.dst + (size_t)i * A.orow;
    if (CH == 4) { if (live) reinterpret_cast<uint32_t *>(drow)[j] = pb_epi_px(E, s_lut, px, i, j); }
    else if (quads3 && (j | 3) < A.dw) {
      // four lanes' 3-byte pixels = three dwords: every lane reads its quad's four values (DPP quad_perm broadcasts), packs them, and lanes 0..2 of the quad
      // store one dword each -- coalesced dword stores instead of three byte stores per pixel
      uint32_t q[4], w0, w1, w2;
      q[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0x00, 0xF, 0xF, true); q[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0x55, 0xF, 0xF, true);
      q[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0xAA, 0xF, 0xF, true); q[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0xFF, 0xF, 0xF, true);
      pack3(q, w0, w1, w2);
      const int l = lane & 3;
      if (l < 3) reinterpret_cast<uint32_t *>(drow)[3 * (j >> 2) + l] = l == 0 ? w0 : l == 1 ? w1 : w2;
    } else if (live) { uint8_t *d = drow + 3 * (size_t)j; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
  };
  if (NPC && NY) {
    // tap rows known at compile time: this row's weight vectors are in `cur` (requested before the window was staged, or during the previous row's taps); the next
    // row's go out into `nxt` before the taps.  The two register sets change roles from row to row (no take-over copies)
    pb_u4 wn[(NPC && NY) ? NY : 1];
    auto do_row = [&](int r_, const pb_u4 *cur, pb_u4 *nxt) {
      const int i = i0 + r_;
      const long long y = (long long)i * A.y_step + A.yoff;
      const int ys = (int)(y >> 16);
      const pb_u4 *wp = winp + (ys + A.ty0 - ys0) * A.wpairs + pidx;
      if (!ONE && r_ + 4 < A.tile_h && i + 4 < A.dh) {
        const uint32_t win = (uint32_t)(((int)(((long long)(i + 4) * A.y_step + A.yoff) >> 12) & 15) * NY * 32 + (lane & 31));
#pragma unroll
        for (int ty = 0; ty < NY; ty++) nxt[ty] = pairs4[win + 32 * ty];
      }
      unsigned r = 0, g = 0, b = 0, a = 0;
#pragma unroll
      for (int ty = 0; ty < NY; ty++) {
        uint32_t wq[4];
        take(cur[ty], wq);
#pragma unroll
        for (int k = 0; k < NPC; k++) {
          const pb_u4 dd = wp[ty * A.wpairs + k];
          r = pb_dot2(dd.x, wq[k], r); g = pb_dot2(dd.y, wq[k], g); b = pb_dot2(dd.z, wq[k], b);
          if (CH == 4 && !OPQ) a = pb_dot2(dd.w, wq[k], a);
        }
        // long filters: a tap row's window reads stay next to its taps.  Left alone the compiler runs the alpha sums of all rows first and sinks the colour sums behind
        // the alpha test of pb_finish_px: every window vector stays live (4 x 6 x 4 = 96 registers) and the occupancy goes from 8 waves to 3
        if (NY >= 3) { asm volatile("" : "+v"(r), "+v"(g), "+v"(b), "+v"(a)); __builtin_amdgcn_sched_barrier(0); }
      }
      emit(i, r, g, b, a);
    };
    for (int r_ = wave; r_ < A.tile_h && i0 + r_ < A.dh; r_ += 8) {
      do_row(r_, wv, wn);
      if (ONE || r_ + 4 >= A.tile_h || i0 + r_ + 4 >= A.dh) break;
      do_row(r_ + 4, wn, wv);
    }
    return;
  }
  for (int r_ = wave; r_ < A.tile_h; r_ += 4) {
    const int i = i0 + r_;
    if (i >= A.dh) break;
    const long long y = (long long)i * A.y_step + A.yoff;
    const int ys = (int)(y >> 16), yph = (int)(y >> 12) & 15;
    const pb_u4 *wp = winp + (ys + A.ty0 - ys0) * A.wpairs + pidx;
    const uint32_t wi = (uint32_t)(yph * A.ny_eff * 32 + (NPC ? (lane & 31) : xph * 2 + par));       // [y phase][tap row][x phase][parity]; NPC: the vector this lane REQUESTS (see take())
    unsigned r = 0, g = 0, b = 0, a = 0;
    for (int ty = 0; ty < A.ny_eff; ty++, wp += A.wpairs) {
      if (NPC) {
        uint32_t wq[4];
        take(pairs4[wi + 32 * ty], wq);
#pragma unroll
        for (int k = 0; k < (NPC ? NPC : 1); k++) {
          const pb_u4 dd = wp[k];
          r = pb_dot2(dd.x, wq[k], r); g = pb_dot2(dd.y, wq[k], g); b = pb_dot2(dd.z, wq[k], b);
          if (CH == 4 && !OPQ) a = pb_dot2(dd.w, wq[k], a);
        }
        continue;
      }
      for (int qd = 0; qd < A.nq; qd++) {
        const pb_u4 w = pairs4[(wi + 32 * ty) * A.nq + qd];
        const pb_u4 d0 = wp[4 * qd], d1 = wp[4 * qd + 1], d2 = wp[4 * qd + 2], d3 = wp[4 * qd + 3];
        r = pb_dot2(d0.x, w.x, r); g = pb_dot2(d0.y, w.x, g); b = pb_dot2(d0.z, w.x, b);
        r = pb_dot2(d1.x, w.y, r); g = pb_dot2(d1.y, w.y, g); b = pb_dot2(d1.z, w.y, b);
        r = pb_dot2(d2.x, w.z, r); g = pb_dot2(d2.y, w.z, g); b = pb_dot2(d2.z, w.z, b);
        r = pb_dot2(d3.x, w.w, r); g = pb_dot2(d3.y, w.w, g); b = pb_dot2(d3.z, w.w, b);
        if (CH == 4 && !OPQ) { a = pb_dot2(d0.w, w.x, a); a = pb_dot2(d1.w, w.y, a); a = pb_dot2(d2.w, w.z, a); a = pb_dot2(d3.w, w.w, a); }
      }
    }
    emit(i, r, g, b, a);
  }
}

// =====================================================================================================================================================
// k_pb_gather -- 4-byte pixels at INTEGER reduction ratios other than 2:1 (3:1, 4:1, ...: one phase for the whole frame), no LDS window and no barrier: a lane
// owns one destination pixel and reads its taps where they lie (one or two 4-byte-aligned vector loads per tap row; neighbouring lanes' windows overlap in
// L1), premultiplies them as 16-bit pairs (SDWA) and accumulates two taps per v_dot2_u32_u16 with weight pairs aligned on the FIRST tap.  The weights are
// wave-uniform: scalar loads, SGPR operands of the dot products.  The loads of tap row t + 1 are issued before the arithmetic of row t.
// profiles/r03/pb_gather_ab.txt: 4K -> 720p HYPER 18.8 -> 13.3 us, BILINEAR 12.4 -> 8.7, 4K -> 960x540 16.7 -> 11.0 against k_pb_pairs.  With per-lane weights
// (ratios with several phases; from device memory or from an LDS copy of the table, both built and measured) the same kernel LOSES to the LDS window
// (4K -> 1706x960 31.4 / 25.6 against 24.7 us, 1080p -> 720p 13.7 / 14.3 against 10.1, enlargements 15-40 % slower): every source pixel is premultiplied once per
// destination pixel that uses it, and the taps come through L1 at half the LDS rate -- those ratios keep k_pb_pairs.
// =====================================================================================================================================================
struct PbGatherArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, sw, sh, dw, dh;
  int x_step, y_step, xoff, yoff;
  int tx0, ty0, ny_eff;
};                                     // + gp, device: [16][16][ny_eff][RL] weight pairs (tap tx0 + 2k | tap tx0 + 2k + 1 << 16), zero padded
typedef pb_u4 pb_u4a __attribute__((aligned(4)));
typedef pb_u2 pb_u2a __attribute__((aligned(4)));

// OPQ (LGPU_INTERP_OPAQUE): the taps go in as byte pairs (one v_perm per pair and channel instead of two SDWA products), no alpha sums, T >> 16 (k_pb_pairs<.., OPQ>)
template <int NP, typename EA = PbNoEpi, int OPQ = 0>
__global__ __launch_bounds__(256) void k_pb_gather(const PbGatherArgs A_, const uint32_t *__restrict__ gp, const PbFrames F, const EA E) {
  PB_FRAME_ARGS(PbGatherArgs);
  __shared__ uint8_t s_lut[pb_has_epi<EA>::value ? 256 : 4];
  if (pb_has_epi<EA>::value) { pb_epi_stage(s_lut, E); __syncthreads(); }      // (the only barrier of this kernel, and only in the chain's form)
  constexpr int RL = NP <= 2 ? 2 : 4;                  // weight dwords per tap row
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x * 64 + lane, i = blockIdx.y * 4 + wave;
  if (i >= A.dh) return;
  const bool live = j < A.dw;
  const long long x = (long long)(live ? j : A.dw - 1) * A.x_step + A.xoff, y = (long long)i * A.y_step + A.yoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15, ys = (int)(y >> 16) + A.ty0, yph = (int)(y >> 12) & 15;
  const int p0 = xs + A.tx0;
  const bool fast = p0 >= 0 && p0 + 2 * NP <= A.sw;
  const uint32_t *wrow = gp + (uint32_t)((yph * 16 + __builtin_amdgcn_readfirstlane(xph)) * A.ny_eff) * RL;      // the same phase in every lane (host-checked)
  struct Row { uint32_t q[2 * NP]; uint32_t w[RL]; };
  auto fetch = [&](int ty) -> Row {
    Row R;
    const uint8_t *row = A.src + (size_t)pb_clamp(ys + ty, A.sh - 1) * A.irow;
    if (fast) {
      const uint8_t *p = row + 4 * (size_t)p0;
      if (NP == 1) { const pb_u2 v = *reinterpret_cast<const pb_u2a *>(p); R.q[0] = v.x; R.q[1] = v.y; }
      else {
        const pb_u4 v = *reinterpret_cast<const pb_u4a *>(p);
        R.q[0] = v.x; R.q[1] = v.y; R.q[2] = v.z; R.q[3] = v.w;
        if (NP == 3) { const pb_u2 u = *reinterpret_cast<const pb_u2a *>(p + 16); R.q[4 % (2 * NP)] = u.x; R.q[5 % (2 * NP)] = u.y; }
        if (NP == 4) { const pb_u4 u = *reinterpret_cast<const pb_u4a *>(p + 16); R.q[4 % (2 * NP)] = u.x; R.q[5 % (2 * NP)] = u.y; R.q[6 % (2 * NP)] = u.z; R.q[7 % (2 * NP)] = u.w; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 2 * NP; k++) R.q[k] = reinterpret_cast<const uint32_t *>(row)[pb_clamp(p0 + k, A.sw - 1)];
    }
    if (RL == 2) { const pb_u2 v = *reinterpret_cast<const pb_u2 *>(wrow + ty * RL); R.w[0] = v.x; R.w[1] = v.y; }
    else { const pb_u4 v = *reinterpret_cast<const pb_u4 *>(wrow + ty * RL); R.w[0] = v.x; R.w[1] = v.y; R.w[2 % RL] = v.z; R.w[3 % RL] = v.w; }
    return R;
  };
  unsigned r = 0, g = 0, b = 0, a = 0;
  Row cur = fetch(0);
  for (int ty = 0; ty < A.ny_eff; ty++) {
    Row nxt = cur;
    if (ty + 1 < A.ny_eff) nxt = fetch(ty + 1);
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const uint32_t q0 = cur.q[2 * k], q1 = cur.q[2 * k + 1], w = cur.w[k];
      if (OPQ) {
        r = pb_dot2(__builtin_amdgcn_perm(q1, q0, 0x0C040C00u), w, r); g = pb_dot2(__builtin_amdgcn_perm(q1, q0, 0x0C050C01u), w, g); b = pb_dot2(__builtin_amdgcn_perm(q1, q0, 0x0C060C02u), w, b);
        continue;
      }
      r = pb_dot2(pb_premul_pair<0>(q0, q1), w, r); g = pb_dot2(pb_premul_pair<1>(q0, q1), w, g); b = pb_dot2(pb_premul_pair<2>(q0, q1), w, b);
      a = pb_dot2(__builtin_amdgcn_perm(q1, q0, 0x0C070C03u), w, a);
    }
    cur = nxt;
  }
  if (live) reinterpret_cast<uint32_t *>(A.dst + (size_t)i * A.orow)[j] = pb_epi_px(E, s_lut, OPQ ? ((r >> 16) | ((g >> 16) << 8) | ((b >> 16) << 16) | 0xFF000000u) : pb_finish_px<4>(r, g, b, a, false, 0u), i, j);
}

// =====================================================================================================================================================
// k_pb_up -- 4-byte pixels, both sides enlarged (any ratio but the exact 1:2 of k_pb_double): a destination pixel needs at most 4 x 4 taps, consecutive
// destination rows share their source rows, and sixteen phases per axis are all the weights there are.  A lane owns ONE destination column and walks a band of
// destination rows with its tap window in REGISTERS (NY source rows x NP premultiplied pixel pairs): a source row is read (one 4-byte-aligned vector load where
// the taps lie, the next row always in flight) and premultiplied once per lane and band, when the walk crosses into it, instead of once per destination row; the
// pair table of all 256 phases sits in LDS (8 KB at 4 x 4 taps), so a destination pixel costs two ds_read_b128, NY x NP x 4 dot products and the library's
// double-precision un-premultiply.  k_pb_pairs on the same frames is LDS-bound (8 window reads + 4 weight vectors from device memory per destination pixel).
// =====================================================================================================================================================
struct PbUpArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, sw, sh, dw, dh;
  int x_step, y_step, xoff, yoff;
  int tx0, ty0, rb;                    // rb: destination rows per band
};

// OPQ (lgpu_pixbuf_scale with LGPU_INTERP_OPAQUE: the caller states that every source pixel has alpha 255): the window holds the colour bytes as 16-bit pairs (one
// v_perm per pair and channel, no alpha product), three dot products per pair, and the library's un-premultiply is T >> 16 (k_pb_pairs<.., OPQ>; pb_opaque_check)
template <int NP, int NY, int OPQ = 0, typename EA = PbNoEpi>
__global__ __launch_bounds__(256) void k_pb_up(const PbUpArgs A_, const uint32_t *__restrict__ gp, const PbFrames F, const EA E) {
  PB_FRAME_ARGS(PbUpArgs);
  __shared__ uint8_t s_lut[pb_has_epi<EA>::value ? 256 : 4];
  pb_epi_stage(s_lut, E);                              // (the barrier behind the weight table covers it)
  constexpr int RL = 2, PW = NY * RL;                  // dwords per tap row / per phase of the pair table (rows of 2 dwords for NP <= 2)
  __shared__ __attribute__((aligned(16))) uint32_t s_w[256 * PW];
  for (int e = threadIdx.x; e < 256 * PW; e += 256) s_w[e] = gp[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cb = blockIdx.x * 4 + wave;
  if (cb * 64 >= A.dw) return;
  const int j = cb * 64 + lane;
  const bool live = j < A.dw;
  const long long x = (long long)(live ? j : A.dw - 1) * A.x_step + A.xoff;
  const int xs = (int)(x >> 16), xph = (int)(x >> 12) & 15;
  const int p0 = xs + A.tx0;
  const bool fast = p0 >= 0 && p0 + 2 * NP <= A.sw;
  struct Raw { uint32_t q[2 * NP]; };
  auto load_raw = [&](int v) -> Raw {                  // the taps of virtual source row v (rows beyond the frame repeat its first / last row)
    Raw R;
    const uint8_t *row = A.src + (size_t)pb_clamp(v, A.sh - 1) * A.irow;
    if (fast) {
      const uint8_t *p = row + 4 * (size_t)p0;
      if (NP == 1) { const pb_u2 t = *reinterpret_cast<const pb_u2a *>(p); R.q[0] = t.x; R.q[1] = t.y; }
      else { const pb_u4 t = *reinterpret_cast<const pb_u4a *>(p); R.q[0] = t.x; R.q[1] = t.y; R.q[2 % (2 * NP)] = t.z; R.q[3 % (2 * NP)] = t.w; }
    } else {
#pragma unroll
      for (int k = 0; k < 2 * NP; k++) R.q[k] = reinterpret_cast<const uint32_t *>(row)[pb_clamp(p0 + k, A.sw - 1)];
    }
    return R;
  };
  uint32_t win[NY][4 * NP];
  auto premul = [&](const Raw &R, uint32_t *o) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if (OPQ) {
        o[4 * k] = __builtin_amdgcn_perm(R.q[2 * k + 1], R.q[2 * k], 0x0C040C00u); o[4 * k + 1] = __builtin_amdgcn_perm(R.q[2 * k + 1], R.q[2 * k], 0x0C050C01u);
        o[4 * k + 2] = __builtin_amdgcn_perm(R.q[2 * k + 1], R.q[2 * k], 0x0C060C02u); o[4 * k + 3] = 0u;
        continue;
      }
      o[4 * k] = pb_premul_pair<0>(R.q[2 * k], R.q[2 * k + 1]); o[4 * k + 1] = pb_premul_pair<1>(R.q[2 * k], R.q[2 * k + 1]);
      o[4 * k + 2] = pb_premul_pair<2>(R.q[2 * k], R.q[2 * k + 1]); o[4 * k + 3] = __builtin_amdgcn_perm(R.q[2 * k + 1], R.q[2 * k], 0x0C070C03u);
    }
  };
  const int i0 = blockIdx.y * A.rb, i1 = min(i0 + A.rb, A.dh);
  int cur = (int)(((long long)i0 * A.y_step + A.yoff) >> 16) + A.ty0;          // virtual source row of win[0]
  {
    Raw r0[NY];
#pragma unroll
    for (int t = 0; t < NY; t++) r0[t] = load_raw(cur + t);
#pragma unroll
    for (int t = 0; t < NY; t++) premul(r0[t], win[t]);
  }
  Raw nxt = load_raw(cur + NY);
  const uint32_t *wbase = s_w + xph * PW;
  for (int i = i0; i < i1; i++) {
    const long long y = (long long)i * A.y_step + A.yoff;
    const int vs = (int)(y >> 16) + A.ty0, yph = (int)(y >> 12) & 15;
    if (vs != cur) {                                   // wave-uniform; the walk enters the next source row (a step <= 1 never skips one)
#pragma unroll
      for (int t = 0; t + 1 < NY; t++)
#pragma unroll
        for (int c = 0; c < 4 * NP; c++) win[t][c] = win[t + 1][c];
      premul(nxt, win[NY - 1]);
      cur = vs;
      nxt = load_raw(cur + NY);
    }
    const uint32_t *w = wbase + yph * 16 * PW;
    uint32_t wv[PW];
    if (PW % 4 == 0) {
#pragma unroll
      for (int c = 0; c < PW / 4; c++) { const pb_u4 t = reinterpret_cast<const pb_u4 *>(w)[c]; wv[4 * c] = t.x; wv[4 * c + 1] = t.y; wv[4 * c + 2] = t.z; wv[4 * c + 3] = t.w; }
    } else {
#pragma unroll
      for (int c = 0; c < PW / 2; c++) { const pb_u2 t = reinterpret_cast<const pb_u2 *>(w)[c]; wv[2 * c] = t.x; wv[2 * c + 1] = t.y; }
    }
    unsigned r = 0, g = 0, b = 0, a = 0;
#pragma unroll
    for (int t = 0; t < NY; t++)
#pragma unroll
      for (int k = 0; k < NP; k++) {
        const uint32_t ww = wv[t * RL + k];
        r = pb_dot2(win[t][4 * k], ww, r); g = pb_dot2(win[t][4 * k + 1], ww, g); b = pb_dot2(win[t][4 * k + 2], ww, b);
        if (!OPQ) a = pb_dot2(win[t][4 * k + 3], ww, a);
      }
    if (live) reinterpret_cast<uint32_t *>(A.dst + (size_t)i * A.orow)[j] = pb_epi_px(E, s_lut, OPQ ? ((r >> 16) | ((g >> 16) << 8) | ((b >> 16) << 16) | 0xFF000000u) : pb_finish_px<4>(r, g, b, a, false, 0u), i, j);
  }
}

// ---- host: the per-phase weight tables ----------------------------------------------------------------------------------------------------
struct PbDim { int n; double offset; std::vector<double> w; };   // w[phase * n + tap]

static double ramp_integral(double lo, double hi) {   // integral of t over [lo, hi] intersected with [0, 1]
  if (lo > 0.) { if (lo >= 1.) return 0.; } else { if (hi <= 0.) return 0.; lo = 0.; }
  const double top = hi < 1. ? hi : 1.;
  return 0.5 * (top * top - lo * lo);
}

static PbDim pb_dimension(int interp, double scale) {
  PbDim d;
  const bool hyper = interp == 3, grow = scale > 1.0;
  d.n = hyper ? (int)ceil(1 / scale + 3.0) : grow ? 2 : (int)ceil(1.0 + 1.0 / scale);
  d.offset = hyper ? -1.0 : grow ? 0.5 * (1 / scale - 1) : 0.0;
  d.w.resize((size_t)16 * d.n);
  for (int ph = 0; ph < 16; ph++) {
    const double x = (double)ph / 16, a = x + 1 / scale;
    for (int i = 0; i < d.n; i++) {
      double w;
      if (hyper) w = (ramp_integral(0.5 + i - a, 0.5 + i - x) + ramp_integral(1.5 + x - i, 1.5 + a - i)) * scale;
      else if (grow) w = (((i == 0) ? (1 - x) : x) / scale) * scale;
      else if (i < x) w = (i + 1 > x) ? (fmin(i + 1, a) - x) * scale : 0.;
      else w = (a > i) ? (fmin(i + 1, a) - i) * scale : 0.;
      d.w[(size_t)ph * d.n + i] = w;
    }
  }
  return d;
}

// rounding residue of one phase spread from the last tap backwards until the table sums to 65536
static void pb_fix_sum(int *w, int count, int total) {
  const int correction = 65536 - total;
  int remaining = correction;
  for (int d = 1, c = correction; c != 0 && remaining != 0; d++, c = correction / d)
    for (int i = count - 1; i >= 0 && c != 0 && remaining != 0; i--)
      if (w[i] + c >= 0) {
        w[i] += c;
        remaining -= c;
        if ((0 < remaining && remaining < c) || (0 > remaining && remaining > c)) c = remaining;
      }
}

struct PbTable { int n_x, n_y, xoff, yoff, uniform_x; int *table_d; std::vector<int> host;
                 int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0, nq = 0; uint32_t *pairs_d = nullptr; uint32_t *gpairs_d = nullptr;      // pairs_d: the k_pb_pairs form of the table (nullptr: a weight needs 17 bits)
                 // cache bookkeeping (under g_pb_mu): the streams that have launched with this table, calls between lookup and launch, age
                 std::vector<hipStream_t> users; int pins = 0; unsigned long long stamp = 0; };
// The tables of a geometry are cached per (device, interp, sw, sh, dw, dh) -- but BOUNDED: a compositor that animates its layers' scale or an interactive zoom asks
// for a new geometry every frame.  At most LGPU_PB_CACHE_MAX entries (64) stay; the least recently used one that no call holds is retired.  Device memory comes
// from the stream-ordered pool on the CALLING stream (no null-stream copy, no device-wide synchronisation); the builder waits for its own uploads before the entry
// becomes visible (a new geometry only -- and no cross-stream event waits later, which a stream under graph capture cannot take); a retired entry's memory is freed
// stream-ordered behind the last launch of every stream that used it.  A table is built outside the lock.
static std::mutex g_pb_mu;
static std::map<std::tuple<int, int, int, int, int, int>, PbTable *> g_pb_tables;
static unsigned long long g_pb_clock = 0;

// `ordered` false: the caller could not order `st` behind every user of the table (or st itself is gone) -- the device is drained first, then the blocks are freed outright
static void pb_free_device(PbTable *t, hipStream_t st, bool ordered = true) {
  void *blocks[3] = {t->table_d, t->pairs_d, t->gpairs_d};
  for (void *b : blocks) {
    if (!b) continue;
    if (ordered && hipFreeAsync(b, st) == hipSuccess) continue;
    (void)hipGetLastError();
    if (ordered) { (void)hipDeviceSynchronize(); ordered = false; }       // a destroyed home stream (hipFreeAsync failed): everything it and the others launched is over after this
    (void)hipFree(b);
  }
  t->table_d = nullptr; t->pairs_d = nullptr; t->gpairs_d = nullptr;
  (void)hipGetLastError();
}
// an entry no longer in the map and held by no call: its memory goes back behind everything its users have enqueued
static void pb_retire(PbTable *t) {
  // (entries are only ever retired on their own device: the eviction loop in pb_table looks at the current device's keys alone, so the events below are
  // created where the streams live)
  hipStream_t home = t->users.empty() ? nullptr : t->users[0];
  bool ordered = true;
  for (size_t i = 1; i < t->users.size(); i++) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ordered = false; (void)hipGetLastError(); continue; }
    if (hipEventRecord(e, t->users[i]) == hipSuccess) {
      if (hipStreamWaitEvent(home, e, 0) != hipSuccess) ordered = false;       // the home stream is gone, or the wait could not be placed: drain instead
    }                                                                          // (a USER stream that is gone has finished its work: nothing to wait for)
    (void)hipEventDestroy(e);
    (void)hipGetLastError();
  }
  if (!ordered) (void)hipDeviceSynchronize();
  pb_free_device(t, home, ordered);
  delete t;
}
static int pb_upload(const void *host, size_t bytes, hipStream_t st, void **out) {
  *out = nullptr;
  int rc = lgpu_malloc_ordered(out, bytes, st);
  if (rc) return rc;
  if (hipMemcpyAsync(*out, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
    set_error("upload of a scaler table failed: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFreeAsync(*out, st);
    *out = nullptr;
    return LGPU_E_HIP;
  }
  return LGPU_OK;
}

static int pb_build(int interp, int sw, int sh, int dw, int dh, PbTable *t, bool upload, hipStream_t st = nullptr) {
  const PbDim fx = pb_dimension(interp, (double)dw / sw), fy = pb_dimension(interp, (double)dh / sh);
  t->n_x = fx.n; t->n_y = fy.n;
  t->xoff = (int)floor(fx.offset * 65536); t->yoff = (int)floor(fy.offset * 65536);
  t->table_d = nullptr;
  if ((long long)fx.n * fy.n > 1000) return LGPU_E_UNSUPPORTED;       // the library's two-step scaler takes over there
  const int nn = fx.n * fy.n;
  t->host.assign((size_t)256 * nn, 0);
  for (int yp = 0; yp < 16; yp++)
    for (int xp = 0; xp < 16; xp++) {
      int *pw = t->host.data() + (size_t)(yp * 16 + xp) * nn, total = 0;
      for (int i = 0; i < fy.n; i++)
        for (int j = 0; j < fx.n; j++) {
          const double weight = fx.w[(size_t)xp * fx.n + j] * fy.w[(size_t)yp * fy.n + i] * 1.0 * 65536 + 0.5;
          pw[i * fx.n + j] = (int)weight;
          total += (int)weight;
        }
      pb_fix_sum(pw, nn, total);
    }
  int rc;
  if (upload && (rc = pb_upload(t->host.data(), t->host.size() * sizeof(int), st, (void **)&t->table_d))) return rc;
  // the phases this geometry's destination pixels take, the bounding box of their non-zero taps, and -- when every weight of those phases fits 16 bits -- the
  // table again as aligned tap PAIRS for both start parities (k_pb_pairs)
  const int x_step = (int)(65536 / ((double)dw / sw)), y_step = (int)(65536 / ((double)dh / sh));
  if (x_step > 0 && y_step > 0) {
    bool xp[16] = {false}, yp[16] = {false};
    for (int j = 0; j < dw; j++) xp[(((long long)j * x_step + t->xoff) >> 12) & 15] = true;
    for (int i = 0; i < dh; i++) yp[(((long long)i * y_step + t->yoff) >> 12) & 15] = true;
    int tx0 = t->n_x, tx1 = 0, ty0 = t->n_y, ty1 = 0, wmax = 0;
    for (int y = 0; y < 16; y++)
      for (int x = 0; x < 16; x++) {
        if (!xp[x] || !yp[y]) continue;
        const int *w = t->host.data() + (size_t)(y * 16 + x) * nn;
        for (int ty = 0; ty < t->n_y; ty++)
          for (int tx = 0; tx < t->n_x; tx++)
            if (w[ty * t->n_x + tx]) {
              tx0 = tx < tx0 ? tx : tx0; tx1 = tx + 1 > tx1 ? tx + 1 : tx1; ty0 = ty < ty0 ? ty : ty0; ty1 = ty + 1 > ty1 ? ty + 1 : ty1;
              wmax = w[ty * t->n_x + tx] > wmax ? w[ty * t->n_x + tx] : wmax;
            }
      }
    t->tx0 = tx0; t->tx1 = tx1; t->ty0 = ty0; t->ty1 = ty1;
    if (upload && wmax < 65536 && tx1 > tx0) {
      const int n_eff = tx1 - tx0, ny_eff = ty1 - ty0, np = (n_eff + 2) / 2, nq = (np + 3) / 4, rowlen = 4 * nq;
      std::vector<uint32_t> pr((size_t)16 * 16 * 2 * ny_eff * rowlen, 0u);
      for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++)
          for (int par = 0; par < 2; par++) {
            const int *w = t->host.data() + (size_t)(y * 16 + x) * nn;

            for (int ty = 0; ty < ny_eff; ty++)
              for (int i = 0; i < np; i++) {
                const int t0 = tx0 + 2 * i - par, t1 = t0 + 1;        // the two taps of aligned pair i when the first tap sits on an even (par 0) / odd (par 1) source pixel
                const uint32_t w0 = (t0 >= tx0 && t0 < tx1) ? (uint32_t)w[(ty0 + ty) * t->n_x + t0] : 0u, w1 = (t1 >= tx0 && t1 < tx1) ? (uint32_t)w[(ty0 + ty) * t->n_x + t1] : 0u;
                pr[((size_t)(y * ny_eff + ty) * 32 + x * 2 + par) * rowlen + i] = w0 | (w1 << 16);
              }
          }
      if ((rc = pb_upload(pr.data(), pr.size() * sizeof(uint32_t), st, (void **)&t->pairs_d))) { pb_free_device(t, st); return rc; }
      (void)hipStreamSynchronize(st);          // pr is a local: its bytes must have left before it goes
      t->nq = nq;
      // the same weights as pairs aligned on the first used tap (k_pb_gather: one form, no parity), rows of 2 or 4 dwords
      const int gnp = (n_eff + 1) / 2;
      if (gnp <= 4) {
        const int rl = gnp <= 2 ? 2 : 4;
        std::vector<uint32_t> gp((size_t)256 * ny_eff * rl, 0u);
        for (int ph = 0; ph < 256; ph++) {
          const int *w = t->host.data() + (size_t)ph * nn;
          for (int ty = 0; ty < ny_eff; ty++)
            for (int i = 0; i < gnp; i++) {
              const int t0 = tx0 + 2 * i, t1 = t0 + 1;
              const uint32_t w0 = (uint32_t)w[(ty0 + ty) * t->n_x + t0], w1 = t1 < tx1 ? (uint32_t)w[(ty0 + ty) * t->n_x + t1] : 0u;
              gp[((size_t)ph * ny_eff + ty) * rl + i] = w0 | (w1 << 16);
            }
        }
        if ((rc = pb_upload(gp.data(), gp.size() * sizeof(uint32_t), st, (void **)&t->gpairs_d))) { pb_free_device(t, st); return rc; }
        (void)hipStreamSynchronize(st);
      }
    }
  }
  return LGPU_OK;
}

// lookup (or build) + pin: the caller launches on `st` and then lets go with pb_unpin() -- or simply lets a PbPin go out of scope
static void pb_unpin(const PbTable *t) {
  if (!t) return;
  std::lock_guard<std::mutex> lk(g_pb_mu);
  const_cast<PbTable *>(t)->pins--;
}
struct PbPin {
  const PbTable *t = nullptr;
  ~PbPin() { pb_unpin(t); }
};
static int pb_table(int interp, int sw, int sh, int dw, int dh, hipStream_t st, PbPin *pin) {
  int dev = 0;
  LGPU_HIP(hipGetDevice(&dev));
  const auto key = std::make_tuple(dev, interp, sw, sh, dw, dh);
  auto take = [&](PbTable *t) {          // under the lock
    t->pins++;
    t->stamp = ++g_pb_clock;
    bool known = false;
    for (hipStream_t u : t->users) known = known || u == st;
    if (!known) t->users.push_back(st);
    pin->t = t;
  };
  {
    std::lock_guard<std::mutex> lk(g_pb_mu);
    auto it = g_pb_tables.find(key);
    if (it != g_pb_tables.end()) take(it->second);
  }
  if (!pin->t) {
    PbTable *t = new PbTable();
    int rc = pb_build(interp, sw, sh, dw, dh, t, true, st);          // outside the lock
    if (rc != LGPU_OK && rc != LGPU_E_UNSUPPORTED) { delete t; return rc; }
    if (t->table_d && hipStreamSynchronize(st) != hipSuccess) {        // the uploads are over before anybody else can find the entry
      set_error("scaler table: upload failed: %s", hipGetErrorString(hipGetLastError()));
      pb_free_device(t, st);
      delete t;
      return LGPU_E_HIP;
    }
    t->users.push_back(st);
    std::vector<PbTable *> out_;
    {
      std::lock_guard<std::mutex> lk(g_pb_mu);
      auto it = g_pb_tables.find(key);
      if (it != g_pb_tables.end()) { out_.push_back(t); take(it->second); }       // another thread was faster: its entry is the one
      else {
        g_pb_tables.emplace(key, t);
        t->pins = 1; t->stamp = ++g_pb_clock;
        pin->t = t;
        int cap = tune(TUNE_PB_CACHE_MAX);
        if (cap < 1) cap = 64;
        while ((int)g_pb_tables.size() > cap) {           // the least recently used entry nobody holds
          auto victim = g_pb_tables.end();
          for (auto j = g_pb_tables.begin(); j != g_pb_tables.end(); ++j)       // of THIS device: its streams and pool are the current ones here
            if (std::get<0>(j->first) == std::get<0>(key) && j->second->pins == 0 && (victim == g_pb_tables.end() || j->second->stamp < victim->second->stamp)) victim = j;
          if (victim == g_pb_tables.end()) break;
          out_.push_back(victim->second);
          g_pb_tables.erase(victim);
        }
      }
    }
    for (PbTable *o : out_) pb_retire(o);                  // outside the lock
  }
  return pin->t->table_d ? LGPU_OK : LGPU_E_UNSUPPORTED;
}
extern "C" int lgpu_debug_pixbuf_cache_entries(void) {
  std::lock_guard<std::mutex> lk(g_pb_mu);
  return (int)g_pb_tables.size();
}


// zero taps are common (the 5th row / column of an integer-ratio HYPER table, the far taps of most phases): only the phases this call's destination pixels
// take are looked at, and the tap loops run over the bounding box of their non-zero weights
static void pb_used_taps(const PbTable *t, int x_step, int y_step, int dw, int dh, PbArgs *a) {
  bool xp[16] = {false}, yp[16] = {false};
  for (int j = 0; j < dw; j++) xp[(((long long)j * x_step + t->xoff) >> 12) & 15] = true;
  for (int i = 0; i < dh; i++) yp[(((long long)i * y_step + t->yoff) >> 12) & 15] = true;
  int tx0 = t->n_x, tx1 = 0, ty0 = t->n_y, ty1 = 0;
  for (int y = 0; y < 16; y++)
    for (int x = 0; x < 16; x++) {
      if (!xp[x] || !yp[y]) continue;
      const int *w = t->host.data() + (size_t)(y * 16 + x) * t->n_x * t->n_y;
      for (int ty = 0; ty < t->n_y; ty++)
        for (int tx = 0; tx < t->n_x; tx++)
          if (w[ty * t->n_x + tx]) { tx0 = tx < tx0 ? tx : tx0; tx1 = tx + 1 > tx1 ? tx + 1 : tx1; ty0 = ty < ty0 ? ty : ty0; ty1 = ty + 1 > ty1 ? ty + 1 : ty1; }
    }
  a->tx0 = tx0; a->tx1 = tx1; a->ty0 = ty0; a->ty1 = ty1;
}

// the exact-2:1 fast path: geometry, alignment, and the table really being the outer product the kernel evaluates
static bool pb_half_ok(const PbTable *t, int interp, int sw, int sh, int dw, int dh, uintptr_t src_bits, uintptr_t dst_bits, int *hyper, int *ashift) {
  if (sw != 2 * dw || sh != 2 * dh || (sw & 3) || (src_bits & 15) || (dst_bits & 7)) return false;
  const int vi = interp == 3 ? 7 : 1, vo = interp == 3 ? 1 : 0, scale = interp == 3 ? 256 : 16384, first = interp == 3 ? 0 : 1;
  if (t->xoff != (interp == 3 ? -65536 : 0) || t->yoff != t->xoff) return false;
  const int *w = t->host.data();                       // phase (0, 0)
  for (int ty = 0; ty < t->n_y; ty++)
    for (int tx = 0; tx < t->n_x; tx++) {
      const int ky = ty + first, kx = tx + first;      // position in the 4-tap vector [vo vi vi vo] that starts at source pixel 2X - 1
      const int wy = (ky < 0 || ky > 3) ? 0 : (ky == 0 || ky == 3) ? vo : vi, wx = (kx < 0 || kx > 3) ? 0 : (kx == 0 || kx == 3) ? vo : vi;
      if (w[ty * t->n_x + tx] != scale * wy * wx) return false;
    }
  *hyper = interp == 3; *ashift = interp == 3 ? 8 : 2;
  return true;
}

// the exact 1:2 enlargement: every phase the destination pixels take must be 4096 * [wy0 wy1] x [wx0 wx1] with (1, 3) or (3, 1) on two adjacent taps, placed so that
// out(2i) = P[i-1] + 3 P[i] and out(2i+1) = 3 P[i] + P[i+1] (what k_pb_double evaluates); anything else keeps the general kernels
static bool pb_double_ok(const PbTable *t, int x_step, int y_step) {
  if (x_step != 32768 || y_step != 32768) return false;
  for (int jy = 0; jy < 2; jy++)
    for (int jx = 0; jx < 2; jx++) {
      const long long x = (long long)jx * x_step + t->xoff, y = (long long)jy * y_step + t->yoff;
      const int xs = (int)(x >> 16), ys = (int)(y >> 16), xph = (int)(x >> 12) & 15, yph = (int)(y >> 12) & 15;
      const int *w = t->host.data() + (size_t)(yph * 16 + xph) * t->n_x * t->n_y;
      // destination pixel jx (0: even column, 1: odd) must weight source pixels (jx == 0 ? -1, 0 : 0, 1) with (1, 3) / (3, 1); same vertically
      for (int ty = 0; ty < t->n_y; ty++)
        for (int tx = 0; tx < t->n_x; tx++) {
          const int sx = xs + tx, sy = ys + ty;
          const int wx = jx == 0 ? (sx == -1 ? 1 : sx == 0 ? 3 : 0) : (sx == 0 ? 3 : sx == 1 ? 1 : 0);
          const int wy = jy == 0 ? (sy == -1 ? 1 : sy == 0 ? 3 : 0) : (sy == 0 ? 3 : sy == 1 ? 1 : 0);
          if (w[ty * t->n_x + tx] != 4096 * wx * wy) return false;
        }
    }
  return true;
}

// bands of th or th + 1 rows that cover dh exactly
static void pb_half_bands(PbHalfArgs *a, int bands) {
  if (bands < 1) bands = 1;
  if (bands > a->dh) bands = a->dh;
  a->bands = bands; a->th = a->dh / bands; a->rem = a->dh - a->th * bands;
}

static void pb_half_geometry(PbHalfArgs *a, int ntracks, int blur = 0, int opaque = 0) {
  // strips of 64 storing lanes on 128-byte lines (k_pb_half<.., ALIGNED>; the two outer taps of a strip from one extra 4-byte load in lanes 0 and 63, which ride into
  // the lane exchange for free).  Round 3 measured them 3 % lighter on traffic and 13 % heavier on arithmetic: a draw.  With round 4's arithmetic (buffer addressing,
  // five-operation reciprocal, no register moves, the edge taps through DPP's kept destination) they win clearly: 16 tracks 166.5 -> 155.0 us, 8 tracks 85.7 -> 81.1,
  // one frame equal (profiles/r04/al_ab1.txt, interleaved).  LGPU_PBH_ALIGNED=0 keeps the feeder-lane strips.
  a->aligned = blur ? 0 : 1;
  if (!blur && tune(TUNE_PBH_ALIGNED) >= 0) a->aligned = tune(TUNE_PBH_ALIGNED) ? 1 : 0;
  a->strips = (int)cdiv((unsigned)a->dw, blur ? 120 : a->aligned ? 128 : 124);
  a->cgroups = (a->strips + 3) / 4;
  a->ntracks = ntracks;
  // Band height.  Short bands win (profiles/r03/pbh_sweep*.txt, r04/al_ab1.txt: 4-6 rows within 1 %, 8 rows 5 % behind, 12 rows further).  One frame: 6 rows.  A launch of
  // more than one generation of workgroups: 5 rows -- under the round-robin work order (pb_chain_half) the band count that fills whole generations no longer matters and
  // finer bands sweep the frame more evenly (profiles/r04/band_count_sweep.txt: 16 tracks 160 bands 143.6-144.4 us, 192 142.2-142.7, **216 141.6**, 256 143.4, 288 143.1;
  // rounds 3 / 4 chose 160 = whole generations).  With the blur a band computes th + 4 scaled rows: its own rule below.
  const int per_cu = blur ? 4 : 8;                       // resident workgroups per CU (98 / <= 64 VGPRs)
  const long long slots = (long long)device_cus() * per_cu, cols = (long long)a->cgroups * ntracks;
  int bands = (int)cdiv((unsigned)a->dh, 6u);
  if (blur) {
    // The straight-line walk (round 5) runs at four workgroups per CU (98 registers).  ONE workgroup per CU and track (two for a single frame) is what every track count
    // from 1 to 16 wants: 1920 wide = 4 column groups x 64 bands of ~17 rows, four tracks = one whole generation.  Against the 48 / 184 bands of before (bands of 22 rows
    // for full-device launches, 6 rows below): 1 track 17.5 -> 16.8 us, 2 28.7 -> 26.2, 3 38.8 -> 34.1, 4 49.4 -> 42.2, 6 72 -> 67, 8 92.0 -> 86.3, 12 134.0 -> 126.1,
    // 16 169.3 -> 165.2; more bands (80 .. 128) and fewer (32 .. 48) lose at every count (profiles/r05/late/blur_generations.txt).  A multiple of 8: every XCD the same
    // number of a track's bands.  Bands of 6 .. 34 rows.
    const int per_track = device_cus() * (ntracks == 1 ? 2 : 1);
    bands = std::max(8, per_track / std::max(1, a->cgroups) / 8 * 8);
    bands = std::min(bands, std::max(1, a->dh / 6));
    bands = std::max(bands, (int)cdiv((unsigned)a->dh, 34u));
    // The all-opaque instantiation (170 vector instructions per step instead of 232, 82 registers) is no longer bound by its arithmetic: shorter bands -- more waves
    // in flight, four halo rows or not -- win.  16 tracks: 64 bands 160.8 us, 80 155.9, 112 154.3, **120 152.1 / 136 151.8**, 144 156.1, 184 157.3; 8 tracks 84.0 /
    // 81.1 / 81.8 / **79.2** / 79.6 / 81.9 / 81.5; one frame 21.1 (64) .. 16.9 (120) .. **14.9 (184)** (profiles/r06/blur_opaque_bands.txt).  The general kernel
    // (arithmetic-bound) loses with them: 171.6 us at 120 bands against 162.2 at 64.
    if (opaque) bands = ntracks == 1 ? std::max(8, a->dh / 6 / 8 * 8) : std::max(8, a->dh / 9 / 8 * 8);
  }
  if (!blur && cols * bands > slots) bands = 8 * std::max(1, (a->dh + 20) / 40);       // ~5 rows per band, a multiple of 8: every XCD then owns the same number of bands (pb_chain_half)
  { const int v = tune(TUNE_PBH_TH); if (v >= 1 && v <= 1024) bands = (int)cdiv((unsigned)a->dh, (unsigned)v); else if (v > 100000) bands = v - 100000; }       // tuning probe / tests: bands of (about) v rows, or 100000 + the number of bands
  pb_half_bands(a, bands);
}

static unsigned pb_half_grid(const PbHalfArgs &a) { return a.row_major == 2 ? 8u * (unsigned)(a.cgroups * a.ntracks * a.bgroup) * cdiv(cdiv((unsigned)a.bands, (unsigned)a.bgroup), 8u) : 8u * cdiv((unsigned)(a.cgroups * a.bands * a.ntracks), 8u); }

// the fused chain on the pixbuf arithmetic (lgpu_chain with LGPU_INTERP_PIXBUF): convert -> gdk-pixbuf 2:1 scale -> chroma blend -> gamma LUT in one launch.
// LGPU_E_UNSUPPORTED when the geometry is not the exact aligned 2:1 case (the caller then runs the stages one by one).
int pb_chain_half(const lgpu_chain_params *pr, const lgpu_canvas *cv, const lgpu_chain_track *tracks, int ntracks, hipStream_t st, const uint8_t *amounts = nullptr) {
  const int interp = pr->interp & 0xFF;
  if (interp != 2 && interp != 3) return LGPU_E_UNSUPPORTED;
  PbPin pin;
  int rc = pb_table(interp, pr->sw, pr->sh, pr->dw, pr->dh, st, &pin);
  if (rc) return rc;
  const PbTable *t = pin.t;
  uintptr_t sb = (uintptr_t)pr->irow, db = (uintptr_t)pr->orow | (uintptr_t)pr->irow2;
  for (int i = 0; i < ntracks; i++) { sb |= (uintptr_t)tracks[i].src_d; db |= (uintptr_t)tracks[i].dst_d | (uintptr_t)tracks[i].layer2_d; }
  PbHalfArgs a;
  if (cv && (cv->offs_x & 1)) return LGPU_E_UNSUPPORTED;        // 8-byte stores: the frame must start on an even canvas column
  if (!pb_half_ok(t, interp, pr->sw, pr->sh, pr->dw, pr->dh, sb, db, &a.hyper, &a.ashift)) return LGPU_E_UNSUPPORTED;
  {
    // k_pb_half addresses its frames through 32-bit buffer offsets (row * rowstride as a scalar offset): a plane of 2 GiB or more -- a sub-rectangle of a huge atlas
    // with its rowstride -- keeps the general kernels and their 64-bit row addresses
    const long long out_rows = cv ? cv->nheight : pr->dh, lim = 1ll << 31;
    if ((long long)pr->sh * pr->irow >= lim || out_rows * pr->orow >= lim || out_rows * pr->irow2 >= lim) return LGPU_E_UNSUPPORTED;
  }
  if ((rc = get_kscale(&a.kscale))) return rc;
  a.sw = pr->sw; a.sh = pr->sh; a.irow = pr->irow; a.dw = pr->dw; a.dh = pr->dh; a.orow = pr->orow;
  a.swap_rb = pr->swap_rb ? 1 : 0; a.blend = 1; a.irow2 = pr->irow2; a.use_lut = pr->use_lut ? 1 : 0; a.bf = (uint32_t)pr->bf & 0xFF; a.bf_d = pr->param_block_d;
  a.bf_tracks = amounts ? 1 : 0;
  a.nt_out = 1;
  // (non-temporal loads for a band's inner source rows were measured and dropped: a gain only on buffers the memory-side cache still holds, profiles/r04/nt_cold_ab.txt)
  // Work order (PBH_ORDER; profiles/r04/order_ab.txt, interleaved on cold buffers).  0: a column group's bands one after the other, an XCD owning a contiguous run (rounds 3 / 4);
  // 1: the column groups of a band one after the other -- consecutive workgroups read a band across the whole row: 16 tracks 147.5 -> 142.3 us on one box, 152 -> 146 on another;
  // 2: that, with the bands of a track dealt round robin to the XCDs, so that the whole device sweeps ONE frame at a time like a linear stream does: 151-152 -> 144.2, 8 tracks
  // 76.8 -> 74.3.  With the 5x5 gaussian in the chain neighbouring bands share ten source rows instead of two and want the same L2: 186 us with 1, 189 with 2 -- it keeps 1.
  a.row_major = tune(TUNE_PBH_ORDER) >= 0 && tune(TUNE_PBH_ORDER) <= 2 ? tune(TUNE_PBH_ORDER) : -1;       // -1: decided below, once the geometry is known
  // ... in groups of `bgroup` neighbouring bands per XCD turn.  What matters is that every XCD gets the SAME number of bands of a track (profiles/r04/group_ab.txt: 216 bands
  // in groups of 1 / 3 / 9 / 27 -- 27 / 9 / 3 / 1 turns per XCD -- 138.2-140.3 us; groups of 2 / 4 / 5 / 13 / 25, which leave some XCDs a turn short, 143 / 143 / 152 / 166 / 194);
  // among the balanced ones the largest group re-reads least: L2 -> fabric reads 769.5 / 698.8 / 675.2 / 667.3 MB per launch against 663.6 MB of source + layer 2.
  // So: an eighth of the track's bands per XCD when the band count is a multiple of 8 (pb_half_geometry makes it one for full-device launches), else band by band.
  a.bgroup = tune(TUNE_PBH_GROUP) >= 1 && tune(TUNE_PBH_GROUP) <= 4096 ? tune(TUNE_PBH_GROUP) : 0;
  pb_half_geometry(&a, ntracks, pr->do_blur ? 1 : 0, (pr->do_blur && (pr->interp & LGPU_INTERP_OPAQUE)) ? 1 : 0);
  // a launch that fits one generation of workgroups (one or two 4K tracks) is a matter of latency, not of streaming order: order 1 there (graph replay, one frame 12.09 us
  // against 12.45 with order 2 and 12.24 with order 0; config 3 12.19 / 12.62 / 12.33; two tracks 20.0 / 21.05 / 20.5)
  // With the gaussian (four resident workgroups per CU, one per CU and track): order 2 from a whole generation on -- 4 tracks 42.3 -> 41.9 us, 8 88.7 -> 86.9, 16 166.5 -> 164.5;
  // one and two tracks keep order 1 (16.9 / 26.4 against 17.1 / 27.2).  Round 4's kernel, with its taller bands, had it the other way round.
  if (a.row_major < 0) a.row_major = ((long long)a.cgroups * a.bands * ntracks >= (long long)device_cus() * (pr->do_blur ? 4 : 8) + (pr->do_blur ? 0 : 1)) ? 2 : 1;
  if (a.bgroup <= 0) a.bgroup = (a.bands % 8 == 0) ? a.bands / 8 : 1;
  a.cw = a.ch = a.ox = a.oy = 0; a.bar_blocks = 0;
  if (cv) { a.cw = cv->nwidth; a.ch = cv->nheight; a.ox = cv->offs_x; a.oy = cv->offs_y; a.bar_blocks = (int)cdiv((unsigned)(a.cw * a.ch - a.dw * a.dh), 1024u); }
  a.main_blocks = (int)pb_half_grid(a);
  a.bar_first = (a.bar_blocks * ntracks + 7) & ~7;
  PbTracks T;
  for (int i = 0; i < ntracks; i++) { T.src[i] = tracks[i].src_d; T.l2[i] = tracks[i].layer2_d; T.dst[i] = tracks[i].dst_d; T.bf[i] = amounts ? amounts[i] : 0; }
  const Lut8 l = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
  const dim3 grid((unsigned)(a.main_blocks + a.bar_first));
  // Workgroups per CU (PBH_OCC): a launch of more than one generation runs FIVE workgroups per CU instead of the eight its registers allow -- unused dynamic LDS is what
  // holds the others back.  Fewer bands in flight = a narrower window of the frames being streamed at any moment (profiles/r04/occupancy_sweep.txt: 8 / 7 / 6 / 5 / 4 / 3
  // per CU 142.3 / 142.2 / 141.4 / 139.2 / 142.3 / 180 us per 16-track launch, 8 tracks 74.1 -> 72.8; with the gaussian 184 -> 187, so that chain keeps its six).
  int occ = (!pr->do_blur && (long long)a.cgroups * a.bands * ntracks > (long long)device_cus() * 8) ? 5 : 0;
  if (tune(TUNE_PBH_OCC) >= 0) occ = tune(TUNE_PBH_OCC);
  const size_t occ_lds = occ > 0 && occ < 16 ? (size_t)(160 * 1024) / (size_t)occ - 3072 : 0;
  const bool noblend = (pr->interp & LGPU_INTERP_NOBLEND) != 0;        // (never with the gaussian: pb_chain keeps that pair staged)
#define PBH_LAUNCH(HY, BL, AL, SW) { if (noblend && !(BL)) hipLaunchKernelGGL((k_pb_half<(BL) ? 1 : 2, HY, BL, AL, SW>), grid, dim3(256), occ_lds, st, a, T, l); \
                                     else hipLaunchKernelGGL((k_pb_half<1, HY, BL, AL, SW>), grid, dim3(256), occ_lds, st, a, T, l); }
#define PBH_SWAP(HY, BL, AL) do { if (a.swap_rb) PBH_LAUNCH(HY, BL, AL, 1) else PBH_LAUNCH(HY, BL, AL, 0) } while (0)
#define PBH_OPAQUE(HY, SW) hipLaunchKernelGGL((k_pb_half<1, HY, 1, 0, SW, 1>), grid, dim3(256), occ_lds, st, a, T, l)
  if (pr->do_blur && (pr->interp & LGPU_INTERP_OPAQUE)) {
    // the caller's word that every source pixel is opaque: the vector-bound gaussian chain sheds a quarter of its instructions (pb_half_hrow)
    if ((rc = pb_opaque_check())) return rc;
    if (a.hyper) { if (a.swap_rb) PBH_OPAQUE(1, 1); else PBH_OPAQUE(1, 0); }
    else { if (a.swap_rb) PBH_OPAQUE(0, 1); else PBH_OPAQUE(0, 0); }
  } else if (pr->do_blur) {
    if (a.hyper) PBH_SWAP(1, 1, 0); else PBH_SWAP(0, 1, 0);
  } else if (a.aligned) {
    if (a.hyper) PBH_SWAP(1, 0, 1); else PBH_SWAP(0, 0, 1);
  } else {
    if (a.hyper) PBH_SWAP(1, 0, 0); else PBH_SWAP(0, 0, 0);
  }
#undef PBH_OPAQUE
#undef PBH_SWAP
#undef PBH_LAUNCH
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// lgpu_chain / lgpu_chain_canvas.  One fused launch for the exact aligned 2:1 case on the pixbuf arithmetic (blur stage and letterbox canvas included); otherwise the
// stages run one after the other through stream-ordered scratch frames: [bars] -> scale (into the canvas) -> [5x5 gaussian] -> [R <-> B] + chroma blend + gamma LUT.
// The channel swap commutes with the scalers and the gaussian (all treat the three colour bytes alike), so it rides in the last kernel.
int pb_scale_fused(const uint8_t *const *srcs, uint8_t *const *dsts, int n, int irow, int sw, int sh, int orow, int dw, int dh, int interp, hipStream_t st, const PbEpi *epi);
int pb_chain(const lgpu_chain_params *pr, const lgpu_canvas *cv, const lgpu_chain_track *tracks, int ntracks, hipStream_t st, const uint8_t *amounts) {
  int rc;
  const bool pixbuf = (pr->interp & LGPU_INTERP_PIXBUF) != 0, noblend = (pr->interp & LGPU_INTERP_NOBLEND) != 0;
  if (pr->sw == pr->dw && pr->sh == pr->dh) {             // no resize stage (lgpu_chain_amounts only; the gaussian is not offered here)
    if (pr->do_blur) { set_error("lgpu_chain_amounts: the gaussian needs the resize stage"); return LGPU_E_UNSUPPORTED; }
    PbEpi e;
    PbTracks T;
    for (int i = 0; i < ntracks; i++) { T.src[i] = tracks[i].src_d; T.l2[i] = tracks[i].layer2_d; T.dst[i] = tracks[i].dst_d; T.bf[i] = e.bf[i] = amounts ? amounts[i] : 0; e.l2[i] = tracks[i].layer2_d; }
    e.bf_d = amounts ? nullptr : pr->param_block_d; e.bf0 = (uint32_t)pr->bf & 0xFF; e.irow2 = pr->irow2; e.swap_rb = pr->swap_rb ? 1 : 0; e.use_lut = pr->use_lut ? 1 : 0;
    e.bf_tracks = amounts ? 1 : 0; e.blend = noblend ? 0 : 1; e.lut = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
    const int cw = cv ? cv->nwidth : pr->dw, ch = cv ? cv->nheight : pr->dh;
    hipLaunchKernelGGL(k_pb_flat_n, dim3(cdiv((unsigned)cw, 64), cdiv((unsigned)ch, 4), (unsigned)ntracks), dim3(256), 0, st, T, e, pr->irow, pr->orow, cw, ch,
                       cv ? cv->offs_x : 0, cv ? cv->offs_y : 0, pr->dw, pr->dh);
    if (hipGetLastError() != hipSuccess) { set_error("k_pb_flat_n launch failed"); return LGPU_E_HIP; }
    return LGPU_OK;
  }
  if (pixbuf && !(cv && pr->do_blur) && !(noblend && pr->do_blur)) {
    rc = pb_chain_half(pr, cv, tracks, ntracks, st, amounts);         // one launch
    if (tune_on(TUNE_PLAN_DEBUG)) fprintf(stderr, "pb_chain: one-launch form rc %d (%s)\n", rc, rc ? lgpu_last_error() : "ok");
    if (rc != LGPU_E_UNSUPPORTED) return rc;
  }
  // any other ratio, no gaussian, no canvas: the scaler of that ratio with the chain's last stages in its store -- one launch as well
  // (a letterbox canvas: the scaled frame lands at its place in the canvas, its layer-2 pixels are the canvas's there, and the bars are one more small launch)
  if (pixbuf && !pr->do_blur && !tune_on(TUNE_PB_CHAIN_GROUP)) {
    PbEpi e;
    PbTracks T;
    const uint8_t *srcs[LGPU_CHAIN_MAX_TRACKS];
    uint8_t *dsts[LGPU_CHAIN_MAX_TRACKS];
    const size_t doff = cv ? (size_t)cv->offs_y * pr->orow + 4 * (size_t)cv->offs_x : 0, loff = cv ? (size_t)cv->offs_y * pr->irow2 + 4 * (size_t)cv->offs_x : 0;
    for (int i = 0; i < ntracks; i++) {
      srcs[i] = tracks[i].src_d; dsts[i] = tracks[i].dst_d + doff; e.l2[i] = tracks[i].layer2_d + loff; e.bf[i] = amounts ? amounts[i] : 0;
      T.src[i] = nullptr; T.l2[i] = tracks[i].layer2_d; T.dst[i] = tracks[i].dst_d; T.bf[i] = e.bf[i];
    }
    e.bf_d = amounts ? nullptr : pr->param_block_d; e.bf0 = (uint32_t)pr->bf & 0xFF; e.irow2 = pr->irow2; e.swap_rb = pr->swap_rb ? 1 : 0; e.use_lut = pr->use_lut ? 1 : 0;
    e.bf_tracks = amounts ? 1 : 0; e.blend = noblend ? 0 : 1; e.lut = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
    rc = pb_scale_fused(srcs, dsts, ntracks, pr->irow, pr->sw, pr->sh, pr->orow, pr->dw, pr->dh, (pr->interp & 0xFF) | (pr->interp & LGPU_INTERP_OPAQUE), st, &e);
    if (tune_on(TUNE_PLAN_DEBUG)) fprintf(stderr, "pb_chain: scaler with the chain's last stages rc %d\n", rc);
    if (rc == LGPU_OK && cv && (cv->nwidth != pr->dw || cv->nheight != pr->dh)) {
      for (int i = 0; i < ntracks; i++) e.l2[i] = tracks[i].layer2_d;      // the bars read layer 2 at canvas coordinates
      hipLaunchKernelGGL(k_pb_bars_epi, dim3(cdiv((unsigned)cv->nwidth, 64), cdiv((unsigned)cv->nheight, 4), (unsigned)ntracks), dim3(256), 0, st, T, e, pr->orow,
                         cv->nwidth, cv->nheight, cv->offs_x, cv->offs_y, pr->dw, pr->dh);
      if (hipGetLastError() != hipSuccess) { set_error("k_pb_bars_epi launch failed"); return LGPU_E_HIP; }
    }
    if (rc != PB_NOT_FUSED) return rc;
  }
  const int interp = pr->interp & 0xFF;
  const int cw = cv ? cv->nwidth : pr->dw, ch = cv ? cv->nheight : pr->dh, ox = cv ? cv->offs_x : 0, oy = cv ? cv->offs_y : 0;
  const size_t per = (size_t)cw * 4 * ch;
  // The tracks go through the stages in GROUPS (as many as fit 256 MB of scratch frames): the scaler serves a group in one launch (lgpu_pixbuf_scale_batch: the
  // weight table, the launch and the ramp of a generation are paid once; 16 x 1080p -> 720p: 114 us against 16 x 11), and so does the last kernel.
  int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)ntracks, ((size_t)256 << 20) / per));
  if (tune(TUNE_PB_CHAIN_GROUP) >= 1) group = std::min(group, tune(TUNE_PB_CHAIN_GROUP));      // measurement: LGPU_PB_CHAIN_GROUP=1 is the track-by-track walk
  void *sa = nullptr, *sb = nullptr;
  if ((rc = lgpu_malloc_ordered(&sa, per * group, st))) return rc;
  if (pr->do_blur && (rc = lgpu_malloc_ordered(&sb, per * group, st))) { lgpu_free_ordered(sa, st); return rc; }
  const Lut8 l = pack_lut(pr->use_lut ? pr->lut8 : nullptr);
  const uint8_t black[4] = {0, 0, 0, 255};
  for (int g0 = 0; g0 < ntracks && !rc; g0 += group) {
    const int n = std::min(group, ntracks - g0);
    const uint8_t *srcs[LGPU_CHAIN_MAX_TRACKS];
    uint8_t *inner[LGPU_CHAIN_MAX_TRACKS];
    PbTracks T;
    for (int i = 0; i < n; i++) {
      uint8_t *frame = (uint8_t *)sa + (size_t)i * per;
      srcs[i] = tracks[g0 + i].src_d; inner[i] = frame + (size_t)oy * cw * 4 + (size_t)ox * 4;
      T.src[i] = nullptr; T.l2[i] = tracks[g0 + i].layer2_d; T.dst[i] = tracks[g0 + i].dst_d; T.bf[i] = amounts ? amounts[g0 + i] : 0;
      if (cv && !rc) rc = lgpu_letterbox_bars(frame, cw * 4, cw, ch, 4, black, ox, oy, pr->dw, pr->dh, st);
    }
    if (!rc) {
      if (pixbuf) rc = lgpu_pixbuf_scale_batch(srcs, inner, n, pr->irow, pr->sw, pr->sh, cw * 4, pr->dw, pr->dh, 4, interp | (pr->interp & LGPU_INTERP_OPAQUE), st);
      else for (int i = 0; i < n && !rc; i++) rc = lgpu_resize(srcs[i], pr->irow, pr->sw, pr->sh, inner[i], cw * 4, pr->dw, pr->dh, 4, interp, nullptr, st);
    }
    const uint8_t *trk = (const uint8_t *)sa;
    if (!rc && pr->do_blur) {
      for (int i = 0; i < n && !rc; i++) rc = lgpu_gauss5((const uint8_t *)sa + (size_t)i * per, cw * 4, (uint8_t *)sb + (size_t)i * per, cw * 4, cw, ch, 4, st);
      trk = (const uint8_t *)sb;
    }
    if (rc) break;
    hipLaunchKernelGGL(k_pb_epilogue_n, dim3(cdiv((unsigned)cw, 64), cdiv((unsigned)ch, 4), (unsigned)n), dim3(256), 0, st, trk, per, cw * 4, T, (amounts ? 1 : 0) | (noblend ? 2 : 0), pr->irow2,
                       pr->orow, cw, ch, pr->swap_rb ? 1 : 0, (uint32_t)pr->bf & 0xFF, amounts ? nullptr : pr->param_block_d, pr->use_lut ? 1 : 0, l);
    if (hipGetLastError() != hipSuccess) { set_error("k_pb_epilogue_n launch failed"); rc = LGPU_E_HIP; }
  }
  lgpu_free_ordered(sa, st);
  if (sb) lgpu_free_ordered(sb, st);
  return rc;
}

}  // namespace lgpu

using namespace lgpu;

// measurement hook (bench.py's roofline.box_class): the chain's own ALGORITHMIC bytes as a bare stream -- every source and layer-2 byte of the tracks read once with
// 16-byte loads, every destination byte written once with non-temporal 16-byte stores (an XOR fold of what was read, so nothing can be elided) -- timed with HIP events
// on `stream` around `reps` launches.  What this box's memory system gives this read : write mix with no arithmetic and no re-reads; the destinations are left dirty.
__global__ __launch_bounds__(256) void k_chain_stream_probe(const PbTracks T, int ntracks, unsigned src_q, unsigned dst_q) {      // sizes in 16-byte quads per track
  const unsigned per = dst_q, nthreads = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int t = 0; t < ntracks; t++) {
    const pb_u4 *s = reinterpret_cast<const pb_u4 *>(T.src[t]), *l = reinterpret_cast<const pb_u4 *>(T.l2[t]);
    pb_u4 *d = reinterpret_cast<pb_u4 *>(T.dst[t]);
    for (unsigned i = tid; i < per; i += nthreads) {
      pb_u4 a = __builtin_nontemporal_load(l + i);
      const unsigned n = src_q / dst_q;          // 4 source quads per destination quad at 2:1
      for (unsigned j = 0; j < n; j++) { const pb_u4 b = __builtin_nontemporal_load(s + (size_t)j * per + i); a.x ^= b.x; a.y ^= b.y; a.z ^= b.z; a.w ^= b.w; }
      __builtin_nontemporal_store(a, d + i);
    }
  }
}
extern "C" int lgpu_debug_stream_probe(const lgpu_chain_params *pr, const lgpu_chain_track *tracks, int ntracks, int reps, float *ms_total, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(pr && tracks && ms_total && reps > 0 && ntracks > 0 && ntracks <= LGPU_CHAIN_MAX_TRACKS, "bad arguments");
  const size_t sb = (size_t)pr->irow * pr->sh, db = (size_t)pr->orow * pr->dh;
  LGPU_REQUIRE(pr->irow == pr->sw * 4 && pr->orow == pr->dw * 4 && pr->irow2 == pr->orow && db % 16 == 0 && sb % db == 0, "compact frames whose source is a whole multiple of the destination");
  PbTracks T;
  for (int i = 0; i < ntracks; i++) { T.src[i] = tracks[i].src_d; T.l2[i] = tracks[i].layer2_d; T.dst[i] = tracks[i].dst_d; }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  LGPU_HIP(hipEventCreate(&e0));
  LGPU_HIP(hipEventCreate(&e1));
  LGPU_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_chain_stream_probe, dim3((unsigned)device_cus() * 8u), dim3(256), 0, st, T, ntracks, (unsigned)(sb / 16), (unsigned)(db / 16));
  LGPU_HIP(hipEventRecord(e1, st));
  LGPU_HIP(hipEventSynchronize(e1));
  LGPU_HIP(hipEventElapsedTime(ms_total, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// test hook: how many integers a of [lo, hi) give pb_recip(a) != 1.0 / (double)a on the device (0 for the whole 24-bit range, tests/test_pixbuf_scale.py)
extern "C" int lgpu_debug_recip_check(uint32_t lo, uint32_t hi, unsigned long long *mismatches) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(mismatches && hi > lo && hi <= (1u << 24), "range must lie inside [0, 2^24)");
  unsigned long long *d = nullptr;
  LGPU_HIP(hipMalloc((void **)&d, sizeof *d));
  LGPU_HIP(hipMemset(d, 0, sizeof *d));
  hipLaunchKernelGGL(k_pb_recip_check, dim3(cdiv(hi - lo, 256u)), dim3(256), 0, 0, lo, hi, d);
  const hipError_t e = hipMemcpy(mismatches, d, sizeof *d, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error("lgpu_debug_recip_check: %s", hipGetErrorString(e)); return LGPU_E_HIP; }
  return LGPU_OK;
}

extern "C" int lgpu_pixbuf_weights(int interp, int sw, int sh, int dw, int dh, int *n_x, int *n_y, int *xoff, int *yoff, int32_t *table, size_t table_ints) {
  if ((interp != 2 && interp != 3) || sw < 1 || sh < 1 || dw < 1 || dh < 1 || !n_x || !n_y || !xoff || !yoff) return LGPU_E_BADARG;
  PbTable t;
  const int rc = pb_build(interp, sw, sh, dw, dh, &t, false);
  *n_x = t.n_x; *n_y = t.n_y; *xoff = t.xoff; *yoff = t.yoff;
  if (rc) return rc;
  if (table) {
    if (table_ints < t.host.size()) return LGPU_E_BADARG;
    __builtin_memcpy(table, t.host.data(), t.host.size() * sizeof(int));
  }
  return LGPU_OK;
}

// n frames of one geometry in one launch of whichever kernel the geometry selects (n == 1: lgpu_pixbuf_scale)
// epi: the chain's last stages in the store (PbEpi) -- served by the pair, gather and enlargement kernels; PB_NOT_FUSED (nothing launched) where another kernel has the ratio
static int pb_scale_n(const uint8_t *const *srcs, uint8_t *const *dsts, int n, int irow, int sw, int sh, int orow, int dw, int dh, int channels, int interp, void *stream,
                      const PbEpi *epi = nullptr) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(srcs && dsts && n >= 1 && n <= LGPU_CHAIN_MAX_TRACKS, "1..64 frames");
  LGPU_REQUIRE(sw > 0 && sh > 0 && dw > 0 && dh > 0, "empty geometry");
  PbFrames F;
  uintptr_t sbits = 0, dbits = 0;                       // alignment of the launch = of its least aligned frame
  for (int i = 0; i < n; i++) {
    LGPU_REQUIRE(srcs[i] && dsts[i], "null frame");
    LGPU_REQUIRE(srcs[i] != dsts[i], "scaling cannot run in place");
    F.src[i] = srcs[i]; F.dst[i] = dsts[i];
    sbits |= (uintptr_t)srcs[i]; dbits |= (uintptr_t)dsts[i];
  }
  const uint8_t *const src_d = srcs[0];
  uint8_t *const dst_d = dsts[0];
  LGPU_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (no alpha) or 4 (alpha)");
  const bool opaque = (interp & LGPU_INTERP_OPAQUE) != 0;          // the caller's word that every source pixel has alpha 255: the pair kernel's lighter form (k_pb_pairs<.., OPQ>)
  interp &= ~LGPU_INTERP_OPAQUE;
  LGPU_REQUIRE(interp == 0 || interp == 2 || interp == 3, "interp must be 0 (NEAREST), 2 (BILINEAR) or 3 (HYPER)");
  LGPU_REQUIRE(irow >= sw * channels && orow >= dw * channels, "rowstride smaller than a row");
  LGPU_REQUIRE(sw < 32768 && sh < 32768 && dw < 32768 && dh < 32768, "frame sides must stay below 32768 (16.16 positions)");
  if (channels == 4) LGPU_REQUIRE(((sbits | dbits | (unsigned)irow | (unsigned)orow) & 3) == 0, "4-byte pixels must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (epi && (channels != 4 || interp == 0 || (dw == sw && dh == sh))) return PB_NOT_FUSED;
  if (dw == sw && dh == sh) {                           // gdk_pixbuf_scale_simple: a plain copy
    for (int i = 0; i < n && !rc; i++) rc = lgpu_copy_rows(dsts[i], orow, srcs[i], irow, sw * channels, sh, stream);
    return rc;
  }
  const double scale_x = (double)dw / sw, scale_y = (double)dh / sh;
  const int x_step = (int)(65536 / scale_x), y_step = (int)(65536 / scale_y);
  if (x_step == 0 || y_step == 0) { set_error("lgpu_pixbuf_scale: enlargement beyond 65536x"); return LGPU_E_UNSUPPORTED; }
  const dim3 grid(cdiv((unsigned)dw, 64), cdiv((unsigned)dh, 4), (unsigned)n), block(256);
  if (interp == 0) {
    if (channels == 4) hipLaunchKernelGGL(k_pb_nearest<4>, grid, block, 0, st, F, irow, sw, sh, orow, dw, dh, x_step, y_step);
    else hipLaunchKernelGGL(k_pb_nearest<3>, grid, block, 0, st, F, irow, sw, sh, orow, dw, dh, x_step, y_step);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  PbPin pin;
  rc = pb_table(interp, sw, sh, dw, dh, st, &pin);
  const PbTable *t = pin.t;
  if (rc) {
    if (rc == LGPU_E_UNSUPPORTED) set_error("lgpu_pixbuf_scale: %dx%d -> %dx%d needs %d x %d taps; the library's two-step scaler is not covered", sw, sh, dw, dh, t->n_x, t->n_y);
    return rc;
  }
  if (!epi && channels == 4 && dw == 2 * sw && dh == 2 * sh && (sw & 1) == 0 && ((sbits | (unsigned)irow) & 7) == 0 && ((dbits | (unsigned)orow) & 15) == 0 &&
      pb_double_ok(t, x_step, y_step) && !tune_on(TUNE_PB_NO_DOUBLE)) {
    PbHalfArgs h;
    h.sw = sw; h.sh = sh; h.irow = irow; h.dw = dw; h.dh = dh; h.orow = orow;
    h.strips = (int)cdiv((unsigned)sw, 124); h.cgroups = (h.strips + 3) / 4;
    h.th = 3;                      // measured (1080p -> 4K): 16.5 us at 2-3 source rows per band, 18.2 us at 4, 22.8 at 8, 33.4 at 16, 52.7 at 32 -- per-wave latency, as in k_pb_half
    h.bands = (int)cdiv((unsigned)sh, (unsigned)h.th); h.ntracks = 1;
    hipLaunchKernelGGL(k_pb_double<0>, dim3(8u * cdiv((unsigned)(h.cgroups * h.bands), 8u), 1, (unsigned)n), dim3(256), 0, st, h, F);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if (channels == 3 && sw == 2 * dw && sh == 2 * dh && (sw & 7) == 0 && ((sbits | dbits | (unsigned)irow | (unsigned)orow) & 3) == 0 && !tune_on(TUNE_PB_NO_HALF3)) {
    PbHalfArgs h;
    // the same table check as the 4-byte kernel (alignment arguments that always pass: this kernel's own are checked above)
    if (pb_half_ok(t, interp, sw, sh, dw, dh, 0, 0, &h.hyper, &h.ashift)) {
      h.sw = sw; h.sh = sh; h.irow = irow; h.dw = dw; h.dh = dh; h.orow = orow;
      h.strips = (int)cdiv((unsigned)dw, 120); h.cgroups = (h.strips + 3) / 4; h.th = 6; h.bands = (int)cdiv((unsigned)dh, 6u); h.ntracks = 1;
      const dim3 g3(8u * cdiv((unsigned)(h.cgroups * h.bands), 8u), 1, (unsigned)n);
      if (h.hyper) hipLaunchKernelGGL(k_pb_half3<1>, g3, dim3(256), 0, st, h, F);
      else hipLaunchKernelGGL(k_pb_half3<0>, g3, dim3(256), 0, st, h, F);
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if (channels == 4 && !epi) {
    PbHalfArgs h;
    if ((long long)sh * irow < (1ll << 31) && (long long)dh * orow < (1ll << 31) &&        // 32-bit buffer offsets in k_pb_half (see pb_chain_half)
        pb_half_ok(t, interp, sw, sh, dw, dh, sbits | (uintptr_t)irow, dbits | (uintptr_t)orow, &h.hyper, &h.ashift)) {
      h.sw = sw; h.sh = sh; h.irow = irow; h.dw = dw; h.dh = dh; h.orow = orow;
      h.swap_rb = 0; h.blend = 0; h.irow2 = 0; h.use_lut = 0; h.bf = 0; h.bf_d = nullptr; h.nt_out = 0; h.row_major = tune(TUNE_PBH_ORDER) >= 0 && tune(TUNE_PBH_ORDER) <= 2 ? tune(TUNE_PBH_ORDER) : 1; h.bgroup = tune(TUNE_PBH_GROUP) >= 1 && tune(TUNE_PBH_GROUP) <= 4096 ? tune(TUNE_PBH_GROUP) : 0;
      pb_half_geometry(&h, n);
      if (tune(TUNE_PBH_ORDER) < 0 && (long long)h.cgroups * h.bands * n > (long long)device_cus() * 8) h.row_major = 2;      // more than one generation: the chain's sweep order (pb_chain_half)
      if (h.bgroup <= 0) h.bgroup = (h.bands % 8 == 0) ? h.bands / 8 : 1;
      PbTracks T;
      for (int i = 0; i < n; i++) { T.src[i] = srcs[i]; T.l2[i] = nullptr; T.dst[i] = dsts[i]; }
      h.kscale = nullptr; h.cw = h.ch = h.ox = h.oy = 0; h.bar_blocks = 0; h.bar_first = 0; h.main_blocks = (int)pb_half_grid(h);
      if (h.aligned) {
        if (h.hyper) hipLaunchKernelGGL((k_pb_half<0, 1, 0, 1>), dim3(pb_half_grid(h)), dim3(256), 0, st, h, T, pack_lut(nullptr));
        else hipLaunchKernelGGL((k_pb_half<0, 0, 0, 1>), dim3(pb_half_grid(h)), dim3(256), 0, st, h, T, pack_lut(nullptr));
      } else if (h.hyper) hipLaunchKernelGGL((k_pb_half<0, 1, 0>), dim3(pb_half_grid(h)), dim3(256), 0, st, h, T, pack_lut(nullptr));
      else hipLaunchKernelGGL((k_pb_half<0, 0, 0>), dim3(pb_half_grid(h)), dim3(256), 0, st, h, T, pack_lut(nullptr));
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  const bool no_pairs = tune_on(TUNE_PB_NO_PAIRS);        // tests: the one-tap-per-operation kernels at any ratio
  const bool no_gather = tune_on(TUNE_PB_NO_GATHER);
  // both sides enlarged: the register-window walk
  const bool no_up = tune_on(TUNE_PB_NO_UP);
  if (channels == 4 && t->gpairs_d && !no_pairs && !no_up && x_step <= 65536 && y_step <= 65536) {
    const int unp = (t->tx1 - t->tx0 + 1) / 2, uny = t->ty1 - t->ty0;
    if (unp >= 1 && unp <= 2 && uny >= 1 && uny <= 4) {
      PbUpArgs ua;
      ua.src = src_d; ua.dst = dst_d; ua.irow = irow; ua.orow = orow; ua.sw = sw; ua.sh = sh; ua.dw = dw; ua.dh = dh;
      ua.x_step = x_step; ua.y_step = y_step; ua.xoff = t->xoff; ua.yoff = t->yoff; ua.tx0 = t->tx0; ua.ty0 = t->ty0;
      ua.rb = (long long)dw * dh >= 6000000 ? 8 : 6;      // per-wave time rules (profiles/r03/pb_up_ab.txt: 720p -> 1080p 11.6 us at 6 rows per band, 15.3 at 16, 38 at 64)
      { const int v = tune(TUNE_PB_UP_RB); if (v >= 1 && v <= 4096) ua.rb = v; }      // tuning probe
      const dim3 gu(cdiv(cdiv((unsigned)dw, 64), 4), cdiv((unsigned)dh, (unsigned)ua.rb), (unsigned)n);
      const uint32_t *gp = t->gpairs_d;
      if (opaque && (rc = pb_opaque_check())) return rc;
#define PB_UP(NP_, NY_) { if (epi) { if (opaque) hipLaunchKernelGGL((k_pb_up<NP_, NY_, 1, PbEpi>), gu, block, 0, st, ua, gp, F, *epi); else hipLaunchKernelGGL((k_pb_up<NP_, NY_, 0, PbEpi>), gu, block, 0, st, ua, gp, F, *epi); } \
                          else if (opaque) hipLaunchKernelGGL((k_pb_up<NP_, NY_, 1, PbNoEpi>), gu, block, 0, st, ua, gp, F, PbNoEpi{}); else hipLaunchKernelGGL((k_pb_up<NP_, NY_, 0, PbNoEpi>), gu, block, 0, st, ua, gp, F, PbNoEpi{}); }
      if (unp == 1) { if (uny == 1) PB_UP(1, 1) else if (uny == 2) PB_UP(1, 2) else if (uny == 3) PB_UP(1, 3) else PB_UP(1, 4) }
      else { if (uny == 1) PB_UP(2, 1) else if (uny == 2) PB_UP(2, 2) else if (uny == 3) PB_UP(2, 3) else PB_UP(2, 4) }
#undef PB_UP
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  // integer reductions (one phase for the whole frame): the barrier-free kernel with scalar weights; every other ratio keeps the LDS window
  if (channels == 4 && t->gpairs_d && !no_pairs && !no_gather && (x_step & 0xFFFF) == 0 && (y_step & 0xFFFF) == 0) {
    PbGatherArgs ga;
    ga.src = src_d; ga.dst = dst_d; ga.irow = irow; ga.orow = orow; ga.sw = sw; ga.sh = sh; ga.dw = dw; ga.dh = dh;
    ga.x_step = x_step; ga.y_step = y_step; ga.xoff = t->xoff; ga.yoff = t->yoff; ga.tx0 = t->tx0; ga.ty0 = t->ty0; ga.ny_eff = t->ty1 - t->ty0;
    const int gnp = (t->tx1 - t->tx0 + 1) / 2;
    const uint32_t *gp = t->gpairs_d;
    if (opaque && (rc = pb_opaque_check())) return rc;
#define PB_GATHER(NP_) { if (epi) { if (opaque) hipLaunchKernelGGL((k_pb_gather<NP_, PbEpi, 1>), grid, block, 0, st, ga, gp, F, *epi); else hipLaunchKernelGGL((k_pb_gather<NP_, PbEpi, 0>), grid, block, 0, st, ga, gp, F, *epi); } \
                         else if (opaque) hipLaunchKernelGGL((k_pb_gather<NP_, PbNoEpi, 1>), grid, block, 0, st, ga, gp, F, PbNoEpi{}); else hipLaunchKernelGGL((k_pb_gather<NP_, PbNoEpi, 0>), grid, block, 0, st, ga, gp, F, PbNoEpi{}); }
    if (gnp == 1) PB_GATHER(1) else if (gnp == 2) PB_GATHER(2) else if (gnp == 3) PB_GATHER(3) else PB_GATHER(4)
#undef PB_GATHER
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  if (t->pairs_d && !no_pairs) {
    PbPairArgs pa;
    pa.src = src_d; pa.dst = dst_d; pa.irow = irow; pa.orow = orow; pa.sw = sw; pa.sh = sh; pa.dw = dw; pa.dh = dh;
    pa.x_step = x_step; pa.y_step = y_step; pa.xoff = t->xoff; pa.yoff = t->yoff; pa.n_x = t->n_x; pa.tx0 = t->tx0; pa.ty0 = t->ty0; pa.ny_eff = t->ty1 - t->ty0; pa.nq = t->nq;
    pa.pairs = t->pairs_d;
    pa.rnd = (t->n_x == 2 && t->n_y == 2 && channels == 3) ? 0x8000u : 0xffffu;
    // pairs per window row: the span of 64 destination pixels' first taps, the taps of the last one (padded to whole groups of four pairs), one pair of alignment slack
    pa.wpairs = ((int)((((63LL * x_step + 65535) >> 16) + 1) / 2) + 4 * t->nq + 3 + 3) & ~3;       // whole quads of pairs, and wx0 is rounded down to a multiple of 4 below

    pa.tile_h = 0;
    size_t lds_cap = (tune(TUNE_PB_LDS_KB) > 0 ? (size_t)tune(TUNE_PB_LDS_KB) : 24) * 1024;     // 6 workgroups per CU: the per-lane weight loads want occupancy more than the window wants rows (profiles/r03/pb_pairs_lds_sweep.txt)
    for (int th = 16; th >= 1; th >>= 1) {
      const int wh = (int)(((long long)(th - 1) * y_step + 65535) >> 16) + pa.ny_eff + 1;
      if ((size_t)pa.wpairs * wh * 16 <= lds_cap) { pa.tile_h = th; pa.win_h = wh; break; }
    }
    if (pa.tile_h) {
      dim3 g(cdiv((unsigned)dw, 64), cdiv((unsigned)dh, (unsigned)pa.tile_h), (unsigned)n);
      pa.gx = (int)g.x; pa.gy = (int)g.y; pa.per_xcd = 0;
      // Tile order.  Counters on 4K -> 1706x960 (profiles/r06/fetch_size_calibration.md: FETCH_SIZE reports HALF of what is read for this kernel's 640-byte window
      // segments exactly as for a full-wave stream -- 128-byte requests) put its L2 -> fabric reads at 2.0 x the source: with the tiles of a row dealt round robin to
      // the XCDs, the window rows that vertically neighbouring tiles share (5 of 14) are fetched by another XCD's L2 again.  LGPU_PB_TILE_ORDER=1: every XCD a
      // contiguous run of the row-major tile sequence.
      if (tune(TUNE_PB_TILE_ORDER) != 0 && g.x * g.y >= 64) { pa.per_xcd = (int)cdiv(g.x * g.y, 8u); g = dim3(8u * (unsigned)pa.per_xcd, 1, (unsigned)n); }
      const size_t lds = (size_t)pa.wpairs * pa.win_h * 16;
      const int np = (t->tx1 - t->tx0 + 2) / 2;
#define PB_PRE(CHN, NP_, NY_, OQ, EAT, EAV) { if (pa.tile_h <= 4) hipLaunchKernelGGL((k_pb_pairs<CHN, NP_, NY_, 1, OQ, EAT>), g, block, lds, st, pa, F, EAV); else hipLaunchKernelGGL((k_pb_pairs<CHN, NP_, NY_, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV); }
#define PB_PAIRS(CHN, OQ, EAT, EAV)                                                                                       \
      { const int ny = t->ty1 - t->ty0;                                                                                   \
        const bool pre = !tune_on(TUNE_PB_NO_PRE);                                                                          \
        if (np == 2 && ny == 2) PB_PRE(CHN, 2, 2, OQ, EAT, EAV)                                                              \
        else if (np == 2 && ny == 3) PB_PRE(CHN, 2, 3, OQ, EAT, EAV)                                                         \
        else if (np == 3 && ny == 3) PB_PRE(CHN, 3, 3, OQ, EAT, EAV)                                                         \
        else if (pre && np == 3 && ny == 4) PB_PRE(CHN, 3, 4, OQ, EAT, EAV)                                                  \
        else if (pre && np == 3 && ny == 5) PB_PRE(CHN, 3, 5, OQ, EAT, EAV)                                                  \
        else if (pre && np == 4 && ny == 6) PB_PRE(CHN, 4, 6, OQ, EAT, EAV)                                                  \
        else if (np == 1) hipLaunchKernelGGL((k_pb_pairs<CHN, 1, 0, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV);            \
        else if (np == 2) hipLaunchKernelGGL((k_pb_pairs<CHN, 2, 0, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV);            \
        else if (np == 3) hipLaunchKernelGGL((k_pb_pairs<CHN, 3, 0, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV);            \
        else if (np == 4) hipLaunchKernelGGL((k_pb_pairs<CHN, 4, 0, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV);            \
        else hipLaunchKernelGGL((k_pb_pairs<CHN, 0, 0, 0, OQ, EAT>), g, block, lds, st, pa, F, EAV); }
      if (channels == 4 && opaque && (rc = pb_opaque_check())) return rc;
      if (epi) { if (opaque) { PB_PAIRS(4, 1, PbEpi, *epi) } else { PB_PAIRS(4, 0, PbEpi, *epi) } }
      else if (channels == 4 && opaque) { PB_PAIRS(4, 1, PbNoEpi, PbNoEpi{}) } else if (channels == 4) { PB_PAIRS(4, 0, PbNoEpi, PbNoEpi{}) } else { PB_PAIRS(3, 0, PbNoEpi, PbNoEpi{}) }
#undef PB_PAIRS
#undef PB_PRE
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  if (epi) return PB_NOT_FUSED;
  PbArgs a;
  a.src = src_d; a.dst = dst_d; a.irow = irow; a.orow = orow; a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
  a.x_step = x_step; a.y_step = y_step; a.xoff = t->xoff; a.yoff = t->yoff; a.n_x = t->n_x; a.n_y = t->n_y; a.table = t->table_d;
  a.rnd = (t->n_x == 2 && t->n_y == 2 && channels == 3) ? 0x8000u : 0xffffu;
  pb_used_taps(t, x_step, y_step, dw, dh, &a);
  a.win_w = (int)((63LL * x_step + 65535) >> 16) + t->n_x;
  a.tile_h = 0;
  for (int th = 16; th >= 1; th >>= 1) {
    const int wh = (int)(((long long)(th - 1) * y_step + 65535) >> 16) + t->n_y;
    if ((size_t)a.win_w * wh * 4 <= 48 * 1024) { a.tile_h = th; a.win_h = wh; break; }
  }
  const bool uniform = (x_step & 0xFFFF) == 0;
  if (a.tile_h) {
    const dim3 g(cdiv((unsigned)dw, 64), cdiv((unsigned)dh, (unsigned)a.tile_h), (unsigned)n);
    const size_t lds = (size_t)a.win_w * a.win_h * 4;
    if (channels == 4) {
      if (uniform) hipLaunchKernelGGL((k_pb_window<4, 1>), g, block, lds, st, a, F);
      else hipLaunchKernelGGL((k_pb_window<4, 0>), g, block, lds, st, a, F);
    } else {
      if (uniform) hipLaunchKernelGGL((k_pb_window<3, 1>), g, block, lds, st, a, F);
      else hipLaunchKernelGGL((k_pb_window<3, 0>), g, block, lds, st, a, F);
    }
  } else {
    if (channels == 4) hipLaunchKernelGGL(k_pb_direct<4>, grid, block, 0, st, a, F);
    else hipLaunchKernelGGL(k_pb_direct<3>, grid, block, 0, st, a, F);
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// what lgpu_pixbuf_scale would answer for this geometry, without launching anything (the weight table is built and cached on the way, so the scale that follows
// finds it): LGPU_OK, LGPU_E_BADARG, or LGPU_E_UNSUPPORTED for a reduction past the library's one-step range.  The layer seam asks before it records a scale
// for later (deferred execution, layer_seam.cpp): a recorded call must not turn into a refusal.
int lgpu::pb_scale_fused(const uint8_t *const *srcs, uint8_t *const *dsts, int n, int irow, int sw, int sh, int orow, int dw, int dh, int interp, hipStream_t st, const PbEpi *epi) {
  return pb_scale_n(srcs, dsts, n, irow, sw, sh, orow, dw, dh, 4, interp, (void *)st, epi);
}
extern "C" int lgpu_pixbuf_scale_check(int sw, int sh, int dw, int dh, int channels, int interp, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(sw > 0 && sh > 0 && dw > 0 && dh > 0, "empty geometry");
  LGPU_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (no alpha) or 4 (alpha)");
  LGPU_REQUIRE(interp == 0 || interp == 2 || interp == 3, "interp must be 0 (NEAREST), 2 (BILINEAR) or 3 (HYPER)");
  LGPU_REQUIRE(sw < 32768 && sh < 32768 && dw < 32768 && dh < 32768, "frame sides must stay below 32768 (16.16 positions)");
  if (dw == sw && dh == sh) return LGPU_OK;
  const double scale_x = (double)dw / sw, scale_y = (double)dh / sh;
  if ((int)(65536 / scale_x) == 0 || (int)(65536 / scale_y) == 0) { set_error("lgpu_pixbuf_scale: enlargement beyond 65536x"); return LGPU_E_UNSUPPORTED; }
  if (interp == 0) return LGPU_OK;
  PbPin pin;
  rc = pb_table(interp, sw, sh, dw, dh, (hipStream_t)stream, &pin);
  if (rc == LGPU_E_UNSUPPORTED) set_error("lgpu_pixbuf_scale: %dx%d -> %dx%d: the library's two-step scaler is not covered", sw, sh, dw, dh);
  return rc;
}
extern "C" int lgpu_pixbuf_scale(const uint8_t *src_d, int irow, int sw, int sh, uint8_t *dst_d, int orow, int dw, int dh, int channels, int interp,
                                 void *stream) {
  return pb_scale_n(&src_d, &dst_d, 1, irow, sw, sh, orow, dw, dh, channels, interp, stream);
}
// lgpu_pixbuf_scale for nframes frames of one geometry (the layers of the live tracks of one plan step: src/effects-weed.c:1850-2425 runs one instance per track per
// tick; compositor.c:262-266 scales every layer of a frame): ONE launch
extern "C" int lgpu_pixbuf_scale_batch(const uint8_t *const *src_d, uint8_t *const *dst_d, int nframes, int irow, int sw, int sh, int orow, int dw, int dh, int channels,
                                       int interp, void *stream) {
  return pb_scale_n(src_d, dst_d, nframes, irow, sw, sh, orow, dw, dh, channels, interp, stream);
}

// resize -> letterbox -> blend (-> gamma) as one call: BASELINE config 3's chain.  letterbox_layer (src/colourspace.c:15343-15567) centres the scaled frame on an
// opaque black canvas; here the canvas never exists on its own: the scaled frame is blended and stored at its place, the bars are blended black.
static int chain_canvas_impl(const lgpu_chain_params *params, const lgpu_canvas *canvas, const lgpu_chain_track *tracks, int ntracks, void *stream, const uint8_t *amounts, bool flat_ok = false) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(params && canvas && tracks && ntracks > 0 && ntracks <= LGPU_CHAIN_MAX_TRACKS, "1..64 tracks");
  LGPU_REQUIRE(params->sw > 0 && params->sh > 0 && params->dw > 0 && params->dh > 0, "empty geometry");
  LGPU_REQUIRE(canvas->nwidth >= params->dw && canvas->nheight >= params->dh && canvas->offs_x >= 0 && canvas->offs_y >= 0 &&
               canvas->offs_x + params->dw <= canvas->nwidth && canvas->offs_y + params->dh <= canvas->nheight, "the scaled frame must lie inside the canvas");
  LGPU_REQUIRE(params->irow >= params->sw * 4 && params->orow >= canvas->nwidth * 4 && params->irow2 >= canvas->nwidth * 4, "rowstride smaller than a row");
  LGPU_REQUIRE(((params->irow | params->orow | params->irow2) & 3) == 0, "rowstrides must be multiples of 4");
  LGPU_REQUIRE(flat_ok || !(params->sw == params->dw && params->sh == params->dh), "chain needs a resize stage");
  for (int i = 0; i < ntracks; i++) {
    LGPU_REQUIRE(tracks[i].src_d && tracks[i].layer2_d && tracks[i].dst_d, "null track pointer");
    LGPU_REQUIRE((((uintptr_t)tracks[i].src_d | (uintptr_t)tracks[i].layer2_d | (uintptr_t)tracks[i].dst_d) & 3) == 0, "frames must be 4-byte aligned");
    LGPU_REQUIRE(tracks[i].src_d != tracks[i].dst_d, "the chain cannot run in place");
  }
  return pb_chain(params, canvas, tracks, ntracks, (hipStream_t)stream, amounts);
}
extern "C" int lgpu_chain_canvas(const lgpu_chain_params *params, const lgpu_canvas *canvas, const lgpu_chain_track *tracks, int ntracks, void *stream) {
  return chain_canvas_impl(params, canvas, tracks, ntracks, stream, nullptr);
}
// lgpu_chain / lgpu_chain_canvas (canvas may be NULL) on the pixbuf arithmetic with a blend amount PER TRACK (amounts[ntracks], 0..255; params->bf and
// params->param_block_d are not used): the tracks of a tick share geometry and gamma table, not necessarily their transition amount -- one launch all the same
extern "C" int lgpu_chain_amounts(const lgpu_chain_params *params, const lgpu_canvas *canvas, const lgpu_chain_track *tracks, int ntracks, const uint8_t *amounts, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(params && (params->interp & LGPU_INTERP_PIXBUF), "lgpu_chain_amounts serves the gdk-pixbuf arithmetic (LGPU_INTERP_PIXBUF)");
  const bool noblend = (params->interp & LGPU_INTERP_NOBLEND) != 0;
  LGPU_REQUIRE(amounts || noblend, "null amounts");
  lgpu_chain_params p0 = *params;
  p0.param_block_d = nullptr;
  lgpu_chain_track tr[LGPU_CHAIN_MAX_TRACKS];
  if (noblend) {
    // no layer 2: the checks below see the destination in its place (same alignment and geometry rules), the kernels never read it
    LGPU_REQUIRE(tracks && ntracks > 0 && ntracks <= LGPU_CHAIN_MAX_TRACKS, "1..64 tracks");
    for (int i = 0; i < ntracks; i++) { tr[i] = tracks[i]; tr[i].layer2_d = tracks[i].dst_d; }
    tracks = tr; p0.irow2 = p0.orow; amounts = nullptr;
  }
  const lgpu_canvas whole = {p0.dw, p0.dh, 0, 0};
  if (!canvas && p0.sw == p0.dw && p0.sh == p0.dh) canvas = &whole;      // no resize stage: the canvas form's checks (lgpu_chain_check insists on a resize)
  if (canvas) return chain_canvas_impl(&p0, canvas, tracks, ntracks, stream, amounts, true);
  if ((rc = lgpu_chain_check(&p0, tracks, ntracks))) return rc;
  return pb_chain(&p0, nullptr, tracks, ntracks, (hipStream_t)stream, amounts);
}
