// stencil.hip -- the 3x3 stencil effects of the reference plugin set:
//   "softlight"     lives-plugins/weed-plugins/softlight.c:62-141   (planar YUV, luma plane only)
//   "edge detect"   lives-plugins/weed-plugins/edge.c:129-248       (packed RGB, global Otsu threshold)
// Both are HBM-bound byte work: one LDS tile with its halo per workgroup, lane = 4 consecutive pixels.
#include "lgpu_common.h"
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

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

// lo <= x <= hi as ONE v_med3_i32.  Written as two selects the compiler emits a compare, a v_cndmask_b32_e32 on vcc and a v_min per clamp, and that v_cndmask form
// issues at less than a quarter of the rate of the other vector operations on gfx950 (tools/valu_rates2.hip): the two clamps were a quarter of this kernel's time
__device__ __forceinline__ int sl_clamp(int x, int lo, int hi) {
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(lo), "v"(hi));
  return d;
}
// the planes that are only copied (softlight.c:143-151) ride in the same launch: blockIdx.z = 1 .. ncopy copies 64 x 16 tiles of plane z
// batched (lgpu_fx_batch): blockIdx.z = frame * (n + 1) + plane, the frame's planes from the table
struct SoftCopy { int irow[3], orow[3], w, h, n; };
__global__ __launch_bounds__(kBlock) void k_softlight(const FxFrames F, int irow, int orow, int width, int height,
                                                        int ymin, int ymax, SoftCopy cp) {
  const int frame = blockIdx.z / (cp.n + 1), plane = blockIdx.z - frame * (cp.n + 1);
  const uint8_t *src = F.in0[frame][0];
  uint8_t *dst = F.out[frame][0];
  __shared__ __attribute__((aligned(4))) uint8_t s[(kStH + 2) * (kStW + 8)];              // rows y0-1 .. y0+kStH, columns x0-4 .. x0+kStW+3
  constexpr int P = kStW + 8;
  const int x0 = blockIdx.x * kStW, y0 = blockIdx.y * kStH;
  if (plane) {
    const int pz = plane - 1, ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4, y = y0 + ly, x = x0 + lx;
    if (y >= cp.h || x >= cp.w) return;
    const uint8_t *ps = F.in0[frame][plane] + (size_t)y * cp.irow[pz] + x;
    uint8_t *pd = F.out[frame][plane] + (size_t)y * cp.orow[pz] + x;
    if (x + 4 <= cp.w && (((uintptr_t)ps | (uintptr_t)pd) & 3) == 0) *reinterpret_cast<uint32_t *>(pd) = *reinterpret_cast<const uint32_t *>(ps);
    else for (int j = 0; j < 4 && x + j < cp.w; j++) pd[j] = ps[j];
    return;
  }
  // window as dwords, both of a thread's requests issued before the first is stored (a rolled byte loop paid five dependent round trips per tile)
  {
    constexpr int DW = P / 4, ND = (kStH + 2) * DW, NI = (ND + kBlock - 1) / kBlock;        // 18 dwords per row, 324 per window, 2 per thread
    const bool al = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)irow) & 3) == 0;
    uint32_t v[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const int i = threadIdx.x + k * kBlock, r = i / DW, c = i - r * DW;
      int sy = y0 - 1 + r;
      const int sx = x0 - 4 + 4 * c;
      sy = sy < 0 ? 0 : sy >= height ? height - 1 : sy;        // clamped fetches only feed border outputs, which are copies
      v[k] = 0;
      if (i < ND) {
        const uint8_t *row = src + (size_t)sy * irow;
        if (al && sx >= 0 && sx + 4 <= width) v[k] = *reinterpret_cast<const uint32_t *>(row + sx);
        else
          for (int j = 0; j < 4; j++) { int xx = sx + j; xx = xx < 0 ? 0 : xx >= width ? width - 1 : xx; v[k] |= (uint32_t)row[xx] << (8 * j); }
      }
    }
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const int i = threadIdx.x + k * kBlock;
      if (i < ND) reinterpret_cast<uint32_t *>(s)[i] = v[k];
    }
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
      sum = sl_clamp(sum, ymin, ymax);
      sum = (64 * sum + 192 * v) >> 8;
      v = sl_clamp(sum, ymin, ymax);
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

// k_softlight_s -- the same filter on 4-aligned luma planes without LDS and without a barrier (round 4).  A lane owns FOUR columns (one dword of luma) and a wave a band
// of RB rows: the RB + 2 rows the band needs are requested up front, a row's left / right neighbours come from the adjacent lanes (DPP wave shifts + v_alignbyte; lanes
// 0 and 63 only feed their neighbours), the 3 x 3 sums run on pixel PAIRS in packed 16-bit lanes (|row0| <= 765, |row1| <= 1275), row0^2 + row1^2 is one
// v_dot2_i32_i16, the integer square root is the truncated float square root (exact for n < 2.3 M: checked exhaustively on the host, tests/test_host_cpu.py), and
// (64 s + 192 v) >> 8 == (s + 3 v) >> 2.  57 -> ~22 vector operations per pixel; one 1080p 4:2:0 frame 6.2 -> see profiles/r04/ops_roofline.md.
typedef short sl_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short sl_u2 __attribute__((ext_vector_type(2)));
template <int RB>
__global__ __launch_bounds__(kBlock) void k_softlight_s(const FxFrames F, int irow, int orow, int width, int height, int ymin, int ymax, SoftCopy cp) {
  const int frame = blockIdx.z / (cp.n + 1), plane = blockIdx.z - frame * (cp.n + 1);
  const uint8_t *src = F.in0[frame][0];
  uint8_t *dst = F.out[frame][0];
  if (plane) {          // the planes that are only copied: 256 x 16 tiles of plane `plane`, sixteen bytes per thread
    const int pz = plane - 1, ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 16, y = blockIdx.y * 16 + ly, x = blockIdx.x * 256 + lx;
    if (y >= cp.h || x >= cp.w) return;
    const uint8_t *ps = F.in0[frame][plane] + (size_t)y * cp.irow[pz] + x;
    uint8_t *pd = F.out[frame][plane] + (size_t)y * cp.orow[pz] + x;
    if (x + 16 <= cp.w && (((uintptr_t)ps | (uintptr_t)pd) & 15) == 0) *reinterpret_cast<uint4 *>(pd) = *reinterpret_cast<const uint4 *>(ps);
    else for (int j = 0; j < 16 && x + j < cp.w; j++) pd[j] = ps[j];
    return;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = width >> 2;                                       // dwords per row (width % 4 == 0 on this path)
  if ((int)blockIdx.x * 62 >= nq) return;                          // the grid is as wide as the widest plane wants it
  const int q = blockIdx.x * 62 - 1 + lane;                        // this lane's dword; lanes 0 and 63 are feeders
  const int qc = q < 0 ? 0 : q >= nq ? nq - 1 : q;
  const int y0 = (blockIdx.y * 4 + wave) * RB;
  if (y0 >= height) return;
  const int rows = min(RB, height - y0);
  const bool out_lane = lane >= 1 && lane <= 62 && q < nq;
  uint32_t in[RB + 2];
#pragma unroll
  for (int r = 0; r < RB + 2; r++) {
    int sy = y0 - 1 + r;
    sy = __builtin_amdgcn_readfirstlane(sy < 0 ? 0 : sy >= height ? height - 1 : sy);      // clamped rows only feed border rows, which are copies
    in[r] = reinterpret_cast<const uint32_t *>(src + (size_t)sy * irow)[qc];
  }
  // per source row: the row shifted right / left by one pixel, then everything as pixel pairs in 16-bit lanes: e = (pixel 0, pixel 2), o = (pixel 1, pixel 3)
  uint32_t Le[RB + 2], Lo[RB + 2], Ce[RB + 2], Co[RB + 2], Re[RB + 2], Ro[RB + 2];
#pragma unroll
  for (int r = 0; r < RB + 2; r++) {
    const uint32_t c = in[r];
    const uint32_t ql = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0x138, 0xF, 0xF, true), qr = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0x130, 0xF, 0xF, true);
    const uint32_t l = __builtin_amdgcn_alignbyte(c, ql, 3), rr = __builtin_amdgcn_alignbyte(qr, c, 1);      // byte k = pixel k - 1 / pixel k + 1
    Le[r] = l & 0x00FF00FFu; Lo[r] = (l >> 8) & 0x00FF00FFu;
    Ce[r] = c & 0x00FF00FFu; Co[r] = (c >> 8) & 0x00FF00FFu;
    Re[r] = rr & 0x00FF00FFu; Ro[r] = (rr >> 8) & 0x00FF00FFu;
  }
  // frame-border pixels keep their value: byte mask of this lane's border columns
  const uint32_t xmask = (q == 0 ? 0x000000FFu : 0u) | (q == nq - 1 ? 0xFF000000u : 0u);
  auto pk = [](uint32_t v) -> sl_s2 { return __builtin_bit_cast(sl_s2, v); };
  auto un = [](sl_s2 v) -> uint32_t { return __builtin_bit_cast(uint32_t, v); };
  const sl_u2 lo2 = {(unsigned short)ymin, (unsigned short)ymin}, hi2 = {(unsigned short)ymax, (unsigned short)ymax};
#pragma unroll
  for (int r = 0; r < RB; r++) {
    if (r >= rows) break;
    const int y = y0 + r;
    uint32_t out = in[r + 1];
    if (y > 0 && y < height - 1) {                                  // uniform
      uint32_t res2[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {                                 // h = 0: pixels 0 and 2, h = 1: pixels 1 and 3
        const sl_s2 AL = pk(h ? Lo[r] : Le[r]), AC = pk(h ? Co[r] : Ce[r]), AR = pk(h ? Ro[r] : Re[r]);
        const sl_s2 CL = pk(h ? Lo[r + 1] : Le[r + 1]), CC = pk(h ? Co[r + 1] : Ce[r + 1]), CR = pk(h ? Ro[r + 1] : Re[r + 1]);
        const sl_s2 BL = pk(h ? Lo[r + 2] : Le[r + 2]), BC = pk(h ? Co[r + 2] : Ce[r + 2]), BR = pk(h ? Ro[r + 2] : Re[r + 2]);
        // softlight.c:115-120 as written: row0 = (S[+1][-1] - S[-1][-1]) + 2 (S[+1][0] - S[-1][0]) + (S[+1][+1] - S[+1][-1]); row1's third term is a SUM
        const sl_s2 row0 = (BC - AC) * (short)2 + (BR - AL);
        const sl_s2 row1 = (CR - CL) * (short)2 + (AR - AL) + (BR + BL);
        const uint32_t r0 = un(row0), r1 = un(row1);
        const uint32_t xa = __builtin_amdgcn_perm(r1, r0, 0x05040100u), xb = __builtin_amdgcn_perm(r1, r0, 0x07060302u);      // (row0, row1) of the pair's first / second pixel
        // the two square roots one by one, everything after them on the pair in packed 16-bit lanes (sq <= 1660: 3 sq, (3 sq / 2) 3 and sum + 3 v all fit 16 bits)
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_sdot2(pk(xa), pk(xa), 0, false), n1 = (uint32_t)__builtin_amdgcn_sdot2(pk(xb), pk(xb), 0, false);
        const uint32_t sq0 = (uint32_t)__fsqrt_rn((float)n0), sq1 = (uint32_t)__fsqrt_rn((float)n1);
        sl_u2 t = __builtin_bit_cast(sl_u2, sq0 | (sq1 << 16));
        t = (t * (unsigned short)3) >> (unsigned short)1;             // 3 sq / 2
        t = (t * (unsigned short)3) >> (unsigned short)1;             // (.. * 384) >> 8
        t = __builtin_elementwise_min(__builtin_elementwise_max(t, lo2), hi2);
        t = (t + __builtin_bit_cast(sl_u2, CC) * (unsigned short)3) >> (unsigned short)2;      // (64 sum + 192 v) >> 8
        t = __builtin_elementwise_min(__builtin_elementwise_max(t, lo2), hi2);
        res2[h] = __builtin_bit_cast(uint32_t, t);
      }
      const uint32_t calc = res2[0] | (res2[1] << 8);                // pixels (0, 2) | pixels (1, 3) << 8
      out = (calc & ~xmask) | (out & xmask);
    }
    if (out_lane) reinterpret_cast<uint32_t *>(dst + (size_t)y * orow)[q] = out;
  }
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
  unsigned int ticket;        // k_edge_reduce_otsu: workgroups that have added their share of the slices (the last one runs the scan and resets it)
};

// luma (pass 0) or byte pass-1 of the pixel; gradient magnitude map + histogram
template <int PS>
__global__ __launch_bounds__(kBlock) void k_edge_map(const uint8_t *src, int irow, int width, int height, int order, int pass,
                                                       const int32_t *gluma, uint16_t *map, unsigned int *slices, int vec) {
  __shared__ uint8_t l[(kStH + 4) * (kStW + 4)];
  __shared__ int32_t s_luma[768];
  __shared__ unsigned int s_hist[1024];
  __shared__ unsigned int s_sum;
  constexpr int P = kStW + 4;
  if (pass == 0) for (int i = threadIdx.x; i < 768; i += kBlock) s_luma[i] = gluma[i];
  for (int i = threadIdx.x; i < 1024; i += kBlock) s_hist[i] = 0;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  // persistent over tiles: the histogram stays in LDS and is flushed once per workgroup (global atomics on 1017 hot addresses
  // were most of this kernel's time with one flush per tile)
  const int tiles_x = (width + kStW - 1) / kStW, ntiles = tiles_x * ((height + kStH - 1) / kStH);
  unsigned int lsum = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int x0 = (tile % tiles_x) * kStW, y0 = (tile / tiles_x) * kStH;
  __syncthreads();                                            // the previous tile's readers are done with `l`
  if (PS == 4 && vec) {
    // one dword per pixel, and all of a thread's window pixels requested before the first is used: the rolled loop below pays a global round trip
    // per pixel (five to six per tile and thread), which was most of this kernel's 37 us
    constexpr int NI = ((kStH + 4) * P + kBlock - 1) / kBlock;
    uint32_t px[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const int i = threadIdx.x + k * kBlock, r = i / P, c = i - r * P;
      const int sy = y0 - 2 + r, sx = x0 - 2 + c;
      px[k] = 0;
      if (i < (kStH + 4) * P && sy >= 0 && sy < height && sx >= 0 && sx < width) px[k] = *reinterpret_cast<const uint32_t *>(src + (size_t)sy * irow + (size_t)sx * 4);
    }
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const int i = threadIdx.x + k * kBlock, r = i / P, c = i - r * P;
      const int sy = y0 - 2 + r, sx = x0 - 2 + c;
      uint8_t v = 0;
      if (sy >= 0 && sy < height && sx >= 0 && sx < width) {
        if (pass == 0) {        // calc_luma(), libweed/weed-plugin-utils.c:924-934
          const int c0 = (px[k] >> (order == 2 ? 8 : 0)) & 0xFF, c1 = (px[k] >> (order == 2 ? 16 : 8)) & 0xFF, c2 = (px[k] >> (order == 2 ? 24 : 16)) & 0xFF;
          const int32_t t = order == 1 ? (s_luma[c2] + s_luma[256 + c1] + s_luma[512 + c0]) : (s_luma[c0] + s_luma[256 + c1] + s_luma[512 + c2]);
          v = (uint8_t)(t >> 16);
        } else v = (uint8_t)(px[k] >> (8 * (pass - 1)));
      }
      if (i < (kStH + 4) * P) l[i] = v;
    }
  } else
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
  }   // tiles
  if (lsum) atomicAdd(&s_sum, lsum);
  __syncthreads();
  // the workgroup's histogram goes to its own slice, plain stores: k_edge_hist_reduce adds the slices up.  (Every workgroup adding its 1017 bins
  // to the same 1017 global counters cost more than the map itself: 54 us per frame with 1024 workgroups, 79 with 2048, 47 with 512.)
  unsigned int *sl = slices + (size_t)blockIdx.x * 1025;
  for (int i = threadIdx.x; i < 1024; i += kBlock) sl[i] = s_hist[i];
  if (threadIdx.x == 0) sl[1024] = s_sum;
}
// slices -> EdgeState: workgroup = (256 bins, one 64th of the slices: at most 16 per thread, all requested before they are added -- with 16 chunks and a rolled
// loop this kernel took 25.6 us, every load a round trip of its own)
constexpr int kEdgeChunks = 64;
__global__ __launch_bounds__(256) void k_edge_hist_reduce(const unsigned int *slices, int nslices, EdgeState *st) {
  const int bin = (blockIdx.x & 3) * 256 + threadIdx.x, chunk = blockIdx.x >> 2;
  unsigned int v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { const int sidx = chunk + k * kEdgeChunks; v[k] = sidx < nslices ? slices[(size_t)sidx * 1025 + bin] : 0u; }
  unsigned int acc = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += v[k];
  if (acc) atomicAdd(&st->hist[bin], acc);
  if ((blockIdx.x & 3) == 0 && threadIdx.x < 16) {
    const int sidx = chunk + (int)threadIdx.x * kEdgeChunks;
    const unsigned int t = sidx < nslices ? slices[(size_t)sidx * 1025 + 1024] : 0u;
    if (t) atomicAdd(&st->sum, (unsigned long long)t);
  }
}

// The Otsu scan of edge.c:184-205 with 1024 threads.  The reference walks t = 0..1016 serially keeping running sums and the
// first record high of dif; here the running sums are exclusive-inclusive prefix sums of the histogram (exact in uint64),
// every t evaluates its dif with the same IEEE double operations in the same order, and the serial "if (dif > difmax)" chain
// is the smallest t that attains the maximum -- provided that maximum beats the difmax carried in from the previous pass.
__global__ __launch_bounds__(1024) void k_edge_otsu(EdgeState *st, unsigned long long count, int first) {
  __shared__ unsigned long long s_n[1024], s_w[1024];
  __shared__ double s_d[1024];
  __shared__ unsigned int s_t[1024];
  const unsigned int t = threadIdx.x;
  const unsigned long long pr = t < 1017 ? st->hist[t] : 0ull;
  s_n[t] = pr; s_w[t] = pr * t;
  __syncthreads();
  for (unsigned int off = 1; off < 1024; off <<= 1) {          // inclusive scan (Hillis-Steele): sum over 0..t
    const unsigned long long n = t >= off ? s_n[t - off] : 0ull, w = t >= off ? s_w[t - off] : 0ull;
    __syncthreads();
    s_n[t] += n; s_w[t] += w;
    __syncthreads();
  }
  // first pass of a frame: the carried accumulators start from zero (edge.c:146-149 are function-scope locals); this replaces a reset launch per frame --
  // the histogram and the sum are left zeroed by the previous scan (below) and by the allocation
  const unsigned long long bh0 = (first ? 0ull : st->bh) + st->sum, nbh0 = (first ? 0ull : st->nbh) + count, bl0 = first ? 0ull : st->bl, nbl0 = first ? 0ull : st->nbl;
  double dif = -1.;
  bool valid = false;
  if (t >= 1 && t < 1017) {
    const unsigned long long bl = bl0 + s_w[t], nbl = nbl0 + s_n[t], bh = bh0 - s_w[t], nbh = nbh0 - s_n[t];
    const double abh = __ddiv_rn((double)bh, (double)nbh);
    const double abl = __ddiv_rn((double)bl, (double)nbl);
    const double d = __dsub_rn(abh, abl);
    dif = __dmul_rn(__dmul_rn((double)(nbl * nbh), d), d);
    valid = dif == dif;                                         // a NaN never wins a `>` comparison
  }
  s_d[t] = valid ? dif : -1.;                                   // dif >= 0 whenever it is a number
  s_t[t] = t;
  __syncthreads();
  for (unsigned int off = 512; off > 0; off >>= 1) {            // max, ties to the smaller t
    if (t < off) {
      const double o = s_d[t + off];
      if (o > s_d[t] || (o == s_d[t] && s_t[t + off] < s_t[t])) { s_d[t] = o; s_t[t] = s_t[t + off]; }
    }
    __syncthreads();
  }
  if (t == 0) {
    double difmax = first ? 0. : st->difmax;
    unsigned int threshmax = first ? 0u : st->threshmax;
    if (s_d[0] > difmax) { difmax = s_d[0]; threshmax = s_t[0]; }
    st->bh = bh0 - s_w[1016]; st->nbh = nbh0 - s_n[1016]; st->bl = bl0 + s_w[1016]; st->nbl = nbl0 + s_n[1016];
    st->difmax = difmax; st->threshmax = threshmax; st->thresh = threshmax;
    st->sum = 0;
  }
  __syncthreads();
  st->hist[t] = 0;
}

// copywalpha (edge.c:93-125) on every pixel: codes per colour byte 0 = black, 1 = source, 2 = white, 3 = leave
template <int PS>
__global__ __launch_bounds__(kBlock) void k_edge_paint(const uint8_t *src, int irow, uint8_t *dst, int orow, int width, int height,
                                                         int pass, int mode, int aoffs, int inplace, const uint16_t *map,
                                                         const EdgeState *st, int vec) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= width) return;
  const unsigned int thresh = st->thresh;
  for (int y = blockIdx.y; y < height; y += gridDim.y) {
    const uint8_t *s = src + (size_t)y * irow + (size_t)x * PS;
    uint8_t *d = dst + (size_t)y * orow + (size_t)x * PS;
    const bool edge = map[(size_t)y * width + x] >= thresh;
    if (PS == 4 && vec) {
      // the same byte rules on one dword: colour bytes = source / black / white (pass 0) or one channel forced to 255 (passes 1..3, edges only);
      // the alpha byte is the source's (copied when out of place, already there when in place)
      const uint32_t sp = *reinterpret_cast<const uint32_t *>(s);
      const uint32_t cmask = aoffs == 1 ? 0xFFFFFF00u : 0x00FFFFFFu;
      if (pass == 0) {
        const uint32_t colour = edge ? (mode == 1 ? cmask : (sp & cmask)) : 0u;
        *reinterpret_cast<uint32_t *>(d) = colour | (sp & ~cmask);
      } else if (edge) {
        const uint32_t old_ = *reinterpret_cast<const uint32_t *>(d);
        const uint32_t forced = 0xFFu << (8 * ((aoffs == 1 ? 1 : 0) + pass - 1));
        *reinterpret_cast<uint32_t *>(d) = ((old_ | forced) & cmask) | ((inplace ? old_ : sp) & ~cmask);
      }
      continue;
    }
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

// ---- 4-byte pixels on 16-byte aligned rows, width % 4 == 0 (round 4; profiles/r04/pmc_before/pmc_edge.md: map 14.8 + reduce 5.0 + scan 6.3 + paint 8.3 us per 1080p pass) ----
// k_edge_map4: tiles of 128 x 32 pixels; the window (one pixel of halo, rows of 34 quads) is staged as LUMA BYTES, a 16-byte load and one LDS dword per quad;
// a thread owns 4 columns x 4 rows and reads three dwords per window row: its pixels' 3-byte neighbourhoods are byte alignments of those (v_alignbyte), their
// sums one v_dot4_u32_u8 and the horizontal difference one SDWA subtract -- 32 byte reads per pixel quad became 3 dword reads.  Same arithmetic as k_edge_map:
// integer gradients, the float square root and the 0.94 product in the reference's order.
constexpr int kEmW = 128, kEmQ = kEmW / 4 + 2;       // 34 quads = 136 luma bytes per window row: columns x0 - 4 .. x0 + 131
// EH: rows per tile (8 thread rows x EH / 8 rows each).  One 1080p frame is a single generation of workgroups whichever EH: the launch lasts as long as ONE
// workgroup's chain of latencies (window loads, tables, luma, gradients, histogram flush), so the window loads go out before anything else, the tables are
// staged while they fly, and nothing on the way is a per-lane branch: window quads outside the frame are loaded from a clamped address (their luma is never a
// neighbour of an interior pixel), pixels outside the interior compute like the others and add 0 to the histogram (first form: 47 exec-mask branches per tile,
// each behind its own s_waitcnt).  Frame offsets are 32-bit (the host checks height * rowstride < 2^31).
template <int EH>
__global__ __launch_bounds__(kBlock) void k_edge_map4(const uint8_t *src, int irow, int width, int height, int order, int pass,
                                                        const int32_t *gluma, uint16_t *map, int mpitch, unsigned int *slices, unsigned int *ticket) {
  // the first map launch of a call clears the reduce kernel's ticket (ordered before that launch by the stream): a launch that died half way in an EARLIER call
  // cannot leave the persistent state with a count that picks the wrong "last" workgroup for ever after
  if (pass == 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicExch(ticket, 0u);
  constexpr int RT = EH / 8;                                 // rows per thread
  constexpr int NQ = (EH + 2) * kEmQ, NI = (NQ + kBlock - 1) / kBlock;       // quads of a window, per thread: all requested before the first is used
  __shared__ __attribute__((aligned(16))) uint32_t l[NI * kBlock];           // (EH + 2) rows of kEmQ dwords, padded to whole rounds of the workgroup
  __shared__ int32_t s_luma[768];
  __shared__ unsigned int s_hist[1024];
  __shared__ unsigned int s_sum;
  const int tiles_x = (width + kEmW - 1) / kEmW, ntiles = tiles_x * ((height + EH - 1) / EH);
  const int cgx = threadIdx.x & 31, rg = threadIdx.x >> 5;
  // quad i = tid + 256 k of the window: row i / 34, column i % 34 (i * 1928 >> 16 == i / 34 below 1,400: 24-bit multiplies only)
  uint32_t wr[NI], wc[NI];
#pragma unroll
  for (int k = 0; k < NI; k++) {
    const uint32_t i = threadIdx.x + k * kBlock;
    wr[k] = __umul24(i, 1928u) >> 16; wc[k] = i - __umul24(wr[k], (uint32_t)kEmQ);
  }
  auto window = [&](int tile, uint4 q[NI]) {
    const int ty = tile / tiles_x, x0 = (tile - ty * tiles_x) * kEmW, y0 = ty * EH;
#pragma unroll
    for (int k = 0; k < NI; k++) {
      int sy = y0 - 1 + (int)wr[k], sx = x0 - 4 + 4 * (int)wc[k];
      sy = sy < 0 ? 0 : sy > height - 1 ? height - 1 : sy;
      sx = sx < 0 ? 0 : sx > width - 4 ? width - 4 : sx;
      q[k] = *reinterpret_cast<const uint4 *>(src + __umul24((uint32_t)sy, (uint32_t)irow) + 4u * (uint32_t)sx);
    }
  };
  uint4 q[NI];
  window((int)blockIdx.x < ntiles ? (int)blockIdx.x : 0, q);
  {
    int32_t t0 = 0, t1 = 0, t2 = 0;
    if (pass == 0) { t0 = gluma[threadIdx.x]; t1 = gluma[256 + threadIdx.x]; t2 = gluma[512 + threadIdx.x]; }
#pragma unroll
    for (int i = 0; i < 4; i++) s_hist[threadIdx.x + 256 * i] = 0;
    if (threadIdx.x == 0) s_sum = 0;
    s_luma[threadIdx.x] = t0; s_luma[256 + threadIdx.x] = t1; s_luma[512 + threadIdx.x] = t2;
  }
  unsigned int lsum = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty = tile / tiles_x, x0 = (tile - ty * tiles_x) * kEmW, y0 = ty * EH;
    __syncthreads();                                          // tables staged / the previous tile's readers are done with `l`
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint32_t px[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t b;
        if (pass == 0) {        // calc_luma(), libweed/weed-plugin-utils.c:924-934
          const int c0 = (px[j] >> (order == 2 ? 8 : 0)) & 0xFF, c1 = (px[j] >> (order == 2 ? 16 : 8)) & 0xFF, c2 = (px[j] >> (order == 2 ? 24 : 16)) & 0xFF;
          const int32_t t = order == 1 ? (s_luma[c2] + s_luma[256 + c1] + s_luma[512 + c0]) : (s_luma[c0] + s_luma[256 + c1] + s_luma[512 + c2]);
          b = (uint32_t)(t >> 16) & 0xFFu;
        } else b = (px[j] >> (8 * (pass - 1))) & 0xFFu;
        v |= b << (8 * j);
      }
      l[threadIdx.x + k * kBlock] = v;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) window(tile + gridDim.x, q);      // the next tile's window, in flight during this tile's gradients
    // window row r = frame row y0 - 1 + r; this thread's pixels are window bytes 4 cgx + 4 .. + 7
    int hs[RT + 2][4], hd[RT + 2][4];
#pragma unroll
    for (int r = 0; r < RT + 2; r++) {
      const uint32_t *lw = l + (RT * rg + r) * kEmQ + cgx;
      const uint32_t d0 = lw[0], d1 = lw[1], d2 = lw[2];
      const uint32_t w[4] = {__builtin_amdgcn_alignbyte(d1, d0, 3), d1, __builtin_amdgcn_alignbyte(d2, d1, 1), __builtin_amdgcn_alignbyte(d2, d1, 2)};      // bytes x - 1, x, x + 1 (, x + 2)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        hs[r][j] = (int)__builtin_amdgcn_udot4(w[j], 0x00010101u, 0u, false);
        int dd;
        asm("v_sub_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_0" : "=v"(dd) : "v"(w[j]));
        hd[r][j] = dd;
      }
    }
    const int xb = x0 + 4 * cgx;
    bool xin[4];
#pragma unroll
    for (int j = 0; j < 4; j++) xin[j] = xb + j >= 2 && xb + j < width - 2;
#pragma unroll
    for (int ry = 0; ry < RT; ry++) {
      const int y = y0 + RT * rg + ry;
      const bool yin = y >= 2 && y < height - 2;
      uint32_t vals[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int v0 = hd[ry][j] + hd[ry + 1][j] + hd[ry + 2][j];       // sum over rows y-1..y+1 of (l[x+1] - l[x-1])
        const int v1 = hs[ry + 2][j] - hs[ry][j];                        // sum over columns x-1..x+1 of (l[y+1] - l[y-1])
        const float m = __fmul_rn(__fsqrt_rn(__fadd_rn((float)(v0 * v0), (float)(v1 * v1))), 0.94f);
        const bool in = yin && xin[j];
        const unsigned int val = in ? (unsigned int)m & 0xFFFFu : 0u;
        atomicAdd(&s_hist[val & 1023], in ? 1u : 0u);
        lsum += val;
        vals[j] = val;
      }
      if (y < height && xb < width)
        *reinterpret_cast<uint2 *>(map + __umul24((uint32_t)y, (uint32_t)mpitch) + (uint32_t)xb) = make_uint2(vals[0] | (vals[1] << 16), vals[2] | (vals[3] << 16));
    }
  }
  atomicAdd(&s_sum, lsum);
  __syncthreads();
  unsigned int *sl = slices + (size_t)blockIdx.x * 1025;
#pragma unroll
  for (int i = 0; i < 4; i++) sl[threadIdx.x + 256 * i] = s_hist[threadIdx.x + 256 * i];
  if (threadIdx.x == 0) sl[1024] = s_sum;
}

// k_edge_hist_reduce and k_edge_otsu as ONE launch: every workgroup adds its share of the slices to the state's histogram, the LAST one to finish (a ticket
// counter behind a device-scope fence) runs the scan with its 256 threads, four bins each: local sums, a Hillis-Steele scan over the 256 thread totals, every
// bin's dif in the same IEEE double operations and order as k_edge_otsu, the maximum with ties to the smaller bin.
__global__ __launch_bounds__(256) void k_edge_reduce_otsu(const unsigned int *slices, int nslices, EdgeState *st, unsigned long long count, int first) {
  {
    const int bin = (blockIdx.x & 3) * 256 + threadIdx.x, chunk = blockIdx.x >> 2;
    unsigned int v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const int sidx = chunk + k * kEdgeChunks; v[k] = sidx < nslices ? slices[(size_t)sidx * 1025 + bin] : 0u; }
    unsigned int acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) acc += v[k];
    // Every read-modify-write below is a device-scope atomic that RETURNS its old value: the value comes back from the place the operation was performed, so once
    // a thread holds it the addition is done as far as any later device-scope atomic is concerned.  That is all the ticket needs -- no device-scope fence, which on
    // this part writes the whole L2 back (the map kernel has just left 4 MB of dirty lines there: 23 us per launch with __threadfence() here, measured).
    // HARDWARE ASSUMPTION (gfx950, the only target build.sh compiles for): agent-scope returning RMWs are performed at the memory side, in issue order per
    // wave, so "old value returned" implies "performed" for every later agent-scope atomic.  Formally (C++ memory model) these relaxed atomics do not
    // synchronise; on another part or compiler, use __hip_atomic_fetch_add(&st->ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) and pay the fence.
    unsigned int r0 = 0;
    unsigned long long r1 = 0;
    if (acc) r0 = atomicAdd(&st->hist[bin], acc);
    if ((blockIdx.x & 3) == 0 && threadIdx.x < 16) {
      const int sidx = chunk + (int)threadIdx.x * kEdgeChunks;
      const unsigned int t = sidx < nslices ? slices[(size_t)sidx * 1025 + 1024] : 0u;
      if (t) r1 = atomicAdd(&st->sum, (unsigned long long)t);
    }
    asm volatile("" :: "v"(r0), "v"(r1));                     // the returns are waited for (s_waitcnt) before the barrier below
  }
  __shared__ unsigned int s_last;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last workgroup reads the histogram and the sum with device-scope atomic loads (the other workgroups' additions were performed where those look); the
  // carried accumulators were written by an earlier launch
  __shared__ unsigned long long s_n[256], s_w[256], s_tot[2];
  __shared__ double s_d[256];
  __shared__ unsigned int s_t[256];
  const unsigned int t = threadIdx.x;
  unsigned long long n[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned int bin = 4 * t + i;
    const unsigned long long pr = bin < 1017 ? (unsigned long long)__hip_atomic_load(&st->hist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    n[i] = pr + (i ? n[i - 1] : 0ull); w[i] = pr * bin + (i ? w[i - 1] : 0ull);      // inclusive within the thread
  }
  s_n[t] = n[3]; s_w[t] = w[3];
  __syncthreads();
  for (unsigned int off = 1; off < 256; off <<= 1) {
    const unsigned long long a = t >= off ? s_n[t - off] : 0ull, b = t >= off ? s_w[t - off] : 0ull;
    __syncthreads();
    s_n[t] += a; s_w[t] += b;
    __syncthreads();
  }
  const unsigned long long en = t ? s_n[t - 1] : 0ull, ew = t ? s_w[t - 1] : 0ull;      // the threads before this one
#pragma unroll
  for (int i = 0; i < 4; i++) { n[i] += en; w[i] += ew; }
  if (t == 254) { s_tot[0] = n[0]; s_tot[1] = w[0]; }                                   // bin 1016 = 4 * 254: the sums over 0 .. 1016
  const unsigned long long sum = __hip_atomic_load(&st->sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long bh0 = (first ? 0ull : st->bh) + sum, nbh0 = (first ? 0ull : st->nbh) + count, bl0 = first ? 0ull : st->bl, nbl0 = first ? 0ull : st->nbl;
  double best = -1.;
  unsigned int best_t = 4 * t;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned int bin = 4 * t + i;
    if (bin >= 1 && bin < 1017) {
      const unsigned long long bl = bl0 + w[i], nbl = nbl0 + n[i], bh = bh0 - w[i], nbh = nbh0 - n[i];
      const double abh = __ddiv_rn((double)bh, (double)nbh);
      const double abl = __ddiv_rn((double)bl, (double)nbl);
      const double d = __dsub_rn(abh, abl);
      const double dif = __dmul_rn(__dmul_rn((double)(nbl * nbh), d), d);
      if (dif == dif && dif > best) { best = dif; best_t = bin; }                        // a NaN never wins; ties stay with the smaller bin
    }
  }
  s_d[t] = best; s_t[t] = best_t;
  __syncthreads();
  for (unsigned int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      const double o = s_d[t + off];
      if (o > s_d[t] || (o == s_d[t] && s_t[t + off] < s_t[t])) { s_d[t] = o; s_t[t] = s_t[t + off]; }
    }
    __syncthreads();
  }
  if (t == 0) {
    double difmax = first ? 0. : st->difmax;
    unsigned int threshmax = first ? 0u : st->threshmax;
    if (s_d[0] > difmax) { difmax = s_d[0]; threshmax = s_t[0]; }
    st->bh = bh0 - s_tot[1]; st->nbh = nbh0 - s_tot[0]; st->bl = bl0 + s_tot[1]; st->nbl = nbl0 + s_tot[0];
    st->difmax = difmax; st->threshmax = threshmax; st->thresh = threshmax;
    st->sum = 0;
    st->ticket = 0;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) st->hist[4 * t + i] = 0;
}

// k_edge_paint on quads: 16 bytes of source, 8 of the map, (passes 1..3: 16 of the destination,) one 16-byte store per thread
__global__ __launch_bounds__(kBlock) void k_edge_paint4(const uint8_t *src, int irow, uint8_t *dst, int orow, int wq, int height, int pass, int mode, int aoffs, int inplace,
                                                          const uint16_t *map, int mpitch, const EdgeState *st, uint32_t qmagic) {
  const unsigned int thresh = st->thresh;
  const uint32_t cmask = aoffs == 1 ? 0xFFFFFF00u : 0x00FFFFFFu;
  const uint32_t forced = pass ? 0xFFu << (8 * ((aoffs == 1 ? 1 : 0) + pass - 1)) : 0u;
  const uint32_t nq = (uint32_t)wq * (uint32_t)height;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < nq; i += gridDim.x * kBlock) {
    uint32_t y = __umulhi(i, qmagic);
    uint32_t xq = i - y * (uint32_t)wq;
    if (xq >= (uint32_t)wq) { xq -= wq; y++; }
    const uint4 sp = *reinterpret_cast<const uint4 *>(src + (size_t)y * irow + 16 * (size_t)xq);
    const uint2 mv = *reinterpret_cast<const uint2 *>(map + (size_t)y * mpitch + 4 * (size_t)xq);
    uint4 *dp = reinterpret_cast<uint4 *>(dst + (size_t)y * orow + 16 * (size_t)xq);
    const bool e[4] = {(mv.x & 0xFFFFu) >= thresh, (mv.x >> 16) >= thresh, (mv.y & 0xFFFFu) >= thresh, (mv.y >> 16) >= thresh};
    const uint32_t s4[4] = {sp.x, sp.y, sp.z, sp.w};
    uint32_t o[4];
    if (pass == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = (e[j] ? (mode == 1 ? cmask : (s4[j] & cmask)) : 0u) | (s4[j] & ~cmask);
    } else {
      if (!(e[0] | e[1] | e[2] | e[3])) continue;
      const uint4 od = *dp;
      const uint32_t d4[4] = {od.x, od.y, od.z, od.w};
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = e[j] ? (((d4[j] | forced) & cmask) | ((inplace ? d4[j] : s4[j]) & ~cmask)) : d4[j];
    }
    *dp = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// blurzoom (RadioacTV): lives-plugins/weed-plugins/blurzoom.c:74-101, :106-145, :149-192, :201-237, :345-421
// ---------------------------------------------------------------------------------------------------------------------
// background subtract + threshold (:74-101), OR of the motion mask into the feedback plane (:371-379)
__global__ __launch_bounds__(kBlock) void k_bz_update(const uint32_t *src, int irow, int16_t *bg, uint8_t *buf, int vw, int vh, int bw, int ml,
                                                        int threshold, int do_or) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= vw) return;
  for (int y = blockIdx.y; y < vh; y += gridDim.y) {
    const uint32_t p = src[(size_t)y * irow + x];
    const int sum = (int)((p & 0xff0000) >> 15) + (int)((p & 0xff00) >> 6) + (int)(p & 0xff);
    const int v = sum - (int)bg[(size_t)y * vw + x];
    bg[(size_t)y * vw + x] = (int16_t)sum;
    const uint32_t d = (uint32_t)(((v + threshold) >> 24) | ((threshold - v) >> 24)) & 0xFFu;   // 0xFF where |v| > threshold
    if (do_or && x >= ml && x < ml + bw) buf[(size_t)y * bw + (x - ml)] |= (uint8_t)(d >> 3);
  }
}
// 4-neighbour average minus one, -1 wraps to 0 (:149-168); the border of the result plane is never written
__global__ __launch_bounds__(kBlock) void k_bz_blur(uint8_t *buf, int bw, int bh) {
  const int x = blockIdx.x * kBlock + threadIdx.x + 1;
  if (x >= bw - 1) return;
  const size_t area = (size_t)bw * bh;
  for (int y = blockIdx.y + 1; y < bh - 1; y += gridDim.y) {
    const uint8_t *p = buf + (size_t)y * bw + x;
    uint8_t v = (uint8_t)((p[-bw] + p[-1] + p[1] + p[bw]) / 4 - 1);
    if (v == 255) v = 0;
    buf[area + (size_t)y * bw + x] = v;
  }
}
// the serial pointer walk of zoom() (:171-192) as a gather: row starts and inclusive column step counts are prefix sums
__global__ __launch_bounds__(kBlock) void k_bz_zoom(uint8_t *buf, const int32_t *rowstart, const int32_t *colcum, int bw, int bh) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= bw) return;
  const size_t area = (size_t)bw * bh;
  const int cc = colcum[x];
  for (int y = blockIdx.y; y < bh; y += gridDim.y) buf[(size_t)y * bw + x] = buf[area + (size_t)(rowstart[y] + cc)];
}
// palette add with per-byte saturation (:398-414)
__global__ __launch_bounds__(kBlock) void k_bz_color(const uint32_t *src, int irow, uint32_t *dst, int orow, const uint8_t *buf, const uint32_t *pal,
                                                       int vw, int vh, int bw, int ml, int pattern) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= vw) return;
  for (int y = blockIdx.y; y < vh; y += gridDim.y) {
    const uint32_t s = src[(size_t)y * irow + x];
    uint32_t o = s;
    if (x >= ml && x < ml + bw) {
      uint32_t a = s & 0xfefeffu, b = pal[32 * pattern + buf[(size_t)y * bw + (x - ml)]];
      a += b;
      b = a & 0x10101u;
      o = (s & 0xff000000u) | ((a | (b - (b >> 8))) & 0xffffffu);
    }
    dst[(size_t)y * orow + x] = o;
  }
}

// per (device, stream) scratch: gradient map + state
struct EdgeScratch { uint16_t *map = nullptr; size_t cap = 0; EdgeState *st = nullptr; unsigned int *slices = nullptr; };
static std::mutex g_edge_mu;
// held across the launches of one multi-launch sequence (edge passes, in-place deinterlace): host threads that share a stream
// must not interleave their sequences on the shared scratch
static std::mutex g_seq_mu;
static std::map<std::pair<int, void *>, EdgeScratch> g_edge;


static std::mutex g_deint_mu;
static std::map<std::pair<int, void *>, std::pair<uint8_t *, size_t>> g_deint;   // per (device, stream) snapshot for in-place calls

// deinterlace (deinterlace.c:45-308): one thread per pixel triple of one odd row r < height - 2; writes rows r - 1 and r
struct DeintArgs {
  const uint8_t *src;
  uint8_t *dst;
  int irow, orow, ntrip, height;
  int psize, pcpy, green, packed422, copy_alpha;
  int dword, inplace;       // 4-byte pixels (not packed 4:2:2) in 4-byte aligned rows: the triple as three dwords per row
};
__global__ __launch_bounds__(kBlock) void k_deinterlace(DeintArgs a) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.ntrip) return;
  const int x = t * 3 * a.psize, xc = x + 2 * a.psize;
  const int npairs = (a.height - 2) >> 1;                                   // odd rows 1, 3, .. < height - 2
  for (int pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
    const int r = 2 * pr + 1;
    const uint8_t *r0 = a.src + (size_t)(r - 1) * a.irow, *r1 = r0 + a.irow, *r2 = r1 + a.irow, *r3 = r2 + a.irow;
    if (a.dword) {
      // the same byte rules on dwords: colour bytes 0..2 of the three pixels come from row r (top, written to row r - 1) and from row r + 1 or the byte-wise
      // mean of rows r and r + 2 (bottom, written to row r); byte 3 of the top pixels is not written (in place: what the snapshot holds; out of place: what the
      // destination holds), byte 3 of the bottom pixels is row r's when out of place, untouched in place
      const uint32_t *q0 = reinterpret_cast<const uint32_t *>(r0) + 3 * t, *q1 = reinterpret_cast<const uint32_t *>(r1) + 3 * t;
      const uint32_t *q2 = reinterpret_cast<const uint32_t *>(r2) + 3 * t, *q3 = reinterpret_cast<const uint32_t *>(r3) + 3 * t;
      uint32_t *o0 = reinterpret_cast<uint32_t *>(a.dst + (size_t)(r - 1) * a.orow) + 3 * t, *o1 = reinterpret_cast<uint32_t *>(a.dst + (size_t)r * a.orow) + 3 * t;
      uint32_t p0[3], p1[3], p2[3], p3[3], keep[3];
#pragma unroll
      for (int p = 0; p < 3; p++) { p0[p] = q0[p]; p1[p] = q1[p]; p2[p] = q2[p]; p3[p] = q3[p]; }
#pragma unroll
      for (int p = 0; p < 3; p++) keep[p] = a.inplace ? p0[p] : o0[p];
      const int sh = 8 * a.green;
      const int m1 = (int)(((p0[0] >> sh) & 0xFF) + ((p0[2] >> sh) & 0xFF)) >> 1, m2 = (int)(((p2[0] >> sh) & 0xFF) + ((p2[2] >> sh) & 0xFF)) >> 1;
      const int m3 = (int)(((p1[0] >> sh) & 0xFF) + ((p1[2] >> sh) & 0xFF)) >> 1, m4 = (int)(((p3[0] >> sh) & 0xFF) + ((p3[2] >> sh) & 0xFF)) >> 1;
      const bool mixd = abs(m1 - m2) + abs(m3 - m4) < abs(m1 - m4) + abs(m3 - m2);
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const uint32_t avg = (p1[p] & p3[p]) + (((p1[p] ^ p3[p]) & 0xFEFEFEFEu) >> 1);          // (x + y) >> 1 per byte
        const uint32_t bot = mixd ? avg : p2[p];
        o0[p] = (p1[p] & 0x00FFFFFFu) | (keep[p] & 0xFF000000u);
        o1[p] = (bot & 0x00FFFFFFu) | (p1[p] & 0xFF000000u);          // out of place: row r's byte 3 (copy_alpha); in place: the byte that is there already (the same one)
      }
      continue;
    }
    int m1, m2, m3, m4;
    if (a.packed422) {
      const int yo = a.packed422 == 1 ? 1 : 0;
      m1 = (r0[x + yo] + r0[x + yo + 2] + r0[xc + yo] + r0[xc + yo + 2]) >> 2;
      m2 = (r2[x + yo] + r2[x + yo + 2] + r2[xc + yo] + r2[xc + yo + 2]) >> 2;
      m3 = (r1[x + yo] + r1[x + yo + 2] + r1[xc + yo] + r1[xc + yo + 2]) >> 2;
      m4 = (r3[x + yo] + r3[x + yo + 2] + r3[xc + yo] + r3[xc + yo + 2]) >> 2;
    } else {
      m1 = (r0[x + a.green] + r0[xc + a.green]) >> 1; m2 = (r2[x + a.green] + r2[xc + a.green]) >> 1;
      m3 = (r1[x + a.green] + r1[xc + a.green]) >> 1; m4 = (r3[x + a.green] + r3[xc + a.green]) >> 1;
    }
    const bool mix = abs(m1 - m2) + abs(m3 - m4) < abs(m1 - m4) + abs(m3 - m2);
    // in place the triple's own bytes are read before they are written: gather everything first
    uint8_t top[12], bot[12], al[3] = {0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < a.pcpy) {
          const int i = x + p * a.psize + k;
          top[p * 4 + k] = r1[i];
          bot[p * 4 + k] = mix ? (uint8_t)((r1[i] + r3[i]) >> 1) : r2[i];
        }
    if (a.copy_alpha) { al[0] = r1[x + 3]; al[1] = r1[x + 7]; al[2] = r1[x + 11]; }
    uint8_t *o0 = a.dst + (size_t)(r - 1) * a.orow, *o1 = o0 + a.orow;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < a.pcpy) { o0[x + p * a.psize + k] = top[p * 4 + k]; o1[x + p * a.psize + k] = bot[p * 4 + k]; }
    if (a.copy_alpha) { o1[x + 3] = al[0]; o1[x + 7] = al[1]; o1[x + 11] = al[2]; }
  }
}

// RGBdelay / YUVdelay (RGBdelay.c:135-416): the output is the 8-bit wrap-around SUM over the enabled cached frames of a
// per-frame, per-channel LUT of the cached pixel.  One launch per processed frame: every job's three LUTs sit in LDS
// (<= 51 x 768 B), a thread walks one pixel through all jobs (wave-uniform loop) and writes it once.
struct RgbdJob {
  const uint8_t *frame;      // compact cached frame (pitch = 3 * width), or the source itself in direct mode
  uint8_t lut[3][256];
  uint8_t b[3];              // channel enabled
  uint8_t cross;             // 2: red / blue swapped between this cached frame and the output
  uint8_t pad[4];
};
struct RgbdArgs {
  const RgbdJob *jobs;       // device
  int njobs;
  uint8_t *dst;
  int orow, width, height;
  int direct, pitch;         // direct: no cache (tcache == 0, :254-309): one job, source pitch, untouched channels stay
  int inplace, ymin, uvmin;
  int reclamp;               // YUV clamped: final LUT pass (:393-403); its two tables follow the jobs
};
struct RgbdSmall { RgbdJob j[4]; };                           // up to 4 table entries travel as kernel arguments (no copy, no fence)
template <bool SMALL>
__global__ __launch_bounds__(kBlock) void k_rgbdelay(RgbdArgs a, RgbdSmall sm) {
  extern __shared__ uint8_t s_rgbd[];                         // [njobs (+1 for the reclamp pair)][3][256]
  const int nl = a.njobs + (a.reclamp ? 1 : 0);
  const RgbdJob *jobs = SMALL ? sm.j : a.jobs;
  for (int i = threadIdx.x; i < nl * 192; i += kBlock)
    reinterpret_cast<uint32_t *>(s_rgbd)[i] = reinterpret_cast<const uint32_t *>(jobs[i / 192].lut)[i % 192];
  __syncthreads();
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (!a.direct && x >= a.width && 3 * x < a.orow) {          // memset(dst, 0, dframesize) (:311) reaches the row padding too
    for (int y = blockIdx.y; y < a.height; y += gridDim.y)
      for (int b = 3 * x; b < 3 * x + 3 && b < a.orow; b++) a.dst[(size_t)y * a.orow + b] = 0;
  }
  if (x >= a.width) return;
  for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    uint8_t *d = a.dst + (size_t)y * a.orow + 3 * x;
    if (a.direct) {
      const RgbdJob &j = jobs[0];
      const uint8_t *p = j.frame + (size_t)y * a.pitch + 3 * x;
      const uint8_t *l = s_rgbd;
      uint8_t v[3] = {p[0], p[1], p[2]};
#pragma unroll
      for (int c = 0; c < 3; c++) {
        uint8_t o;
        bool wr = true;
        if (j.b[c]) o = l[c * 256 + v[c]];
        else if (a.inplace) o = (uint8_t)(c == 0 ? a.ymin : a.uvmin);
        else { wr = false; o = 0; }
        if (wr) {
          if (a.reclamp) o = s_rgbd[a.njobs * 768 + (c ? 256 : 0) + o];
          d[c] = o;
        } else if (a.reclamp) d[c] = s_rgbd[a.njobs * 768 + (c ? 256 : 0) + d[c]];      // the final pass maps whatever the frame holds
      }
    } else {
      uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
      const size_t off = (size_t)y * a.pitch + 3 * x;
      for (int k = 0; k < a.njobs; k++) {
        const RgbdJob &j = jobs[k];
        const uint8_t *p = j.frame + off;
        const uint8_t *l = s_rgbd + k * 768;
        const int cr = j.cross;
        if (j.b[0]) acc0 += l[p[cr]];
        if (j.b[1]) acc1 += l[256 + p[1]];
        if (j.b[2]) acc2 += l[512 + p[2 - cr]];
      }
      uint8_t o0 = (uint8_t)acc0, o1 = (uint8_t)acc1, o2 = (uint8_t)acc2;
      if (a.reclamp) { const uint8_t *r = s_rgbd + a.njobs * 768; o0 = r[o0]; o1 = r[256 + o1]; o2 = r[256 + o2]; }
      d[0] = o0; d[1] = o1; d[2] = o2;
    }
  }
}
// the accumulate mode with one lane per FOUR pixels (12 bytes = three dwords per cached frame, three dword stores): rows of 3 * width and
// orow bytes both multiples of 4.  Lanes past the pixels zero the row padding (memset :311).
template <bool SMALL>
__global__ __launch_bounds__(kBlock) void k_rgbdelay4(RgbdArgs a, RgbdSmall sm) {
  extern __shared__ uint8_t s_rgbd[];
  const int nl = a.njobs + (a.reclamp ? 1 : 0);
  const RgbdJob *jobs = SMALL ? sm.j : a.jobs;
  for (int i = threadIdx.x; i < nl * 192; i += kBlock)
    reinterpret_cast<uint32_t *>(s_rgbd)[i] = reinterpret_cast<const uint32_t *>(jobs[i / 192].lut)[i % 192];
  __syncthreads();
  const int g = blockIdx.x * kBlock + threadIdx.x;            // group of 4 pixels = 12 bytes
  const int b0 = 12 * g, wb = 3 * a.width;
  if (b0 >= a.orow) return;
  for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
    uint32_t *d = reinterpret_cast<uint32_t *>(a.dst + (size_t)y * a.orow + b0);
    if (b0 >= wb) {                                           // padding only
      for (int q = 0; q < 3 && b0 + 4 * q < a.orow; q++) d[q] = 0;
      continue;
    }
    uint32_t acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0;
    const size_t off = (size_t)y * a.pitch + b0;
    for (int k = 0; k < a.njobs; k++) {
      const RgbdJob &j = jobs[k];
      const uint32_t *p = reinterpret_cast<const uint32_t *>(j.frame + off);
      const uint32_t w0 = p[0], w1 = (b0 + 4 < wb) ? p[1] : 0, w2 = (b0 + 8 < wb) ? p[2] : 0;
      const uint8_t *l = s_rgbd + k * 768;
      const int cr = j.cross;
      auto byte = [&](int i) -> uint32_t { const uint32_t w = i < 4 ? w0 : i < 8 ? w1 : w2; return (w >> (8 * (i & 3))) & 0xFF; };
#pragma unroll
      for (int px = 0; px < 4; px++) {
        if (j.b[0]) acc[3 * px] += l[byte(3 * px + cr)];
        if (j.b[1]) acc[3 * px + 1] += l[256 + byte(3 * px + 1)];
        if (j.b[2]) acc[3 * px + 2] += l[512 + byte(3 * px + 2 - cr)];
      }
    }
    if (a.reclamp) {
      const uint8_t *r = s_rgbd + a.njobs * 768;
#pragma unroll
      for (int i = 0; i < 12; i++) acc[i] = r[((i % 3) ? 256 : 0) + (acc[i] & 0xFF)];
    }
    uint32_t o[3];
#pragma unroll
    for (int q = 0; q < 3; q++) o[q] = (acc[4 * q] & 0xFF) | ((acc[4 * q + 1] & 0xFF) << 8) | ((acc[4 * q + 2] & 0xFF) << 16) | ((acc[4 * q + 3] & 0xFF) << 24);
    // wb % 4 == 0: a dword below wb holds pixel bytes only, one at or above it padding only
    for (int q = 0; q < 3 && b0 + 4 * q < a.orow; q++) d[q] = (b0 + 4 * q < wb) ? o[q] : 0;
  }
}
__global__ __launch_bounds__(kBlock) void k_rgbd_snapshot(const uint8_t *src, int irow, uint8_t *frame, int wb, int height) {
  const int x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= wb) return;
  for (int y = blockIdx.y; y < height; y += gridDim.y) frame[(size_t)y * wb + x] = src[(size_t)y * irow + x];
}
}  // namespace lgpu

using namespace lgpu;

// nframes frames of one geometry (lgpu_softlight: one; lgpu_fx_batch: the instances of the filter on the live tracks of a tick) in ONE launch
int lgpu::softlight_n(const FxFrames &F, int nframes, const int irow[4], const int orow[4], int width, int height, int palette, int unclamped, hipStream_t st) {
  LGPU_REQUIRE(irow && orow, "null rowstride tables");
  LGPU_REQUIRE(palette == 544 || palette == 545 || palette == 522 || palette == 512 || palette == 513,
               "palette must be YUV444P, YUVA4444P, YUV422P, YUV420P or YVU420P (softlight.c:162-164)");
  LGPU_REQUIRE(width >= 3 && height >= 3, "softlight needs at least 3 x 3 samples");
  const int nplanes = palette == 545 ? 4 : 3;
  uintptr_t bits = 0;
  for (int f = 0; f < nframes; f++)
    for (int i = 0; i < nplanes; i++) {
      LGPU_REQUIRE(F.in0[f][i] && F.out[f][i] && F.in0[f][i] != F.out[f][i], "null plane, or in place (the filter is not CAN_DO_INPLACE)");
      if (i == 0) bits |= (uintptr_t)F.in0[f][0] | (uintptr_t)F.out[f][0];
    }
  LGPU_REQUIRE(irow[0] >= width && orow[0] >= width, "rowstride smaller than a row");
  // the other planes are copied (softlight.c:143-151) by extra blocks of the same launch; alpha of YUVA4444P has the chroma geometry of 4:4:4
  SoftCopy cp = {};
  cp.w = (palette == 512 || palette == 513 || palette == 522) ? width >> 1 : width;
  cp.h = (palette == 512 || palette == 513) ? height >> 1 : height;
  cp.n = nplanes - 1;
  for (int i = 1; i < nplanes; i++) { cp.irow[i - 1] = irow[i]; cp.orow[i - 1] = orow[i]; }
  // 4-aligned luma planes: the register form (k_softlight_s); everything else the LDS tile kernel
  if ((width & 3) == 0 && width >= 8 && ((bits | (unsigned)irow[0] | (unsigned)orow[0]) & 3) == 0 && !tune_on(TUNE_SOFT_NO_S)) {
    // rows per wave: 2 for one frame (more waves for a launch of one generation), 4 for a batch (16 x 1080p: 45.6 -> 38.5 us)
    const int rbt = tune(TUNE_SOFT_RB), rb = (rbt & 15) == 4 ? 4 : (rbt & 15) == 2 ? 2 : nframes >= 4 ? 4 : 2;
    const unsigned strips = cdiv((unsigned)(width >> 2), 62u), bands = cdiv((unsigned)height, (unsigned)(4 * rb));
    const unsigned cx = cdiv((unsigned)cp.w, 256u), cy = cdiv((unsigned)cp.h, 16u);     // the copy planes' tiles must fit the same grid
    dim3 gs(strips > cx ? strips : cx, bands > cy ? bands : cy, (unsigned)(nplanes * nframes));
    if (rbt >= 16) { gs = dim3(strips, bands, (unsigned)nframes); cp.n = 0; }          // probe: luma only
    if (rb == 4) hipLaunchKernelGGL(k_softlight_s<4>, gs, dim3(kBlock), 0, st, F, irow[0], orow[0], width, height, unclamped ? 0 : 16, unclamped ? 255 : 235, cp);
    else hipLaunchKernelGGL(k_softlight_s<2>, gs, dim3(kBlock), 0, st, F, irow[0], orow[0], width, height, unclamped ? 0 : 16, unclamped ? 255 : 235, cp);
    LGPU_CHECK_LAUNCH();
    return LGPU_OK;
  }
  const dim3 grid(cdiv((unsigned)width, kStW), cdiv((unsigned)height, kStH), (unsigned)(nplanes * nframes));
  hipLaunchKernelGGL(k_softlight, grid, dim3(kBlock), 0, st, F, irow[0], orow[0], width, height, unclamped ? 0 : 16, unclamped ? 255 : 235, cp);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

extern "C" int lgpu_softlight(const uint8_t *const src_d[4], const int irow[4], uint8_t *const dst_d[4], const int orow[4],
                              int width, int height, int palette, int unclamped, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && irow && orow, "null plane tables");
  FxFrames F = {};
  const int nplanes = palette == 545 ? 4 : 3;
  for (int i = 0; i < nplanes; i++) { F.in0[0][i] = src_d[i]; F.out[0][i] = dst_d[i]; }
  return softlight_n(F, 1, irow, orow, width, height, palette, unclamped, (hipStream_t)stream);
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
  std::lock_guard<std::mutex> seq(g_seq_mu);
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
    if (!e.st) { LGPU_HIP(hipMalloc((void **)&e.st, sizeof(EdgeState))); LGPU_HIP(hipMemset(e.st, 0, sizeof(EdgeState))); }
    if (!e.slices) LGPU_HIP(hipMalloc((void **)&e.slices, (size_t)1024 * 1025 * sizeof(unsigned int)));      // one histogram slice per map workgroup
    sc = e;
  }
  const unsigned long long count = (width > 4 && height > 4) ? (unsigned long long)(width - 4) * (unsigned long long)(height - 4) : 0ull;
  const unsigned ntiles = cdiv((unsigned)width, kStW) * cdiv((unsigned)height, kStH);
  const dim3 tgrid(ntiles < 1024u ? ntiles : 1024u);
  const dim3 pgrid(cdiv((unsigned)width, kBlock), (unsigned)(height < 1024 ? height : 1024));
  const int vec4 = (psize == 4 && ((((uintptr_t)src_d | (uintptr_t)dst_d | (uintptr_t)irow | (uintptr_t)orow) & 3) == 0)) ? 1 : 0;
  // 4-byte pixels, 16-byte aligned rows, whole quads: three launches per pass (map by quads, reduce + scan, paint by quads); the scratch map's rows are padded to quads
  const bool fast4 = psize == 4 && !(width & 3) && width >= 8 && (unsigned long long)height * (unsigned)irow < (1ull << 31) && irow < (1 << 24) && width < (1 << 22) && height < (1 << 22) && ((((uintptr_t)src_d | (uintptr_t)dst_d | (uintptr_t)irow | (uintptr_t)orow) & 15) == 0) && !tune_on(TUNE_EDGE_NO_S);
  if (fast4) {
    // tiles of 32 rows (16 with EDGE_TH = 16: measured equal on one 1080p frame)
    const int eh = tune(TUNE_EDGE_TH) > 0 ? tune(TUNE_EDGE_TH) : 32;
    const unsigned nt4 = cdiv((unsigned)width, (unsigned)kEmW) * cdiv((unsigned)height, (unsigned)(eh == 16 ? 16 : 32));
    const dim3 g4(nt4 < 1024u ? nt4 : 1024u);
    const unsigned wq = (unsigned)width >> 2;
    const unsigned long long quads = (unsigned long long)wq * (unsigned)height;
    if (quads < (1ull << 31)) {
      const uint32_t qmagic = (uint32_t)((1ull << 32) / wq - (wq == 1 ? 1 : 0));
      const unsigned pcap = (unsigned)device_cus() * 8u, pneed = (unsigned)((quads + kBlock - 1) / kBlock);
      const dim3 pg(pneed < pcap ? pneed : pcap);
      for (int pass = 0; pass < 4; pass++) {
        if (eh == 16) hipLaunchKernelGGL(k_edge_map4<16>, g4, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, width, sc.slices, &sc.st->ticket);
        else hipLaunchKernelGGL(k_edge_map4<32>, g4, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, width, sc.slices, &sc.st->ticket);
        hipLaunchKernelGGL(k_edge_reduce_otsu, dim3(4 * kEdgeChunks), dim3(256), 0, st, sc.slices, (int)g4.x, sc.st, count, pass == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_edge_paint4, pg, dim3(kBlock), 0, st, src_d, irow, dst_d, orow, (int)wq, height, pass, mode, aoffs, inplace, sc.map, width, sc.st, qmagic);
        if (mode < 2) break;
      }
      LGPU_CHECK_LAUNCH();
      return LGPU_OK;
    }
  }
  for (int pass = 0; pass < 4; pass++) {
    if (psize == 4) hipLaunchKernelGGL(k_edge_map<4>, tgrid, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, sc.slices, vec4);
    else hipLaunchKernelGGL(k_edge_map<3>, tgrid, dim3(kBlock), 0, st, src_d, irow, width, height, order, pass, device_tables()->luma, sc.map, sc.slices, 0);
    hipLaunchKernelGGL(k_edge_hist_reduce, dim3(4 * kEdgeChunks), dim3(256), 0, st, sc.slices, (int)tgrid.x, sc.st);      // tgrid.x <= 1024 = 16 * kEdgeChunks slices
    hipLaunchKernelGGL(k_edge_otsu, dim3(1), dim3(1024), 0, st, sc.st, count, pass == 0 ? 1 : 0);
    if (psize == 4) hipLaunchKernelGGL(k_edge_paint<4>, pgrid, dim3(kBlock), 0, st, src_d, irow, dst_d, orow, width, height, pass, mode, aoffs, inplace, sc.map, sc.st, vec4);
    else hipLaunchKernelGGL(k_edge_paint<3>, pgrid, dim3(kBlock), 0, st, src_d, irow, dst_d, orow, width, height, pass, mode, aoffs, inplace, sc.map, sc.st, 0);
    if (mode < 2) break;
  }
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// ---- blurzoom: stateful handle ----------------------------------------------------------------------------------------
struct lgpu_blurzoom {
  int vw, vh, bw, bh, ml, snap_time, snap_interval, threshold, device;
  int16_t *bg;
  uint8_t *buf;
  uint32_t *snap, *pal;
  int32_t *rowstart, *colcum;
};

extern "C" void lgpu_blurzoom_destroy(lgpu_blurzoom *z) {
  if (!z) return;
  (void)hipFree(z->bg); (void)hipFree(z->buf); (void)hipFree(z->snap); (void)hipFree(z->pal); (void)hipFree(z->rowstart); (void)hipFree(z->colcum);
  delete z;
}

extern "C" int lgpu_blurzoom_create(int width, int height, int palette, lgpu_blurzoom **out) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(out, "null result pointer");
  LGPU_REQUIRE(width >= 32 && height >= 3 && width / 32 <= 255, "blurzoom needs 32 <= width < 8192 (blurzoom.c:262-263)");
  LGPU_REQUIRE(palette == 3 || palette == 4, "palette must be RGBA32 (3) or BGRA32 (4)");
  lgpu_blurzoom *z = new lgpu_blurzoom();
  z->vw = width; z->vh = height;
  const int blocks = width / 32;
  z->bw = blocks * 32; z->bh = height; z->ml = (width - z->bw) / 2;
  z->snap_time = 0; z->snap_interval = 3; z->threshold = 40 * 7;
  LGPU_HIP(hipGetDevice(&z->device));
  const size_t area = (size_t)z->bw * z->bh;
  // setTable (:106-145), restated as prefix sums of the reference's own step tables
  const double RATIO = 0.95;
  const int HW = z->bw / 2, HH = z->bh / 2;
  std::vector<int32_t> colcum((size_t)z->bw), rowstart((size_t)z->bh);
  int prevptr = (int)(0.5 + RATIO * (-HW) + HW), steps = 0;
  for (int x = 0; x < z->bw; x++) {
    const int ptr = (int)(0.5 + RATIO * (x - HW) + HW);
    if (ptr != prevptr) steps++;
    prevptr = ptr;
    colcum[(size_t)x] = steps;
  }
  {
    const int tx = (int)(0.5 + RATIO * (-HW) + HW), xx = (int)(0.5 + RATIO * (z->bw - 1 - HW) + HW);
    int ty = (int)(0.5 + RATIO * (-HH) + HH);
    long pos = (long)ty * z->bw + tx;                 // blurzoomy[0]
    long prev = (long)ty * z->bw + xx;
    rowstart[0] = (int32_t)pos;
    for (int y = 1; y < z->bh; y++) {
      ty = (int)(0.5 + RATIO * (y - HH) + HH);
      pos += steps;                                   // the walk ended `steps` past the row's start
      pos += (long)ty * z->bw + tx - prev;            // blurzoomy[y]
      prev = (long)ty * z->bw + xx;
      rowstart[(size_t)y] = (int32_t)pos;
    }
  }
  for (int y = 0; y < z->bh; y++) {                   // the walk must stay inside the result plane
    const long lo = rowstart[(size_t)y], hi = lo + steps;
    if (lo < 0 || hi >= (long)area) { delete z; set_error("blurzoom zoom table leaves the feedback plane"); return LGPU_E_BADARG; }
  }
  uint32_t P[128];
  __builtin_memset(P, 0, sizeof P);
  {                                                    // makePalette (:201-237)
    const int COLORS = 32, DELTA = 255 / (COLORS / 2 - 1);
    for (int i = 0; i < COLORS / 2; i++) {
      const uint32_t d = (uint32_t)(i * DELTA);
      if (palette == 3) { P[i] = d << 16; P[COLORS * 2 + i] = d; } else { P[i] = d; P[COLORS * 2 + i] = d << 16; }
      P[COLORS + i] = d << 8;
      if (palette == 3) { P[i + COLORS / 2] = (255u << 16) | d << 8 | d; P[COLORS * 2 + i + COLORS / 2] = 255u | d << 16 | d << 8; }
      else { P[i + COLORS / 2] = 255u | d << 16 | d << 8; P[COLORS * 2 + i + COLORS / 2] = (255u << 16) | d << 8 | d; }
      P[COLORS + i + COLORS / 2] = (255u << 8) | d << 16 | d;
    }
    for (int i = 0; i < COLORS; i++) P[COLORS * 3 + i] = (uint32_t)(255 * i / COLORS) * 0x10101u;
    for (int i = 0; i < 128; i++) P[i] &= 0xfefeffu;
  }
  bool ok = hipMalloc((void **)&z->bg, sizeof(int16_t) * (size_t)width * height) == hipSuccess &&
            hipMalloc((void **)&z->buf, area * 2) == hipSuccess && hipMalloc((void **)&z->snap, 4 * (size_t)width * height) == hipSuccess &&
            hipMalloc((void **)&z->pal, sizeof P) == hipSuccess && hipMalloc((void **)&z->rowstart, 4 * (size_t)z->bh) == hipSuccess &&
            hipMalloc((void **)&z->colcum, 4 * (size_t)z->bw) == hipSuccess;
  ok = ok && hipMemset(z->bg, 0, sizeof(int16_t) * (size_t)width * height) == hipSuccess && hipMemset(z->buf, 0, area * 2) == hipSuccess &&
       hipMemset(z->snap, 0, 4 * (size_t)width * height) == hipSuccess &&
       hipMemcpy(z->pal, P, sizeof P, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(z->rowstart, rowstart.data(), 4 * (size_t)z->bh, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(z->colcum, colcum.data(), 4 * (size_t)z->bw, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { lgpu_blurzoom_destroy(z); set_error("blurzoom: device allocation failed"); return LGPU_E_NOMEM; }
  *out = z;
  return LGPU_OK;
}

extern "C" int lgpu_blurzoom_process(lgpu_blurzoom *z, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int mode, int pattern,
                                     void *stream) {
  LGPU_REQUIRE(z && src_d && dst_d, "null handle or frame");
  LGPU_REQUIRE(mode >= 0 && mode <= 3 && pattern >= 0 && pattern <= 3, "mode 0..3 (normal, strobe, strobe2, trigger), pattern 0..3 (blue, green, red, white)");
  LGPU_REQUIRE(!(irow & 3) && !(orow & 3) && irow >= z->vw * 4 && orow >= z->vw * 4, "rowstrides must be multiples of 4 and cover a row");
  LGPU_REQUIRE(!((reinterpret_cast<uintptr_t>(src_d) | reinterpret_cast<uintptr_t>(dst_d)) & 3), "frames must be 4-byte aligned");
  if ((mode == 1 || mode == 2) && irow != z->vw * 4) {
    set_error("blurzoom strobe modes read the snapshot with the source's row padding in the reference (blurzoom.c:391-396): compact rows only");
    return LGPU_E_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(src_d);
  const int vw = z->vw, vh = z->vh, bw = z->bw, bh = z->bh;
  const dim3 vgrid(cdiv((unsigned)vw, kBlock), (unsigned)(vh < 1024 ? vh : 1024)), bgrid(cdiv((unsigned)bw, kBlock), (unsigned)(bh < 1024 ? bh : 1024));
  if (mode != 2 || z->snap_time <= 0) {
    const int feed = (mode == 0 || z->snap_time <= 0) ? 1 : 0;
    hipLaunchKernelGGL(k_bz_update, vgrid, dim3(kBlock), 0, st, src, irow / 4, z->bg, z->buf, vw, vh, bw, z->ml, z->threshold, feed);
    if (feed && (mode == 1 || mode == 2))
      LGPU_HIP(hipMemcpy2DAsync(z->snap, (size_t)vw * 4, src_d, (size_t)irow, (size_t)vw * 4, (size_t)vh, hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(k_bz_blur, bgrid, dim3(kBlock), 0, st, z->buf, bw, bh);
  hipLaunchKernelGGL(k_bz_zoom, bgrid, dim3(kBlock), 0, st, z->buf, z->rowstart, z->colcum, bw, bh);
  const bool from_snap = (mode == 1 || mode == 2);
  hipLaunchKernelGGL(k_bz_color, vgrid, dim3(kBlock), 0, st, from_snap ? z->snap : src, from_snap ? vw : irow / 4, reinterpret_cast<uint32_t *>(dst_d),
                     orow / 4, z->buf, z->pal, vw, vh, bw, z->ml, pattern);
  LGPU_CHECK_LAUNCH();
  if (mode == 1 || mode == 2) { z->snap_time--; if (z->snap_time < 0) z->snap_time = z->snap_interval; }
  return LGPU_OK;
}


extern "C" int lgpu_deinterlace(const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(src_d && dst_d && width > 0 && height > 0, "null frame or empty geometry");
  lgpu::DeintArgs a = {};
  const bool inplace = (src_d == dst_d);
  switch (palette) {
  case 1: case 2: a.psize = 3; a.pcpy = 3; a.green = 1; break;
  case 588: a.psize = 3; a.pcpy = 3; break;
  case 3: case 4: a.psize = 4; a.pcpy = 3; a.green = 1; a.copy_alpha = !inplace; break;
  case 589: a.psize = 4; a.pcpy = 3; a.copy_alpha = !inplace; break;
  case 5:
    a.psize = 4; a.pcpy = 3; a.green = 2;
    if (!inplace) { lgpu::set_error("lgpu_deinterlace: ARGB32 out of place is declined (deinterlace.c:119: the triple walk drifts by one byte per triple)"); return LGPU_E_UNSUPPORTED; }
    break;
  case 564: a.psize = 4; a.pcpy = 4; a.packed422 = 1; break;
  case 565: a.psize = 4; a.pcpy = 4; a.packed422 = 2; break;
  default:
    lgpu::set_error("lgpu_deinterlace: palette %d is not taken (planar frames: the reference's pixel_size() is 0 and its loop does nothing; YUV444P dereferences an unset pointer, deinterlace.c:146)", palette);
    return LGPU_E_UNSUPPORTED;
  }
  LGPU_REQUIRE(irow >= width * a.psize && orow >= width * a.psize, "rowstride smaller than a row");
  a.ntrip = (width + 2) / 3;
  if (a.ntrip * 3 * a.psize > irow || a.ntrip * 3 * a.psize > orow) {
    lgpu::set_error("lgpu_deinterlace: width %d is not a multiple of 3 and the rows have no room for the last partial triple (the reference then writes into the next row)", width);
    return LGPU_E_UNSUPPORTED;
  }
  a.src = src_d; a.dst = dst_d; a.irow = irow; a.orow = orow; a.height = height;
  const int npairs = (height - 2) >> 1;
  if (npairs < 1) return LGPU_OK;
  std::unique_lock<std::mutex> seq(g_seq_mu, std::defer_lock);
  if (inplace) {
    seq.lock();
    // the serial reference reads rows r + 1, r + 2 before the next row pair overwrites them: read from a snapshot
    int dev = 0;
    LGPU_HIP(hipGetDevice(&dev));
    const size_t bytes = (size_t)irow * height;
    uint8_t *snap;
    {
      std::lock_guard<std::mutex> lk(g_deint_mu);
      auto &e = g_deint[std::make_pair(dev, stream)];
      if (e.second < bytes) {
        if (e.first) { LGPU_HIP(hipStreamSynchronize((hipStream_t)stream)); LGPU_HIP(hipFree(e.first)); e.first = nullptr; e.second = 0; }
        if (hipMalloc((void **)&e.first, bytes) != hipSuccess) { lgpu::set_error("lgpu_deinterlace: hipMalloc(%zu) failed", bytes); return LGPU_E_NOMEM; }
        e.second = bytes;
      }
      snap = e.first;
    }
    LGPU_HIP(hipMemcpyAsync(snap, src_d, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    a.src = snap;
  }
  a.inplace = inplace ? 1 : 0;
  // whole triples only (width a multiple of 3: a partial last triple keeps the byte walk), 4-byte pixels whose copied bytes are 0..2
  a.dword = (a.psize == 4 && !a.packed422 && a.pcpy == 3 && (width % 3) == 0 && ((((uintptr_t)a.src | (uintptr_t)dst_d | (uintptr_t)irow | (uintptr_t)orow) & 3) == 0) &&
             (inplace || a.copy_alpha)) ? 1 : 0;
  const dim3 grid(cdiv((unsigned)a.ntrip, kBlock), (unsigned)(npairs < 2048 ? npairs : 2048));
  hipLaunchKernelGGL(lgpu::k_deinterlace, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  LGPU_CHECK_LAUNCH();
  return LGPU_OK;
}

// ---- RGBdelay / YUVdelay handle: the frame ring lives in HBM, the ring bookkeeping on the host (RGBdelay.c:24-32, :54-91, :200-228) ----
struct lgpu_rgbdelay {
  int ccache = 0, tcache = 0, width = 0, height = 0, device = 0;
  uint8_t *cache[51] = {nullptr};
  int is_bgr[51] = {0};
  lgpu::RgbdJob *jobs_d = nullptr;           // 52 slots
  lgpu::RgbdJob *jobs_h = nullptr;           // pinned staging copy of the table; `copied` fences its reuse
  hipEvent_t copied = nullptr;
  bool in_flight = false;
};

extern "C" void lgpu_rgbdelay_destroy(lgpu_rgbdelay *s);
extern "C" int lgpu_rgbdelay_create(lgpu_rgbdelay **out) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(out, "null result pointer");
  lgpu_rgbdelay *s = new lgpu_rgbdelay();
  LGPU_HIP(hipGetDevice(&s->device));
  if (hipMalloc((void **)&s->jobs_d, sizeof(lgpu::RgbdJob) * 52) != hipSuccess || hipHostMalloc((void **)&s->jobs_h, sizeof(lgpu::RgbdJob) * 52) != hipSuccess ||
      hipEventCreateWithFlags(&s->copied, hipEventDisableTiming) != hipSuccess) {
    lgpu_rgbdelay_destroy(s);
    lgpu::set_error("lgpu_rgbdelay_create: allocation failed");
    return LGPU_E_NOMEM;
  }
  *out = s;
  return LGPU_OK;
}
extern "C" void lgpu_rgbdelay_destroy(lgpu_rgbdelay *s) {
  if (!s) return;
  for (int i = 0; i < 51; i++) if (s->cache[i]) (void)hipFree(s->cache[i]);
  (void)hipFree(s->jobs_d);
  if (s->jobs_h) (void)hipHostFree(s->jobs_h);
  if (s->copied) (void)hipEventDestroy(s->copied);
  delete s;
}
static void rgbd_make_lut(uint8_t *lut, double val, int min) {            // make_lut, RGBdelay.c:36-52 (double arithmetic on the host)
  int mina = min, minb = 0;
  double rnd = 0.5;
  if (min < 0) { mina = 0; minb = -min; rnd += (double)minb; }
  for (int i = 0; i < 256; i++) {
    double rval = (double)(i - mina) * val + rnd;
    if (rval < 0.) rval = 0.;
    if (rval > 255.) rval = 255.;
    lut[i] = (uint8_t)rval;
  }
}
extern "C" int lgpu_rgbdelay_process(lgpu_rgbdelay *s, const uint8_t *src_d, int irow, uint8_t *dst_d, int orow, int width, int height, int palette,
                                     int yuv_clamped, int maxcache, const int *on, const double *strength, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  LGPU_REQUIRE(s && src_d && dst_d && on && strength && width > 0 && height > 0, "null handle / frame / parameter tables or empty geometry");
  LGPU_REQUIRE(palette == 1 || palette == 2 || palette == 588, "palette must be RGB24 (1), BGR24 (2) or YUV888 (588)");
  LGPU_REQUIRE(irow >= width * 3 && orow >= width * 3, "rowstride smaller than a row");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> seq(g_seq_mu);
  const int wb = width * 3, inplace = (src_d == dst_d), is_bgr = (palette == 2), is_yuv = (palette == 588);
  if (s->width != width || s->height != height) {                 // the reference's in channel is REINIT_ON_SIZE_CHANGE: a fresh instance
    LGPU_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 51; i++) { if (s->cache[i]) (void)hipFree(s->cache[i]); s->cache[i] = nullptr; s->is_bgr[i] = 0; }
    s->tcache = s->ccache = 0; s->width = width; s->height = height;
  }
  if (maxcache < 0) maxcache = 0; else if (maxcache > 50) maxcache = 50;
  int maxneeded = 0;
  for (int i = 1; i < maxcache; i++) if (on[3 * i] || on[3 * i + 1] || on[3 * i + 2]) maxneeded = i + 1;
  if (maxneeded != s->tcache) {
    LGPU_HIP(hipStreamSynchronize(st));
    for (int i = s->tcache; i > maxneeded; i--) { (void)hipFree(s->cache[i - 1]); s->cache[i - 1] = nullptr; }
    for (int i = s->tcache; i < maxneeded; i++)
      if (hipMalloc((void **)&s->cache[i], (size_t)wb * height) != hipSuccess) {
        for (int k = s->tcache; k < i; k++) { (void)hipFree(s->cache[k]); s->cache[k] = nullptr; }
        lgpu::set_error("lgpu_rgbdelay_process: hipMalloc of a cache frame failed");
        return LGPU_E_NOMEM;
      }
    s->tcache = maxneeded;
    if (s->ccache > s->tcache) s->ccache = s->tcache;
  }
  uint8_t *tmpcache = s->tcache > 1 ? s->cache[s->tcache - 1] : nullptr;
  double tstr[3] = {0., 0., 0.}, yscale = 1., uvscale = 1.;
  for (int i = s->tcache - 1; i >= 0; i--) {
    if (i > 0) { s->cache[i] = s->cache[i - 1]; s->is_bgr[i] = s->is_bgr[i - 1]; }
    for (int c = 0; c < 3; c++) if (on[3 * i + c]) tstr[c] += strength[i];
  }
  s->is_bgr[0] = is_bgr;
  if (s->tcache > 0) {
    const dim3 g(cdiv((unsigned)wb, kBlock), (unsigned)(height < 1024 ? height : 1024));
    hipLaunchKernelGGL(lgpu::k_rgbd_snapshot, g, dim3(kBlock), 0, st, src_d, irow, tmpcache, wb, height);
    s->cache[0] = tmpcache;
  }
  for (int c = 0; c < 3; c++) if (tstr[c] < 1.) tstr[c] = 1.;
  int yuvmin = 0, uvmin = 0;
  if (is_yuv && yuv_clamped) { yuvmin = 16; uvmin = 16; yscale = 255. / 219.; uvscale = 255. / 224.; }
  if (s->in_flight) { LGPU_HIP(hipEventSynchronize(s->copied)); s->in_flight = false; }     // the previous table has left the staging buffer
  lgpu::RgbdArgs a = {};
  a.jobs = s->jobs_d; a.dst = dst_d; a.orow = orow; a.width = width; a.height = height; a.inplace = inplace; a.ymin = yuvmin; a.uvmin = uvmin;
  int n = 0;
  auto fill_job = [&](lgpu::RgbdJob &j, const uint8_t *frame, int frame_is_bgr, int idx, int cross) {
    int b[3] = {on[3 * idx] != 0, on[3 * idx + 1] != 0, on[3 * idx + 2] != 0}, red = 0, blue = 2;
    if (frame_is_bgr) { const int t = b[0]; b[0] = b[2]; b[2] = t; red = 2; blue = 0; }
    const double cstr = strength[idx];
    rgbd_make_lut(j.lut[red], cstr / tstr[0] * yscale, yuvmin);
    rgbd_make_lut(j.lut[1], cstr / tstr[1] * uvscale, yuvmin);
    rgbd_make_lut(j.lut[blue], cstr / tstr[2] * uvscale, yuvmin);
    j.frame = frame; j.b[0] = (uint8_t)b[0]; j.b[1] = (uint8_t)b[1]; j.b[2] = (uint8_t)b[2]; j.cross = (uint8_t)cross;
  };
  if (s->tcache == 0) {
    fill_job(s->jobs_h[0], src_d, is_bgr, 0, 0);
    n = 1; a.direct = 1; a.pitch = irow;
  } else {
    for (int j = 0; j < s->tcache; j++) {
      const int k = (j <= s->ccache) ? j : s->ccache;
      if (!on[3 * j] && !on[3 * j + 1] && !on[3 * j + 2] && j > 0) continue;
      const int cross = ((!is_bgr && s->is_bgr[j]) || (is_bgr && !s->is_bgr[j])) ? 2 : 0;
      fill_job(s->jobs_h[n++], s->cache[k], s->is_bgr[j], j, cross);
    }
    a.pitch = wb;                                                                   // the kernel also zeroes the row padding (:311)
  }
  if (is_yuv && yuvmin == 16) {
    lgpu::RgbdJob &r = s->jobs_h[n];
    rgbd_make_lut(r.lut[0], 1. / yscale, -yuvmin);
    rgbd_make_lut(r.lut[1], 1. / uvscale, -yuvmin);
    a.reclamp = 1;
  }
  a.njobs = n;
  const int nl = n + a.reclamp;
  const size_t lds = (size_t)nl * 768;
  // x covers the pixels and, in the accumulate mode, the row padding (3 bytes per lane)
  const unsigned span = a.direct ? (unsigned)width : (unsigned)((orow + 2) / 3);
  const dim3 grid(cdiv(span, kBlock), (unsigned)(height < 1024 ? height : 1024));
  lgpu::RgbdSmall sm = {};
  const bool quad = !a.direct && (wb & 3) == 0 && (orow & 3) == 0 && ((uintptr_t)dst_d & 3) == 0;
  const dim3 grid4(cdiv((unsigned)((orow + 11) / 12), kBlock), grid.y);
  if (nl <= 4) {
    for (int i = 0; i < nl; i++) sm.j[i] = s->jobs_h[i];
    if (quad) hipLaunchKernelGGL(lgpu::k_rgbdelay4<true>, grid4, dim3(kBlock), lds, st, a, sm);
    else hipLaunchKernelGGL(lgpu::k_rgbdelay<true>, grid, dim3(kBlock), lds, st, a, sm);
  } else if (quad) {
    LGPU_HIP(hipMemcpyAsync(s->jobs_d, s->jobs_h, sizeof(lgpu::RgbdJob) * nl, hipMemcpyHostToDevice, st));
    LGPU_HIP(hipEventRecord(s->copied, st));
    s->in_flight = true;
    if (lds > 48 * 1024) LGPU_HIP(hipFuncSetAttribute((const void *)lgpu::k_rgbdelay4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lgpu::k_rgbdelay4<false>, grid4, dim3(kBlock), lds, st, a, sm);
  } else {
    LGPU_HIP(hipMemcpyAsync(s->jobs_d, s->jobs_h, sizeof(lgpu::RgbdJob) * nl, hipMemcpyHostToDevice, st));
    LGPU_HIP(hipEventRecord(s->copied, st));
    s->in_flight = true;
    if (lds > 48 * 1024) LGPU_HIP(hipFuncSetAttribute((const void *)lgpu::k_rgbdelay<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lgpu::k_rgbdelay<false>, grid, dim3(kBlock), lds, st, a, sm);
  }
  LGPU_CHECK_LAUNCH();
  if (s->ccache < s->tcache) s->ccache++;
  return LGPU_OK;
}
