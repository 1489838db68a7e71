// tools/dma_ceiling.hip -- how fast can LDS-DMA (global_load_lds_dwordx4) stream a 4K frame set out of HBM on this box, with nothing consuming the data?
// Persistent workgroups, NW loader waves each, every wave keeps up to DEPTH requests of 1 KB in flight into a private LDS ring (counted vmcnt).
// Patterns: 0 = linear (a request = 1 KB contiguous), 1 = window-shaped (a request = 64 x 16 B walking 544-byte row segments of a 38-row window, rows
// 15,360 bytes apart, windows in the tile order of k_half8s), optionally with a writer wave storing a quarter of the bytes (the chain's result stream).
// build: hipcc --offload-arch=gfx950 -O2 tools/dma_ceiling.hip -o tools/_dma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef const __attribute__((address_space(1))) void *gptr;
typedef __attribute__((address_space(3))) void *lptr;

template <int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const uint8_t *src, size_t bytes, int nw, int pattern, uint8_t *dst, int writer, int irow, int tiles_x, int tiles_y, int nframes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < nw) {
    uint8_t *ring = smem + wave * (DEPTH * 1024);
    if (pattern == 0) {
      const size_t nreq = bytes >> 10;
      const size_t per = (nreq + (size_t)gridDim.x * nw - 1) / ((size_t)gridDim.x * nw);
      const size_t r0 = ((size_t)blockIdx.x * nw + wave) * per, r1 = r0 + per < nreq ? r0 + per : nreq;
      int k = 0;
      for (size_t r = r0; r < r1; r++, k++) {
        if (k >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        __builtin_amdgcn_global_load_lds((gptr)(src + (r << 10) + lane * 16), (lptr)(ring + (k % DEPTH) * 1024), 16, 0, 0);
      }
    } else if (pattern == 4) {
      // 128-column x 8-row tiles: windows of 22 rows x 1,056 bytes (66 chunks a row), the same bytes per tile as 64 x 16 with twice as long row segments
      const int tx4 = 15, ty4 = 135, ntiles = tx4 * ty4 * nframes;
      const int xcd = blockIdx.x & 7, stride = (gridDim.x >> 3) * nw, chunk = (ntiles + 7) >> 3;
      const int wend = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
      int k = 0;
      for (int t = xcd * chunk + (blockIdx.x >> 3) * nw + wave; t < wend; t += stride) {
        const int f = t / (tx4 * ty4), tt = t - f * tx4 * ty4, ty = tt / tx4, tx = tt - ty * tx4;
        const uint8_t *base = src + (size_t)f * irow * 2160;
        for (int q = 0; q < 23; q++, k++) {
          const int c = q * 64 + lane, r = c / 66, ch = c - r * 66;
          int sy = 16 * ty - 3 + r; sy = sy < 0 ? 0 : sy > 2159 ? 2159 : sy;
          int x = 1024 * tx - 16 + ch * 16; x = x < 0 ? 0 : x > irow - 16 ? irow - 16 : x;
          if (k >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
          if (c < 22 * 66) __builtin_amdgcn_global_load_lds((gptr)(base + (size_t)sy * irow + x), (lptr)(ring + (k % DEPTH) * 1024), 16, 0, 0);
        }
      }
    } else if (pattern >= 2) {
      // pattern 2: every loader wave walks DOWN 64-column strips over a contiguous run of the column-major tile list (k_half8r): 17 requests (32 rows) per tile;
      // pattern 3: the same, but every wave takes whole strips (68 tiles), so that waves on neighbouring strips read the same source rows at the same time
      const int per = tiles_x * tiles_y, ntiles = per * nframes;
      const int nwv = gridDim.x * nw, me = blockIdx.x / 8 + (blockIdx.x & 7) * (gridDim.x / 8);       // XCD-major numbering of the workgroups
      const int wv = me * nw + wave;
      long L0, L1;
      if (pattern == 2) { L0 = (long)ntiles * wv / nwv; L1 = (long)ntiles * (wv + 1) / nwv; }
      else { const int nstrips = tiles_x * nframes; L0 = (long)wv * tiles_y; L1 = wv < nstrips ? L0 + tiles_y : L0; }
      int k = 0;
      for (long L = L0; L < L1; L++) {
        const int f = (int)(L / per), rem = (int)(L - (long)f * per), tx = rem / tiles_y, ty = rem - tx * tiles_y;
        const uint8_t *base = src + (size_t)f * irow * 2160;
        for (int q = 0; q < 17; q++, k++) {
          const int c = q * 64 + lane, r = c / 34, ch = c - r * 34;
          int sy = 32 * ty - 3 + r; sy = sy < 0 ? 0 : sy > 2159 ? 2159 : sy;
          int x = 512 * tx - 16 + ch * 16; x = x < 0 ? 0 : x > irow - 16 ? irow - 16 : x;
          if (k >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
          __builtin_amdgcn_global_load_lds((gptr)(base + (size_t)sy * irow + x), (lptr)(ring + (k % DEPTH) * 1024), 16, 0, 0);
        }
      }
    } else {
      // windows: tile (tx, ty) of frame f reads rows 32 ty - 3 .. + 37 (clamped), bytes 512 tx - 16 .. + 543; 21 requests per window
      const int ntiles = tiles_x * tiles_y * nframes;
      const int xcd = blockIdx.x & 7, stride = (gridDim.x >> 3) * nw, chunk = (ntiles + 7) >> 3;
      const int wend = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
      int k = 0;
      for (int t = xcd * chunk + (blockIdx.x >> 3) * nw + wave; t < wend; t += stride) {
        const int f = t / (tiles_x * tiles_y), tt = t - f * tiles_x * tiles_y, ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const uint8_t *base = src + (size_t)f * irow * 2160;
        for (int q = 0; q < 21; q++, k++) {
          const int c = q * 64 + lane, r = c / 34, ch = c - r * 34;
          int sy = 32 * ty - 3 + r; sy = sy < 0 ? 0 : sy > 2159 ? 2159 : sy;
          int x = 512 * tx - 16 + ch * 16; x = x < 0 ? 0 : x > irow - 16 ? irow - 16 : x;
          if (k >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
          if (c < 38 * 34) __builtin_amdgcn_global_load_lds((gptr)(base + (size_t)sy * irow + x), (lptr)(ring + (k % DEPTH) * 1024), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (writer && wave == nw) {
    if (writer == 1) {
      // the result stream: a quarter of the source bytes, 16-byte stores, linear
      const size_t n16 = (bytes >> 2) >> 4;
      for (size_t i = (size_t)blockIdx.x * 64 + lane; i < n16; i += (size_t)gridDim.x * 64) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(1, 2, 3, 4);
    } else {
      // writer 2: as k_half8s stores its tiles -- 64 x 16 result pixels per tile, a request = 4 rows x 256 bytes, rows 7,680 bytes apart, tiles in the readers' order
      const int ntiles = tiles_x * tiles_y * nframes;
      const int xcd = blockIdx.x & 7, stride = (gridDim.x >> 3), chunk = (ntiles + 7) >> 3;
      const int wend = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
      for (int t = xcd * chunk + (blockIdx.x >> 3); t < wend; t += stride) {
        const int f = t / (tiles_x * tiles_y), tt = t - f * tiles_x * tiles_y, ty = tt / tiles_x, tx = tt - ty * tiles_x;
        uint8_t *base = dst + (size_t)f * 7680 * 1080;
        for (int k2 = 0; k2 < 4; k2++) {
          int oy = 16 * ty + k2 * 4 + (lane >> 4); oy = oy > 1079 ? 1079 : oy;
          uint4 *wp = reinterpret_cast<uint4 *>(base + (size_t)oy * 7680 + (size_t)tx * 256 + (lane & 15) * 16);
          typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
          const u32x4w val = {1, 2, 3, 4};
          if (writer == 2) *wp = make_uint4(1, 2, 3, 4);
          else if (writer == 3) __builtin_nontemporal_store(val, reinterpret_cast<u32x4w *>(wp));
          else if (writer == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(wp), "v"(val) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(wp), "v"(val) : "memory");
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  const int nframes = 16, irow = 15360;
  const size_t bytes = (size_t)nframes * irow * 2160;
  uint8_t *src, *dst;
  hipMalloc(&src, bytes * 2); hipMalloc(&dst, bytes / 2);
  hipMemset(src, 1, bytes * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("pattern waves/WG WGs depth writer : us per pass, read GB/s (+ write GB/s)\n");
  for (int pattern : {1})
    for (int writer : {0, 2, 3, 4, 5})
      for (int nw : {2})
        for (int wgs : {512})
          for (int depth : {16}) {
            const size_t lds = (size_t)nw * depth * 1024;
            if (lds * (wgs / 256) > 150 * 1024) continue;
            auto launch = [&](int it) {
              const uint8_t *s = src + (it & 1) * bytes;
#define L(D) hipLaunchKernelGGL(k_stream<D>, dim3(wgs), dim3((nw + 1) * 64), lds, 0, s, bytes, nw, pattern, dst, writer, irow, 30, 68, nframes)
              if (depth == 8) L(8); else if (depth == 16) L(16); else L(32);
            };
            for (int i = 0; i < 6; i++) launch(i);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int reps = 20;
            for (int i = 0; i < reps; i++) launch(i);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            const double rbytes = pattern == 0 ? (double)bytes : pattern == 1 ? 30.0 * 68 * nframes * 38 * 544 : pattern == 4 ? 15.0 * 135 * nframes * 22 * 1056 : 30.0 * 68 * nframes * 32 * 544;
            printf("%d %d %d %d %d : %.1f us, %.0f GB/s%s\n", pattern, nw, wgs, depth, writer, us, rbytes / us / 1e3, writer ? " + write" : "");
            if (writer) printf("        (+ %.0f GB/s written; read + write %.0f GB/s)\n", bytes / 4.0 / us / 1e3, (rbytes + bytes / 4.0) / us / 1e3);
          }
  return 0;
}
