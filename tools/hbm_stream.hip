// tools/hbm_stream.hip -- what this box's memory system delivers to hand-written streams of the chain kernel's shape (tools/, not product).
//
//   read   : pure read stream, global_load_dwordx4 per lane, 4 loads in flight per lane, xor-reduced so that nothing is elided
//   copy   : 1 read : 1 write (the guide's float4 copy: 6.29 TB/s on its box)
//   mix5   : 5 reads : 1 write in 16-byte units -- the byte mix of one lgpu_chain launch (33.2 MB source + 8.3 MB layer 2 read, 8.3 MB written per frame)
// each as plain / non-temporal accesses, over a working set the size of the bench's step (~800 MB, 3x the 256 MiB Infinity Cache), on two rotating
// buffer sets.  Output: GB/s of bytes moved (read + written), to be compared with roofline.achieved of bench.py on the same box.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_hbm_stream tools/hbm_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT>
__device__ __forceinline__ u4 ld(const u4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT>
__device__ __forceinline__ void st(u4 *p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride over 16-byte units; UNROLL independent loads per lane before the first use
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u4 *src, size_t n, u4 *sink) {
  u4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= v[u];
  }
  for (; i < n; i += stride) acc ^= ld<NT>(src + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;       // never true for the random fill; keeps the loads
}

template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const u4 *src, u4 *dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) st<NT>(dst + i + u * stride, v[u]);
  }
  for (; i < n; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}

// 5 reads (4 from the big stream + 1 from a second stream) : 1 write
template <int NT>
__global__ __launch_bounds__(256) void k_mix5(const u4 *big, const u4 *small, u4 *dst, size_t nout) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += stride) {
    const u4 a = ld<NT>(big + i), b = ld<NT>(big + nout + i), c = ld<NT>(big + 2 * nout + i), d = ld<NT>(big + 3 * nout + i), e = ld<NT>(small + i);
    st<NT>(dst + i, a ^ b ^ c ^ d ^ e);
  }
}

// rows: the chain kernel's own access shape -- 16 frames of 3840 x 2160 x 4 (pitch 15,360 B) in one buffer; a wave owns a strip of LPL KB of a band of 12 rows and
// walks it row by row (LPL 16-byte loads per lane and row, the next row requested before the current one is consumed), plus one 8-byte read and one 8-byte write
// per lane and PAIR of rows from / to two 1920 x 1080 x 4 frames (layer 2, destination): the launch's 5 : 1 : 1 byte mix with its 2-D locality instead of a
// linear stream.  Work order: wave w -> (frame, strip, band), band-minor, as k_pb_half numbers it.
template <int LPL, int NT, int MAP = 0>
__global__ __launch_bounds__(256) void k_rows(const u4 *src, const uint2 *l2, uint2 *dst, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int strips = 15360 / (1024 * LPL), bands = 180;
  int band = wave % bands, strip = (wave / bands) % strips, frame = wave / (bands * strips);
  if (MAP == 1) {                                  // a workgroup = four adjacent strips of one band, workgroups band-minor: k_pb_half's own numbering
    const int cgroups = (strips + 3) / 4, wg = blockIdx.x, w = threadIdx.x >> 6;
    band = wg % bands; strip = ((wg / bands) % cgroups) * 4 + w; frame = wg / (bands * cgroups);
    if (strip >= strips || frame >= 16) return;
  }
  const size_t fbase = (size_t)frame * 2160 * 15360 / 16;
  const u4 *p = src + fbase + (size_t)(band * 12) * (15360 / 16) + (size_t)strip * (64 * LPL) + lane;
  u4 acc = {0, 0, 0, 0};
  u4 cur[LPL], nxt[LPL];
#pragma unroll
  for (int u = 0; u < LPL; u++) cur[u] = ld<NT>(p + 64 * u);
  for (int r = 0; r < 12; r++) {
    const u4 *q = p + (size_t)(r + 1) * (15360 / 16);
#pragma unroll
    for (int u = 0; u < LPL; u++) nxt[u] = r + 1 < 12 ? ld<NT>(q + 64 * u) : cur[u];
#pragma unroll
    for (int u = 0; u < LPL; u++) acc ^= cur[u];
    if (r & 1) {
      const size_t o = (size_t)frame * 1080 * 960 + (size_t)(band * 6 + (r >> 1)) * 960 + (size_t)strip * (64 * LPL) + lane;      // 8-byte units: a 1920 x 4 row has 960
#pragma unroll
      for (int u = 0; u < LPL; u++) {
        const uint2 a = l2[o + 64 * u];
        uint2 w; w.x = a.x ^ acc.x ^ acc.z; w.y = a.y ^ acc.y ^ acc.w;
        dst[o + 64 * u] = w;
      }
    }
#pragma unroll
    for (int u = 0; u < LPL; u++) cur[u] = nxt[u];
  }
}

// the same 1 KB strips with PF rows requested ahead instead of one (registers instead of waves as the place where bytes wait)
template <int PF>
__global__ __launch_bounds__(256) void k_rows_pf(const u4 *src, const uint2 *l2, uint2 *dst, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int strips = 15, bands = 180;
  const int band = wave % bands, strip = (wave / bands) % strips, frame = wave / (bands * strips);
  const u4 *p = src + (size_t)frame * 2160 * 15360 / 16 + (size_t)(band * 12) * (15360 / 16) + (size_t)strip * 64 + lane;
  u4 acc = {0, 0, 0, 0};
  u4 q[PF];
#pragma unroll
  for (int u = 0; u < PF; u++) q[u] = *(p + (size_t)u * (15360 / 16));
#pragma unroll
  for (int r = 0; r < 12; r++) {
    const u4 cur = q[r % PF];
    if (r + PF < 12) q[r % PF] = *(p + (size_t)(r + PF) * (15360 / 16));
    acc ^= cur;
    if (r & 1) {
      const size_t o = (size_t)frame * 1080 * 960 + (size_t)(band * 6 + (r >> 1)) * 960 + (size_t)strip * 64 + lane;
      const uint2 a = l2[o];
      uint2 w; w.x = a.x ^ acc.x ^ acc.z; w.y = a.y ^ acc.y ^ acc.w;
      dst[o] = w;
    }
  }
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char **argv) {
  const size_t out_units = (size_t)16 * 1920 * 1080 * 4 / 16;          // 16 frames of 1920x1080x4 in 16-byte units = 132.7 MB
  const size_t big_units = 4 * out_units;                              // 530.8 MB
  const int sets = 2, reps = 40;
  u4 *big[2], *small[2], *dst[2], *sink;
  for (int s = 0; s < sets; s++) {
    CK(hipMalloc(&big[s], big_units * 16)); CK(hipMalloc(&small[s], out_units * 16)); CK(hipMalloc(&dst[s], big_units * 16));
    CK(hipMemset(big[s], 0x5A + s, big_units * 16)); CK(hipMemset(small[s], 0x33 + s, out_units * 16)); CK(hipMemset(dst[s], 0, big_units * 16));
  }
  CK(hipMalloc(&sink, 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grids[] = {cus * 4, cus * 8, cus * 16, cus * 32};
  printf("# %d CUs; working set per launch: read %.1f MB / copy %.1f MB / mix5 %.1f MB; %d launches each on %d rotating buffer sets\n", cus, big_units * 16 / 1e6,
         2 * big_units * 16 / 1e6, 6 * out_units * 16 / 1e6, reps, sets);
  auto run = [&](const char *name, double bytes, auto launch) {
    for (int g : grids) {
      for (int w = 0; w < 6; w++) launch(g, w & 1);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) launch(g, r & 1);
      CK(hipEventRecord(e1, 0));
      const float ms = time_ms(e0, e1) / reps;
      printf("%-22s grid %6d  %8.2f us  %8.1f GB/s\n", name, g, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    }
  };
  run("read plain x4", big_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_read<0, 4>), dim3(g), dim3(256), 0, 0, big[s], big_units, sink); });
  run("read nt x4", big_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_read<1, 4>), dim3(g), dim3(256), 0, 0, big[s], big_units, sink); });
  run("read plain x8", big_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_read<0, 8>), dim3(g), dim3(256), 0, 0, big[s], big_units, sink); });
  run("copy plain x4", 2 * big_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_copy<0, 4>), dim3(g), dim3(256), 0, 0, big[s], dst[s], big_units); });
  run("copy nt x4", 2 * big_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_copy<1, 4>), dim3(g), dim3(256), 0, 0, big[s], dst[s], big_units); });
  run("mix5 (5R:1W) plain", 6 * out_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_mix5<0>), dim3(g), dim3(256), 0, 0, big[s], small[s], dst[s], out_units); });
  run("mix5 (5R:1W) nt", 6 * out_units * 16.0, [&](int g, int s) { hipLaunchKernelGGL((k_mix5<1>), dim3(g), dim3(256), 0, 0, big[s], small[s], dst[s], out_units); });
  // the chain kernel's 2-D shape: strips of 1 / 2 / 4 KB per wave and row
  {
    const double bytes = 16.0 * (3840.0 * 2160 * 4 + 2 * 1920.0 * 1080 * 4);
    auto rows = [&](const char *name, int lpl, auto launch) {
      const int nwaves = 16 * (15360 / (1024 * lpl)) * 180;
      for (int w = 0; w < 6; w++) launch((nwaves + 3) / 4, w & 1, nwaves);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) launch((nwaves + 3) / 4, r & 1, nwaves);
      CK(hipEventRecord(e1, 0));
      const float ms = time_ms(e0, e1) / reps;
      printf("%-22s waves %6d  %8.2f us  %8.1f GB/s\n", name, nwaves, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    };
    rows("rows 1 KB strips", 1, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows<1, 0>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    rows("rows 2 KB strips", 2, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows<2, 0>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    rows("rows 4 KB strips", 4, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows<4, 0>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    {
      const int nwg = 16 * 4 * 180;               // 16 frames x ceil(15 / 4) strip groups x 180 bands
      for (int w = 0; w < 6; w++) hipLaunchKernelGGL((k_rows<1, 0, 1>), dim3(nwg), dim3(256), 0, 0, big[w & 1], (const uint2 *)small[w & 1], (uint2 *)dst[w & 1], 1 << 30);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_rows<1, 0, 1>), dim3(nwg), dim3(256), 0, 0, big[r & 1], (const uint2 *)small[r & 1], (uint2 *)dst[r & 1], 1 << 30);
      CK(hipEventRecord(e1, 0));
      const float ms = time_ms(e0, e1) / reps;
      printf("%-22s wgs   %6d  %8.2f us  %8.1f GB/s\n", "rows 1 KB, wg = 4 adjacent strips", nwg, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    }
    rows("rows 1 KB, 2 rows ahead", 1, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows_pf<2>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    rows("rows 1 KB, 4 rows ahead", 1, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows_pf<4>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    rows("rows 1 KB, 12 rows ahead", 1, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows_pf<12>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
    rows("rows 1 KB strips nt", 1, [&](int g, int s, int n) { hipLaunchKernelGGL((k_rows<1, 1>), dim3(g), dim3(256), 0, 0, big[s], (const uint2 *)small[s], (uint2 *)dst[s], n); });
  }
  return 0;
}
