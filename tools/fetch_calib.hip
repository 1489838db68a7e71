// tools/fetch_calib.hip -- what rocprofv3's FETCH_SIZE says about a KNOWN number of bytes, per access pattern (review item: the HBM traffic of the two-dimensional
// gdk-pixbuf kernels was "uncalibrated": MI355X_MICROARCH.md says gfx950 reports HALF of a wide coalesced read stream, and nobody had said which kind the window
// staging of k_pb_pairs is).  Every kernel reads each byte of a 2 GiB buffer exactly ONCE (8 x the memory-side cache: all of it comes from HBM):
//   stream16     full waves, 16 bytes per lane, consecutive lanes consecutive addresses (the headline kernel's source rows)
//   seg640       the window staging of k_pb_pairs<4,4,6,1> at 4K -> 1706x960: per wave and source row ONE segment of 640 bytes -- 40 active lanes x 16 bytes, 24 idle --
//                segments of a row adjacent, rows 15,360 bytes apart, a workgroup's four waves on consecutive rows
//   seg640x4     the same with a lane reading 4 bytes (160 active lanes' worth per segment: the byte-pair path's request size)
//   stream4      full waves, 4 bytes per lane
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_fetch_calib tools/fetch_calib.hip ; run under  rocprofv3 --pmc FETCH_SIZE  (and, separately, TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream16(const u4 *p, size_t n16, unsigned *sink) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * 256) { const u4 v = __builtin_nontemporal_load(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void stream4(const unsigned *p, size_t n4, unsigned *sink) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i < n4; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(p + i);
  if (acc == 0x12345u) *sink = acc;
}
// rows of `row` bytes, `segs` segments of 640 bytes per row (segs * 640 <= row: the rest of a row is never read and is not counted); a wave takes (row r, segment s)
__global__ __launch_bounds__(256) void seg640(const uint8_t *p, int rows, int row, int segs, unsigned *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long w = ((long long)blockIdx.x * 4 + wave);          // (row block of 4, segment): a workgroup's waves read the SAME segment of 4 consecutive rows
  const long long rb = w / 4 / segs, s = (w / 4) % segs;
  const long long r = rb * 4 + (w & 3);
  unsigned acc = 0;
  if (r < rows && lane < 40) {
    const u4 v = *reinterpret_cast<const u4 *>(p + r * row + s * 640 + lane * 16);
    acc = v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void seg640x4(const uint8_t *p, int rows, int row, int segs, unsigned *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long w = ((long long)blockIdx.x * 4 + wave);
  const long long rb = w / 4 / segs, s = (w / 4) % segs;
  const long long r = rb * 4 + (w & 3);
  unsigned acc = 0;
  if (r < rows)
    for (int k = lane; k < 160; k += 64) acc ^= *reinterpret_cast<const unsigned *>(p + r * row + s * 640 + k * 4);
  if (acc == 0x12345u) *sink = acc;
}

int main(int argc, char **argv) {
  const size_t bytes = 2ull << 30;
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  uint8_t *d = nullptr;
  unsigned *sink = nullptr;
  if (hipMalloc((void **)&d, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(d, 1, bytes);
  hipDeviceSynchronize();
  const int row = 15360, rows = (int)(bytes / row), segs = row / 640;          // 24 segments of 640 bytes = the whole row
  const long long waves = (long long)(rows / 4) * 4 * segs;
  printf("{\"buffer_bytes\": %zu, \"seg_rows\": %d, \"seg_bytes_read\": %lld}\n", bytes, rows / 4 * 4, (long long)(rows / 4 * 4) * segs * 640);
  for (int i = 0; i < reps; i++) {
    hipLaunchKernelGGL(stream16, dim3(256 * 32), dim3(256), 0, 0, (const u4 *)d, bytes / 16, sink);
    hipLaunchKernelGGL(seg640, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, d, rows / 4 * 4, row, segs, sink);
    hipLaunchKernelGGL(seg640x4, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, d, rows / 4 * 4, row, segs, sink);
    hipLaunchKernelGGL(stream4, dim3(256 * 32), dim3(256), 0, 0, (const unsigned *)d, bytes / 4, sink);
  }
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  return 0;
}
