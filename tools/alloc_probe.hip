// tools/alloc_probe.hip -- host cost of device allocation paths on this box: hipMalloc / hipFree against the stream-ordered pool (hipMallocAsync / hipFreeAsync)
// build: hipcc --offload-arch=gfx950 -O2 tools/alloc_probe.hip -o tools/_alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_touch(uint8_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  hipFree(nullptr);
  const size_t sizes[] = {1u << 20, 2400000, 8294400, 33177600};
  for (size_t n : sizes) {
    void *p = nullptr;
    for (int warm = 0; warm < 2; warm++) { hipMalloc(&p, n); hipFree(p); }
    double t0 = now();
    for (int i = 0; i < 20; i++) { hipMalloc(&p, n); hipFree(p); }
    double t1 = now();
    // with a busy stream: does hipMalloc / hipFree wait for the device?
    void *big; hipMalloc(&big, 1u << 28);
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_touch, dim3((1u << 28) / 256), dim3(256), 0, 0, (uint8_t *)big, (size_t)1 << 28);
    double t2 = now();
    hipMalloc(&p, n);
    double t3 = now();
    hipFree(p);
    double t4 = now();
    hipDeviceSynchronize();
    hipMemPool_t mp; hipDeviceGetDefaultMemPool(&mp, 0);
    uint64_t thr = UINT64_MAX; hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &thr);
    for (int warm = 0; warm < 2; warm++) { hipMallocAsync(&p, n, 0); hipFreeAsync(p, 0); }
    hipDeviceSynchronize();
    double t5 = now();
    for (int i = 0; i < 20; i++) { hipMallocAsync(&p, n, 0); hipFreeAsync(p, 0); }
    double t6 = now();
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_touch, dim3((1u << 28) / 256), dim3(256), 0, 0, (uint8_t *)big, (size_t)1 << 28);
    double t7 = now();
    hipMallocAsync(&p, n, 0);
    hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint8_t *)p, n);
    hipFreeAsync(p, 0);
    double t8 = now();
    hipDeviceSynchronize();
    hipFree(big);
    printf("%9zu bytes: hipMalloc+hipFree %.1f us idle | busy stream: hipMalloc %.1f us, hipFree %.1f us | hipMallocAsync+hipFreeAsync %.1f us idle, %.1f us (alloc + launch + free) behind a busy stream\n",
           n, (t1 - t0) / 20, t3 - t2, t4 - t3, (t6 - t5) / 20, t8 - t7);
  }
  return 0;
}
