// tools/pcie_probe.hip -- ways to move one PAGEABLE 1080p RGBA plane (8.3 MB, what LiVES' frame allocator hands out) across PCIe and back:
// plain hipMemcpyAsync, two pinned 4 MB staging chunks with the CPU copy overlapped (what lgpu_upload / lgpu_download do), hipHostRegister around a direct DMA.
// build: hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o tools/_pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 8294400;
  void *d; hipMalloc(&d, n);
  const int reps = 10;
  char *h[reps];
  for (int i = 0; i < reps; i++) { h[i] = (char *)malloc(n + 64); memset(h[i], i, n); }
  // 1. pageable, plain
  double t0 = now();
  for (int i = 0; i < reps; i++) { hipMemcpyAsync(d, h[i], n, hipMemcpyHostToDevice, 0); hipStreamSynchronize(0); }
  double t1 = now();
  for (int i = 0; i < reps; i++) { hipMemcpyAsync(h[i], d, n, hipMemcpyDeviceToHost, 0); hipStreamSynchronize(0); }
  double t2 = now();
  printf("pageable hipMemcpyAsync          : up %.0f us (%.1f GB/s), down %.0f us (%.1f GB/s)\n", (t1 - t0) / reps, n / ((t1 - t0) / reps) / 1e3, (t2 - t1) / reps, n / ((t2 - t1) / reps) / 1e3);
  // 2. staged through two pinned 4 MB chunks
  const size_t ck = 4u << 20;
  void *st[2]; hipEvent_t ev[2];
  for (int k = 0; k < 2; k++) { hipHostMalloc(&st[k], ck, hipHostMallocDefault); hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); }
  t0 = now();
  for (int i = 0; i < reps; i++) {
    int k = 0; bool busy[2] = {false, false};
    for (size_t off = 0; off < n; off += ck, k ^= 1) {
      const size_t m = n - off < ck ? n - off : ck;
      if (busy[k]) hipEventSynchronize(ev[k]);
      memcpy(st[k], h[i] + off, m);
      hipMemcpyAsync((char *)d + off, st[k], m, hipMemcpyHostToDevice, 0);
      hipEventRecord(ev[k], 0); busy[k] = true;
    }
    hipStreamSynchronize(0);
  }
  t1 = now();
  printf("two pinned 4 MB staging chunks   : up %.0f us (%.1f GB/s)\n", (t1 - t0) / reps, n / ((t1 - t0) / reps) / 1e3);
  // 3. register, DMA, unregister
  double treg = 0, tcpy = 0, tunreg = 0, tdown = 0;
  for (int i = 0; i < reps; i++) {
    double a = now();
    hipError_t e = hipHostRegister(h[i], n, hipHostRegisterDefault);
    double b = now();
    if (e != hipSuccess) { printf("hipHostRegister failed: %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpyAsync(d, h[i], n, hipMemcpyHostToDevice, 0); hipStreamSynchronize(0);
    double c = now();
    hipMemcpyAsync(h[i], d, n, hipMemcpyDeviceToHost, 0); hipStreamSynchronize(0);
    double c2 = now();
    hipHostUnregister(h[i]);
    double dd = now();
    treg += b - a; tcpy += c - b; tdown += c2 - c; tunreg += dd - c2;
  }
  printf("hipHostRegister around the DMA   : register %.0f us, up %.0f us (%.1f GB/s), down %.0f us (%.1f GB/s), unregister %.0f us\n", treg / reps, tcpy / reps, n / (tcpy / reps) / 1e3,
         tdown / reps, n / (tdown / reps) / 1e3, tunreg / reps);
  // 3b. downloads into FRESH (never touched) pageable blocks, which is what a seam call's new host plane is: plain, and through the staging chunks
  {
    char *f[reps];
    for (int i = 0; i < reps; i++) f[i] = (char *)malloc(n + 64);
    t0 = now();
    for (int i = 0; i < reps; i++) { hipMemcpyAsync(f[i], d, n, hipMemcpyDeviceToHost, 0); hipStreamSynchronize(0); }
    t1 = now();
    printf("download into untouched pageable : plain %.0f us (%.1f GB/s)\n", (t1 - t0) / reps, n / ((t1 - t0) / reps) / 1e3);
    for (int i = 0; i < reps; i++) { free(f[i]); f[i] = (char *)malloc(n + 64); }
    t0 = now();
    for (int i = 0; i < reps; i++) {
      const size_t nch = (n + ck - 1) / ck;
      hipMemcpyAsync(st[0], d, n < ck ? n : ck, hipMemcpyDeviceToHost, 0); hipEventRecord(ev[0], 0);
      for (size_t c = 0; c < nch; c++) {
        const int k = (int)(c & 1);
        const size_t off = c * ck, m = n - off < ck ? n - off : ck;
        if (c + 1 < nch) { const size_t o2 = (c + 1) * ck, m2 = n - o2 < ck ? n - o2 : ck; hipMemcpyAsync(st[k ^ 1], (char *)d + o2, m2, hipMemcpyDeviceToHost, 0); hipEventRecord(ev[k ^ 1], 0); }
        hipEventSynchronize(ev[k]);
        memcpy(f[i] + off, st[k], m);
      }
    }
    t1 = now();
    printf("download into untouched pageable : staged %.0f us (%.1f GB/s)\n", (t1 - t0) / reps, n / ((t1 - t0) / reps) / 1e3);
    for (int i = 0; i < reps; i++) free(f[i]);
  }
  // 4. already pinned
  void *hp; hipHostMalloc(&hp, n, hipHostMallocDefault); memset(hp, 3, n);
  t0 = now();
  for (int i = 0; i < reps; i++) { hipMemcpyAsync(d, hp, n, hipMemcpyHostToDevice, 0); hipStreamSynchronize(0); }
  t1 = now();
  for (int i = 0; i < reps; i++) { hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, 0); hipStreamSynchronize(0); }
  t2 = now();
  printf("page-locked (hipHostMalloc) frame: up %.0f us (%.1f GB/s), down %.0f us (%.1f GB/s)\n", (t1 - t0) / reps, n / ((t1 - t0) / reps) / 1e3, (t2 - t1) / reps, n / ((t2 - t1) / reps) / 1e3);
  return 0;
}
