// tools/corun_probe.hip -- does a small kernel with N KB of LDS get a CU while the persistent chain kernel (2 x 71.9 KB workgroups on every CU) is running?
// Built by tools/corun_probe.py with hipcc; TEST TOOL, not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void k_probe(unsigned long long *out) {
  extern __shared__ unsigned char lds[];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() + lds[7];
}
extern "C" int probe_launch(void *stream, unsigned long long *out_d, int blocks, int lds_bytes) {
  hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, out_d);
  return (int)hipGetLastError();
}
