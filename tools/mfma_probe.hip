// mfma_probe.hip -- determines the A/B operand layout of v_mfma_i32_16x16x64_i8 on gfx950 (tooling).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int int4v __attribute__((ext_vector_type(4)));
// lane l provides 16 bytes of A and 16 bytes of B exactly as laid out in abuf/bbuf (lane-major)
__global__ void k(const int4v *abuf, const int4v *bbuf, int *d) {
  int l = threadIdx.x;
  int4v a = abuf[l], b = bbuf[l], c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) d[l * 4 + i] = c[i];
}
int main() {
  signed char A[16][64], B[64][16];
  srand(1);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 64; k++) A[i][k] = (signed char)(rand() % 256 - 128);
  for (int k = 0; k < 64; k++) for (int j = 0; j < 16; j++) B[k][j] = (signed char)(rand() % 256 - 128);
  int ref[16][16];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += A[i][k] * B[k][j]; ref[i][j] = s; }
  // candidate layouts: lane l, byte e (0..15) -> k index
  for (int cand = 0; cand < 3; cand++) {
    signed char ab[64][16], bb[64][16];
    for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) {
      int k;
      if (cand == 0) k = 16 * (l >> 4) + e;                        // contiguous 16
      else if (cand == 1) k = 8 * (l >> 4) + (e & 7) + 32 * (e >> 3);  // two K=32 halves, 8 contiguous each
      else k = 4 * (l >> 4) + (e & 3) + 16 * (e >> 2);             // four K=16 quarters
      ab[l][e] = A[l & 15][k];
      bb[l][e] = B[k][l & 15];
    }
    int4v *da, *db; int *dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, ab, 1024, hipMemcpyHostToDevice); hipMemcpy(db, bb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    int out[64][4];
    hipMemcpy(out, dd, 64 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) if (out[l][i] != ref[(l >> 4) * 4 + i][l & 15]) bad++;
    printf("candidate %d: %d mismatches (C/D map row=(l>>4)*4+i, col=l&15)\n", cand, bad);
  }
  return 0;
}
