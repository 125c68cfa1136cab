// Probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (run on the GPU box: hipcc --offload-arch=gfx950 -o /tmp/p this && /tmp/p)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/, int* rowmap) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];     // assumed: A[i = l % 16][k = l / 16]
  const double b = B[(l / 16) * 16 + l % 16];    // assumed: B[k = l / 16][j = l % 16]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; v++) D[l * 4 + v] = c[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; i++) for (int kk = 0; kk < 4; kk++) hA[i * 4 + kk] = 1 + i + 100 * kk;
  for (int kk = 0; kk < 4; kk++) for (int j = 0; j < 16; j++) hB[kk * 16 + j] = 0.5 + j * 0.25 + 7 * kk;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int kk = 0; kk < 4; kk++) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; int* dm;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dm, 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dm);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  // find, for every (lane, v), the (i, j) of the reference it equals
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
    const double x = hD[l * 4 + v];
    if (x != ref[(4 * (l / 16) + v) * 16 + l % 16]) ok1 = 0;      // hypothesis 1: i = 4 (l / 16) + v, j = l % 16
    if (x != ref[(l % 16) * 16 + 4 * (l / 16) + v]) ok2 = 0;      // hypothesis 2: transposed
  }
  printf("layout: i = 4*(lane/16)+v, j = lane%%16 : %s ; transposed : %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
  if (!ok1 && !ok2) for (int l = 0; l < 64; l += 5) for (int v = 0; v < 4; v++) { const double x = hD[l * 4 + v]; for (int q = 0; q < 256; q++) if (ref[q] == x) printf("lane %d v %d -> i %d j %d\n", l, v, q / 16, q % 16); }
  return 0;
}
