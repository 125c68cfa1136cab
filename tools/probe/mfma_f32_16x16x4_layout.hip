// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950, found by one-hot experiments (prints, for every A lane and B lane, the row / column of D
// it feeds, and which A lanes pair with which B lanes in the k sum).  hipcc --offload-arch=gfx950 -O2 -o mfma_f32_layout mfma_f32_16x16x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}
int main() {
  float *a, *b, *d; hipMallocManaged(&a, 256); hipMallocManaged(&b, 256); hipMallocManaged(&d, 1024);
  // 1. D position -> (i, j): A[i][k] = i + 1 for all k needs the A layout; instead find rows / columns by one-hot lanes
  int arow[64], bcol[64];
  for (int L = 0; L < 64; L++) {      // A one-hot at lane L, B all ones: D = row i(L) filled with 1
    for (int l = 0; l < 64; l++) { a[l] = l == L; b[l] = 1.f; }
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b, d); hipDeviceSynchronize();
    printf("A lane %2d feeds D entries (lane:reg):", L); int cnt = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) if (d[l * 4 + r] != 0.f) { if (cnt < 4) printf(" %d:%d", l, r); cnt++; }
    printf(" ... %d entries\n", cnt);
  }
  for (int L = 0; L < 64; L++) {      // B one-hot at lane L, A all ones: D = column j(L) filled
    for (int l = 0; l < 64; l++) { b[l] = l == L; a[l] = 1.f; }
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b, d); hipDeviceSynchronize();
    printf("B lane %2d feeds D entries (lane:reg):", L); int cnt = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) if (d[l * 4 + r] != 0.f) { if (cnt < 4) printf(" %d:%d", l, r); cnt++; }
    printf(" ... %d entries\n", cnt);
  }
  // 2. which B lanes share k with A lane 0, 16, 32, 48 (and 1)
  for (int La : {0, 1, 16, 32, 48}) {
    printf("A lane %2d pairs (same k) with B lanes:", La);
    for (int Lb = 0; Lb < 64; Lb++) {
      for (int l = 0; l < 64; l++) { a[l] = l == La; b[l] = l == Lb; }
      hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b, d); hipDeviceSynchronize();
      bool nz = false; for (int q = 0; q < 256; q++) nz |= d[q] != 0.f;
      if (nz) printf(" %d", Lb);
    }
    printf("\n");
  }
  return 0;
}
