// Micro-benchmark: cost of LDS ds_add_f64 (no return) / plain ds_read + ds_write on gfx950 as a function of the number of active lanes
// and the address pattern.  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_f64 lds_atomic_f64.hip && ./lds_atomic_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_double;
template <int MODE>   // 0: atomic add, distinct addresses per lane; 1: atomic add, all lanes one address; 2: atomic, lanes in groups of 4 share an address; 3: plain RMW distinct; 4: groups of 16 share an address
__global__ void k(unsigned long long* out, int active, double* sink, int zero) {
  __shared__ double buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = 0;
  __syncthreads();
  lds_double* b = (lds_double*)buf;
  const int lane = threadIdx.x;
  const int base = MODE == 1 ? zero : (MODE == 2 ? (lane >> 2) * 33 : (MODE == 4 ? (lane >> 4) * 33 : lane * 33));
  unsigned long long t0 = clock64();
  if (lane < active) {
#pragma unroll 16
    for (int i = 0; i < 256; i++) {
      const int a = (base + i * 7) & 4095;
      if (MODE == 3) b[a] = b[a] + 1.0 + i;
      else (void)__hip_atomic_fetch_add(&b[a], 1.0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  if (lane == 0) out[0] = t1 - t0;
  double s = 0; for (int i = lane; i < 4096; i += 64) s += buf[i];
  sink[lane] = s;
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 64 * 8);
  const char* names[5] = {"atomic distinct", "atomic same address", "atomic groups of 4", "plain rmw distinct", "atomic groups of 16"};
  for (int mode = 0; mode < 5; mode++)
    for (int active : {1, 4, 8, 16, 32, 64}) {
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, active, sink, 0);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, active, sink, 0);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, active, sink, 0);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, active, sink, 0);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, active, sink, 0);
        hipDeviceSynchronize();
      }
      unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      printf("%-22s active %2d: %6.1f clock64 ticks per op\n", names[mode], active, (double)h / 256.0);
    }
  return 0;
}
