// ISA probe: the tree solver of the generic engine (gen_tree.h) as a kernel of its own, for instruction / register / source-line statistics
// (tools/probe/tree_probe.sh).  Not part of the library.
#include <hip/hip_runtime.h>
#include "../../d3il_amd/csrc/gen_step.h"
using namespace d3il;
__global__ __launch_bounds__(64) void k_tree_probe(double* scratch, double* state, unsigned* out, int n, int stride, int warm) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x & 63, col = lane & (GEN_LANES - 1), l = lane / GEN_LANES;
  const int e = blockIdx.x * GEN_LANES + col;
  PushScratch sc{(push_lds_double*)(smem + col), (push_glb_double*)(scratch + (size_t)blockIdx.x * GG_BLOCK * GEN_LANES + 2 * col), GEN_LANES, (push_glb_double*)(state + (size_t)42 * stride + e), stride};
  unsigned fl = 0;
  if (e < n && l < g_gen_consts.nb) fl = gen_tree_solve<1>(g_gen_consts, sc, l, warm != 0);
  out[blockIdx.x * 64 + lane] = fl;
}
