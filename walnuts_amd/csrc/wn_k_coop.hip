// wn_k_coop.hip -- kernels of the cooperative MFMA layout (wn_policy_coop.h): 16 chains x W waves
// per workgroup, dense-Gaussian gradient as v_mfma_f64_16x16x4_f64 with the precision matrix
// resident in registers.
#include "wn_launch.h"
#include "wn_policy_coop.h"

namespace wn {

// FLOW: adaptLeapFrogFlowD (adaptiveIntegrators.py:246-356; round 5): a micro step is two ticks of the same machine -- the leapfrog
// half, then the gradient at the midpoint -- so the group's gradient (and its barriers) is still one call per tick for every chain
template <int NT, int W, int KS, int TK, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__(W * 64) void wn_coop_kernel(Params prm) {
  __shared__ double lds[CoopPolicy<NT, W, KS, TK>::LDS_DOUBLES];
  CoopPolicy<NT, W, KS, TK> p(prm, lds);
  Engine<CoopPolicy<NT, W, KS, TK>, ADAPT, WARM, FLOW> eng(p, prm);
  eng.run();
}

template <int NT, int W, int KS, int TK>
static hipError_t go(int adapt_, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid((prm.C + 15) / 16), block(W * 64);
  const bool adapt = adapt_ != 0;
#ifndef WN_ONLY_FIXED
  if (adapt_ == 2) {  // adaptLeapFrogFlowD
    if constexpr (NT <= 2) {
      if (warm) hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, true, true, true>), grid, block, 0, st, prm);
      else hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, true, false, true>), grid, block, 0, st, prm);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, false, true>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (adapt) hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, true, false>), grid, block, 0, st, prm);
  else
#endif
    hipLaunchKernelGGL((wn_coop_kernel<NT, W, KS, TK, false, false>), grid, block, 0, st, prm);
  return hipGetLastError();
}

size_t coop_work_doubles(int NT, int W, int C, int M) {
  size_t groups = ((size_t)C + 15) / 16;
  return groups * (size_t)((M + 1) * 2 + 5) * W * (NT * 4) * 64;
}

hipError_t launch_coop(const KernelChoice& kc, int adapt_, bool warm, const Params& prm, hipStream_t st) {
  if (kc.W != 4) return hipErrorInvalidValue;
  const int adapt = adapt_;
  // p0 = k-steps (= ceil(D/4) rounded up to an instantiated value)
#ifdef WN_EXP_W8  // experiment: eight waves per group, one 16-row tile each (two waves per SIMD)
  if (kc.p1 == CK_DENSE && kc.p0 == 25) return go<1, 8, 25, CK_DENSE>(adapt, warm, prm, st);
#elif defined(WN_ONLY_MAIN)
  if (kc.p1 == CK_DENSE && kc.p0 == 25) return go<2, 4, 25, CK_DENSE>(adapt, warm, prm, st);
#else
  if (kc.p1 == CK_DENSE) {
    if (kc.p0 == 16) return go<1, 4, 16, CK_DENSE>(adapt, warm, prm, st);
    if (kc.p0 == 25) return go<2, 4, 25, CK_DENSE>(adapt, warm, prm, st);
    if (kc.p0 == 32) return go<2, 4, 32, CK_DENSE>(adapt, warm, prm, st);
  } else {
    if (kc.p0 == 16) return go<1, 4, 16, CK_DIAG>(adapt, warm, prm, st);
    if (kc.p0 == 32) return go<2, 4, 32, CK_DIAG>(adapt, warm, prm, st);
    if (kc.p0 == 64) return go<4, 4, 64, CK_DIAG>(adapt, warm, prm, st);
  }
#endif
  return hipErrorInvalidValue;
}

}  // namespace wn
