// wn_k_lane.hip -- kernels of the one-chain-per-lane layout (small D: funnel10, 2-D toy targets,
// low-dimensional Gaussians).  One wavefront = 64 chains; no LDS, no barriers.
#include "wn_launch.h"
#include "wn_policy_lane.h"
#include "wn_targets.h"

namespace wn {

template <int DMAX, class Tgt, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__(64) void wn_lane_kernel(Params prm) {
  __shared__ double lds[LanePolicy<DMAX, Tgt>::WARM_DOUBLES];
  const int cl = blockIdx.x * 64 + threadIdx.x;
  LanePolicy<DMAX, Tgt> p(prm, cl, lds);
  Engine<LanePolicy<DMAX, Tgt>, ADAPT, WARM, FLOW> eng(p, prm);
  eng.run();
}

template <int DMAX, class Tgt>
static hipError_t go(int adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid((prm.C + 63) / 64), block(64);
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, true, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, true, false, true>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, false, true>), grid, block, 0, st, prm);
  } else {
    if (adapt) hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, true, false>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_lane_kernel<DMAX, Tgt, false, false>), grid, block, 0, st, prm);
  }
  return hipGetLastError();
}

size_t lane_work_doubles(int dmax, int C, int M) {
  size_t cpad = ((size_t)C + 63) / 64 * 64;
  return ((size_t)(M + 1) * 2 + COLD_N + 4) * dmax * cpad;
}

hipError_t launch_lane(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
#ifdef WN_NO_LANE  // diagnostic variant builds (tools/build_variant.sh) skip the lane kernels
  return hipErrorInvalidValue;
#else
#ifdef WN_LANE_ONLY_FUNNEL
  if (kc.p1 == T_FUNNEL10) return go<11, TgtFunnel10>(adapt, warm, prm, st);
  return hipErrorInvalidValue;
#endif
  switch (kc.p1) {
    case T_STDGAUSS:
    case T_DIAGGAUSS:
      switch (kc.p0) {
        case 4: return go<4, TgtDiag<4>>(adapt, warm, prm, st);
        case 8: return go<8, TgtDiag<8>>(adapt, warm, prm, st);
        case 16: return go<16, TgtDiag<16>>(adapt, warm, prm, st);
        case 24: return go<24, TgtDiag<24>>(adapt, warm, prm, st);
      }
      break;
    case T_DENSEGAUSS:
      switch (kc.p0) {
        case 4: return go<4, TgtDense<4>>(adapt, warm, prm, st);
        case 8: return go<8, TgtDense<8>>(adapt, warm, prm, st);
        case 16: return go<16, TgtDense<16>>(adapt, warm, prm, st);
      }
      break;
    case T_FUNNEL10: return go<11, TgtFunnel10>(adapt, warm, prm, st);
    case T_CORRGAUSS: return go<2, TgtCorrGauss>(adapt, warm, prm, st);
    case T_SMILE: return go<2, TgtSmile>(adapt, warm, prm, st);
    case T_FUNNEL1: return go<2, TgtFunnel1>(adapt, warm, prm, st);
    case T_MODFUNNEL: return go<2, TgtModFunnel>(adapt, warm, prm, st);
  }
  return hipErrorInvalidValue;
#endif
}

}  // namespace wn
