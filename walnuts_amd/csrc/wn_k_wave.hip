// wn_k_wave.hip -- persistent kernels of the one-chain-per-wavefront layout (wn_policy_wave.h) for
// fused, HBM-bound targets with 256 < D <= 1024 (std / diagonal Gaussian, StockWatson).  One workgroup = one
// wave = one chain; no LDS, no barriers.
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

template <int NE, class Tgt, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__(64) void wn_wave_kernel(Params prm) {
  __shared__ double lds[Tgt::LDS_DOUBLES > 0 ? Tgt::LDS_DOUBLES : 1];
  WavePolicy<NE, Tgt> p(prm, blockIdx.x, lds);
  Engine<WavePolicy<NE, Tgt>, ADAPT, WARM, FLOW> eng(p, prm);
  eng.run();
}

template <int NE, class Tgt>
static hipError_t go(int adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64);
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false, true>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, false, true>), grid, block, 0, st, prm);
  } else {
    if (adapt) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, false, false>), grid, block, 0, st, prm);
  }
  return hipGetLastError();
}

size_t wave_work_doubles(int NE, int C, int M) {
  if (NE == 12) return WavePolicy<12, WTgtDiag<12>>::work_doubles(C, M);
  return NE <= 8 ? WavePolicy<8, WTgtDiag<8>>::work_doubles(C, M) : WavePolicy<16, WTgtDiag<16>>::work_doubles(C, M);
}

hipError_t launch_wave(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (kc.p1 == T_DIAGGAUSS || kc.p1 == T_STDGAUSS) {
    if (kc.p0 == 8) return go<8, WTgtDiag<8>>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go<16, WTgtDiag<16>>(adapt, warm, prm, st);
  }
  if (kc.p1 == T_STOCKWATSON && kc.p0 == 12) return go<12, WTgtStockWatson<12>>(adapt, warm, prm, st);
  return hipErrorInvalidValue;
}

}  // namespace wn
