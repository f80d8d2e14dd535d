// wn_k_wave.hip -- persistent kernels of the one-chain-per-wavefront layout (wn_policy_wave.h) for
// fused, HBM-bound targets with 256 < D <= 1024 (std / diagonal Gaussian, StockWatson).  One workgroup = one
// wave = one chain; no LDS, no barriers.
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

template <int NE, class Tgt, bool ADAPT, bool WARM, bool FLOW = false, int W = 1, bool CL = false, int SPEC = -1>
__global__ __launch_bounds__(64 * W) void wn_wave_kernel(Params prm) {
  using P = WavePolicy<NE, Tgt, W, CL>;
  __shared__ double lds[P::LDS_DOUBLES > 0 ? P::LDS_DOUBLES : 1];
  extern __shared__ double wn_wave_dyn_lds[];  // P::VLDS: the chain's vectors (P::dyn_lds_bytes)
  P p(prm, blockIdx.x, lds, wn_wave_dyn_lds);
  Engine<P, ADAPT, WARM, FLOW, SPEC> eng(p, prm);
  if constexpr (SPEC >= 0 && W == 1 && !CL) eng.run_nested();  // one chain per wave: nested loops instead of the tick loop
  else eng.run();
}

template <class T, class = void> struct wave_spec_engine { static constexpr bool value = false; };
template <class T> struct wave_spec_engine<T, typename std::enable_if<T::SPEC_ENGINE>::type> { static constexpr bool value = true; };

template <int NE, class Tgt, int W = 1, bool CL = false>
static hipError_t go(int adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(CL ? prm.C * prm.cl_g : prm.C), block(64 * W);
  const size_t dyn = WavePolicy<NE, Tgt, W, CL>::dyn_lds_bytes(prm.M);
  // the small targets (one register per vector) and StockWatson are tick-latency bound: plain sampling runs of the two integrators the
  // reference's experiments use on them get an engine with the run-time switches compiled out (Engine SPEC)
  // (round 4: StockWatson's fixedLeapFrog run too -- mainSW.py:55-62, 2047 leaves of one evaluation each per iteration: the tick loop's
  //  flag tests are branches, and a wave alone on its SIMD pays 25-55 cycles for each, tools/branch_cost.hip)
  if constexpr (wave_spec_engine<Tgt>::value) if (adapt == 0 && !warm && prm.mode == MODE_SAMPLE && !prm.trace_n && !prm.omin && prm.integrator == I_FIXED) {
    hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, false, false, false, W, CL, I_FIXED>), grid, block, dyn, st, prm);
    return hipGetLastError();
  }
  if constexpr (WavePolicy<NE, Tgt, W, CL>::VLDS || wave_spec_engine<Tgt>::value) if (adapt == 1 && !warm && prm.mode == MODE_SAMPLE && !prm.trace_n && !prm.omin) {
    if (prm.integrator == I_ADAPT_R2P) {
      hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false, false, W, CL, I_ADAPT_R2P>), grid, block, dyn, st, prm);
      return hipGetLastError();
    }
    if (prm.integrator == I_ADAPT_D) {
      hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false, false, W, CL, I_ADAPT_D>), grid, block, dyn, st, prm);
      return hipGetLastError();
    }
  }
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, true, true, W, CL>), grid, block, dyn, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false, true, W, CL>), grid, block, dyn, st, prm);
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, true, false, W, CL>), grid, block, dyn, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, false, true, false, W, CL>), grid, block, dyn, st, prm);
  } else {
    if (adapt) hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, true, false, false, W, CL>), grid, block, dyn, st, prm);
    else hipLaunchKernelGGL((wn_wave_kernel<NE, Tgt, false, false, false, W, CL>), grid, block, dyn, st, prm);
  }
  return hipGetLastError();
}

size_t wave_work_doubles(int NE, int C, int M, int G) {
  if (NE == 64) return WavePolicy<16, WTgtDiag<16, 256>, 4>::work_doubles(C, M, G);  // four waves per workgroup, D <= 4096 * G
  if (NE == 1) return WavePolicy<1, WTgtFunnel10>::work_doubles(C, M);
  if (NE == 2) return WavePolicy<2, WTgtDiag<2>>::work_doubles(C, M);  // (wn_k_wavelds.hip)
  if (NE == 12) return WavePolicy<12, WTgtDiag<12>>::work_doubles(C, M);
  return NE <= 8 ? WavePolicy<8, WTgtDiag<8>>::work_doubles(C, M) : WavePolicy<16, WTgtDiag<16>>::work_doubles(C, M);
}

hipError_t launch_wave(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (wavelds_geom(kc)) return launch_wavelds(kc, adapt, warm, prm, st);
  if (kc.p1 == T_DIAGGAUSS || kc.p1 == T_STDGAUSS) {
    if (kc.p0 == 8) return go<8, WTgtDiag<8>>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go<16, WTgtDiag<16>>(adapt, warm, prm, st);
    if (kc.p0 == 64 && kc.G > 1) return go<16, WTgtDiag<16, 256>, 4, true>(adapt, warm, prm, st);  // cluster of workgroups per chain
    if (kc.p0 == 64) return go<16, WTgtDiag<16, 256>, 4>(adapt, warm, prm, st);  // DP / 64 = 64: four waves per chain
  }
  if (kc.p1 == T_FUNNEL10 && kc.p0 == 1) return go<1, WTgtFunnel10>(adapt, warm, prm, st);
  if (kc.p0 == 1) {
    switch (kc.p1) {
      case T_CORRGAUSS: return go<1, WTgtPair<TgtCorrGauss>>(adapt, warm, prm, st);
      case T_SMILE: return go<1, WTgtPair<TgtSmile>>(adapt, warm, prm, st);
      case T_FUNNEL1: return go<1, WTgtPair<TgtFunnel1>>(adapt, warm, prm, st);
      case T_MODFUNNEL: return go<1, WTgtPair<TgtModFunnel>>(adapt, warm, prm, st);
    }
  }
  // (four waves per chain were tried for StockWatson: 4.43 -> 4.08 us per tick at 256 chains, 2.2x slower
  //  at 1024 -- the tick is scalar bookkeeping and f64 exp, not vector work; not kept)
  if (kc.p1 == T_STOCKWATSON && kc.p0 == 12) return go<12, WTgtStockWatson<12>>(adapt, warm, prm, st);
  return hipErrorInvalidValue;
}

}  // namespace wn
