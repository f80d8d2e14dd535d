// wn_k_split.hip -- split-mode kernels of the one-chain-per-wavefront layout: the chain state
// machine advances by one tick per launch (Engine::split_tick) and the gradient is evaluated between
// launches, either by the batched FP64-MFMA kernel below (dense Gaussian, 128 < D <= 1024: the
// precision matrix cannot stay on chip, SURVEY.md 8d cfg 4) or by the caller (wn_ext_* entry points:
// the reference's lpFun plugin as a batched device callable).
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

// DPW = DP / 64 (8 or 16); every chain is served by a workgroup of SPLIT_W waves, DPW/SPLIT_W registers per vector
constexpr int SPLIT_W = 2;

template <int DPW, int EXT, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__(64 * SPLIT_W) void wn_split_begin_kernel(Params prm) {
  using P = WavePolicy<DPW / SPLIT_W, WTgtExt<DPW / SPLIT_W, EXT>, SPLIT_W>;
  __shared__ double lds[P::LDS_DOUBLES > 0 ? P::LDS_DOUBLES : 1];
  P p(prm, blockIdx.x, lds);
  Engine<P, ADAPT, WARM, FLOW> eng(p, prm);
  eng.split_begin();
}

template <int DPW, int EXT, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__(64 * SPLIT_W) void wn_split_tick_kernel(Params prm) {
  using P = WavePolicy<DPW / SPLIT_W, WTgtExt<DPW / SPLIT_W, EXT>, SPLIT_W>;
  __shared__ double lds[P::LDS_DOUBLES > 0 ? P::LDS_DOUBLES : 1];
  P p(prm, blockIdx.x, lds);
  Engine<P, ADAPT, WARM, FLOW> eng(p, prm);
#ifdef WN_SPAN  // diagnostic build: wall-clock start / end of every workgroup of launch 100
  const unsigned long long span0 = __builtin_amdgcn_s_memrealtime();
#endif
  const bool was_done = p.sc->sp_stage == 2;
  const bool more = eng.split_tick();
  if (!more && !was_done && threadIdx.x == 0) atomicAdd(prm.sp_done, 1);
#ifdef WN_SPAN
  if (prm.stamps && threadIdx.x == 0) {
    unsigned long long* o = prm.stamps + (size_t)blockIdx.x * 16;
    if (o[6]++ == 100) { o[13] = span0; o[14] = __builtin_amdgcn_s_memrealtime(); }
  }
#endif
}

template <int NE, int EXT>
static hipError_t go_begin(int adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64 * SPLIT_W);
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, true, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, true, false, true>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, false, true>), grid, block, 0, st, prm);
  } else {
    if (adapt) hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, true, false>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_begin_kernel<NE, EXT, false, false>), grid, block, 0, st, prm);
  }
  return hipGetLastError();
}
template <int NE, int EXT>
static hipError_t go_tick(int adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64 * SPLIT_W);
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, true, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, true, false, true>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, true, true>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, false, true>), grid, block, 0, st, prm);
  } else {
    if (adapt) hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, true, false>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_split_tick_kernel<NE, EXT, false, false>), grid, block, 0, st, prm);
  }
  return hipGetLastError();
}

size_t split_work_doubles(int NE, int C, int M) {
  return NE <= 8 ? WavePolicy<8, WTgtExt<8, EXT_LP>>::work_doubles(C, M) : WavePolicy<16, WTgtExt<16, EXT_LP>>::work_doubles(C, M);
}

// kc.p0 = NE (8 or 16), kc.p1 = EXT_HALF_QG (dense Gaussian through the GEMM) or EXT_LP (caller)
hipError_t launch_split_begin(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (kc.p1 == EXT_HALF_QG) {
    if (kc.p0 == 8) return go_begin<8, EXT_HALF_QG>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_begin<16, EXT_HALF_QG>(adapt, warm, prm, st);
  } else {
    if (kc.p0 == 8) return go_begin<8, EXT_LP>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_begin<16, EXT_LP>(adapt, warm, prm, st);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_split_tick(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (kc.p1 == EXT_HALF_QG) {
    if (kc.p0 == 8) return go_tick<8, EXT_HALF_QG>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_tick<16, EXT_HALF_QG>(adapt, warm, prm, st);
  } else {
    if (kc.p0 == 8) return go_tick<8, EXT_LP>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_tick<16, EXT_LP>(adapt, warm, prm, st);
  }
  return hipErrorInvalidValue;
}
}  // namespace wn
