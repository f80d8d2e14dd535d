// wn_k_split.hip -- split-mode kernels of the one-chain-per-wavefront layout: the chain state
// machine advances by one tick per launch (Engine::split_tick) and the gradient is evaluated between
// launches, either by the batched FP64-MFMA kernel below (dense Gaussian, 128 < D <= 1024: the
// precision matrix cannot stay on chip, SURVEY.md 8d cfg 4) or by the caller (wn_ext_* entry points:
// the reference's lpFun plugin as a batched device callable).
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

// DPW = DP / 64 (8 or 16).  Adaptive integrators and warm-up: a workgroup of two waves serves a chain, DPW/2 registers
// per vector (their engines hold ten-odd vectors).  fixedLeapFrog: one wave per chain (no cross-wave reduction, no
// barrier) and FOUR chains per workgroup -- with one chain per workgroup the dispatcher needed 10.7 us to start the
// 1024 workgroups of a tick whose waves each run for 8 us.
template <bool ADAPT, bool WARM> struct SplitShape { static constexpr int SW = (ADAPT || WARM) ? 2 : 1, CPW = (ADAPT || WARM) ? 1 : 4, THREADS = 64 * SW * CPW; };

template <int DPW, int EXT, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__((SplitShape<ADAPT, WARM>::THREADS)) void wn_split_begin_kernel(Params prm) {
  constexpr int SW = SplitShape<ADAPT, WARM>::SW, CPW = SplitShape<ADAPT, WARM>::CPW;
  using P = WavePolicy<DPW / SW, WTgtExt<DPW / SW, EXT>, SW>;
  __shared__ double lds[P::LDS_DOUBLES > 0 ? P::LDS_DOUBLES : 1];
  const int chain = CPW > 1 ? (int)(blockIdx.x * CPW + (threadIdx.x >> 6)) : (int)blockIdx.x;
  if (CPW > 1 && chain >= prm.C) return;  // (one wave per chain: no barrier behind this)
  P p(prm, chain, lds);
  Engine<P, ADAPT, WARM, FLOW> eng(p, prm);
  eng.split_begin();
}

template <int DPW, int EXT, bool ADAPT, bool WARM, bool FLOW = false>
__global__ __launch_bounds__((SplitShape<ADAPT, WARM>::THREADS)) void wn_split_tick_kernel(Params prm) {
  constexpr int SW = SplitShape<ADAPT, WARM>::SW, CPW = SplitShape<ADAPT, WARM>::CPW;
  using P = WavePolicy<DPW / SW, WTgtExt<DPW / SW, EXT>, SW>;
  __shared__ double lds[P::LDS_DOUBLES > 0 ? P::LDS_DOUBLES : 1];
  const int chain = CPW > 1 ? (int)(blockIdx.x * CPW + (threadIdx.x >> 6)) : (int)blockIdx.x;
  if (CPW > 1 && chain >= prm.C) return;
  P p(prm, chain, lds);
  Engine<P, ADAPT, WARM, FLOW> eng(p, prm);
#ifdef WN_SPAN  // diagnostic build: wall-clock start / end of every workgroup of launch 100
  const unsigned long long span0 = __builtin_amdgcn_s_memrealtime();
#endif
  const bool was_done = p.sc->sp_stage == 2;
  const bool more = eng.split_tick();
  if (prm.xzero) p.ext_zero_g();
  const bool first_lane = CPW > 1 ? (threadIdx.x & 63) == 0 : threadIdx.x == 0;
  if (!more && !was_done && first_lane) atomicAdd(prm.sp_done, 1);
#ifdef WN_SPAN
  if (prm.stamps && first_lane) {
    unsigned long long* o = prm.stamps + (size_t)chain * 16;
    if (o[6]++ == 100) { o[13] = span0; o[14] = __builtin_amdgcn_s_memrealtime(); }
  }
#endif
}

#define WN_SPLIT_GO(KERNEL, A_, W_, ...)                                                                      \
  {                                                                                                           \
    constexpr int SW_ = SplitShape<A_, W_>::SW, CPW_ = SplitShape<A_, W_>::CPW;                               \
    hipLaunchKernelGGL((KERNEL<NE, EXT, A_, W_, ##__VA_ARGS__>), dim3((prm.C + CPW_ - 1) / CPW_), dim3(64 * SW_ * CPW_), 0, st, prm); \
  }
template <int NE, int EXT>
static hipError_t go_begin(int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) WN_SPLIT_GO(wn_split_begin_kernel, true, true, true)
    else WN_SPLIT_GO(wn_split_begin_kernel, true, false, true)
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) WN_SPLIT_GO(wn_split_begin_kernel, true, true)
    else WN_SPLIT_GO(wn_split_begin_kernel, false, true)
  } else {
    if (adapt) WN_SPLIT_GO(wn_split_begin_kernel, true, false)
    else WN_SPLIT_GO(wn_split_begin_kernel, false, false)
  }
  return hipGetLastError();
}
template <int NE, int EXT>
static hipError_t go_tick(int adapt, bool warm, const Params& prm, hipStream_t st) {
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) WN_SPLIT_GO(wn_split_tick_kernel, true, true, true)
    else WN_SPLIT_GO(wn_split_tick_kernel, true, false, true)
    return hipGetLastError();
  }
  if (warm) {
    if (adapt) WN_SPLIT_GO(wn_split_tick_kernel, true, true)
    else WN_SPLIT_GO(wn_split_tick_kernel, false, true)
  } else {
    if (adapt) WN_SPLIT_GO(wn_split_tick_kernel, true, false)
    else WN_SPLIT_GO(wn_split_tick_kernel, false, false)
  }
  return hipGetLastError();
}
#undef WN_SPLIT_GO

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
