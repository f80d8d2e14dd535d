// wn_k_split.hip -- split-mode kernels of the one-chain-per-wavefront layout: the chain state
// machine advances by one tick per launch (Engine::split_tick) and the gradient is evaluated between
// launches, either by the batched FP64-MFMA kernel below (dense Gaussian, 128 < D <= 1024: the
// precision matrix cannot stay on chip, SURVEY.md 8d cfg 4) or by the caller (wn_ext_* entry points:
// the reference's lpFun plugin as a batched device callable).
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

template <int NE, int EXT, bool ADAPT, bool WARM>
__global__ __launch_bounds__(64) void wn_split_begin_kernel(Params prm) {
  using P = WavePolicy<NE, WTgtExt<NE, EXT>>;
  P p(prm, blockIdx.x);
  Engine<P, ADAPT, WARM> eng(p, prm);
  eng.split_begin();
}

template <int NE, int EXT, bool ADAPT, bool WARM>
__global__ __launch_bounds__(64) void wn_split_tick_kernel(Params prm) {
  using P = WavePolicy<NE, WTgtExt<NE, EXT>>;
  P p(prm, blockIdx.x);
  Engine<P, ADAPT, WARM> eng(p, prm);
  const bool was_done = p.sc->sp_stage == 2;
  const bool more = eng.split_tick();
  if (!more && !was_done && (threadIdx.x & 63) == 0) atomicAdd(prm.sp_done, 1);
}

// grad = -A q for every chain: xg[c][d] = -sum_k A[d][k] xq[c][k]   (A symmetric, zero-padded to
// DP x DP; xq, xg are [Cpad][DP] with zero padding).  v_mfma_f64_16x16x4_f64 with M = output index d,
// N = chain, K = k.  Both operands are read "along k": lane l loads 32 contiguous bytes
// row (l%16), k0 + 4*(l/16) .. +3, which serve four k-steps (the sum over k does not care which k
// sits in which MFMA slot as long as A and B agree).  Workgroup tile 64 x 64, wave tile 32 x 32.
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wn_dense_grad_kernel(const double* __restrict__ A, const double* __restrict__ xq,
                                                            double* __restrict__ xg, int DP, int Kr, int tiles_d,
                                                            int tiles_c, const int* done, int C) {
  if (*done >= C) return;  // every chain has finished: nothing reads the result
  const int T = tiles_d * tiles_c, per = (T + 7) / 8;
  // consecutive blocks go to different XCDs: give each XCD a contiguous range of tiles (same chains,
  // neighbouring rows of A) so its L2 sees each operand row once
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int ct = t / tiles_d, dt = t % tiles_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d0 = dt * 64 + (wave & 1) * 32, c0 = ct * 64 + (wave >> 1) * 32;
  const int r = lane & 15, kq = 4 * (lane >> 4);
  const double* a0 = A + (size_t)(d0 + r) * DP + kq;
  const double* a1 = a0 + (size_t)16 * DP;
  const double* b0 = xq + (size_t)(c0 + r) * DP + kq;
  const double* b1 = b0 + (size_t)16 * DP;
  d4_t acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
  d4_t fa0 = *(const d4_t*)a0, fa1 = *(const d4_t*)a1, fb0 = *(const d4_t*)b0, fb1 = *(const d4_t*)b1;
  for (int k0 = 0; k0 < Kr; k0 += 16) {
    const int kn = k0 + 16 < Kr ? k0 + 16 : k0;  // prefetch the next k-block while this one multiplies
    const d4_t na0 = *(const d4_t*)(a0 + kn), na1 = *(const d4_t*)(a1 + kn);
    const d4_t nb0 = *(const d4_t*)(b0 + kn), nb1 = *(const d4_t*)(b1 + kn);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[s], fb0[s], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[s], fb1[s], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[s], fb0[s], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[s], fb1[s], acc11, 0, 0, 0);
    }
    fa0 = na0; fa1 = na1; fb0 = nb0; fb1 = nb1;
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: register r of lane l holds row 4*r + l/16 (output index),
  // column l%16 (chain)
  double* o0 = xg + (size_t)(c0 + r) * DP + d0 + (lane >> 4);
  double* o1 = o0 + (size_t)16 * DP;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o0[4 * j] = -acc00[j];
    o1[4 * j] = -acc01[j];
    o0[16 + 4 * j] = -acc10[j];
    o1[16 + 4 * j] = -acc11[j];
  }
}

template <int NE, int EXT>
static hipError_t go_begin(bool adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64);
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
static hipError_t go_tick(bool adapt, bool warm, const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64);
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
hipError_t launch_split_begin(const KernelChoice& kc, bool adapt, bool warm, const Params& prm, hipStream_t st) {
  if (kc.p1 == EXT_HALF_QG) {
    if (kc.p0 == 8) return go_begin<8, EXT_HALF_QG>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_begin<16, EXT_HALF_QG>(adapt, warm, prm, st);
  } else {
    if (kc.p0 == 8) return go_begin<8, EXT_LP>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_begin<16, EXT_LP>(adapt, warm, prm, st);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_split_tick(const KernelChoice& kc, bool adapt, bool warm, const Params& prm, hipStream_t st) {
  if (kc.p1 == EXT_HALF_QG) {
    if (kc.p0 == 8) return go_tick<8, EXT_HALF_QG>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_tick<16, EXT_HALF_QG>(adapt, warm, prm, st);
  } else {
    if (kc.p0 == 8) return go_tick<8, EXT_LP>(adapt, warm, prm, st);
    if (kc.p0 == 16) return go_tick<16, EXT_LP>(adapt, warm, prm, st);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_dense_grad(const double* A, const double* xq, double* xg, int DP, int D, int Cpad, const int* done, int C,
                             hipStream_t st) {
  const int tiles_d = DP / 64, tiles_c = Cpad / 64, T = tiles_d * tiles_c, per = (T + 7) / 8;
  const int Kr = (D + 15) / 16 * 16;
  hipLaunchKernelGGL(wn_dense_grad_kernel, dim3(8 * per), dim3(256), 0, st, A, xq, xg, DP, Kr, tiles_d, tiles_c, done, C);
  return hipGetLastError();
}

}  // namespace wn
