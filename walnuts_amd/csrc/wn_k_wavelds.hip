// wn_k_wavelds.hip -- dense Gaussian of 2 .. 128 dimensions with an ADAPTIVE integrator (adaptLeapFrogD / R2P / YoshidaD / FlowD, with or
// without warm-up, traced runs, orbit statistics, wn_integrator_step): one chain per wavefront, the precision matrix in LDS, shared by
// the K chains of a workgroup (round 5; the reference's adaptive integrators on the headline geometry,
// adaptiveIntegrators.py:65-137 / 246-356 / 361-475 with lpFun = the dense Gaussian of SURVEY 8d cfg 2).
//
// Why not the MFMA layouts.  The role-split kernel (wn_k_fast4.hip) is fixedLeapFrog's: its leaf is one tick of a group of four chains,
// and its register file is full.  The cooperative layout (wn_k_coop.hip: 16 chains x 4 waves in lockstep, v_mfma_f64_16x16x4) does run
// the adaptive integrators -- at 2.8e8 evaluations/s on 4096 chains of the 100-D problem, an eighth of what its MFMAs could do: one
// workgroup per CU, one wave per SIMD, a tick machine that waits at two barriers per gradient while the chains of a group sit in
// different passes of their searches.  An adaptive macro step is per-chain control flow; what it needs from the hardware is many
// independent instruction streams, not a wide tile.
//
// So: the one-stream engine of the wave layout (wn_chain.h Engine<WavePolicy<2, ..>>: every integrator, warm-up, traces, orbit
// statistics -- nothing new to prove but the gradient), element i of a vector in lane i & 63 of register i >> 6, and
//     g = -A q  as  sum_j (-A)[:, j] q_j :  q_j is a v_readlane (the loop index is uniform), column j of -A two 512-byte LDS reads (rows
//     0..63 and 64..127, conflict-free), two FMAs: 6 instructions per column, ~600 per gradient at D = 100.
// -A is staged ONCE per workgroup, column-major with 128 doubles per column (rows and columns >= D are zero: the lanes that hold no
// element, and the last block of eight columns, need no mask): 128 KB, one workgroup of K = 8 chains per CU, two waves per SIMD.  Neither the
// LDS array (1.6 k cycles per gradient and SIMD) nor the vector ALU (1.8 k) is saturated: DESIGN 4.1b has the measured cost and what was tried.
#include "wn_launch.h"
#include "wn_policy_wave.h"
#include "wn_target_denselds.h"

namespace wn {

template <int NE, int K, bool ADAPT, bool WARM, bool FLOW, int SPEC>
__global__ __launch_bounds__(64 * K) void wn_wavelds_kernel(Params prm) {
  using Tgt = WTgtDenseLDS<NE>;
  using P = WavePolicy<NE, Tgt, 1, false>;
  __shared__ double wn_a_lds[Tgt::STAGE_DOUBLES];
  Tgt::stage(prm, wn_a_lds, (int)threadIdx.x, 64 * K);
  __syncthreads();
  const int chain = blockIdx.x * K + (int)(threadIdx.x >> 6);
  if (chain >= prm.C) return;
  P p(prm, chain, wn_a_lds, nullptr);
  Engine<P, ADAPT, WARM, FLOW, SPEC> eng(p, prm);
  if constexpr (SPEC >= 0) eng.run_nested();  // one chain per wave: nested loops instead of the tick loop
  else eng.run();
}

bool wavelds_geom(const KernelChoice& kc) { return kc.layout == 2 && kc.p0 == 2 && kc.p1 == T_DENSEGAUSS && kc.W == 1 && kc.G == 1; }

hipError_t launch_wavelds(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st) {
  (void)kc;
  constexpr int K = 8;
  dim3 grid((prm.C + K - 1) / K), block(64 * K);
  if (adapt == 0) return hipErrorInvalidValue;  // (fixedLeapFrog on this geometry is the role-split kernels')
  if (adapt == 1 && !warm && prm.mode == MODE_SAMPLE && !prm.trace_n && !prm.omin) {
    if (prm.integrator == I_ADAPT_R2P) { hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, false, false, I_ADAPT_R2P>), grid, block, 0, st, prm); return hipGetLastError(); }
    if (prm.integrator == I_ADAPT_D) { hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, false, false, I_ADAPT_D>), grid, block, 0, st, prm); return hipGetLastError(); }
  }
  if (adapt == 2) {  // adaptLeapFrogFlowD
    if (warm) hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, true, true, -1>), grid, block, 0, st, prm);
    else hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, false, true, -1>), grid, block, 0, st, prm);
    return hipGetLastError();
  }
  if (warm) hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, true, false, -1>), grid, block, 0, st, prm);
  else hipLaunchKernelGGL((wn_wavelds_kernel<2, K, true, false, false, -1>), grid, block, 0, st, prm);
  return hipGetLastError();
}

}  // namespace wn
