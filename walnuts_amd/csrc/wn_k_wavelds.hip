// wn_k_wavelds.hip -- dense Gaussian of 65 .. 128 dimensions with an ADAPTIVE integrator (adaptLeapFrogD / R2P / YoshidaD / FlowD, with or
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
// element, and the last block of eight columns, need no mask): 128 KB, one workgroup of K = 8 chains per CU, two waves per SIMD.  The bound is the vector ALU (400 of the 600
// instructions), not LDS bandwidth (12.8 KB per column pair and wave).
#include "wn_launch.h"
#include "wn_policy_wave.h"

namespace wn {

template <int NE>
struct WTgtDenseLDS {
  static_assert(NE == 2, "65 <= D <= 128: two registers per lane");
  static constexpr int EXT = EXT_NONE;
  static constexpr int LDS_DOUBLES = 0;  // (the matrix is the kernel's own shared array, handed to init() as `lds`)
  static constexpr int VEC_LDS = 0;
  static constexpr bool SPEC_ENGINE = true;  // plain sampling runs of the leapfrog integrators get the nested driver (wn_chain.h run_nested)
  typedef __attribute__((address_space(3))) double lds_double;
  const lds_double* A;  // (-A)[row][col] at A[col * 128 + row]
  int D;
  __device__ __forceinline__ void init(const Params& prm, int l0, double* lds) {
    A = (const lds_double*)lds + (l0 & 63);
    D = prm.D;
  }
  // grad = -A q, lp = 1/2 q.grad (targetDistr.py: the Gaussian's log density up to its constant); returns this lane's share of q.grad
  __device__ __forceinline__ double grad(const double* q, double* g, int) const {
    // eight columns at a time: their 16 LDS reads in flight together, then the 8 broadcasts and the 16 FMAs (a column at a time the loop
    // waited for the LDS and for the v_readlane -> SGPR -> VALU hazard at every column: 10 k cycles per gradient; columns D .. 127 of the
    // staged matrix are zero, so the last block needs no mask).  Two accumulators per row half: the FMA chain is not the critical path.
    double g0a = 0.0, g0b = 0.0, g1a = 0.0, g1b = 0.0;
    const int dpad = (D + 7) & ~7;
#pragma unroll 1
    for (int j0 = 0; j0 < dpad; j0 += 8) {
      double a0[8], a1[8], qj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0[u] = A[(j0 + u) * 128]; a1[u] = A[(j0 + u) * 128 + 64]; }
      const double qs = j0 < 64 ? q[0] : q[1];  // (uniform: a block of eight columns lies in one register)
#pragma unroll
      for (int u = 0; u < 8; ++u) qj[u] = wn_readlane(qs, (j0 + u) & 63);
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        g0a = fma(a0[u], qj[u], g0a); g1a = fma(a1[u], qj[u], g1a);
        g0b = fma(a0[u + 1], qj[u + 1], g0b); g1b = fma(a1[u + 1], qj[u + 1], g1b);
      }
    }
    const double g0 = g0a + g0b, g1 = g1a + g1b;
    g[0] = g0; g[1] = g1;
    return q[0] * g0 + q[1] * g1;
  }
  __device__ __forceinline__ static double lp_finish(double r) { return 0.5 * r; }
};

template <int NE, int K, bool ADAPT, bool WARM, bool FLOW, int SPEC>
__global__ __launch_bounds__(64 * K) void wn_wavelds_kernel(Params prm) {
  using Tgt = WTgtDenseLDS<NE>;
  using P = WavePolicy<NE, Tgt, 1, false>;
  __shared__ double wn_a_lds[128 * 128];
  {  // -A, column-major, 128 rows per column (zero beyond D)
    const int D = prm.D;
    for (int idx = threadIdx.x; idx < 128 * 128; idx += 64 * K) {
      const int j = idx >> 7, l = idx & 127;
      wn_a_lds[idx] = (l < D && j < D) ? -prm.tpar[(size_t)l * D + j] : 0.0;
    }
  }
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
