// wn_target_denselds.h -- the dense Gaussian as a target of the one-chain-per-wavefront layout, its matrix in LDS (wn_k_wavelds.hip: the
// one-stream engine; wn_k_orbit.hip: the trajectory-first kernel's wide mode).  See wn_k_wavelds.hip for the why.
#pragma once
#include "wn_policy_wave.h"

namespace wn {

template <int NE>
struct WTgtDenseLDS {
  static_assert(NE == 2, "D <= 128: two registers per lane (the second one empty up to D = 64)");
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
    if (D <= 64) {  // (uniform) rows 64 .. 127 are empty: half the LDS reads, which are what bounds this loop
#pragma unroll 1
      for (int j0 = 0; j0 < dpad; j0 += 8) {
        double a0[8], qj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = A[(j0 + u) * 128];
#pragma unroll
        for (int u = 0; u < 8; ++u) qj[u] = wn_readlane(q[0], j0 + u);
#pragma unroll
        for (int u = 0; u < 8; u += 2) { g0a = fma(a0[u], qj[u], g0a); g0b = fma(a0[u + 1], qj[u + 1], g0b); }
      }
      const double g0 = g0a + g0b;
      g[0] = g0; g[1] = 0.0;
      return q[0] * g0;
    }
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
  // -A into the workgroup's shared array, column-major, 128 rows per column, zero beyond D (every thread of the workgroup, then a barrier)
  static constexpr int STAGE_DOUBLES = 128 * 128;
  static constexpr bool BATCH_ADAPT = true;  // (wn_k_orbit.hip: an adaptive macro step is an evaluation or two here: batches of 16 leaves pay)
  __device__ __forceinline__ static void stage(const Params& prm, double* lds, int tid, int nthreads) {
    const int D = prm.D;
    for (int idx = tid; idx < 128 * 128; idx += nthreads) {
      const int j = idx >> 7, l = idx & 127;
      lds[idx] = (l < D && j < D) ? -prm.tpar[(size_t)l * D + j] : 0.0;
    }
  }
};



// std / diagonal Gaussian of at most 16 dimensions in the trajectory-first kernel's one-register form (lane i < D of every row of 16 holds
// element i: WTgtDiag's arithmetic, lp = 1/2 q.grad)
struct WTgtDiagRow {
  static constexpr int EXT = EXT_NONE;
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 16;
  double a;
  __device__ __forceinline__ void init(const Params& prm, int lane, double*) { a = (lane & 15) < prm.D ? prm.tpar[lane & 15] : 0.0; }
  __device__ __forceinline__ double grad(const double* q, double* g, int) const {
    g[0] = -(a * q[0]);
    return q[0] * g[0];
  }
  __device__ __forceinline__ static double lp_finish(double r) { return 0.5 * r; }
};

}  // namespace wn
