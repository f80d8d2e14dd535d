// wn_policy_lane.h -- layout policy "one chain per lane": every D-vector of a chain lives in the
// registers of one lane (D <= DMAX, small), so dot products are lane-local loops in the
// reference's own left-to-right order and no reduction or barrier exists.  Sub-tree endpoint
// slots and the cold vectors (passive orbit end, last proposal) are in global memory laid out
// [vector][element][chain] so that the 64 lanes of a wave touch 512 contiguous bytes.
#pragma once
#include "wn_chain.h"

namespace wn {

struct EmptyVec {};

template <int DMAX, class Tgt>
struct LanePolicy {
  static constexpr int NE = DMAX;
  static constexpr int ALIGN_LOG2 = WN_ALIGN_LOG2;  // iteration-start alignment (wn_chain.h run())
  static constexpr bool LANE_RNG = false;           // uniforms come from the engine's own window (wn_chain.h rng_refill)
  static constexpr int NLC = 0;  // no fast cache: every sub-tree left endpoint lives in the slot store
  struct Vec {
    double x[DMAX];
  };
  template <bool B>
  using Opt = typename std::conditional<B, Vec, EmptyVec>::type;

  // warm vectors: LDS on the device ([vector][element][lane], conflict-free), an array on the host
  static constexpr int WARM_DOUBLES = WARM_N * DMAX * 64;
#if defined(__HIP_DEVICE_COMPILE__)
  double* wbase;  // lds + lane
#else
  double wstore[WARM_N * DMAX];
#endif
  int D, cl;
  uint32_t cg;
  size_t cstride;  // padded chain count
  double* work;
  const double* tpar;
  int nslot;

  // workspace doubles needed for C chains
  static size_t work_doubles(int C, int M) {
    size_t cpad = ((size_t)C + 63) / 64 * 64;
    return ((size_t)(M + 1) * 2 + COLD_N + 4) * DMAX * cpad;
  }

  WN_HD LanePolicy(const Params& prm, int chain_local, double* lds = nullptr)
      :
#if defined(__HIP_DEVICE_COMPILE__)
        wbase(lds + (threadIdx.x & 63)),
#endif
        D(prm.D), cl(chain_local), cg(prm.chain0 + (uint32_t)chain_local),
        cstride(((size_t)prm.C + 63) / 64 * 64), work(prm.work), tpar(prm.tpar), nslot(prm.M + 1) {}

  WN_HD bool valid(int e) const { return e < D; }
  WN_HD bool owner(int) const { return true; }
  WN_HD int gidx(int e) const { return e; }
  WN_HD bool writer() const { return true; }
  WN_HD bool wg_writer() const { return true; }
  WN_HD int facs_slot() const { return cl; }
  WN_HD int chain_local() const { return cl; }
  WN_HD uint32_t chain_global() const { return cg; }
  WN_HD bool any(bool b) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(b);
#else
    return b;
#endif
  }
  // at least half of the lanes in `of` satisfy `a` (and at least one does)
  WN_HD bool majority(bool a, bool of) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const int na = __popcll(__ballot(a)), no = __popcll(__ballot(of));
    return na > 0 && 2 * na >= no;
#else
    return a && of;
#endif
  }
  template <int N>
  WN_HD void reduce(double*) const {}
  template <int N>
  WN_HD void reduce_max(double*) const {}
  // (StockWatson's param_constrain for the orbit statistics: the one-chain-per-wavefront layout only)
  template <class F> WN_HD void sw_constrained(const Vec&, F) const {}

  WN_HD void grad(const Vec& q, Vec& g, double& lp) const { lp = Tgt::eval(q.x, g.x, D, tpar); }
  WN_HD double lp_finish(double lp) const { return lp; }

  WN_HD double* vec_ptr(int v, int e) const { return work + ((size_t)v * DMAX + e) * cstride + cl; }
  WN_HD void slot_store(int slot, const Vec& q, const Vec& v) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e)
      if (e < D) { *vec_ptr(2 * slot, e) = q.x[e]; *vec_ptr(2 * slot + 1, e) = v.x[e]; }
  }
  WN_HD void slot_load(int slot, Vec& q, Vec& v) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
      q.x[e] = e < D ? *vec_ptr(2 * slot, e) : 0.0;
      v.x[e] = e < D ? *vec_ptr(2 * slot + 1, e) : 0.0;
    }
  }
  WN_HD void diag_store(double* row, const double* d) const {
#pragma unroll
    for (int k = 0; k < NDIAG; ++k) row[k] = d[k];
  }
  // proposal buffers which = 0/1: (q, grad)
  WN_HD void prop_put(int which, const Vec& q, const Vec& g) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e)
      if (e < D) {
        *vec_ptr(2 * nslot + COLD_N + 2 * which, e) = q.x[e];
        *vec_ptr(2 * nslot + COLD_N + 2 * which + 1, e) = g.x[e];
      }
  }
  WN_HD void prop_get(int which, Vec& q, Vec& g) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
      q.x[e] = e < D ? *vec_ptr(2 * nslot + COLD_N + 2 * which, e) : 0.0;
      g.x[e] = e < D ? *vec_ptr(2 * nslot + COLD_N + 2 * which + 1, e) : 0.0;
    }
  }
  template <int K>
  WN_HD void warm_put(const Vec& x) {
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
#if defined(__HIP_DEVICE_COMPILE__)
      wbase[(K * DMAX + e) * 64] = x.x[e];
#else
      wstore[K * DMAX + e] = x.x[e];
#endif
    }
  }
  template <int K>
  WN_HD void warm_get(Vec& x) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
#if defined(__HIP_DEVICE_COMPILE__)
      x.x[e] = wbase[(K * DMAX + e) * 64];
#else
      x.x[e] = wstore[K * DMAX + e];
#endif
    }
  }
  WN_HD void lslot_store(int, const Vec&, const Vec&) const {}
  WN_HD void lslot_load(int, Vec&, Vec&) const {}
  template <int K>
  WN_HD void cold_put(const Vec& x) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e)
      if (e < D) *vec_ptr(2 * nslot + K, e) = x.x[e];
  }
  template <int K>
  WN_HD void cold_get(Vec& x) const {
#pragma unroll
    for (int e = 0; e < DMAX; ++e) x.x[e] = e < D ? *vec_ptr(2 * nslot + K, e) : 0.0;
  }
};

}  // namespace wn
