// wn_policy_coop.h -- cooperative MFMA layout for Gaussian targets at medium dimension
// (dense precision matrix, D <= 64*NT), gfx950 only.
//
// One workgroup = one group of 16 chains x W wavefronts.  The 16 chains are the N dimension of
// v_mfma_f64_16x16x4_f64; the D coordinates are split into 16-row tiles, NT tiles per wave.
// Every D-vector (q, v, grad, ...) lives in registers in the MFMA C/D layout
//
//     lane l, register r of tile T  <->  coordinate 16*T + 4*r + (l>>4)  of chain (l & 15)
//
// which for the f64 MFMA is also exactly the B-operand layout of k-step s = 4*T + r (lane l
// holds B[k = l>>4][n = l&15]).  So the batched gradient  G[D x 16] = (-A)[D x D] . Q[D x 16]  needs
// no data reshuffle: each wave publishes its own 4*NT k-steps of Q to LDS (512 B contiguous per
// k-step), reads all ceil(D/4) k-steps back as B fragments, and multiplies by its rows of -A, which
// it keeps resident in registers as A fragments for the whole run (NT*ceil(D/4) doubles/lane).
// Dot products are reduced lane -> (xor 16, xor 32) -> across waves through LDS; the per-chain
// scalars are replicated in all 4*W lanes that serve a chain and stay bit-identical because every
// replica reads the same reduced sums.
//
// LDS per workgroup: Q exchange 4*NT*W*512 B (16 KiB at NT=2,W=4) + 2 x 8 x W x 16 x 8 B reduction
// buffers.  HBM traffic is only the sub-tree endpoint slots ([slot][q|v][wave][reg][lane], 512 B
// coalesced per register) -- see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

#include "wn_chain.h"
#include "wn_policy_lane.h"  // EmptyVec

namespace wn {

typedef double d4_t __attribute__((ext_vector_type(4)));

enum { CK_DENSE = 0, CK_DIAG = 1 };

template <int NT, int W, int KS, int TK>
struct CoopPolicy {
  static constexpr int NE = NT * 4;
  static constexpr int KS_MAX = 4 * NT * W;  // k-steps of the padded dimension 16*NT*W
  static constexpr int DPAD = 16 * NT * W;
  // sub-tree levels 2..1+NLC keep their left endpoint (q, v) in LDS: [level][q|v][wave][reg][lane]
#ifndef WN_COOP_NLC
#define WN_COOP_NLC 3
#endif
  static constexpr int NLC = NT <= 2 ? WN_COOP_NLC : 0;
  static constexpr int ALIGN_LOG2 = WN_ALIGN_LOG2;  // iteration-start alignment (wn_chain.h run())
  static constexpr bool LANE_RNG = false;           // uniforms come from the engine's own window (wn_chain.h rng_refill)
  static constexpr int NRED = 4 + 2 * (2 + NLC);
  static constexpr int LDS_Q = KS_MAX * 64;
  static constexpr int LDS_RED = 2 * NRED * W * 16;
  static constexpr int LDS_CACHE = NLC * 2 * W * NE * 64;
  static constexpr int LDS_COLD = 2 * W * NE * 64;  // passive orbit end (q, v), read at every doubling end
  static constexpr int LDS_DOUBLES = LDS_Q + LDS_RED + LDS_CACHE + LDS_COLD;
  static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget (160 KiB per CU)");
  static_assert(KS <= KS_MAX, "k-steps exceed the padded dimension");
  static constexpr int NA = (TK == CK_DENSE) ? NT * KS : 1;
  struct Vec {
    double x[NE];
  };
  template <bool B>
  using Opt = typename std::conditional<B, Vec, EmptyVec>::type;

  int D, lane, wave, cn, lg, cl, ksteps;
  uint32_t cg;
  double* ldsq;
  double* ldsr;
  double* ldsc;
  double* slotbase;
  int redpar;
  double afrag[NA];  // -A fragments: afrag[t*KS+s] = -A[16*T+(l&15)][4*s+(l>>4)]
  double adiag[NE];  // diagonal target: a[gidx(e)]
  Vec warm[WARM_N];  // leaf-pair state / adaptive forward result: registers (unused ones vanish)
  double* ldscold;   // QB, VB in LDS: [q|v][wave][reg][lane]
  double* coldbase;  // GB and the proposal set QP/GP/QPL/GPL: in memory, [vec][wave][reg][lane]

  static constexpr size_t GROUP_DOUBLES(int M) { return (size_t)((M + 1) * 2 + 5) * W * NE * 64; }
  static size_t work_doubles(int C, int M) {
    size_t groups = ((size_t)C + 15) / 16;
    return groups * GROUP_DOUBLES(M);
  }

  __device__ __forceinline__ CoopPolicy(const Params& prm, double* lds) {
    D = prm.D;
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    cn = lane & 15;
    lg = lane >> 4;
    cl = blockIdx.x * 16 + cn;
    cg = prm.chain0 + (uint32_t)cl;
    ksteps = (D + 3) >> 2;
    ldsq = lds;
    ldsr = lds + LDS_Q;
    ldsc = lds + LDS_Q + LDS_RED;
    ldscold = ldsc + LDS_CACHE;
    slotbase = prm.work + (size_t)blockIdx.x * GROUP_DOUBLES(prm.M);
    coldbase = slotbase + (size_t)(prm.M + 1) * 2 * W * NE * 64;
    redpar = 0;
    if (TK == CK_DENSE) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = 16 * (wave * NT + t) + (lane & 15);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int col = 4 * s + (lane >> 4);
          afrag[t * KS + s] = (row < D && col < D) ? -prm.tpar[(size_t)row * D + col] : 0.0;
        }
      }
    } else {
      afrag[0] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) adiag[e] = (TK == CK_DIAG && valid(e)) ? prm.tpar[gidx(e)] : 0.0;
  }

  __device__ __forceinline__ int gidx(int e) const {
    return 16 * (wave * NT + (e >> 2)) + 4 * (e & 3) + lg;
  }
  __device__ __forceinline__ bool valid(int e) const { return gidx(e) < D; }
  __device__ __forceinline__ bool owner(int) const { return true; }
  __device__ __forceinline__ bool writer() const { return wave == 0 && lg == 0; }
  __device__ __forceinline__ bool wg_writer() const { return wave == 0 && lg == 0; }
  __device__ __forceinline__ int facs_slot() const { return cl; }
  __device__ __forceinline__ int chain_local() const { return cl; }
  __device__ __forceinline__ uint32_t chain_global() const { return cg; }
  __device__ __forceinline__ bool any(bool b) const { return __any(b); }
  __device__ __forceinline__ bool majority(bool a, bool of) const {
    const int na = __popcll(__ballot(a)), no = __popcll(__ballot(of));
    return na > 0 && 2 * na >= no;
  }

  // x[l] + x[l^16] + x[l^32] + x[l^48] in every lane, on the VALU (no LDS round trip):
  // v_permlane16_swap exchanges odd rows of one operand with even rows of the other,
  // v_permlane32_swap the upper half of one with the lower half of the other.
  __device__ __forceinline__ static double rowsum4(double x) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    u2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    u2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double y = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(y); hi = __double2hiint(y);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }

  // sum partials over the 4*W lanes that serve a chain; every replica gets the same bits
  template <int N>
  __device__ __forceinline__ void reduce(double* v) {
    static_assert(N <= NRED, "reduce width");
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = rowsum4(v[k]);
    if (W > 1) {
      double* buf = ldsr + (redpar & 1) * (NRED * W * 16);
      redpar++;
      if (lg == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) buf[(k * W + wave) * 16 + cn] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) s += buf[(k * W + w) * 16 + cn];
        v[k] = s;
      }
    }
  }

  // (StockWatson's param_constrain for the orbit statistics: the one-chain-per-wavefront layout only)
  template <class F> __device__ __forceinline__ void sw_constrained(const Vec&, F) const {}
  // max over the 4*W lanes that serve a chain (NaN-free inputs; max is exact, so every replica agrees): the flow errors of
  // adaptLeapFrogFlowD (adaptiveIntegrators.py:276-287)
  __device__ __forceinline__ static double rowmax4(double x) {
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    u2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    u2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double y = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    lo = __double2loint(y); hi = __double2hiint(y);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  }
  template <int N>
  __device__ __forceinline__ void reduce_max(double* v) {
    static_assert(N <= NRED, "reduce width");
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = rowmax4(v[k]);
    if (W > 1) {
      double* buf = ldsr + (redpar & 1) * (NRED * W * 16);
      redpar++;
      if (lg == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) buf[(k * W + wave) * 16 + cn] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double s = buf[(k * W) * 16 + cn];
#pragma unroll
        for (int w = 1; w < W; ++w) s = fmax(s, buf[(k * W + w) * 16 + cn]);
        v[k] = s;
      }
    }
  }

  // grad = -A q for the 16 chains of the group; lp partial = sum_e q_e g_e (lp = 1/2 q.g)
  __device__ __forceinline__ void grad(const Vec& q, Vec& g, double& lp_part) {
    if (TK == CK_DENSE) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int s = 4 * (wave * NT + (e >> 2)) + (e & 3);
        ldsq[s * 64 + lane] = q.x[e];
      }
      __syncthreads();
      d4_t acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double bq = ldsq[s * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t * KS + s], bq, acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g.x[t * 4 + r] = acc[t][r];
      }
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) g.x[e] = -(adiag[e] * q.x[e]);
    }
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) s += q.x[e] * g.x[e];
    lp_part = s;
  }
  __device__ __forceinline__ double lp_finish(double r) const { return 0.5 * r; }

  // Memory-resident vectors are laid out [vector][chain][wave][reg][lane>>4]: the four replicas of a
  // chain write 32 contiguous bytes per register and 256 contiguous bytes (four whole lines) per
  // wave, so a store by ONE chain (a multinomial pick, an endpoint push) dirties whole lines only;
  // with [reg][lane] order the same store scattered 8-byte pieces over 32 lines (measured 672 B of
  // HBM writes per leapfrog step, profiles/r01).
  static constexpr int VEC_DOUBLES = 16 * W * NE * 4;  // one vector of the 16-chain group
  __device__ __forceinline__ double* gvec_ptr(double* base, int vec) const {
    return base + (size_t)vec * VEC_DOUBLES + ((size_t)(cn * W + wave) * NE) * 4 + lg;
  }
  __device__ __forceinline__ void slot_store(int slot, const Vec& q, const Vec& v) const {
    double* pq = gvec_ptr(slotbase, slot * 2);
    double* pv = gvec_ptr(slotbase, slot * 2 + 1);
#pragma unroll
    for (int e = 0; e < NE; ++e) { pq[e * 4] = q.x[e]; pv[e * 4] = v.x[e]; }
  }
  __device__ __forceinline__ void slot_load(int slot, Vec& q, Vec& v) const {
    const double* pq = gvec_ptr(slotbase, slot * 2);
    const double* pv = gvec_ptr(slotbase, slot * 2 + 1);
#pragma unroll
    for (int e = 0; e < NE; ++e) { q.x[e] = pq[e * 4]; v.x[e] = pv[e * 4]; }
  }
  __device__ __forceinline__ double* lslot_ptr(int lev, int which) const {
    return ldsc + ((size_t)((lev * 2 + which) * W + wave) * NE) * 64 + lane;
  }
  __device__ __forceinline__ void lslot_store(int lev, const Vec& q, const Vec& v) const {
    double* pq = lslot_ptr(lev, 0);
    double* pv = lslot_ptr(lev, 1);
#pragma unroll
    for (int e = 0; e < NE; ++e) { pq[e * 64] = q.x[e]; pv[e * 64] = v.x[e]; }
  }
  __device__ __forceinline__ void lslot_load(int lev, Vec& q, Vec& v) const {
    const double* pq = lslot_ptr(lev, 0);
    const double* pv = lslot_ptr(lev, 1);
#pragma unroll
    for (int e = 0; e < NE; ++e) { q.x[e] = pq[e * 64]; v.x[e] = pv[e * 64]; }
  }
  template <int K>
  __device__ __forceinline__ void warm_put(const Vec& x) { warm[K] = x; }
  template <int K>
  __device__ __forceinline__ void warm_get(Vec& x) const { x = warm[K]; }
  template <int K>
  __device__ __forceinline__ double* cold_ptr() const {
    return ldscold + ((size_t)(K * W + wave) * NE) * 64 + lane;  // COLD_QB / COLD_VB
  }
  // one diagnostics row = 24 doubles = three 64-B lines.  Partial-line stores to 16 different rows
  // serialise in the memory system (~17 us per row set measured), so the four replicas of a chain
  // in wave 0 each write 16 B of the same line in ONE instruction: three full-line stores per row.
  __device__ __forceinline__ void diag_store(double* row, const double* d) const {
    if (wave == 0) {
      typedef double d2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d2_t v;
        v[0] = lg == 0 ? d[8 * k] : lg == 1 ? d[8 * k + 2] : lg == 2 ? d[8 * k + 4] : d[8 * k + 6];
        v[1] = lg == 0 ? d[8 * k + 1] : lg == 1 ? d[8 * k + 3] : lg == 2 ? d[8 * k + 5] : d[8 * k + 7];
        *reinterpret_cast<d2_t*>(row + 8 * k + 2 * lg) = v;
      }
    }
  }
  // proposal buffers which = 0/1: (q, grad), in memory after GB
  __device__ __forceinline__ void prop_put(int which, const Vec& q, const Vec& g) const {
    double* pq = gvec_ptr(coldbase, 1 + 2 * which);
    double* pg = gvec_ptr(coldbase, 2 + 2 * which);
#pragma unroll
    for (int e = 0; e < NE; ++e) { pq[e * 4] = q.x[e]; pg[e * 4] = g.x[e]; }
  }
  __device__ __forceinline__ void prop_get(int which, Vec& q, Vec& g) const {
    const double* pq = gvec_ptr(coldbase, 1 + 2 * which);
    const double* pg = gvec_ptr(coldbase, 2 + 2 * which);
#pragma unroll
    for (int e = 0; e < NE; ++e) { q.x[e] = pq[e * 4]; g.x[e] = pg[e * 4]; }
  }
  template <int K>
  __device__ __forceinline__ void cold_put(const Vec& x) {
    if (K == COLD_GB) {
      double* pc = gvec_ptr(coldbase, 0);
#pragma unroll
      for (int e = 0; e < NE; ++e) pc[e * 4] = x.x[e];
    } else {
      double* pc = cold_ptr<K>();
#pragma unroll
      for (int e = 0; e < NE; ++e) pc[e * 64] = x.x[e];
    }
  }
  template <int K>
  __device__ __forceinline__ void cold_get(Vec& x) const {
    if (K == COLD_GB) {
      const double* pc = gvec_ptr(coldbase, 0);
#pragma unroll
      for (int e = 0; e < NE; ++e) x.x[e] = pc[e * 4];
    } else {
      const double* pc = cold_ptr<K>();
#pragma unroll
      for (int e = 0; e < NE; ++e) x.x[e] = pc[e * 64];
    }
  }
};

}  // namespace wn
