// wn_chain.h -- the per-chain WALNUTS/NUTS state machine, written once for every layout.
//
// The reference runs one chain as nested Python loops (WALNUTSpy/WALNUTS.py:189-695 calling an
// integrator from adaptiveIntegrators.py).  On the GPU thousands of chains sit at different tree
// depths, directions and refinement levels, so the loops are turned inside out: every chain is a
// small state machine that advances by exactly ONE gradient evaluation ("tick") and then does its
// own bookkeeping.  A tick is
//
//   A  (chain-local)  start iteration / macro step / micro pass if one is pending
//   B  (uniform)      one leapfrog micro step on the work state + reduce {lp, KE, ...}
//   C  (chain-local)  energy bookkeeping of the macro-step integrator; may finalise the step
//   D  (uniform)      U-turn dot products for chains whose macro step finalised (+ reduce)
//   E  (chain-local)  NUTS bookkeeping: weights, multinomial pick, sub-tree checks, doubling /
//                     iteration end, draws + diagnostics output
//
// "uniform" phases are executed by every lane of a chain group together (they may contain
// barriers in a cooperative layout); "chain-local" phases are plain divergent SIMT code in which
// each lane keeps a replica of its chain's scalars.  The layout policy P says where the D-vectors
// live (registers of one lane, registers of several lanes in MFMA C-layout, ...), how partial dot
// products are reduced, and how the target gradient is evaluated.
//
// Reference semantics reproduced here are cited inline.
#pragma once
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "wn_rng.h"

namespace wn {

enum { T_STDGAUSS = 0, T_FUNNEL10, T_DENSEGAUSS, T_DIAGGAUSS, T_CORRGAUSS, T_SMILE, T_FUNNEL1, T_MODFUNNEL, T_STOCKWATSON = 11 };
enum { I_FIXED = 0, I_ADAPT_D = 1, I_ADAPT_R2P = 2, I_ADAPT_YOSHIDA = 3, I_ADAPT_FLOWD = 4 };
enum { G_IDENTITY = 0, G_Q0_SUMSQ = 1, G_FIRSTK = 2, G_SW_CONSTRAIN = 3 };
enum { MODE_SAMPLE = 0, MODE_KAT = 1 };
enum { NDIAG = 24, TRACE_NSCAL = 12 };
enum { MAX_M = 24 };

// WALNUTSpy/constants.py:13-14
#define WN_LOGZERO (-700.0)

// kernel parameters: keyword arguments of WALNUTS(...) (WALNUTS.py:111-129) + buffers
struct Params {
  int D, C, mode;
  int integrator, M, numIter, minC, maxC, generated, gen_k, dg;
  double H, s, delta, R2Pprob0, logP, log1mP, wtSumThresh;
  uint64_t seed;
  uint32_t chain0;  // global id of local chain 0
  const double* q0;  // [C][D]
  double* samples;   // [C][numIter+1][dg]
  double* diag;      // [C][numIter][24]
  double* trace;     // [C][trace_cap][12+2D] or null
  int trace_cap;
  int* trace_n;   // [C] or null
  double* work;   // layout-specific workspace (sub-tree endpoint slots, cold vectors)
  const double* tpar;
  // MODE_KAT (one macro step per chain; wn_integrator_step)
  const double *kq, *kv, *kg, *kham0, *kh, *ku_r2p;
  const int* kxi;
  double *kq1, *kv1, *kg1, *kdout;
  int* kiout;
  // warm-up adaptation (WALNUTS.py:120-127, 701-712)
  int warmupIter, adaptH, adaptDelta;
  double adaptHtarget, adaptDeltaTarget, adaptDeltaQuantile;
  double* facs;  // [C][2][warmupIter] sorted energyErrorInfFacs, double-buffered
  int adaptQuantBits;  // 0 = the reference's arithmetic; b: the adapted H / delta rounded to b mantissa bits (wn_config.adapt_quant_bits, opt-in)
  // recordOrbitStats (WALNUTS.py:128,182-184): [C][numIter][dg] min / max of generated(q) over the
  // orbit states of each iteration, or null (generated = identity / first-k only)
  double *omin, *omax;
  // split mode (Engine::split_tick): position out / gradient in [C][DP], log density in [C] or
  // null, number of finished chains
  double *xq, *xg;
  const double* xlp;
  int* sp_done;
  int xzero;  // the chain kernel leaves its rows of xg zeroed: the gradient kernel ADDS into them (k split over two workgroups)
  // cluster layout (several workgroups per chain): workgroups per chain, partials, arrival counters, abort flag
  int cl_g;
  double* cl_red;
  unsigned* cl_cnt;
  int* cl_abort;
  int cl_debug_drop;  // test hook (WN_DEBUG_CLUSTER_DROP): slice G-1 skips its arrival at this exchange (0 = never)
  int* dynq;  // chain queue head of a persistent launch (wn_k_orbit.hip, staged targets): zeroed before the launch; null = one chain per wave
  unsigned long long* stamps;  // diagnostic builds (-DWN_STAMPS): [block][wave][8] cycles per phase
};

#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define WN_STAMP(k)                                                                   \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    unsigned long long _t;                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" \
                 : "=s"(_t)::"memory");                                               \
    __builtin_amdgcn_sched_barrier(0);                                                \
    st_[k] += _t - st_last_;                                                          \
    st_last_ = _t;                                                                    \
  } while (0)
#else
#define WN_STAMP(k) do {} while (0)
#endif

#ifndef WN_ALIGN_LOG2
#define WN_ALIGN_LOG2 3
#endif

#define WN_VFOR(e) _Pragma("unroll") for (int e = 0; e < P::NE; ++e)

// cold vector ids (touched only at doubling boundaries)
// QB/VB/GB: passive end of the orbit.  The proposal (q_c-to-be and its gradient) and the proposal
// before the current doubling (qPropLast, WALNUTS.py:292) live in two off-chip buffers that are
// never copied: `ip` names the buffer holding qProp, `ipl` the one holding qPropLast; the first
// pick of a doubling moves to the other buffer, a revert (WALNUTS.py:599,616) just sets ip = ipl.
enum { COLD_QB = 0, COLD_VB, COLD_GB, COLD_N };
// "warm" vectors: touched once per macro step, not per micro step -- first state of the current leaf
// pair (WALNUTS.py:438-439) and, for the adaptive integrators, the accepted forward result while the
// backward passes run (adaptiveIntegrators.py:96-99).  The layout decides registers vs LDS.
enum { WARM_QL = 0, WARM_VL, WARM_QO, WARM_VO, WARM_GO, WARM_N };

// ---- per-chain scalars (replicated in every lane that serves the chain) ---------------------
// A plain struct so that the split-mode kernels (gradient evaluated outside the chain kernel) can
// park a chain between launches with one struct copy.
// Declared through a field list so that parking can copy member by member: an aggregate struct copy
// left the arrays (and whatever shared their memcpy chunks) in scratch / LDS, and scratch-using waves
// are admitted only 64 per XCD on this stack -- the split tick kernel then ran in four rounds.
#define WN_CS_FIELDS(X, XA)                                                                      \
  X(double, H0) X(double, Ha) X(double, Hb) X(double, lp0)                                       \
  X(double, WoldSum) X(double, WnewSum) X(double, lwtSumf) X(double, lwtSumb) X(double, h2)      \
  XA(double, uc, 8) /* uniforms 2*ub0 .. 2*ub0+7 of this (chain, iteration) */                   \
  X(double, keinit_part) X(double, gen_part)                                                     \
  /* diagnostics accumulators (WALNUTS.py:196-207, 660-693) */                                   \
  X(double, orbitLen_) X(double, orbitLenSam_) X(double, indexStat_) X(double, indexStatOld_)    \
  X(double, timeLenF_) X(double, timeLenB_)                                                      \
  X(double, minlw) X(double, maxlw) X(double, Hmin) X(double, Hmax)                              \
  /* macro-step integrator state */                                                              \
  X(double, h) X(double, hh) X(double, Ham0) X(double, HamPass0) X(double, lwtf) X(double, lwt)  \
  X(double, lpn) X(double, Hn) X(double, lpOut) X(double, HOut) X(double, lpCur) X(double, HCur) \
  X(double, lpP) X(double, lpPL)                                                                 \
  X(double, Hc) X(double, dc) /* step size H and tolerance delta of this chain (warm-up adapts) */ \
  /* warm-up only: P2 quantile of log(igrConst) (P2quantile.py), energy-step statistics */       \
  XA(double, p2q, 5) X(double, Hprev) X(double, madv) X(double, igrOut)                          \
  X(double, sp_hh2) /* split mode: half-step factor of the micro step in flight */               \
  /* adaptLeapFrogFlowD: lp / kinetic energy of the leapfrog half of the micro step, running max flow error */ \
  X(double, flp) X(double, fke) X(double, ferr_max)                                              \
  /* recordOrbitStats with generated = [q0, sum q^2]: sum q^2 of the state after the last micro step / of the accepted forward result */ \
  X(double, osq_cur) X(double, osq_out)                                                          \
  X(int, iter) X(uint32_t, Bbits)                                                                \
  X(int, i) X(int, xi) X(int, a) X(int, b) X(int, n) X(int, nleaf)                               \
  X(uint32_t, ku) X(uint32_t, ub0) /* ub0: first Philox block of the cached uniform window */    \
  X(int, nevalF) X(int, nevalB) X(int, L_) X(int, Lold_) X(int, NdS) X(int, NdC)                 \
  X(int, maxFint) X(int, maxBint) X(int, stopCode)                                               \
  X(int, minIf) X(int, maxIf) X(int, minc) X(int, maxc) X(int, nneq) X(int, nzero) X(int, nused) \
  X(int, ph) X(int, c) X(int, im) X(int, nstep) X(int, If) X(int, Ib) X(int, cSim) X(int, maxTry) \
  X(int, nEvF) X(int, nEvB) X(int, cret)                                                         \
  X(int, sub) /* adaptYoshidaD: leapfrog sub-step 0..2 inside the current micro step */          \
  X(uint32_t, firemask)                                                                          \
  X(int, gen_iter) /* iteration whose deferred generated() row is pending */                     \
  X(int, ip) X(int, ipl) X(int, p2n) XA(int, p2m, 5) X(int, ntrace)                              \
  X(int, sp_stage) /* split mode: 0 = waiting for grad(q0), 1 = running, 2 = finished */         \
  X(bool, live) X(bool, needKE) X(bool, needGen) X(bool, bothEndsPassive) X(bool, Hnan)          \
  X(bool, needMacro) X(bool, needPass) X(bool, fin) X(bool, needIter) X(bool, needDoubling)      \
  X(bool, endIter) X(bool, allfin) X(bool, inPass) X(bool, madnan) X(bool, ferr_nan)             \
  X(bool, sp_act) /* split mode: the micro step in flight belongs to an active chain */          \
  X(bool, sp_mid) /* split mode, adaptLeapFrogFlowD: the evaluation in flight is the midpoint one */

#define WN_CS_DECL(T, N) T N;
#define WN_CS_DECLA(T, N, K) T N[K];
#define WN_CS_COPY(T, N) d_.N = s_.N;
#define WN_CS_COPYA(T, N, K) _Pragma("unroll") for (int k_ = 0; k_ < K; ++k_) d_.N[k_] = s_.N[k_];

struct alignas(16) ChainScalars {
  enum { PH_FWD = 0, PH_FWD2 = 1, PH_BWD = 2 };
  WN_CS_FIELDS(WN_CS_DECL, WN_CS_DECLA)
  // member-by-member copy (see above)
  WN_HD static void copy(ChainScalars& d_, const ChainScalars& s_) { WN_CS_FIELDS(WN_CS_COPY, WN_CS_COPYA) }
};

// ------------------------------------------------------------------------------------------
// warm-up adaptation (WALNUTS.py:701-712), shared by the engine below and by the retire side of the headline kernel
// (wn_fast.h).  P2quantile.py:41-86 on registers (every replica of a chain runs it identically); the exact quantile of
// energyErrorInfFacs[0:iterN] (np.quantile, method "linear") reads a sorted copy in memory that only the writer lane
// extends, into the other half of a double buffer, so readers never race with the insertion.
// ------------------------------------------------------------------------------------------
WN_HD bool wn_p2_after(double a, double b) { return a > b || (isnan(a) && !isnan(b)); }  // np.sort order
WN_HD void wn_p2_cswap(double (&p2q)[5], int i, int j) {
  // (select form everywhere below: a store through a run-time choice of two elements' addresses keeps the array -- and with it the
  //  whole bookkeeper -- in scratch memory)
  const double a = p2q[i], b = p2q[j];
  const bool sw = wn_p2_after(a, b);
  p2q[i] = sw ? b : a; p2q[j] = sw ? a : b;
}
WN_HD void wn_p2_push(const Params& prm, double (&p2q)[5], int (&p2m)[5], int& p2n, double xi) {
  p2n++;
  if (p2n <= 5) {  // self.x[npush-1] = xi; (quantile() is not read before npush > 10)
#pragma unroll
    for (int j = 0; j < 5; ++j) p2q[j] = (j == p2n - 1) ? xi : p2q[j];
    if (p2n == 5) {  // np.sort (9-comparator network), self.q = self.x
      wn_p2_cswap(p2q, 0, 1); wn_p2_cswap(p2q, 3, 4); wn_p2_cswap(p2q, 2, 4); wn_p2_cswap(p2q, 2, 3); wn_p2_cswap(p2q, 0, 3);
      wn_p2_cswap(p2q, 0, 2); wn_p2_cswap(p2q, 1, 4); wn_p2_cswap(p2q, 1, 3); wn_p2_cswap(p2q, 1, 2);
    }
    return;
  }
  int k;  // findInterval :31-38 (None when xi == q[4] or xi is NaN -> every marker moves)
  if (xi < p2q[0]) k = 0;
  else if (xi > p2q[4]) k = 5;
  else k = xi < p2q[1] ? 1 : xi < p2q[2] ? 2 : xi < p2q[3] ? 3 : xi < p2q[4] ? 4 : -1;
  p2q[0] = k == 0 ? xi : p2q[0];
  p2q[4] = k == 5 ? xi : p2q[4];
  k = k == 0 ? 1 : k == 5 ? 4 : k;
#pragma unroll
  for (int j = 0; j < 5; ++j) p2m[j] += (j >= k) ? 1 : 0;  // self.n[k:5] += 1
  const double nn = (double)p2n, pp = 1.0 - prm.adaptHtarget;
  const double npp[5] = {1.0, 0.5 * (nn - 1) * pp + 1.0, (nn - 1) * pp + 1.0, (nn - 1) * (1 + pp) / 2.0 + 1, nn};
#pragma unroll
  for (int i = 2; i < 5; ++i) {  // :71-86
    const int ni = p2m[i - 1], nip = p2m[i], nim = p2m[i - 2];
    const double di = npp[i - 1] - (double)ni;
    if ((di >= 1.0 && nip - ni > 1) || (di <= -1.0 && nim - ni < -1)) {
      const int d = di > 0 ? 1 : -1;
      const double qi = p2q[i - 1];
      const double qip = qi + ((double)d / (double)(nip - nim)) *
                                  ((double)(ni - nim + d) * (p2q[i] - qi) / (double)(nip - ni) +
                                   (double)(nip - ni - d) * (qi - p2q[i - 2]) / (double)(ni - nim));
      const double qn = d > 0 ? p2q[i] : p2q[i - 2];
      const int mn = d > 0 ? p2m[i] : p2m[i - 2];
      const double qlin = qi + (double)d * (qn - qi) / (double)(mn - p2m[i - 1]);
      p2q[i - 1] = (p2q[i - 2] < qip && qip < p2q[i]) ? qip : qlin;
      p2m[i - 1] += d;
    }
  }
}
// wn_config.adapt_quant_bits (opt-in, not the reference): x rounded to b explicit mantissa bits, ties away from zero -- pure integer
// arithmetic on the bit pattern, so the CPU checker (quant_bits under oracle/) and every kernel round identically
WN_HD double wn_quant_bits(double x, int b) {
  if (b <= 0 || b >= 52) return x;  // off, or nothing to drop
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  // not finite: unchanged.  (An integer test: `x - x == 0.0` on a freshly computed product is contracted into fma(a, b, -(a * b)) by hipcc's
  // default -ffp-contract=fast, i.e. into the product's rounding error -- round 6's first build left H unrounded for exactly that reason.)
  if (((u >> 52) & 0x7ffull) == 0x7ffull) return x;
  u += 1ull << (51 - b);
  u &= ~((1ull << (52 - b)) - 1ull);
  __builtin_memcpy(&x, &u, 8);
  return x;
}
// `fc`: this chain's (this workgroup's) sorted copies [2][warmupIter]; `writer`: the one lane that extends them
WN_HD void wn_adapt_update(const Params& prm, int iter, double orbitEnergyError, double* fc, bool writer, double& dc, double& Hc,
                                  const double (&p2q)[5], int p2n) {
  if (prm.adaptDelta) {
    const int n0 = iter - 1;  // values stored so far
    const size_t W_ = (size_t)prm.warmupIter;
    const double* cur = fc + (size_t)(n0 & 1) * W_;
    double* nxt = fc + (size_t)((n0 + 1) & 1) * W_;
    const double x = orbitEnergyError / dc;  // :704
    // position of x in the sorted array (NaNs sort last): number of elements that stay before it
    int lo = 0, hi = n0;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const double a = cur[mid];
      if (wn_p2_after(x, a) || x == a) lo = mid + 1; else hi = mid;  // a <= x (NaN-last order)
    }
    const int pos = lo, n = n0 + 1;
    if (iter > 10) {  // :705-707  delta = adaptDeltaTarget / np.quantile(facs[0:iterN], q)
      auto elem = [&](int r) -> double { return r < pos ? cur[r] : (r == pos ? x : cur[r - 1]); };
      double res;
      if (isnan(elem(n - 1))) res = NAN;
      else {
        const double vi = (double)(n - 1) * prm.adaptDeltaQuantile;
        int prev = (int)floor(vi), next = prev + 1;
        double gamma = vi - (double)prev;
        if (vi >= (double)(n - 1)) { prev = n - 1; next = n - 1; gamma = vi + 1.0; }
        if (vi < 0) { prev = 0; next = 0; gamma = vi; }
        const double A = elem(prev), B = elem(next), dba = B - A;
        res = gamma >= 0.5 ? (B - dba * (1 - gamma)) : (A + dba * gamma);
      }
      dc = wn_quant_bits(prm.adaptDeltaTarget / res, prm.adaptQuantBits);
    }
    if (writer) {  // extend the sorted copy into the other buffer
      for (int r = 0; r < pos; ++r) nxt[r] = cur[r];
      nxt[pos] = x;
      for (int r = pos; r < n0; ++r) nxt[r + 1] = cur[r];
    }
  }
  if (prm.adaptH && p2n > 10) Hc = wn_quant_bits(pow(dc, 1.0 / 3.0) * exp(p2q[2]), prm.adaptQuantBits);  // :711-712
}

// FLOW: adaptLeapFrogFlowD (adaptiveIntegrators.py:246-356) -- the search of adaptLeapFrogD with
// c = 0..maxC, two gradient evaluations per micro step (new position, then the midpoint of
// adaptiveIntegrators.py:272) and the flow-error criterion max(Errs) < delta (:294).  Implies ADAPT.
// SPEC >= 0: a plain sampling run with integrator SPEC, known at compile time -- no trace, no orbit statistics, not the
// one-macro-step mode of wn_integrator_step.  The run-time switches on those cost the funnel kernel a sixth of its
// instructions and a third of its scalar-register spills (1.33e8 -> 1.57e8 evaluations/s on SURVEY 8d cfg 3).
template <class P, bool ADAPT, bool WARM = false, bool FLOW = false, int SPEC = -1>
struct Engine : ChainScalars {
  static_assert(!FLOW || ADAPT, "FLOW implies ADAPT");
  WN_HD int integ() const { return SPEC >= 0 ? SPEC : prm.integrator; }
  WN_HD int mode() const { return SPEC >= 0 ? (int)MODE_SAMPLE : prm.mode; }
  WN_HD int* trace_n() const { return SPEC >= 0 ? nullptr : prm.trace_n; }
  WN_HD double* omin() const { return SPEC >= 0 ? nullptr : prm.omin; }
  WN_HD double* omax() const { return SPEC >= 0 ? nullptr : prm.omax; }
  using Vec = typename P::Vec;
  // U-turn checks 0 (full orbit), 1 (leaf pair) and 2..1+NLC (fast-cached sub-tree levels) have
  // their left state on chip; deeper levels are fetched from the slot store in memory.
  static constexpr int NFAST = 2 + P::NLC;
  static constexpr int NPB = ADAPT ? 4 : 4 + 2 * NFAST;  // partials reduced after the micro step
  P& p;
  const Params& prm;

  // ---- D-vectors ----------------------------------------------------------------------------
  Vec qa, va, ga;  // active end of the orbit = start of the current macro step (v forward-time)
  // adaptive integrators only: work state of the running micro pass (the accepted forward result
  // is parked in the layout's warm storage while the backward passes run)
  typename P::template Opt<ADAPT> qw, vw, gw;
  // adaptLeapFrogFlowD only: state before the leapfrog half of the micro step (qold, vold, gold)
  typename P::template Opt<FLOW> qo2, vo2, go2;

  WN_HD Engine(P& p_, const Params& prm_) : p(p_), prm(prm_) {}

  // ------------------------------------------------------------------------------------------
  // random draws in the reference's call order
  // ------------------------------------------------------------------------------------------
  // The k-th uniform of (chain, iteration) is a pure function of k (wn_rng.h); a window of four
  // Philox blocks is slid forward ONCE per tick, at one code site, so the draws themselves are
  // register selects.  A tick consumes at most 5 uniforms (h pair, R2P, pick, accept) < 7.
  // (P::LANE_RNG: the layout keeps a window of uniforms of its own -- one chain per wave, 64 uniforms (32 Philox blocks) at a
  // time, one per lane -- and next_u() asks it.)
  WN_HD void rng_refill() {
    if constexpr (P::LANE_RNG) { p.rng_ensure(prm, iter, ku); return; }
    const uint32_t nb = ku >> 1;
#pragma unroll 1
    while (p.any(live && ub0 != nb)) {
      if (live && ub0 != nb) {
        uc[0] = uc[2]; uc[1] = uc[3]; uc[2] = uc[4]; uc[3] = uc[5]; uc[4] = uc[6]; uc[5] = uc[7];
        ub0++;
        Philox4 r = philox4x32_10(ub0 + 3u, (uint32_t)iter, p.chain_global(), 0u, (uint32_t)prm.seed,
                                  (uint32_t)(prm.seed >> 32));
        uc[6] = u53(r.w[0], r.w[1]);
        uc[7] = u53(r.w[2], r.w[3]);
      }
    }
  }
  WN_HD double next_u() {
    if constexpr (P::LANE_RNG) return p.uniform(prm, iter, ku++);
    const uint32_t k = ku - 2u * ub0;
    ku++;
    const double a01 = (k & 1u) ? uc[1] : uc[0], a23 = (k & 1u) ? uc[3] : uc[2];
    const double a45 = (k & 1u) ? uc[5] : uc[4], a67 = (k & 1u) ? uc[7] : uc[6];
    const double lo = (k & 2u) ? a23 : a01, hi = (k & 2u) ? a67 : a45;
    const double u = (k & 4u) ? hi : lo;
    return k < 8u ? u : NAN;  // outside the window: poison (cannot happen, see rng_refill)
  }
  // numpy legacy uniform(low, high) = low + (high-low)*U
  WN_HD double draw_h() {
    double low = Hc * (1 - prm.s), high = Hc * (1 + prm.s);
    return low + (high - low) * next_u();
  }

  // ------------------------------------------------------------------------------------------
  // iteration start  (WALNUTS.py:196-276)
  // ------------------------------------------------------------------------------------------
  WN_HD void start_iteration() {
    Bbits = 0;
    if constexpr (P::LANE_RNG) {  // :216  B = floor(uniform(0,2,size=M))
      p.rng_ensure(prm, iter, 0u);  // uniforms 0..63 of the new iteration: the M <= 24 direction bits and the first steps
#pragma unroll 1
      for (uint32_t k = 0; k < (uint32_t)prm.M; ++k)
        if ((int)floor(0.0 + (2.0 - 0.0) * p.uniform(prm, iter, k)) != 0) Bbits |= (1u << k);
    } else
#pragma unroll 1
    for (uint32_t blk = 0; 2u * blk < (uint32_t)prm.M; ++blk) {  // :216  B = floor(uniform(0,2,size=M))
      Philox4 r = philox4x32_10(blk, (uint32_t)iter, p.chain_global(), 0u, (uint32_t)prm.seed,
                                (uint32_t)(prm.seed >> 32));
      const double u0 = 0.0 + (2.0 - 0.0) * u53(r.w[0], r.w[1]);
      const double u1 = 0.0 + (2.0 - 0.0) * u53(r.w[2], r.w[3]);
      if ((int)floor(u0) != 0) Bbits |= (1u << (2u * blk));
      if (2u * blk + 1u < (uint32_t)prm.M && (int)floor(u1) != 0) Bbits |= (1u << (2u * blk + 1u));
    }
    ku = (uint32_t)prm.M;
    ub0 = (ku >> 1) - 4u;  // empty window: the refill that follows loads blocks ku/2 .. ku/2+3
    a = 0; b = 0; maxFint = 0; maxBint = 0;  // :230-233
    double kp = 0.0;
    double ztmp[P::NE];
    // rolled on purpose: one copy of philox + Box-Muller in the code instead of NE (this runs
    // once per iteration; the dynamic index costs a few scratch stores)
    if (P::NE > 8) {
#pragma unroll 1
      for (int e = 0; e < P::NE; ++e)
        ztmp[e] = p.valid(e) ? rng_normal(prm.seed, p.chain_global(), (uint32_t)iter, (uint32_t)p.gidx(e)) : 0.0;
    } else {  // few registers per vector: no dynamic index, no scratch
      WN_VFOR(e) ztmp[e] = p.valid(e) ? rng_normal(prm.seed, p.chain_global(), (uint32_t)iter, (uint32_t)p.gidx(e)) : 0.0;
    }
    WN_VFOR(e) {  // :236  v = normal(size=d); both ends start at (qc, v, grad0)  :239-253
      double z = ztmp[e];
      va.x[e] = z;
      kp += z * z;
    }
    p.prop_get(ip, qa, ga);
    orbit_stats(qa, true);
    p.template cold_put<COLD_QB>(qa);
    p.template cold_put<COLD_VB>(va);
    p.template cold_put<COLD_GB>(ga);
    keinit_part = kp;
    needIter = false;
    needKE = true;  // Hs[I0] = -f0 + 0.5*sum(v**2) (:256) completes at the next reduce
    lp0 = lpP;
    WoldSum = 1.0; WnewSum = 0.0; lwtSumb = 0.0; lwtSumf = 0.0;  // :260-264
    stopCode = 0; bothEndsPassive = false;
    nevalF = nevalB = 0; Lold_ = L_ = 0; orbitLen_ = orbitLenSam_ = 0.0; NdS = NdC = 0;
    indexStat_ = indexStatOld_ = 0.0; timeLenF_ = timeLenB_ = 0.0;
    minIf = minc = 1 << 30; maxIf = maxc = -(1 << 30); nneq = nzero = nused = 0;
    minlw = INFINITY; maxlw = -INFINITY;
    i = 0;
    xi = (Bbits & 1u) ? -1 : 1;
    start_doubling_common();
  }

  WN_HD void start_doubling_common() {  // :283-294
    n = 0;
    nleaf = 1 << i;
    ipl = ip;  // qPropLast = qProp  (:292)
    lpPL = lpP;
    Lold_ = L_;
    indexStatOld_ = indexStat_;
    if (i > 0) WnewSum = 0.0;  // :390
    needMacro = true;
  }

  WN_HD void start_doubling() {
    needDoubling = false;
    int nxi = ((Bbits >> i) & 1u) ? -1 : 1;  // :283
    if (nxi != xi) {  // the other end of the orbit becomes the active one
      Vec t;
      p.template cold_get<COLD_QB>(t); p.template cold_put<COLD_QB>(qa); qa = t;
      p.template cold_get<COLD_VB>(t); p.template cold_put<COLD_VB>(va); va = t;
      p.template cold_get<COLD_GB>(t); p.template cold_put<COLD_GB>(ga); ga = t;
      double th = Ha; Ha = Hb; Hb = th;
      xi = nxi;
    }
    start_doubling_common();
  }

  // ------------------------------------------------------------------------------------------
  // macro step / micro pass setup
  // ------------------------------------------------------------------------------------------
  WN_HD void setup_macro() {
    needMacro = false;
    if (mode() == MODE_SAMPLE) {
      if (i == 0 || (n & 1) == 0) {            // :298: one draw; :395: size=2, before the first step of a leaf pair
        h = draw_h();
        if (i != 0) h2 = draw_h();
      } else h = h2;
    }
    Ham0 = Ha;  // Hs[] of the neighbouring stored index = energy at the active end (:302,400,443,487,530)
    nEvF = nEvB = 0;
    if (ADAPT) {
      ph = PH_FWD;
      c = cmin();
      If = prm.maxC;  // adaptiveIntegrators.py:68 / 364
      needPass = true;
    } else {
      // fixedLeapFrog (adaptiveIntegrators.py:49-59) works in place on the active end
      hh = h;
      WN_VFOR(e) va.x[e] = (double)xi * va.x[e];  // vh = xi*v + ...
    }
  }

  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<B_>::type setup_pass() {
    needPass = false;
    sub = 0;
    nstep = 1 << c;
    hh = ldexp(h, -c);  // h / 2**c, adaptiveIntegrators.py:70-71 (exact scaling either way: same bits, one instruction instead of a division)
    im = 0;
    if (ph == PH_BWD) {  // :112-118 from (qOut, -vOut, gOut), Hams[0] = H0b
      p.template warm_get<WARM_QO>(qw); p.template warm_get<WARM_VO>(vw); p.template warm_get<WARM_GO>(gw);
      WN_VFOR(e) vw.x[e] = -vw.x[e];
      HamPass0 = HOut;
    } else {  // :72-76 from (q, xi*v, g), Hams[0] = Ham0
      WN_VFOR(e) { qw.x[e] = qa.x[e]; vw.x[e] = (double)xi * va.x[e]; gw.x[e] = ga.x[e]; }
      HamPass0 = Ham0;
    }
    allfin = isfinite(HamPass0);
    if (WARM) { Hprev = HamPass0; madv = -INFINITY; madnan = false; }
    if (FLOW) { ferr_max = -INFINITY; ferr_nan = false; }
  }
  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<!B_>::type setup_pass() {}

  // references to the state the micro step works on
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type QW() { return qw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type QW() { return qa; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type VW() { return vw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type VW() { return va; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type GW() { return gw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type GW() { return ga; }
  // the finalised macro-step result (q', v' in integration orientation, grad')
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type QN() { return qw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type QN() { return qa; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type VN() { return vw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type VN() { return va; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_, Vec&>::type GN() { return gw; }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_, Vec&>::type GN() { return ga; }

  // ------------------------------------------------------------------------------------------
  // phase C: after one micro step (lp, ke known)
  // ------------------------------------------------------------------------------------------
  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<B_>::type save_out(double lp, double Hi) {
    p.template warm_put<WARM_QO>(qw); p.template warm_put<WARM_VO>(vw); p.template warm_put<WARM_GO>(gw);
    if (WARM) igrOut = hh * pow(madnan ? NAN : madv, -1.0 / 3.0);  // :101 / :399 / :424
    lpOut = lp;
    HOut = Hi;
    osq_out = osq_cur;
  }
  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<!B_>::type save_out(double, double) {}

  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<B_>::type zero_adapt() {
    WN_VFOR(e) { qw.x[e] = vw.x[e] = gw.x[e] = 0.0; }
  }
  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<!B_>::type zero_adapt() {}
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<B_>::type zero_flow() { WN_VFOR(e) { qo2.x[e] = vo2.x[e] = go2.x[e] = 0.0; } }
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<!B_>::type zero_flow() {}

  WN_HD int cmin() const { return FLOW ? 0 : prm.minC; }  // adaptiveIntegrators.py:250,303: range(0, ...)
  WN_HD void start_bwd_or_finalize() {
    if (maxTry >= cmin()) {  // adaptiveIntegrators.py:107 / 440 / 300
      ph = PH_BWD;
      c = cmin();
      needPass = true;
    } else {
      finalize_macro();
    }
  }

  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<B_>::type restore_out() {
    // the backward passes overwrote the work state: bring the accepted forward result back
    p.template warm_get<WARM_QO>(qw); p.template warm_get<WARM_VO>(vw); p.template warm_get<WARM_GO>(gw);
  }
  template <bool B_ = ADAPT>
  WN_HD typename std::enable_if<!B_>::type restore_out() {}

  WN_HD void finalize_macro() {
    fin = true;
    if (ADAPT && ph == PH_BWD) restore_out();
    lpn = ADAPT ? lpOut : lpn;
    Hn = ADAPT ? HOut : Hn;
    if (integ() == I_ADAPT_D || integ() == I_ADAPT_YOSHIDA || integ() == I_ADAPT_FLOWD) {  // :135-137 / :238-240 / :354-356
      cret = If;
      lwt = (If != Ib) ? WN_LOGZERO : 0.0;
    } else if (integ() == I_ADAPT_R2P) {  // :466-475
      double lwtb = WN_LOGZERO;
      if (cSim == Ib) lwtb = prm.logP;
      else if (cSim == Ib + 1) lwtb = prm.log1mP;
      cret = cSim;
      lwt = lwtb - lwtf;
    } else {  // :57-59
      cret = 0; If = 0; Ib = 0; lwt = 0.0;
    }
  }

  // energy bookkeeping of ONE micro step (adaptiveIntegrators.py:79-84): cheap, so several micro
  // steps of a long pass can run back to back before the heavier per-pass / per-step logic
  WN_HD void micro_account(double lp, double ke) {
    if (integ() == I_ADAPT_YOSHIDA && sub < 2) {  // adaptiveIntegrators.py:157-167: not yet a full step
      sub++;
      if (ph == PH_BWD) nEvB++; else nEvF++;
      inPass = true;
      return;
    }
    if (FLOW) {
      if (sub == 0) {  // leapfrog half done: Hams[i] needs nothing more, but the step is not complete
        sub = 1;
        flp = lp; fke = ke;
        if (ph == PH_BWD) nEvB++; else nEvF++;
        inPass = true;
        return;
      }
    }
    // adaptLeapFrogFlowD: the midpoint evaluation only feeds the error estimate (flow_account).
    // (Select form on purpose: reassigning the by-value parameters here was miscompiled by hipcc 7.2
    // in the StockWatson wave kernel -- lpCur came out 0 while Hi was right)
    const double lpS = (FLOW && sub == 1) ? flp : lp, keS = (FLOW && sub == 1) ? fke : ke;
    sub = 0;
    const double Hi = -lpS + 0.5 * keS;  // Hams[i] = -fnew + 0.5*sum(vv*vv)
    lpCur = lpS; HCur = Hi;
    if (WARM) {  // np.max(np.abs(np.diff(Hams))) of the running pass (adaptiveIntegrators.py:101)
      const double df = fabs(Hi - Hprev);
      if (isnan(df)) madnan = true; else madv = df > madv ? df : madv;
      Hprev = Hi;
    }
    if (!isfinite(Hi)) allfin = false;
    im++;
    if (ph == PH_BWD) nEvB++; else nEvF++;
    inPass = im < nstep;
  }

  // ---- adaptLeapFrogFlowD: midpoint evaluation and flow errors (adaptiveIntegrators.py:272-288) ----
  // position whose gradient this tick needs: the new position (leapfrog half) or
  // qMid = 0.5*(qq+qold) + (hh/8)*(vold-vv)  (:272)
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<B_>::type flow_begin(bool act, bool& mid, double& hh2, double& hq) {
    mid = act && sub == 1;
    if (act && sub == 0) { WN_VFOR(e) { qo2.x[e] = qw.x[e]; vo2.x[e] = vw.x[e]; go2.x[e] = gw.x[e]; } }  // :264-266
    if (mid) { hh2 = 0.0; hq = 0.0; }
  }
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<!B_>::type flow_begin(bool, bool& mid, double&, double&) { mid = false; }
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<B_>::type flow_point(bool mid, Vec& x) {
    const double h8 = hh / 8.0;
    WN_VFOR(e) x.x[e] = mid ? (0.5 * (qw.x[e] + qo2.x[e]) + h8 * (vo2.x[e] - vw.x[e])) : qw.x[e];
  }
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<!B_>::type flow_point(bool, Vec&) {}
  // the four errors of :276-287 as local (max, #NaN) pairs; gm = gradient at qMid
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<B_>::type flow_errors(const Vec& gm, double* em, double* en) {
    const double hsq = hh * hh, h6 = hh / 6.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { em[k] = -INFINITY; en[k] = 0.0; }
    WN_VFOR(e) {
      const double s4 = go2.x[e] + gw.x[e] + 4.0 * gm.x[e];
      const double d0 = fabs((qo2.x[e] + hh * vo2.x[e] + hsq * ((1.0 / 6.0) * go2.x[e] + (1.0 / 3.0) * gm.x[e])) - qw.x[e]);  // qf - qq
      const double d1 = fabs((vo2.x[e] + h6 * s4) - vw.x[e]);                                                                // vf - vv
      const double d2 = fabs((qw.x[e] - hh * vw.x[e] + hsq * ((1.0 / 6.0) * gw.x[e] + (1.0 / 3.0) * gm.x[e])) - qo2.x[e]);  // qb - qold
      const double d3 = fabs(-(-vw.x[e] + h6 * s4) - vo2.x[e]);                                                              // vb - vold
      const double dd[4] = {d0, d1, d2, d3};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (isnan(dd[k])) en[k] += 1.0; else em[k] = dd[k] > em[k] ? dd[k] : em[k];
      }
    }
  }
  template <bool B_ = FLOW>
  WN_HD typename std::enable_if<!B_>::type flow_errors(const Vec&, double*, double*) {}
  // err = np.max(..); err = max(err, np.max(..)) x3 (Python's max: x if x > err else err); Errs[i-1] = err;
  // maxErr = np.max(Errs)  (:277-288, 294)
  WN_HD void flow_account(const double* em, const double* en) {
    double err = en[0] > 0.0 ? NAN : em[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const double x = en[k] > 0.0 ? NAN : em[k];
      err = x > err ? x : err;
    }
    if (isnan(err)) ferr_nan = true; else ferr_max = err > ferr_max ? err : ferr_max;
  }

  WN_HD void after_micro(double lp, double ke) {  // fixedLeapFrog: the macro step is one micro step
    nEvF = 1;
    lpn = lp;
    Hn = -lp + 0.5 * ke;
    finalize_macro();
  }

  WN_HD void pass_end() {
    const double lp = lpCur, Hi = HCur;
    // pass complete: all(isfinite(Hams)) and abs(Hams[0]-Hams[-1]) < delta   (:87-94);
    // adaptLeapFrogFlowD: ... and np.max(Errs) < delta  (:294-297)
    bool ok = allfin && (FLOW ? (!ferr_nan && ferr_max < dc) : (fabs(HamPass0 - Hi) < dc));
    if (ph == PH_FWD) {
      bool done = false;
      if (ok) { If = c; done = true; }
      else if (c < prm.maxC) { c++; needPass = true; }
      else { If = prm.maxC; done = true; }  // Q5: last computed state is used silently
      if (done) {
        if (integ() == I_ADAPT_R2P) {
          double u = (mode() == MODE_KAT) ? prm.ku_r2p[p.chain_local()] : next_u();  // :392
          if (u < prm.R2Pprob0) {  // :393-399, 430-433
            cSim = If; save_out(lp, Hi);
            maxTry = If - 1; Ib = If; lwtf = prm.logP;
            start_bwd_or_finalize();
          } else {  // :400-424 re-integrate at c = If+1
            ph = PH_FWD2; c = If + 1; needPass = true;
          }
        } else {  // adaptLeapFrogD :96-111
          cSim = If; save_out(lp, Hi);
          maxTry = If - 1; Ib = If; lwtf = 0.0;
          start_bwd_or_finalize();
        }
      }
    } else if (ph == PH_FWD2) {  // :419-424, 434-437
      cSim = If + 1; save_out(lp, Hi);
      maxTry = prm.maxC; Ib = prm.maxC; lwtf = prm.log1mP;
      start_bwd_or_finalize();
    } else {  // PH_BWD :129-132 / 461-464
      if (ok) { Ib = c; finalize_macro(); }
      else if (c < maxTry) { c++; needPass = true; }
      else finalize_macro();
    }
  }

  // ------------------------------------------------------------------------------------------
  // which U-turn checks does a finalised step need?  check 0 = full orbit (passive end,
  // WALNUTS.py:622), check j>=1 = sub-tree of 2**j leaves ending at this leaf (subTreePlan,
  // WALNUTS.py:22-41: a check [s-2**j+1, s] fires after leaf s for every j with 2**j | s).
  // ------------------------------------------------------------------------------------------
  WN_HD int ctz_leaf() const {  // ctz(n+1) for odd n
    return __builtin_ctz((unsigned)(n + 1));
  }
  WN_HD int nchecks() const {  // number of check indices to evaluate (0 = none)
    if (!(live && fin) || mode() != MODE_SAMPLE) return 0;
    int m = 0;
    if (n == nleaf - 1) m = 1;
    if (i >= 1 && (n & 1)) m = ctz_leaf() + 1;
    return m;
  }
  WN_HD bool need_check(int j) const {
    if (j == 0) return n == nleaf - 1;
    return i >= 1 && (n & 1) && j <= ctz_leaf();
  }
  WN_HD void uturn_dots(const Vec& qo_, const Vec& vo_, double& d1, double& d2) {
    double a1 = 0.0, a2 = 0.0;
    WN_VFOR(e) {
      double dq = QN().x[e] - qo_.x[e];
      a1 += ((double)xi * VN().x[e]) * dq;
      a2 += vo_.x[e] * dq;
    }
    d1 = a1; d2 = a2;
  }
  // dots of the on-chip checks: 0 = passive orbit end, 1 = first state of the leaf pair,
  // 2..NFAST-1 = fast-cached sub-tree left endpoints.  out[2*j], out[2*j+1].
  WN_HD void fast_dots(double* out) {
    const bool on = live && fin && mode() == MODE_SAMPLE;
#pragma unroll
    for (int j = 0; j < NFAST; ++j) {
      double d1 = 0.0, d2 = 0.0;
      if (p.any(on && need_check(j)) && on && need_check(j)) {
        if (j == 0) {
          Vec qo_, vo_;
          p.template cold_get<COLD_QB>(qo_); p.template cold_get<COLD_VB>(vo_);
          uturn_dots(qo_, vo_, d1, d2);
        } else if (j == 1) {
          Vec qo_, vo_;
          p.template warm_get<WARM_QL>(qo_); p.template warm_get<WARM_VL>(vo_);
          uturn_dots(qo_, vo_, d1, d2);
        } else {
          Vec qo_, vo_;
          p.lslot_load(j - 2, qo_, vo_);
          uturn_dots(qo_, vo_, d1, d2);
        }
      }
      out[2 * j] = d1; out[2 * j + 1] = d2;
    }
  }
  WN_HD void fast_fire(const double* dots, int nchk) {
#pragma unroll
    for (int j = 0; j < NFAST; ++j)
      if (nchk > j && need_check(j) &&
          (((double)xi * dots[2 * j] < 0.0) || ((double)xi * dots[2 * j + 1] < 0.0)))
        firemask |= (1u << j);
  }
  WN_HD static int slot_of_leaf(int m) { return m == 0 ? 0 : __builtin_ctz((unsigned)m); }

  // ------------------------------------------------------------------------------------------
  // phase E: NUTS bookkeeping after a finalised macro step (WALNUTS.py:296-648)
  // ------------------------------------------------------------------------------------------
  // orbitMin/orbitMax[:,iterN-1] = np.minimum/np.maximum(..., generated(q))  (WALNUTS.py:274-276,
  // 331-333, 364-366, 434-436, ...): every lane updates the elements it owns, in memory
  WN_HD bool orbit_sumsq() const { return omin() && prm.generated == G_Q0_SUMSQ; }
  // generated = [q[0], sum(q*q)] (mainGaussESS.py:29-30): the second component is a sum over all lanes that serve
  // the chain, so it rides in its own reduction right after the gradient (uniform: prm only)
  WN_HD void orbit_sumsq_reduce(const Vec& q) {
    if (!orbit_sumsq()) return;
    double s[1] = {0.0};
    WN_VFOR(e) s[0] += q.x[e] * q.x[e];
    p.template reduce<1>(s);
    osq_cur = s[0];
  }
  WN_HD void orbit_stats(const Vec& q, bool init) {
    if (!omin()) return;
    const size_t row = ((size_t)p.chain_local() * prm.numIter + (iter - 1)) * prm.dg;
    if (prm.generated == G_Q0_SUMSQ) {
      WN_VFOR(e) if (p.valid(e) && p.owner(e) && p.gidx(e) == 0) orbit_minmax(row, q.x[e], init);
      // the sum of squares of the iteration's first state arrives with its draw (finish_generated)
      if (!init && p.writer()) orbit_minmax(row + 1, ADAPT ? osq_out : osq_cur, false);
      return;
    }
    const int k = prm.generated == G_FIRSTK ? prm.gen_k : prm.D;
    WN_VFOR(e) if (p.valid(e) && p.owner(e) && p.gidx(e) < k) orbit_minmax(row + p.gidx(e), q.x[e], init);
    // generated = param_constrain(q, include_tp=True) of the StockWatson model (mainSW.py:28-29, sw_innov.stan:17-35): the transformed
    // parameters of THIS orbit state -- sigma, gpri, z, x, tau behind the 3T unconstrained ones -- from the layout that holds the series
    if (prm.generated == G_SW_CONSTRAIN) p.sw_constrained(q, [&](int idx, double x) { orbit_minmax(row + (size_t)idx, x, init); });
  }
  WN_HD void orbit_minmax(size_t at, double x, bool init) {
    double* mn = omin() + at;
    double* mx = omax() + at;
    if (init) { *mn = x; *mx = x; }
    else {
      const double a = *mn, b = *mx;
      *mn = (isnan(x) || isnan(a)) ? NAN : (x < a ? x : a);
      *mx = (isnan(x) || isnan(b)) ? NAN : (x > b ? x : b);
    }
  }

  WN_HD void pick_state(int idx) {
    if (ip == ipl) ip ^= 1;  // keep qPropLast intact
    p.prop_put(ip, QN(), GN());
    lpP = lpn;
    L_ = idx;
    indexStat_ = xi > 0 ? timeLenF_ : -timeLenB_;
  }
  WN_HD void revert_prop() {  // qProp = qPropLast (:599-601, 614-616)
    ip = ipl;
    lpP = lpPL;
    L_ = Lold_;
    indexStat_ = indexStatOld_;
  }

  WN_HD void trace_row() {
    if (!trace_n()) return;
    int cl = p.chain_local();
    int row = ntrace++;
    if (prm.trace && row < prm.trace_cap) {
      double* r = prm.trace + ((size_t)cl * prm.trace_cap + row) * (TRACE_NSCAL + 2 * prm.D);
      if (p.writer()) {
        r[0] = iter; r[1] = i; r[2] = xi; r[3] = h; r[4] = Ham0; r[5] = lpn; r[6] = If; r[7] = Ib;
        r[8] = cret; r[9] = lwt; r[10] = nEvF; r[11] = nEvB;
      }
      WN_VFOR(e) if (p.valid(e) && p.owner(e)) {
        r[TRACE_NSCAL + p.gidx(e)] = QN().x[e];
        r[TRACE_NSCAL + prm.D + p.gidx(e)] = (double)xi * VN().x[e];
      }
    }
  }
  unsigned long long st_[16], st_last_;  // diagnostic builds only (-DWN_STAMPS)

  WN_HD void step_done() {
    fin = false;
    const bool first = (i == 0) || ((n & 1) == 0);
    const int idx = xi > 0 ? b + 1 + n : a - 1 - n;
    trace_row();
    if (WARM && iter <= prm.warmupIter && prm.adaptH) {  // igrConstQ.push(np.log(intOut.igrConst)) :313
      const double igr = ADAPT ? igrOut : h * pow(fmax(1.0e-10, fabs(Ham0 - Hn)), -1.0 / 3.0);  // ai.py:59
      p2_push(log(igr));
    }
    // intOut.v is in forward-time orientation (adaptiveIntegrators.py:57,135,473)
    WN_VFOR(e) VN().x[e] = (double)xi * VN().x[e];
    nevalF += nEvF; nevalB += nEvB;
    // per-step records Hs/Ifs/Ibs/cs/lwts feed only min/max/mean diagnostics (:668-692)
    nused++;
    if (ADAPT) {  // fixedLeapFrog: If = Ib = c = 0, lwt = 0 for every step
      minIf = If < minIf ? If : minIf; maxIf = If > maxIf ? If : maxIf;
      minc = cret < minc ? cret : minc; maxc = cret > maxc ? cret : maxc;
      minlw = lwt < minlw ? lwt : minlw; maxlw = lwt > maxlw ? lwt : maxlw;
      nneq += (If != Ib); nzero += (If == 0);
    }
    if (isnan(Hn)) Hnan = true;
    Hmin = Hn < Hmin ? Hn : Hmin; Hmax = Hn > Hmax ? Hn : Hmax;
    {  // select form (a branch on xi is an exec-mask save / restore around two one-line bodies)
      const bool fw = xi > 0;
      const double tf = timeLenF_ + h, tb = timeLenB_ + h;
      maxFint = fw ? idx : maxFint; maxBint = fw ? maxBint : idx;
      timeLenF_ = fw ? tf : timeLenF_; timeLenB_ = fw ? timeLenB_ : tb;
    }
    if (i == 0) orbitLen_ += h;  // :300 (before the integrator call)
    if (!isfinite(Hn)) {         // :316-319, 414-417, 457-459 (Q3, Q4)
      if (first) stopCode = 999;
      endIter = true;
      return;
    }
    {  // i == 0: lwtSum = lwt (:321 / 354); else: lwtSum += lwt, on the backward side only at the first leaf of a pair (:420,507,550; Q2)
      const bool fw = xi > 0, upd = (i == 0) || first || fw;
      const double nf = (i == 0) ? lwt : lwtSumf + lwt, nb = (i == 0) ? lwt : lwtSumb + lwt;
      lwtSumf = (upd && fw) ? nf : lwtSumf;
      lwtSumb = (upd && !fw) ? nb : lwtSumb;
    }
    const double lwtSum = xi > 0 ? lwtSumf : lwtSumb;
    const double Wnew = exp(-Hn + H0 + lwtSum);
    WN_STAMP(8);
    orbit_stats(QN(), false);
    bool pick;
    if (i == 0) {  // :322-328, 368-369: unconditional, no uniform
      WnewSum = Wnew;
      pick = true;
    } else {
      WnewSum += Wnew;
      // :426  the uniform is not drawn when the first clause is false
      pick = WnewSum > prm.wtSumThresh && next_u() < Wnew / WnewSum;
      orbitLen_ += h;  // :432 / 471 / 519 / 561
      if ((n & 1) == 0) {  // statePush (:431,518) -- only sub-tree left endpoints are ever read back
        p.template warm_put<WARM_QL>(QN()); p.template warm_put<WARM_VL>(VN());
        if (i >= 2 && (n & 3) == 0) {
          // left endpoint of sub-trees of 2**j leaves for every j <= i with 2**j | n.  Checks
          // 2..NFAST-1 read it back from the layout's fast cache (one copy per level, reused
          // every 2**j leaves); deeper checks read the long-lived copy in slot ctz(n).
#pragma unroll
          for (int j = 2; j < NFAST; ++j)
            if (j <= i && (n & ((1 << j) - 1)) == 0) p.lslot_store(j - 2, QN(), VN());
          if (n == 0 ? (i >= NFAST) : (__builtin_ctz((unsigned)n) >= NFAST))
            p.slot_store(slot_of_leaf(n), QN(), VN());
        }
      }
    }
    if (pick) pick_state(idx);  // (one call site: the function is inlined)
    WN_STAMP(9);
    // the new state becomes the active end (qp/qm = intOut.q ...)
    if (ADAPT) { WN_VFOR(e) { qa.x[e] = QN().x[e]; va.x[e] = VN().x[e]; ga.x[e] = GN().x[e]; } }
    Ha = Hn;
    bool sub = false;
    if (i >= 1 && (n & 1)) {  // leaf pair + completed inner rows, post-order (:479,568,582-587)
      uint32_t m = ((2u << ctz_leaf()) - 1u) & ~1u;  // checks 1..ctz(n+1)
      sub = (firemask & m) != 0;
    }
    n++;
    if (sub || n == nleaf) doubling_end(sub);  // (one call site: the function is inlined)
    else needMacro = true;
    WN_STAMP(10);
  }

  WN_HD void doubling_end(bool sub) {
    indexStat_ = indexStat_ / (timeLenF_ + timeLenB_);  // :595 (Q6)
    if (sub) {  // :597-605
      revert_prop();
      NdS = i; NdC = i + 1; stopCode = 5;
      endIter = true;
      return;
    }
    if (!(next_u() < WnewSum / WoldSum)) revert_prop();  // :613-616
    const bool joined = (firemask & 1u) != 0;            // :622
    bothEndsPassive = (lwtSumb < WN_LOGZERO + 1.0) && (lwtSumf < WN_LOGZERO + 1.0);  // :624
    if (joined || bothEndsPassive) {  // :625-634
      stopCode = joined ? 4 : -4;
      NdS = NdC = i + 1;
      orbitLenSam_ = orbitLen_;
      endIter = true;
      return;
    }
    WoldSum += WnewSum;  // :641-648
    orbitLenSam_ = orbitLen_;
    NdS = NdC = i + 1;
    if (xi > 0) b += nleaf; else a -= nleaf;
    i++;
    if (i == prm.M) { endIter = true; return; }
    needDoubling = true;
  }

  // ------------------------------------------------------------------------------------------
  // warm-up adaptation (WALNUTS.py:701-712): wn_p2_push / wn_adapt_update above, on this chain's scalars
  // ------------------------------------------------------------------------------------------
  WN_HD void p2_push(double xi) { wn_p2_push(prm, p2q, p2m, p2n, xi); }
  WN_HD void adapt_update(double orbitEnergyError) {
    double* const fc = prm.facs ? prm.facs + (size_t)p.facs_slot() * 2 * (size_t)prm.warmupIter : nullptr;
    wn_adapt_update(prm, iter, orbitEnergyError, fc, p.wg_writer(), dc, Hc, p2q, p2n);
  }

  WN_HD void end_iteration() {  // :653-695
    const int cl = p.chain_local();
    emit_generated(iter);  // samples[:,iterN] = generated(qc), qc = qProp  (loads before the stores below)
    double d[NDIAG];
    d[0] = L_; d[1] = NdS; d[2] = orbitLen_; d[3] = orbitLenSam_; d[4] = maxFint; d[5] = maxBint;
    d[6] = nevalF; d[7] = nevalB;
    d[8] = ADAPT ? minIf : 0; d[9] = ADAPT ? maxIf : 0; d[10] = ADAPT ? minlw : 0.0; d[11] = ADAPT ? maxlw : 0.0;
    d[12] = bothEndsPassive ? 1.0 : 0.0;
    d[13] = ((lwtSumb < WN_LOGZERO + 1.0) || (lwtSumf < WN_LOGZERO + 1.0)) ? 1.0 : 0.0;
    d[14] = ADAPT ? (double)nneq / (double)nused : 0.0;
    d[15] = Hc;
    d[16] = ADAPT ? (double)nzero / (double)nused : 1.0;
    d[17] = Hnan ? NAN : (Hmax - Hmin);
    d[18] = dc; d[19] = (double)stopCode; d[20] = NdC; d[21] = ADAPT ? minc : 0; d[22] = ADAPT ? maxc : 0;
    d[23] = indexStat_;
    if (WARM && iter <= prm.warmupIter) adapt_update(d[17]);
    p.diag_store(prm.diag + ((size_t)cl * prm.numIter + (iter - 1)) * NDIAG, d);
    iter++;
    endIter = false;
    needMacro = false;
    if (iter > prm.numIter) live = false;
    else needIter = true;
  }

  WN_HD void emit_generated(int it) {
    double* row = prm.samples + ((size_t)p.chain_local() * (prm.numIter + 1) + it) * prm.dg;
    Vec qP, gP_;
    p.prop_get(ip, qP, gP_);
    if (prm.generated == G_Q0_SUMSQ) {  // needs a reduction: completed at the next uniform phase
      double s = 0.0;
      WN_VFOR(e) s += qP.x[e] * qP.x[e];
      gen_part = s;
      needGen = true;
      gen_iter = it;
      WN_VFOR(e) if (p.valid(e) && p.owner(e) && p.gidx(e) == 0) row[0] = qP.x[e];
    } else {
      const int k = prm.generated == G_FIRSTK ? prm.gen_k : prm.D;
      WN_VFOR(e) if (p.valid(e) && p.owner(e) && p.gidx(e) < k) row[p.gidx(e)] = qP.x[e];
    }
  }
  WN_HD void finish_generated(double total) {
    if (needGen) {
      if (p.writer()) {
        prm.samples[((size_t)p.chain_local() * (prm.numIter + 1) + gen_iter) * prm.dg + 1] = total;
        // orbitMin / orbitMax[:, iterN-1] start at generated(qc) (WALNUTS.py:274-276): iteration gen_iter + 1 starts here
        if (orbit_sumsq() && gen_iter < prm.numIter) {
          const size_t at = ((size_t)p.chain_local() * prm.numIter + gen_iter) * prm.dg + 1;
          omin()[at] = total; omax()[at] = total;
        }
      }
      needGen = false;
      gen_part = 0.0;
    }
  }

  WN_HD void init_scalars(bool inrange) {
    // every scalar starts defined (reading an indeterminate value is UB even on dead paths)
    Bbits = 0; lp0 = 0.0; WoldSum = 1.0; WnewSum = 0.0; lwtSumf = lwtSumb = 0.0;
    gen_iter = 0; lpP = lpPL = 0.0; ip = ipl = 0;
    Hc = prm.H; dc = prm.delta;
    p2n = 0; Hprev = madv = igrOut = 0.0; madnan = false;
#pragma unroll
    for (int k = 0; k < 5; ++k) { p2q[k] = 0.0; p2m[k] = k + 1; }
    nevalF = nevalB = L_ = Lold_ = NdS = NdC = maxFint = maxBint = stopCode = 0;
    orbitLen_ = orbitLenSam_ = indexStat_ = indexStatOld_ = timeLenF_ = timeLenB_ = 0.0;
    bothEndsPassive = false;
    minIf = minc = 1 << 30; maxIf = maxc = -(1 << 30); nneq = nzero = nused = 0;
    minlw = INFINITY; maxlw = -INFINITY;
    ph = PH_FWD; c = im = nstep = If = Ib = cSim = maxTry = nEvF = nEvB = cret = 0; sub = 0;
    hh = Ham0 = HamPass0 = lwtf = lwt = lpCur = HCur = 0.0;
    allfin = true; inPass = false;
    WN_VFOR(e) { va.x[e] = 0.0; ga.x[e] = 0.0; }
    zero_adapt();
    live = inrange;
    needKE = needGen = needMacro = needPass = fin = needIter = needDoubling = endIter = false;
    keinit_part = gen_part = 0.0;
    ntrace = 0;
    ku = 0; ub0 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) uc[k] = 0.0;
    iter = 1;
    Hnan = false; Hmin = INFINITY; Hmax = -INFINITY;
    firemask = 0;
    h = h2 = 0.0; xi = 1; i = 0; n = 0; nleaf = 1; a = b = 0;
    lpn = Hn = lpOut = HOut = 0.0; H0 = Ha = Hb = 0.0;
    sp_stage = 0; sp_hh2 = 0.0; sp_act = false; sp_mid = false;
    flp = fke = 0.0; ferr_max = -INFINITY; ferr_nan = false;
    osq_cur = osq_out = 0.0;
    zero_flow();
  }


  // ------------------------------------------------------------------------------------------
  // split mode: the gradient is evaluated OUTSIDE the chain kernel -- by a batched MFMA GEMM for
  // dense Gaussians too large to keep A on chip, or by the caller's own batched (logp, grad) on
  // device tensors (the reference's lpFun plugin, targetDistr.py:18).  One launch of the tick kernel
  // runs, for every chain, the part of tick t that follows p.grad() in run() and the part of tick
  // t+1 that precedes it; between launches a chain is parked in memory (P::park / P::unpark).
  // Same arithmetic and same draw order as run(); only one micro step per tick.
  // ------------------------------------------------------------------------------------------
  WN_HD void split_begin() {
    const bool inrange = p.chain_local() < prm.C;
    init_scalars(inrange);
    const int cl = inrange ? p.chain_local() : 0;
    WN_VFOR(e) qa.x[e] = (p.valid(e) && inrange) ? prm.q0[(size_t)cl * prm.D + p.gidx(e)] : 0.0;
    p.ext_put_q(qa);  // grad(q0) is evaluated next
    park();
  }
  // Homes between launches: the work position (qa for fixedLeapFrog, qw otherwise) IS the row of the
  // exchange buffer the gradient kernel reads, and the gradient of an active chain arrives in the
  // other exchange buffer, so neither is parked; everything else goes to the chain's park slots.
  WN_HD void park() {
    // keinit_part / gen_part are per-lane PARTIAL sums, not replicated scalars: park their totals
    // (the next reduce then adds the total in one lane to zeros: the same bits)
    keinit_part = p.park_partial(keinit_part);
    gen_part = p.park_partial(gen_part);
    p.park_scalars(static_cast<const ChainScalars&>(*this));
    p.template park_vec<1>(va);
    if (ADAPT) { p.template park_vec<0>(qa); p.template park_vec<2>(ga); }
    else if (!sp_act) p.template park_vec<2>(ga);
    park_adapt();
  }
  // Loads that do not depend on the parked scalars are issued BEFORE the scalars are read, so a tick
  // starts with one memory round trip, not two; the rest follows once sp_act is known.
  WN_HD void unpark_early() {
    p.template unpark_vec<1>(va);
    if (ADAPT) { p.template unpark_vec<0>(qa); p.template unpark_vec<2>(ga); }
    else p.ext_get_q(qa);
    unpark_adapt_early();
  }
  WN_HD void unpark_late() {
    if (!ADAPT) { if (!sp_act) p.template unpark_vec<2>(ga); }
    unpark_adapt_late();
  }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_>::type park_adapt() {
    p.template park_vec<4>(vw);
    if (FLOW || !sp_act) p.template park_vec<5>(gw);
    park_flow();
  }
  // adaptLeapFrogFlowD: the exchange buffers carry the midpoint on every other tick, so the work
  // position and gradient are parked like everything else, with the pre-step state
  template <bool B_ = FLOW> WN_HD typename std::enable_if<B_>::type park_flow() {
    p.template park_vec<3>(qw); p.template park_vec<6>(qo2); p.template park_vec<7>(vo2); p.template park_vec<8>(go2);
  }
  template <bool B_ = FLOW> WN_HD typename std::enable_if<!B_>::type park_flow() {}
  template <bool B_ = FLOW> WN_HD typename std::enable_if<B_>::type unpark_flow() {
    p.template unpark_vec<3>(qw); p.template unpark_vec<6>(qo2); p.template unpark_vec<7>(vo2); p.template unpark_vec<8>(go2);
  }
  template <bool B_ = FLOW> WN_HD typename std::enable_if<!B_>::type unpark_flow() { p.ext_get_q(qw); }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_>::type park_adapt() {}
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_>::type unpark_adapt_early() {
    unpark_flow();
    p.template unpark_vec<4>(vw);
  }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_>::type unpark_adapt_early() {}
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<B_>::type unpark_adapt_late() {
    if (FLOW || !sp_act) p.template unpark_vec<5>(gw);
  }
  template <bool B_ = ADAPT> WN_HD typename std::enable_if<!B_>::type unpark_adapt_late() {}

  // returns true while the chain still needs ticks
  WN_HD bool split_tick() {
    unpark_early();
    Vec gin;  // the gradient that arrived for the position sent out by the previous launch
    p.ext_load_g(gin);
    p.unpark_scalars(static_cast<ChainScalars&>(*this));
    if (sp_stage == 2) return false;
    rng_refill();  // (P::LANE_RNG: the layout's window of uniforms does not survive the launch boundary)
    keinit_part = p.unpark_partial(keinit_part);
    gen_part = p.unpark_partial(gen_part);
    unpark_late();
#pragma unroll
    for (int k = 0; k < 16; ++k) st_[k] = 0;
    st_last_ = 0;
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    st_last_ = p.t_start;
#endif
    WN_STAMP(0);
    Vec& Q = QW(); Vec& V = VW(); Vec& G = GW();
    double part[NPB > 2 * NFAST ? NPB : 2 * NFAST];
    if (sp_stage == 0) {  // f0, grad0 = lpFun(q0)  (WALNUTS.py:177,249)
      WN_VFOR(e) ga.x[e] = gin.x[e];
      p.ext_lp(qa, ga, part[0]);
      p.template reduce<1>(part);
      lpP = p.lp_finish(part[0]);
      p.prop_put(ip, qa, ga);
      if (live) {
        emit_generated(0);  // samples[:,0] = generated(q0)  (:164)
        if (prm.numIter < 1) live = false; else needIter = true;
      }
      sp_stage = 1;
    } else {
      // ---- B, second half ----
      const bool act = sp_act;
      const double hh2 = sp_hh2;
      const bool mid = FLOW && sp_mid;
      double em[4], en[4];
      if constexpr (FLOW) {
        p.ext_lp(Q, gin, part[0]);
        if (p.any(mid)) flow_errors(gin, em, en);
        WN_VFOR(e) G.x[e] = mid ? G.x[e] : gin.x[e];
      } else {
        WN_VFOR(e) G.x[e] = gin.x[e];
        p.ext_lp(Q, G, part[0]);
      }
      orbit_sumsq_reduce(Q);
      double ke = 0.0;
      WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; ke += V.x[e] * V.x[e]; }
      part[1] = ke; part[2] = keinit_part; part[3] = gen_part;
      WN_STAMP(1);
      if (!ADAPT) {
        fin = act;
        if (p.any(nchecks() > 0)) {
          fast_dots(part + 4);
          WN_STAMP(12);
          p.template reduce<NPB>(part);
        } else {
#pragma unroll
          for (int k = 4; k < NPB; ++k) part[k] = 0.0;
          p.template reduce<4>(part);
        }
      } else {
        if (p.any(needKE || needGen)) p.template reduce<NPB>(part);
        else p.template reduce<2>(part);
      }
      WN_STAMP(2);
      if constexpr (FLOW) if (p.any(mid)) {
        p.template reduce_max<4>(em);
        p.template reduce<4>(en);
        if (mid) flow_account(em, en);
      }
      finish_generated(part[3]);
      if (act) {
        if (needKE) {  // Hs[I0] = -f0 + 0.5*sum(v**2)  (:256); multinomialLscale (:259)
          H0 = -lp0 + 0.5 * part[2];
          Ha = Hb = H0; Ham0 = H0;
          if (ADAPT) { HamPass0 = H0; allfin = isfinite(H0); if (WARM) Hprev = H0; }
          Hnan = isnan(H0); Hmin = Hmax = H0;
          needKE = false; keinit_part = 0.0;
        }
        if (ADAPT) micro_account(p.lp_finish(part[0]), part[1]);
      }
      // ---- C ----
      if (live && !needIter) {
        if (!ADAPT) after_micro(p.lp_finish(part[0]), part[1]);
        else if (!inPass) pass_end();
      }
      WN_STAMP(3);
      // ---- D: U-turn checks ----
      const int nchk = nchecks();
      firemask = 0;
      if (ADAPT) {
        if (p.any(nchk > 0)) {
          fast_dots(part);
          p.template reduce<2 * NFAST>(part);
          fast_fire(part, nchk);
        }
      } else {
        fast_fire(part + 4, nchk);
      }
      for (int j0 = NFAST; p.any(nchk > j0); j0 += 4) {
        double dots[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          double d1 = 0.0, d2 = 0.0;
          if (p.any(nchk > j)) {
            if ((nchk > j) && need_check(j)) {
              Vec qo_, vo_;
              p.slot_load(slot_of_leaf(n + 1 - (1 << j)), qo_, vo_);
              uturn_dots(qo_, vo_, d1, d2);
            }
          }
          dots[2 * jj] = d1; dots[2 * jj + 1] = d2;
        }
        p.template reduce<8>(dots);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          if (nchk > j && need_check(j) &&
              (((double)xi * dots[2 * jj] < 0.0) || ((double)xi * dots[2 * jj + 1] < 0.0)))
            firemask |= (1u << j);
        }
      }
      WN_STAMP(4);
      // ---- E ----
      if (live && fin) {
        step_done();
        if (endIter) end_iteration();
      }
      WN_STAMP(5);
    }
    if (!p.any(live || needGen)) {
      sp_stage = 2;
      if (trace_n() && p.chain_local() < prm.C && p.writer()) trace_n()[p.chain_local()] = ntrace;
      p.park_scalars(static_cast<const ChainScalars&>(*this));
      return false;
    }
    // ---- A of the next tick ----
    if (live && needIter) start_iteration();
    rng_refill();
    if (live && needDoubling) start_doubling();
    if (live && needMacro) setup_macro();
    if (ADAPT && live && needPass) setup_pass();
    WN_STAMP(7);
    // ---- B, first half: kick + drift; the position goes out for the gradient ----
    {
      Vec& Q2 = QW(); Vec& V2 = VW(); Vec& G2 = GW();
      const bool act = live && !needIter;
      double hh2 = act ? 0.5 * hh : 0.0, hq = act ? hh : 0.0;
      if (ADAPT && integ() == I_ADAPT_YOSHIDA && act) {
        const double w = sub == 1 ? -1.702414383919315 : 1.351207191959658;
        hh2 = 0.5 * w * hh; hq = w * hh;
      }
      bool mid;
      flow_begin(act, mid, hh2, hq);
      WN_VFOR(e) { V2.x[e] = V2.x[e] + hh2 * G2.x[e]; Q2.x[e] = Q2.x[e] + hq * V2.x[e]; }
      if constexpr (FLOW) {
        Vec xin;
        flow_point(mid, xin);
        p.ext_put_q(xin);
      } else {
        p.ext_put_q(Q2);
      }
      sp_act = act; sp_hh2 = hh2; sp_mid = mid;
    }
    park();
    WN_STAMP(11);
    st_[6] += 1;
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (prm.stamps && p.writer()) {
      unsigned long long* o = prm.stamps + (size_t)p.chain_local() * 16;
      for (int k = 0; k < 13; ++k) o[k] += st_[k];
      if (o[6] == 100) { o[13] = p.t_real; o[14] = __builtin_amdgcn_s_memrealtime(); }  // wall clock (100 MHz) of launch 100
    }
#endif
    return true;
  }

  // ------------------------------------------------------------------------------------------
  // the tick loop
  // ------------------------------------------------------------------------------------------
  WN_HD void run() {
    const bool inrange = p.chain_local() < prm.C;
    init_scalars(inrange);

    // q_c = q0; f0, grad0 = lpFun(q0)  (WALNUTS.py:177,249)
    const int cl = inrange ? p.chain_local() : 0;
    {
      const double* src = mode() == MODE_KAT ? prm.kq : prm.q0;
      WN_VFOR(e) qa.x[e] = (p.valid(e) && inrange) ? src[(size_t)cl * prm.D + p.gidx(e)] : 0.0;
    }
    if (mode() == MODE_SAMPLE) {
      double part[1];
      p.grad(qa, ga, part[0]);
      p.template reduce<1>(part);
      lpP = p.lp_finish(part[0]);
      p.prop_put(ip, qa, ga);
      if (live) {
        emit_generated(0);  // samples[:,0] = generated(q0)  (:164)
        if (prm.numIter < 1) live = false; else needIter = true;
      }
    } else if (live) {  // one macro step from a caller-supplied state
      WN_VFOR(e) {
        va.x[e] = p.valid(e) ? prm.kv[(size_t)cl * prm.D + p.gidx(e)] : 0.0;
        ga.x[e] = p.valid(e) ? prm.kg[(size_t)cl * prm.D + p.gidx(e)] : 0.0;
      }
      Ha = prm.kham0[cl]; h = prm.kh[cl]; xi = prm.kxi[cl];
      needMacro = true;
    }

#pragma unroll
    for (int k = 0; k < 16; ++k) st_[k] = 0;
    st_last_ = 0;
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    st_last_ = __builtin_readcyclecounter();
#endif
    // fixedLeapFrog: every macro step is one tick, so leaf n of doubling i of an iteration that
    // starts at tick T0 runs at tick T0 + 2**i - 1 + n.  Starting iterations only at T0 = 1 (mod
    // 2**ALIGN_LOG2) makes the leaf parity, the sub-tree check schedule (levels 1..ALIGN_LOG2) and the
    // doubling ends of ALL chains of a group functions of the tick count alone: the expensive
    // divergent paths run on a few tick phases for everybody instead of on most ticks for somebody.
    // Costs (2**ALIGN_LOG2 - 1)/2 idle ticks per iteration; changes no chain's arithmetic.
    constexpr uint32_t ALIGN_MASK = ADAPT ? 0u : ((1u << P::ALIGN_LOG2) - 1u);
    uint32_t T = 0;
    while (p.any(live || needGen)) {
      WN_STAMP(7);
      // ---- A ----
      if (live && needIter && (mode() != MODE_SAMPLE || (T & ALIGN_MASK) == (1u & ALIGN_MASK))) start_iteration();
      T++;
      rng_refill();
      if (live && needDoubling) start_doubling();
      if (live && needMacro) setup_macro();
      if (ADAPT && live && needPass) setup_pass();
      // ---- B: one leapfrog micro step on the work state (adaptive integrators: up to 16 of them
      // back to back while at least half of the live chains are still inside a micro pass, so the
      // per-step bookkeeping below is not paid once per micro step of a 2**c-step pass) ----
      Vec& Q = QW(); Vec& V = VW(); Vec& G = GW();
      double part[NPB > 2 * NFAST ? NPB : 2 * NFAST];
      WN_STAMP(0);
      int kin = 0;
      do {
        const bool act = live && !needIter && (kin == 0 || inPass);
        double hh2 = act ? 0.5 * hh : 0.0, hq = act ? hh : 0.0;
        if (ADAPT && integ() == I_ADAPT_YOSHIDA && act) {
          // 4th-order composition (adaptiveIntegrators.py:143-144): sub-steps firstLast, middle, firstLast
          const double w = sub == 1 ? -1.702414383919315 : 1.351207191959658;
          hh2 = 0.5 * w * hh; hq = w * hh;
        }
        bool mid;
        flow_begin(act, mid, hh2, hq);
        WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; Q.x[e] = Q.x[e] + hq * V.x[e]; }
        double em[4], en[4];
        if constexpr (FLOW) {
          Vec xin, gout;
          flow_point(mid, xin);
          p.grad(xin, gout, part[0]);
          if (p.any(mid)) flow_errors(gout, em, en);
          WN_VFOR(e) G.x[e] = mid ? G.x[e] : gout.x[e];
        } else {
          p.grad(Q, G, part[0]);
        }
        orbit_sumsq_reduce(Q);
        WN_STAMP(1);
        double ke = 0.0;
        WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; ke += V.x[e] * V.x[e]; }
        part[1] = ke; part[2] = keinit_part; part[3] = gen_part;
        if (!ADAPT) {
          // fixedLeapFrog finalises its macro step every tick, so the on-chip U-turn checks ride
          // in the same reduction as the energies
          fin = act;
          if (p.any(nchecks() > 0)) {
            fast_dots(part + 4);
            p.template reduce<NPB>(part);
          } else {  // no chain of the group is at a check leaf on this tick
#pragma unroll
            for (int k = 4; k < NPB; ++k) part[k] = 0.0;
            p.template reduce<4>(part);
          }
        } else {
          // the initial kinetic energy / deferred generated() sum ride along only when one is pending
          if (p.any(needKE || needGen)) p.template reduce<NPB>(part);
          else p.template reduce<2>(part);
        }
        WN_STAMP(2);
        if constexpr (FLOW) if (p.any(mid)) {
          p.template reduce_max<4>(em);
          p.template reduce<4>(en);
          if (mid) flow_account(em, en);
        }
        finish_generated(part[3]);
        if (act) {
          if (needKE) {  // Hs[I0] = -f0 + 0.5*sum(v**2)  (:256); multinomialLscale (:259)
            H0 = -lp0 + 0.5 * part[2];
            Ha = Hb = H0; Ham0 = H0;
            if (ADAPT) { HamPass0 = H0; allfin = isfinite(H0); if (WARM) Hprev = H0; }
            Hnan = isnan(H0); Hmin = Hmax = H0;
            needKE = false; keinit_part = 0.0;
          }
          if (ADAPT) micro_account(p.lp_finish(part[0]), part[1]);
        }
        kin++;
      } while (ADAPT && kin < 16 && p.majority(live && inPass, live));
      // ---- C ----
      if (live && !needIter) {
        if (!ADAPT) after_micro(p.lp_finish(part[0]), part[1]);
        else if (!inPass) pass_end();
      }
      WN_STAMP(3);
      // ---- D: U-turn checks ----
      const int nchk = nchecks();
      firemask = 0;
      if (ADAPT) {
        if (p.any(nchk > 0)) {
          fast_dots(part);
          p.template reduce<2 * NFAST>(part);
          fast_fire(part, nchk);
        }
      } else {
        fast_fire(part + 4, nchk);
      }
      // deep sub-tree levels: left states come from the slot store, four checks per reduction
      for (int j0 = NFAST; p.any(nchk > j0); j0 += 4) {
        double dots[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          double d1 = 0.0, d2 = 0.0;
          if (p.any(nchk > j)) {
            if ((nchk > j) && need_check(j)) {
              Vec qo_, vo_;
              p.slot_load(slot_of_leaf(n + 1 - (1 << j)), qo_, vo_);
              uturn_dots(qo_, vo_, d1, d2);
            }
          }
          dots[2 * jj] = d1; dots[2 * jj + 1] = d2;
        }
        p.template reduce<8>(dots);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          // stopCondition (WALNUTS.py:95-97) with the older state on the side opposite to xi
          if (nchk > j && need_check(j) &&
              (((double)xi * dots[2 * jj] < 0.0) || ((double)xi * dots[2 * jj + 1] < 0.0)))
            firemask |= (1u << j);
        }
      }
      WN_STAMP(4);
      // ---- E ----
      if (live && fin) {
        if (mode() == MODE_SAMPLE) step_done();
        else kat_done();
        if (endIter) end_iteration();
      }
      WN_STAMP(5);
      st_[6] += 1;
    }
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (prm.stamps && (threadIdx.x & 63) == 0) {
      unsigned long long* o = prm.stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
      for (int k = 0; k < 16; ++k) o[k] = st_[k];
    }
#endif
    if (trace_n() && inrange && p.writer()) trace_n()[p.chain_local()] = ntrace;
  }

  // ------------------------------------------------------------------------------------------
  // run() for ONE chain per wavefront, adaptive integrators, plain sampling (SPEC >= 0): the same member functions in
  // the same order, but driven by nested loops -- iteration / doubling / leaf / pass / micro step, as the reference is
  // written -- instead of one flag-driven tick loop.  The tick form exists so that chains SHARING a wave or a workgroup
  // stay in lockstep on the gradient; a chain that has its wave to itself only pays for it: every needX flag is
  // loop-carried state that the compiler keeps as a lane mask and merges with and/andn2/or at every assignment.  Here
  // the program counter is the state.  Same arithmetic, same reductions, same draw order: bit-identical to run().
  // ------------------------------------------------------------------------------------------
  WN_HD void run_nested() {
    static_assert(!FLOW && !WARM && SPEC >= 0, "run_nested: leapfrog integrators, plain sampling");
    const bool inrange = p.chain_local() < prm.C;
    init_scalars(inrange);
    const int cl = inrange ? p.chain_local() : 0;
    WN_VFOR(e) qa.x[e] = (p.valid(e) && inrange) ? prm.q0[(size_t)cl * prm.D + p.gidx(e)] : 0.0;
    double part[NPB > 2 * NFAST ? NPB : 2 * NFAST];
    {  // q_c = q0; f0, grad0 = lpFun(q0)  (WALNUTS.py:177,249)
      p.grad(qa, ga, part[0]);
      p.template reduce<1>(part);
      lpP = p.lp_finish(part[0]);
      p.prop_put(ip, qa, ga);
      if (live) {
        emit_generated(0);  // samples[:,0] = generated(q0)  (:164)
        if (prm.numIter < 1) live = false; else needIter = true;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) st_[k] = 0;
    st_last_ = 0;
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    st_last_ = __builtin_readcyclecounter();
#endif
    Vec& Q = QW(); Vec& V = VW(); Vec& G = GW();
    while (p.any(live)) {                // ---- iterations (WALNUTS.py:189) ----
      start_iteration();
      {  // Hs[I0] = -f0 + 0.5*sum(v**2)  (:256), multinomialLscale (:259), and the deferred generated() sum of the
         // previous draw: one reduction per iteration here (the tick loop lets them ride on the first micro step's)
        part[2] = keinit_part; part[3] = gen_part;
        p.template reduce<2>(part + 2);
        finish_generated(part[3]);
        H0 = -lp0 + 0.5 * part[2];
        Ha = Hb = H0; Ham0 = H0;
        if (ADAPT) { HamPass0 = H0; allfin = isfinite(H0); }
        Hnan = isnan(H0); Hmin = Hmax = H0;
        needKE = false; keinit_part = 0.0;
      }
      for (;;) {                         // ---- doublings (:281) ----
        if (p.any(needDoubling)) start_doubling();
        for (;;) {                       // ---- leaves of the doubling: one macro step each ----
          WN_STAMP(7);
          rng_refill();
          setup_macro();
          // ---- the integrator's searches, written as the reference writes them (adaptiveIntegrators.py:67-137 /
          // 363-475) instead of through pass_end()'s phase machine: forward search over c, adaptLeapFrogR2P's draw
          // and re-integration at c = If+1, backward search from the accepted result ----
          auto run_pass = [&](const int cc) -> bool {  // one micro pass of 2**cc leapfrog steps from the state `ph` names
            c = cc;
            setup_pass();
            WN_STAMP(0);
            do {                         // ---- one leapfrog micro step (:72-84) ----
              double hh2 = 0.5 * hh, hq = hh;
              if (integ() == I_ADAPT_YOSHIDA) {
                const double w = sub == 1 ? -1.702414383919315 : 1.351207191959658;
                hh2 = 0.5 * w * hh; hq = w * hh;
              }
              WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; Q.x[e] = Q.x[e] + hq * V.x[e]; }
              p.grad(Q, G, part[0]);
              double ke = 0.0;
              WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; ke += V.x[e] * V.x[e]; }
              part[1] = ke;
              p.template reduce<2>(part);
              micro_account(p.lp_finish(part[0]), part[1]);
            } while (p.any(inPass));
            WN_STAMP(1);
            // pass complete: all(isfinite(Hams)) and abs(Hams[0]-Hams[-1]) < delta   (:87-94)
            return allfin && (fabs(HamPass0 - HCur) < dc);
          };
          if constexpr (!ADAPT) {      // fixedLeapFrog (adaptiveIntegrators.py:49-59): the macro step is one leapfrog step, in place
            const double hh2 = 0.5 * hh, hq = hh;
            WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; Q.x[e] = Q.x[e] + hq * V.x[e]; }
            p.grad(Q, G, part[0]);
            double ke = 0.0;
            WN_VFOR(e) { V.x[e] = V.x[e] + hh2 * G.x[e]; ke += V.x[e] * V.x[e]; }
            part[1] = ke;
            p.template reduce<2>(part);
            after_micro(p.lp_finish(part[0]), part[1]);
            WN_STAMP(3);
          } else {
            int cc = cmin();             // forward search (:69-94 / 366-389); ph = PH_FWD from setup_macro()
            bool ok;
            for (;;) {
              ok = run_pass(cc);
              if (p.any(ok) || cc >= prm.maxC) break;
              ++cc;
            }
            If = ok ? cc : prm.maxC;     // Q5: the last computed state is used silently
            if (integ() == I_ADAPT_R2P) {
              const double u = next_u();  // :392
              if (p.any(u < prm.R2Pprob0)) {  // :393-399, 430-433
                cSim = If; lpOut = lpCur; HOut = HCur; osq_out = osq_cur;
                maxTry = If - 1; Ib = If; lwtf = prm.logP;
              } else {                   // :400-424, 434-437: re-integrate at c = If+1
                ph = PH_FWD2;
                run_pass(If + 1);
                cSim = If + 1; lpOut = lpCur; HOut = HCur; osq_out = osq_cur;
                maxTry = prm.maxC; Ib = prm.maxC; lwtf = prm.log1mP;
              }
            } else {                     // adaptLeapFrogD :96-111
              cSim = If; lpOut = lpCur; HOut = HCur; osq_out = osq_cur;
              maxTry = If - 1; Ib = If; lwtf = 0.0;
            }
            if (p.any(maxTry >= cmin())) {  // backward search from (qOut, -vOut, gOut)  (:107-132 / 440-464)
              // (the accepted forward result is the work state itself: it is saved only when backward passes are about to overwrite it)
              p.template warm_put<WARM_QO>(qw); p.template warm_put<WARM_VO>(vw); p.template warm_put<WARM_GO>(gw);
              ph = PH_BWD;
              for (cc = cmin();; ++cc) {
                ok = run_pass(cc);
                if (p.any(ok)) { Ib = cc; break; }
                if (p.any(cc >= maxTry)) break;
              }
            }
            needPass = false;
            finalize_macro();            // restores the accepted forward result after backward passes; lwt (:135-137 / 466-475)
            WN_STAMP(3);
          }
          // ---- U-turn checks of the finalised step ----
          const int nchk = nchecks();
          firemask = 0;
          if (p.any(nchk > 0)) {
            fast_dots(part);
            // (widths by need: the full-orbit check 0 is due at the last leaf of a doubling only)
            if (NFAST == 2 && !p.any(need_check(0))) p.template reduce<2>(part + 2);
            else p.template reduce<2 * NFAST>(part);
            fast_fire(part, nchk);
          }
          for (int j0 = NFAST; p.any(nchk > j0); j0 += 4) {  // deep sub-tree levels, four checks per reduction
            double dots[8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = j0 + jj;
              double d1 = 0.0, d2 = 0.0;
              if ((nchk > j) && need_check(j)) {
                Vec qo_, vo_;
                p.slot_load(slot_of_leaf(n + 1 - (1 << j)), qo_, vo_);
                uturn_dots(qo_, vo_, d1, d2);
              }
              dots[2 * jj] = d1; dots[2 * jj + 1] = d2;
            }
            if (p.any(nchk > j0 + 2)) p.template reduce<8>(dots);
            else p.template reduce<4>(dots);  // one or two levels of this group are due
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = j0 + jj;
              if (nchk > j && need_check(j) &&
                  (((double)xi * dots[2 * jj] < 0.0) || ((double)xi * dots[2 * jj + 1] < 0.0)))
                firemask |= (1u << j);
            }
          }
          WN_STAMP(4);
          step_done();                   // NUTS bookkeeping (:296-648); ends the doubling / the iteration when due
          WN_STAMP(5);
          st_[6] += 1;
          if (!p.any(needMacro)) break;
        }
        if (p.any(endIter)) break;
      }
      end_iteration();
    }
#if defined(WN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (prm.stamps && (threadIdx.x & 63) == 0) {
      unsigned long long* o = prm.stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
      for (int k = 0; k < 16; ++k) o[k] = st_[k];
    }
#endif
    if (p.any(needGen)) {  // the deferred generated() sum of the last draw
      part[0] = 0.0; part[1] = 0.0; part[2] = keinit_part; part[3] = gen_part;
      p.template reduce<NPB>(part);
      finish_generated(part[3]);
    }
  }

  WN_HD void kat_done() {  // wn_integrator_step: return integratorReturn fields
    const int cl = p.chain_local();
    WN_VFOR(e) if (p.valid(e) && p.owner(e)) {
      prm.kq1[(size_t)cl * prm.D + p.gidx(e)] = QN().x[e];
      prm.kv1[(size_t)cl * prm.D + p.gidx(e)] = (double)xi * VN().x[e];
      prm.kg1[(size_t)cl * prm.D + p.gidx(e)] = GN().x[e];
    }
    if (p.writer()) {
      int* io = prm.kiout + (size_t)cl * 5;
      io[0] = nEvF; io[1] = nEvB; io[2] = If; io[3] = Ib; io[4] = cret;
      double* dd = prm.kdout + (size_t)cl * 4;
      dd[0] = lpn; dd[1] = lwt; dd[2] = Hn;
      // igrConst (adaptiveIntegrators.py:59 / 101 / 399 / 424 / 300); the adaptive form needs the WARM
      // instantiation, which tracks max|diff(Hams)| of the accepted pass
      dd[3] = ADAPT ? (WARM ? igrOut : NAN) : h * pow(fmax(1.0e-10, fabs(Ham0 - Hn)), -1.0 / 3.0);
    }
    fin = false;
    live = false;
  }
};

}  // namespace wn
