// wn_k_orbit.hip -- one chain per wavefront for targets of at most 16 dimensions (funnel10: BASELINE config 3; the 2-D toys of
// targetDistr.py), plain sampling runs with adaptLeapFrogD / adaptLeapFrogR2P / fixedLeapFrog: TRAJECTORY FIRST, BOOKKEEPING IN BATCHES.
//
// What bounds these launches is one wave's serial latency: the launch lasts as long as its busiest chain (config 3: 3.9e5 gradient
// evaluations against a median of 2.2e4), and that chain is NOT a deep-refinement chain -- 94 % of its macro steps have If = 0 and
// take one evaluation (or four when adaptLeapFrogR2P re-integrates at c = 1); it is a LONG-ORBIT chain, ~580 leaves per iteration.
// In the one-stream engine (wn_chain.h run_nested) 4.8 k of the 6.9 k cycles of such a macro step are the NUTS bookkeeping of
// WALNUTS.py:392-587 -- weights, exp, the multinomial draw, U-turn dots, diagnostics -- executed as scalar code replicated over the
// wave, one instruction per ~5 cycles, between two macro steps of a trajectory that does not depend on it.
//
// Here the wave alternates two phases over a BLOCK of 16 leaves (leaf n of a doubling is entry n & 15 of a ring in LDS):
//   T phase   the integrator calls of up to 16 consecutive leaves back to back (adaptiveIntegrators.py:49-59 / 65-137 / 361-475):
//             step sizes and the R2P uniform come out of the wave's window of uniforms at the index the reference's draw order gives
//             them, assuming every leaf takes its multinomial draw exactly like the leaf before it did (WALNUTS.py:426: the uniform is
//             drawn only when WnewSum > exp(-699)); each leaf's (q', v', lp, H, If, Ib, c, lwt, nEval, h, its multinomial uniform)
//             goes to the ring.
//   R phase   the bookkeeping of those leaves LANE-PARALLEL, lane k = leaf k: the running sums the reference builds leaf by leaf
//             (lwtSum, orbitLen_, timeLen, WnewSum) are serial prefix sums in the reference's order -- x_k + (prefix of k-1), one DPP
//             row shift per leaf, three quantities in three rows of the register -- so every partial sum has the reference's
//             bits; exp, the acceptance ratio's division and the draw test run once for 16 leaves; the sub-tree U-turn checks of
//             levels 1..4 (subTreePlan, WALNUTS.py:22-41: 8 + 4 + 2 + 1 per block, all of them between ring entries), the deeper
//             levels of the block's last leaf (left states from the slot store) and the full-orbit check are one lane each; wavefront
//             ballots turn them into "first leaf whose check fired", "first non-finite energy", "first leaf whose draw differed from
//             the prediction" and "last leaf picked".  Leaves behind the first such event are discarded: a sub-U-turn or a non-finite
//             energy ends the iteration there exactly as WALNUTS.py:316-319 / 597-605 do; a mispredicted draw restarts the T phase
//             at the next leaf with the corrected draw index (the state is in the ring).
// Draw order, nevalF / nevalB, If / Ib / c and every diagnostic are the reference's: leaves the T phase ran ahead of an event are
// never counted.  The arithmetic of a micro step and of every sum is the one-stream engine's, so the two kernels agree bit for bit
// (tests/test_gpu_orbit.py) -- WN_ORBIT_LEGACY=1 runs the old kernel.
//
// A wave that has its SIMD to itself pays 25-55 cycles for EVERY branch, taken or not (tools/branch_cost.hip,
// profiles/r04_branch_cost.txt: a not-taken s_cbranch 26, an exec-mask region 37, v_cmp -> vcc -> branch 40-55, against 5.75 for a
// dependent v_fma_f64), so the T phase is written to be straight: the draws of a whole batch are made lane-parallel before it (one
// Philox block per lane, three ds_bpermute gathers), a leaf's scalars go into lane k of a few registers with selects (the R phase
// reads them where they are), the ring stores need no exec mask (lanes >= 16 write to a dump area), a non-finite energy shortens
// the batch through a scalar select, and the common macro step (first pass accepted, no R2P re-integration, no backward search) is
// one branch away from its end.
//
// LDS per chain (doubles): ring q / v 2 x 16 x 17, dump 64, passive orbit end (q, v, g) 48, two proposals (q) 32, sub-tree slots
// (levels >= 5 and leaf 0) 32 (M + 1): 8.4 KB at M = 12 -- 16 chains per CU, all 4096 resident.
// The proposal keeps q only: the gradient at the start of an iteration is re-evaluated, as the reference does (WALNUTS.py:249).
//
// WIDE MODE (NE > 1; round 5, BASELINE config 5: StockWatson, D = 756 in NE = 12 registers per lane): the same kernel for a chain whose
// vectors fill the wave.  What changes is where vectors live and how a U-turn check is formed, nothing in the bookkeeping:
//   * q, v, g are NE registers per lane (element e of a lane: Tgt::map_gidx, or e * 64 + lane); sums are wave sums in the one-stream
//     engine's order (WavePolicy::reduce), so energies have that engine's bits;
//   * the ring, the passive orbit end, the proposals and the slots are in the chain's workspace ([vector][NE][64]: a register of the wave
//     is 512 contiguous bytes), L2-resident: 16 x 2 ring entries + 3 + 2 + 2 (M + 1) vectors = 67 x 6 KB at M = 14;
//   * a U-turn check is a wave-wide dot: the R phase walks the odd leaves of the batch (right state loaded once), the levels due at each
//     (left state from the ring, levels >= 5 from the slots) and the full-orbit check -- scalar loop, uniform branches.
// The one-stream engine spent 3.1 k of a fixedLeapFrog leaf's 7.3 k cycles on its replicated scalar bookkeeping (mainSW.py:56-62).
#include "wn_launch.h"
#include "wn_policy_wave.h"
#include "wn_orbit.h"
#include "wn_target_denselds.h"

namespace wn {


template <class T_, class = void> struct tgt_staged { static constexpr bool value = false; };
template <class T_> struct tgt_staged<T_, typename std::enable_if<(T_::STAGE_DOUBLES > 0)>::type> { static constexpr bool value = true; };
template <class T_, class = void> struct tgt_batches { static constexpr bool value = false; };
template <class T_> struct tgt_batches<T_, typename std::enable_if<T_::BATCH_ADAPT>::type> { static constexpr bool value = true; };

// K: chains (wavefronts) per workgroup -- wide mode with a target that stages data in LDS for the workgroup's chains (dense Gaussian)
// WARM: the instantiation that also runs the warm-up iterations (WALNUTS.py:701-712: step size H from the P2 quantile of log(igrConst), delta
// from the quantile of the orbits' energy errors -- wn_p2_push / wn_adapt_update of wn_chain.h, the functions every other kernel uses)
template <class Tgt, int INTEG, bool FUSED = false, int NE = 1, int K = 1, bool WARM = false>
__global__ __launch_bounds__(64 * K) void wn_orbit_kernel(Params prm) {
  using namespace orbit;
  constexpr bool WIDE = NE > 1;
  static_assert(!(WIDE && FUSED), "a target's own micro step is a one-register form");
  constexpr int DP = NE * 64;
  extern __shared__ double wn_orbit_lds[];
  lds_d* const L = (lds_d*)wn_orbit_lds;
  static_assert(K == 1 || NE > 1, "several chains per workgroup: wide mode only (the one-register form keeps its vectors in dynamic LDS)");
  const int lane = threadIdx.x & 63;
  const int k16 = lane & 15;
  const int row = lane >> 4;
  // A staged target (the dense Gaussian: the matrix in LDS, ONE workgroup per CU) runs as a persistent grid with a chain queue: a wave whose
  // chain has finished takes the next chain of the launch (one atomic add on `prm.dynq`) instead of idling until the slowest of its
  // workgroup's K chains is done and the next workgroup can start -- the matrix is staged once per workgroup and launch.  A chain's result
  // does not depend on which wave runs it (random stream, workspace, output rows: all keyed by the chain).  Other targets: one chain per wave.
  constexpr bool QUEUE = tgt_staged<Tgt>::value && K > 1 && !WARM;  // (the WARM instantiations sit at the 256-register cap of two waves per SIMD: the loop's extra live state spilled 13 more registers there)
  int cl_ = blockIdx.x * K + (int)(threadIdx.x >> 6);
  const uint32_t seedlo = (uint32_t)prm.seed, seedhi = (uint32_t)(prm.seed >> 32);
  const int D = prm.D, M = prm.M, minC = prm.minC, maxC = prm.maxC;
  Tgt tgt;
  if constexpr (tgt_staged<Tgt>::value) {
    __shared__ double wn_orbit_stage[Tgt::STAGE_DOUBLES];
    Tgt::stage(prm, wn_orbit_stage, (int)threadIdx.x, 64 * K);
    __syncthreads();
    if (cl_ >= prm.C) return;  // (no barrier behind this point in wide mode)
    tgt.init(prm, lane, wn_orbit_stage);
  } else {
    if (K > 1 && cl_ >= prm.C) return;
    tgt.init(prm, lane, nullptr);
  }
  for (;;) {  // (QUEUE: one trip per chain this wave runs; otherwise one trip)
  const int cl = cl_;
  const uint32_t cg = prm.chain0 + (uint32_t)cl;
  if constexpr (!WIDE) {
    for (int k = lane; k < lds_doubles(M); k += 64) L[k] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- the wave's window of uniforms: lane l holds uniform 2 * rb_base + l of iteration rb_iter (wn_rng.h stream 0) ----
  uint32_t rb_iter = 0xFFFFFFFFu, rb_base = 0u;
  double ru0 = 0.0;
  auto rng_ensure = [&](uint32_t it, uint32_t k) {  // uniforms k .. k+7 inside the window afterwards
    const uint32_t blk = (uint32_t)uni_i((int)(k >> 1));
    if (it != rb_iter || blk - rb_base > 28u) {
      rb_iter = it; rb_base = blk;
      Philox4 r = philox4x32_10(blk + (uint32_t)(lane >> 1), it, cg, 0u, seedlo, seedhi);
      ru0 = (lane & 1) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
    }
  };
  auto uni = [&](uint32_t k) -> double { return wn_readlane(ru0, (int)((uint32_t)uni_i((int)k) - 2u * rb_base)); };
  // the sums of a micro step, exactly as the one-stream engine forms them (WavePolicy::reduce<2>, VLDS): two quantities in the
  // rows of one register, one row reduction
  auto reduce2 = [&](double& a, double& b) {
    if constexpr (WIDE) { a = wn_wavesum(a); b = wn_wavesum(b); return; }
    double r = wn_pack32(a, b);
    r = wn_rowsum16(r);
    a = wn_readlane(r, 0) + 0.0;
    b = wn_readlane(r, 32) + 0.0;
  };
  auto reduce1 = [&](double a) -> double {
    if constexpr (WIDE) return wn_wavesum(a);
    else return wn_readlane(wn_rowsum16(a), 0) + 0.0;
  };
#define WN_OV(e) _Pragma("unroll") for (int e = 0; e < NE; ++e)

  // ---- chain state ----
  // vectors: one element per lane (rows 1..3 hold copies of row 0) -- wide mode: NE elements per lane; active end of the orbit, v in
  // forward-time orientation
  double qa[NE], va[NE], ga[NE];
  WN_OV(e) va[e] = 0.0;
  // scalars of WALNUTS.py:189-695 (uniform over the wave)
  double H0 = 0.0, Ha = 0.0, Hb = 0.0, lp0, WoldSum = 1.0, WnewSum = 0.0, lwtSumf = 0.0, lwtSumb = 0.0;
  double orbitLen_ = 0.0, orbitLenSam_ = 0.0, indexStat_ = 0.0, indexStatOld_ = 0.0, timeLenF_ = 0.0, timeLenB_ = 0.0;
  double minlw = INFINITY, maxlw = -INFINITY, Hmin = INFINITY, Hmax = -INFINITY, lpP = 0.0, lpPL = 0.0;
  double gen_sum = 0.0;
  int iter = 1, i = 0, xi = 1, a = 0, b = 0, n = 0, nleaf = 1;
  uint32_t Bbits = 0, ku = 0, khc = 0;
  int nevalF = 0, nevalB = 0, L_ = 0, Lold_ = 0, NdS = 0, NdC = 0, maxFint = 0, maxBint = 0, stopCode = 0;
  int minIf = 1 << 30, maxIf = -(1 << 30), minc = 1 << 30, maxc = -(1 << 30), nneq = 0, nzero = 0, nused = 0;
  int ip = 0, ipl = 0;
  bool Hnan = false, bothEndsPassive = false;
  bool predF = true, predB = true;  // "the next leaf in this direction takes its multinomial draw"
  const bool inrange = cl < prm.C;
  const int clc = inrange ? cl : 0;
  double dlt = prm.delta, Hc = prm.H;  // (WARM: adapted at the end of every warm-up iteration)
  double p2q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  int p2m[5] = {1, 2, 3, 4, 5}, p2n = 0;
  double igr_pass = 0.0;  // WARM: igrConst of the pass run_pass ran last (adaptiveIntegrators.py:101 / 399 / 424)
  (void)p2q; (void)p2m; (void)p2n; (void)igr_pass;

  // element held by register e of this lane (parameter index), or -1
  auto gidx = [&](int e) -> int {
    if constexpr (!WIDE) return k16 < D ? k16 : -1;
    else if constexpr (tgt_mapped<Tgt>::value) return tgt.map_gidx(e, lane);
    else return e * 64 + lane < D ? e * 64 + lane : -1;
  };
  // the chain's stored vectors: LDS offsets (doubles) in the one-register form, vector numbers of the workspace in wide mode
  // (a stored vector is [register pair][lane][2]: one 16-byte access per lane and pair of registers, 1 KB per wave instruction)
  typedef double d2_t __attribute__((ext_vector_type(2)));
  static_assert(!WIDE || NE % 2 == 0, "wide mode stores register pairs");
  d2_t* const Wk = WIDE ? reinterpret_cast<d2_t*>(prm.work + (size_t)clc * wide_doubles(NE, M)) + lane : nullptr;
  constexpr int O_COLDQ = WIDE ? W_COLDQ : COLDQ, O_COLDV = WIDE ? W_COLDV : COLDV, O_COLDG = WIDE ? W_COLDG : COLDG;
  auto o_prop = [&](int w) -> int { return WIDE ? W_PROP + w : PROP + 16 * w; };
  auto o_ringq = [&](int k) -> int { return WIDE ? W_RQ + k : RQ + RS * k; };
  auto o_ringv = [&](int k) -> int { return WIDE ? W_RV + k : RV + RS * k; };
  auto vst = [&](int off, const double* x) {
    if constexpr (WIDE) {
#pragma unroll
      for (int j = 0; j < NE / 2; ++j) { d2_t t; t[0] = x[2 * j]; t[1] = x[2 * j + 1]; Wk[(size_t)off * (DP / 2) + j * 64] = t; }
    } else { if (lane < 16) L[off + lane] = x[0]; }
  };
  auto vld = [&](int off, double* x) {
    if constexpr (WIDE) {
#pragma unroll
      for (int j = 0; j < NE / 2; ++j) { const d2_t t = Wk[(size_t)off * (DP / 2) + j * 64]; x[2 * j] = t[0]; x[2 * j + 1] = t[1]; }
    } else x[0] = L[off + k16];
  };

  // samples[:, it] = generated(q)
  auto emit_generated = [&](int it, const double* qP) {
    double* rowp = prm.samples + ((size_t)clc * (prm.numIter + 1) + it) * prm.dg;
    if constexpr (WIDE) {
      // (arithmetic, not a select of the two kernel arguments: see below)
      const int fk = (int)(prm.generated == G_FIRSTK), sq = (int)(prm.generated == G_Q0_SUMSQ);
      const int kk = sq ? 1 : fk * prm.gen_k + (1 - fk) * D;
      WN_OV(e) { const int gi = gidx(e); if (gi >= 0 && gi < kk) rowp[gi] = qP[e]; }
      if (sq) {
        double s = 0.0;
        WN_OV(e) s += qP[e] * qP[e];
        s = reduce1(s);
        if (lane == 0) rowp[1] = s;
      }
    } else if (prm.generated == G_Q0_SUMSQ) {
      const double s = reduce1(qP[0] * qP[0]);
      if (lane == 0) { rowp[0] = qP[0]; rowp[1] = s; }
    } else {
      // (arithmetic, not a select of the two kernel arguments: hipcc turned the select into a two-entry array in scratch)
      const int fk = (int)(prm.generated == G_FIRSTK);
      const int kk = fk * prm.gen_k + (1 - fk) * D;
      if (lane < 16 && lane < kk && lane < D) rowp[lane] = qP[0];
    }
  };

  // ---- q_c = q0; f0, grad0 = lpFun(q0)  (WALNUTS.py:177,249) ----
  WN_OV(e) { const int gi = gidx(e); qa[e] = (gi >= 0 && inrange) ? prm.q0[(size_t)clc * D + gi] : 0.0; }
  {
    double lp_part = tgt.grad(qa, ga, lane);
    lpP = Tgt::lp_finish(reduce1(lp_part));
    vst(o_prop(ip), qa);
    if (inrange) emit_generated(0, qa);  // samples[:,0] = generated(q0)  (:164)
  }
  bool live = inrange && prm.numIter >= 1;
  int evals_total = 0, prio_now = 0;  // (longest job first: see the end of the iteration)
  unsigned long long st_[16], st_last_ = 0;  // diagnostic builds (-DWN_STAMPS): cycles per phase
#pragma unroll
  for (int k = 0; k < 16; ++k) st_[k] = 0;
#if defined(WN_STAMPS)
  st_last_ = __builtin_readcyclecounter();
#endif

  // one micro pass of 2**cc leapfrog steps (adaptiveIntegrators.py:72-94) from (qs, vs, gs); returns "the pass is accepted"
  // (the sums are the same in every lane)
  double hcur = 0.0;  // step size h of the running macro step
  auto run_pass = [&](const double* qs, const double* vs, const double* gs, double HamStart, int cc, double* q, double* v, double* g, double& lpC,
                      double& HC, int& nev) -> bool {
    WN_OV(e) { q[e] = qs[e]; v[e] = vs[e]; g[e] = gs[e]; }
    const double hh = ldexp(hcur, -cc);  // h / 2**c (exact scaling either way)
    const int nstep = 1 << cc;
    bool allfin = isfinite(HamStart);
    double Hprev = HamStart, madv = -INFINITY;  // WARM: np.max(np.abs(np.diff(Hams))) of this pass
    bool madnan = false;
    const double hh2 = 0.5 * hh, hq = hh;
    int s = nstep;
#pragma unroll 1
    do {  // one branch per micro step
      double lpp, ke;
      if constexpr (FUSED) {  // the target's own micro step: one row reduction instead of two (WTgtFunnel10::leap)
        tgt.leap(q[0], v[0], g[0], hh2, hq, lane, lpp, ke);
      } else {
        WN_OV(e) { v[e] = v[e] + hh2 * g[e]; q[e] = q[e] + hq * v[e]; }
        lpp = tgt.grad(q, g, lane);
        ke = 0.0;
        WN_OV(e) { v[e] = v[e] + hh2 * g[e]; ke += v[e] * v[e]; }
        reduce2(lpp, ke);
        lpp = Tgt::lp_finish(lpp);
      }
      const double Hi = -lpp + 0.5 * ke;  // Hams[i] = -fnew + 0.5*sum(vv*vv)
      lpC = lpp; HC = Hi;
      allfin = allfin && isfinite(Hi);
      if constexpr (WARM) {
        const double df = fabs(Hi - Hprev);
        madnan = madnan || isnan(df);
        madv = df > madv ? df : madv;  // (a NaN difference compares false: it is carried by madnan)
        Hprev = Hi;
      }
    } while (--s);
    if constexpr (WARM) igr_pass = hh * pow(madnan ? NAN : madv, -1.0 / 3.0);
    nev += nstep;
    return uni_b(allfin && (fabs(HamStart - HC) < dlt));  // all(isfinite(Hams)) and abs(Hams[0]-Hams[-1]) < delta  (:87-94)
  };
  // ---- per-leaf records of the running block, lane k (of every row) = leaf k: written by the T phase with selects, read by the R
  // phase where they are ----
  double HnV = 0.0, lpV = 0.0, hV = 0.0, upV = 0.0, urV = 0.0, lgV = 0.0;  // (lgV, WARM: log(igrConst) of the leaf, WALNUTS.py:313)
  (void)lgV;
  int iA = 0, iB = 0;  // If | Ib << 8 | c << 16;  nEvalF | nEvalB << 16
  // ring stores without an exec mask: lanes >= 16 write to the dump area
  const int ring_base = lane < 16 ? RQ + lane : DUMP + lane;
  const int ring_step = lane < 16 ? RS : 0;
  const int ring_vofs = lane < 16 ? RV - RQ : 0;

  while (__any(live)) {  // ---- iterations (WALNUTS.py:189) ----
    WN_STAMP(9);
    // ---- iteration start (:196-276) ----
    rng_ensure((uint32_t)iter, 0u);  // uniforms 0..63: the M <= 24 direction bits and the first steps
    Bbits = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)M; ++k)  // :216  B = floor(uniform(0,2,size=M))
      if ((int)floor(0.0 + (2.0 - 0.0) * uni(k)) != 0) Bbits |= (1u << k);
    Bbits = (uint32_t)uni_i((int)Bbits);
    ku = (uint32_t)M;
    a = 0; b = 0; maxFint = 0; maxBint = 0;
    {
      vld(o_prop(ip), qa);
      // recordOrbitStats (WALNUTS.py:182-184, 274-276): orbitMin / orbitMax[:, iterN-1] start at generated(qc); the one-register form,
      // generated = identity / first k (mainFunnel.py, mainFunnelOrbitLen.py, mainFunnelTransient.py run the funnel this way)
      if constexpr (!WIDE) if (prm.omin) {
        const int fk = (int)(prm.generated == G_FIRSTK);
        const int kk = fk * prm.gen_k + (1 - fk) * D;
        const size_t orow = ((size_t)clc * prm.numIter + (iter - 1)) * prm.dg;
        if (lane < 16 && lane < kk && lane < D) { prm.omin[orow + lane] = qa[0]; prm.omax[orow + lane] = qa[0]; }
      }
      double lpp = tgt.grad(qa, ga, lane);  // f0, grad0 = lpFun(qc)  (:249); its value is lpP already
      (void)lpp;
      // :236  v = normal(size=d)
      if constexpr (NE > 4) {  // (rolled: one copy of Philox + Box-Muller in the code; the register is picked with selects, not a dynamic index)
#pragma unroll 1
        for (int e1 = 0; e1 < NE; ++e1) {
          int gi = -1;
          WN_OV(e) gi = e == e1 ? gidx(e) : gi;
          const double z = gi >= 0 ? rng_normal(prm.seed, cg, (uint32_t)iter, (uint32_t)gi) : 0.0;
          WN_OV(e) va[e] = e == e1 ? z : va[e];
        }
      } else {
        WN_OV(e) { const int gi = gidx(e); va[e] = gi >= 0 ? rng_normal(prm.seed, cg, (uint32_t)iter, (uint32_t)gi) : 0.0; }
      }
      vst(O_COLDQ, qa); vst(O_COLDV, va); vst(O_COLDG, ga);
      double kp = 0.0;
      WN_OV(e) kp += va[e] * va[e];
      double gp = 0.0;
      reduce2(kp, gp);
      lp0 = lpP;
      H0 = -lp0 + 0.5 * kp;  // Hs[I0] = -f0 + 0.5*sum(v**2)  (:256); multinomialLscale (:259)
      Ha = Hb = H0;
      Hnan = isnan(H0); Hmin = Hmax = H0;
    }
    WoldSum = 1.0; WnewSum = 0.0; lwtSumb = 0.0; lwtSumf = 0.0;  // :260-264
    stopCode = 0; bothEndsPassive = false;
    nevalF = nevalB = 0; Lold_ = L_ = 0; orbitLen_ = orbitLenSam_ = 0.0; NdS = NdC = 0;
    indexStat_ = indexStatOld_ = 0.0; timeLenF_ = timeLenB_ = 0.0;
    minIf = minc = 1 << 30; maxIf = maxc = -(1 << 30); nneq = nzero = nused = 0;
    minlw = INFINITY; maxlw = -INFINITY;
    predF = predB = true;
    i = 0;
    xi = (Bbits & 1u) ? -1 : 1;
    bool endIter = false;
    WN_STAMP(8);

    while (!endIter) {  // ---- doublings (:281) ----
      {  // start of doubling i (:283-294)
        const int nxi = ((Bbits >> i) & 1u) ? -1 : 1;
        if (nxi != xi) {  // the other end of the orbit becomes the active one
          double tq[NE], tv[NE], tg[NE];
          vld(O_COLDQ, tq); vld(O_COLDV, tv); vld(O_COLDG, tg);
          vst(O_COLDQ, qa); vst(O_COLDV, va); vst(O_COLDG, ga);
          WN_OV(e) { qa[e] = tq[e]; va[e] = tv[e]; ga[e] = tg[e]; }
          const double th = Ha; Ha = Hb; Hb = th;
          xi = nxi;
        }
        n = 0; nleaf = 1 << i;
        ipl = ip; lpPL = lpP; Lold_ = L_; indexStatOld_ = indexStat_;
        if (i > 0) WnewSum = 0.0;  // :390
      }
      const bool fw = xi > 0;
      bool fire0 = false;  // full-orbit U-turn (:622), due at the last leaf of the doubling
      bool sub = false;

      while (n < nleaf) {  // ---- blocks of the doubling ----
        const int nb0 = n & ~15;
        const int k0 = n & 15;
        int k1 = nleaf - nb0 < 16 ? nleaf - nb0 : 16;
        // wide mode, adaptive integrators: a macro step is tens of evaluations of ~4 k cycles each and the bookkeeping of a leaf a
        // twentieth of that, so nothing is gained by running ahead -- and a mispredicted draw (StockWatson: a fifth of the blocks: the
        // first leaves of a doubling often have weight zero, If != Ib) or a sub-tree U-turn would throw whole macro steps away
        if constexpr (WIDE && INTEG != I_FIXED && !tgt_batches<Tgt>::value) k1 = k0 + 1;
        // =========================== the draws of leaves k0 .. k1-1, lane-parallel ===========================
        // In the reference's order an even leaf of a doubling i >= 1 draws the pair's two step sizes (:395), every leaf its
        // R2P uniform (adaptiveIntegrators.py:392) and -- when WnewSum > exp(-699) -- its multinomial uniform (:426); the single leaf
        // of doubling 0 draws one step size (:298) and its R2P uniform.  With "every leaf of the batch takes / skips the multinomial
        // draw" (pred) the index of every draw is a closed form of the leaf's position; the R phase checks the assumption.
        {
          const bool pred = fw ? predF : predB;
          const uint32_t r_ = INTEG == I_ADAPT_R2P ? 1u : 0u, p_ = (pred && i != 0) ? 1u : 0u;
          const uint32_t ce = (i != 0 ? 2u : 1u) + r_ + p_, co = r_ + p_;
          // (an odd first leaf -- restart inside a pair -- takes the step size drawn with its partner: index khc, just below ku)
          const uint32_t klow = (k0 & 1) ? khc : ku;
          const uint32_t base = (uint32_t)uni_i((int)(klow >> 1));
          {
            Philox4 r = philox4x32_10(base + (uint32_t)(lane >> 1), (uint32_t)iter, cg, 0u, seedlo, seedhi);
            ru0 = (lane & 1) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
            rb_iter = (uint32_t)iter; rb_base = base;
          }
          const uint32_t j = (uint32_t)k16;
          const uint32_t nE = ((j + 1u) >> 1) - (((uint32_t)k0 + 1u) >> 1), nO = (j >> 1) - ((uint32_t)k0 >> 1);  // even / odd leaves in [k0, j)
          const uint32_t kuS = ku + ce * nE + co * nO;  // the leaf's first draw
          const bool evn = (i == 0) || !(j & 1u);
          uint32_t ih = evn ? kuS : kuS - ce + 1u;  // the second step size of the pair was drawn with the first
          ih = (j == (uint32_t)k0 && (k0 & 1)) ? khc : ih;
          const uint32_t ir = evn ? kuS + (i != 0 ? 2u : 1u) : kuS;
          const uint32_t ipk = ir + r_;
          const uint32_t off = 2u * base;
          const double uh = __shfl(ru0, (int)((ih - off) & 63u)), ur = __shfl(ru0, (int)((ir - off) & 63u)), up = __shfl(ru0, (int)((ipk - off) & 63u));
          const double low = Hc * (1 - prm.s), high = Hc * (1 + prm.s);
          hV = low + (high - low) * uh;  // numpy legacy uniform(low, high) = low + (high-low)*U  (WALNUTS.py:298,395)
          urV = ur; upV = up;
        }
        // =========================== T phase: the integrator calls of leaves k0 .. k1-1 ===========================
        {
          int kend = k1;
#pragma unroll 1
          for (int k = k0; k < kend; ++k) {
            hcur = wn_readlane(hV, k);
            WN_STAMP(0);
            const double Ham0 = Ha;  // Hs[] of the neighbouring stored index = energy at the active end (:302,400,443,487,530)
            int nEvF = 0, nEvB = 0, If, Ib, cret;
            double lpn, Hn;
            double qw[NE], vw[NE], gw[NE], lpC, HC;
            double vin[NE];
            WN_OV(e) vin[e] = (double)xi * va[e];  // vh = xi*v + ...
            if (INTEG == I_FIXED) {  // adaptiveIntegrators.py:49-59
              run_pass(qa, vin, ga, Ham0, 0, qw, vw, gw, lpC, HC, nEvF);
              If = Ib = cret = 0; lpn = lpC; Hn = HC;
              if constexpr (WARM) igr_pass = hcur * pow(fmax(1.0e-10, fabs(Ham0 - Hn)), -1.0 / 3.0);  // adaptiveIntegrators.py:59
            } else {
              int cc = minC;  // forward search (:69-94 / 366-389); the first pass is peeled: it is accepted in 94 % of the busiest chains' macro
              // steps, and the likely path then falls through one not-taken branch instead of leaving a loop through two taken ones
              bool ok = run_pass(qa, vin, ga, Ham0, cc, qw, vw, gw, lpC, HC, nEvF);
              if (__builtin_expect(!ok && cc < maxC, 0)) {
                do {
                  ++cc;
                  ok = run_pass(qa, vin, ga, Ham0, cc, qw, vw, gw, lpC, HC, nEvF);
                } while (!ok && cc < maxC);
              }
              If = ok ? cc : maxC;  // Q5: the last computed state is used silently
              WN_STAMP(1);
              // adaptLeapFrogR2P :392: with probability R2Pprob0 keep c = If, else re-integrate at c = If+1
              const bool keep = INTEG != I_ADAPT_R2P || uni_b(wn_readlane(urV, k) < prm.R2Pprob0);
              int cSim = If, maxTry = If - 1;
              Ib = If;
              lpn = lpC; Hn = HC;
              double igr_leaf = igr_pass;  // (the forward pass that is used -- or, below, the re-integration at If + 1; not the backward passes)
              if (__builtin_expect(!keep || maxTry >= minC, 0)) {  // (the common macro step -- first pass accepted, kept -- skips all of this)
                if (!keep) {  // :400-424, 434-437
                  run_pass(qa, vin, ga, Ham0, If + 1, qw, vw, gw, lpC, HC, nEvF);
                  cSim = If + 1; maxTry = maxC; Ib = maxC;
                  lpn = lpC; Hn = HC;
                  igr_leaf = igr_pass;
                }
                if (maxTry >= minC) {  // backward search from (qOut, -vOut, gOut)  (:107-132 / 440-464)
                  double qo[NE], vo[NE], go[NE], vneg[NE];
                  WN_OV(e) { qo[e] = qw[e]; vo[e] = vw[e]; go[e] = gw[e]; vneg[e] = -vw[e]; }
                  double lpB, HB;
                  for (cc = minC;; ++cc) {
                    ok = run_pass(qo, vneg, go, Hn, cc, qw, vw, gw, lpB, HB, nEvB);
                    if (ok) { Ib = cc; break; }
                    if (cc >= maxTry) break;
                  }
                  WN_OV(e) { qw[e] = qo[e]; vw[e] = vo[e]; gw[e] = go[e]; }
                }
              }
              cret = INTEG == I_ADAPT_R2P ? cSim : If;  // :135-137 / 466-475 (lwt follows from If, Ib, c: the R phase forms it)
              igr_pass = igr_leaf;
            }
            WN_STAMP(2);
            // intOut.v is in forward-time orientation (adaptiveIntegrators.py:57,135,473); the new state becomes the active end
            WN_OV(e) { qa[e] = qw[e]; va[e] = (double)xi * vw[e]; ga[e] = gw[e]; }
            Ha = Hn;
            if constexpr (WIDE) { vst(o_ringq(k), qa); vst(o_ringv(k), va); }
            else { L[ring_base + ring_step * k] = qa[0]; L[ring_base + ring_vofs + ring_step * k] = va[0]; }
            {
              const bool mk = k16 == k;
              HnV = mk ? Hn : HnV; lpV = mk ? lpn : lpV;
              if constexpr (WARM) { const double lg = log(igr_pass); lgV = mk ? lg : lgV; }
              iA = mk ? (If | (Ib << 8) | (cret << 16)) : iA;
              iB = mk ? (nEvF | (nEvB << 16)) : iB;
            }
            // forced reject (:316-319): nothing behind a non-finite energy will be used -- the batch ends here
            kend = uni_b(isfinite(Hn)) ? kend : k + 1;
            st_[6] += 1; st_[11] += (unsigned long long)(nEvF + nEvB);
            WN_STAMP(3);
          }
          k1 = kend;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // left endpoint of the sub-trees of 2**j leaves, j >= 5, that start at this block's first leaf (levels 1..4 stay inside the ring)
        if (k0 == 0 && (nb0 == 0 ? i >= 5 : __builtin_ctz((unsigned)nb0) >= 5)) {
          const int sl = nb0 == 0 ? 0 : __builtin_ctz((unsigned)nb0);
          if constexpr (WIDE) {
            double tq[NE], tv[NE];
            vld(o_ringq(0), tq); vld(o_ringv(0), tv);
            vst(W_SLOT + 2 * sl, tq); vst(W_SLOT + 2 * sl + 1, tv);
          } else {
            const double tq = L[RQ + k16], tv = L[RV + k16];
            L[(lane < 16 ? SLOT + 32 * sl : DUMP) + lane] = tq; L[(lane < 16 ? SLOT + 32 * sl + 16 : DUMP) + lane] = tv;
          }
        }
        // =========================== R phase: bookkeeping of leaves k0 .. k1-1, lane k = leaf k ===========================
        const int k = k16;
        const bool inb = k >= k0 && k < k1;
        const double Hn_k = HnV, h_k = hV, up_k = upV, lp_k = lpV;
        const int If_k = iA & 255, Ib_k = (iA >> 8) & 255, c_k = iA >> 16, ef_k = iB & 0xFFFF, eb_k = iB >> 16;
        double lw_k;  // lwt of the macro step (adaptiveIntegrators.py:57-59 / 135-137 / 466-475)
        if (INTEG == I_ADAPT_R2P) {
          const double lwtf = c_k == If_k ? prm.logP : prm.log1mP;
          const double lwtb = c_k == Ib_k ? prm.logP : (c_k == Ib_k + 1 ? prm.log1mP : WN_LOGZERO);
          lw_k = lwtb - lwtf;
        } else if (INTEG == I_ADAPT_D) {
          lw_k = (If_k != Ib_k) ? WN_LOGZERO : 0.0;
        } else lw_k = 0.0;
        const int nk = nb0 + k;
        const bool first_k = (i == 0) || !(nk & 1);
        const int nbat = k1 - k0;
        // serial prefix sums in the reference's order: row 0 lwtSum of this direction (:321,354,420,507,550; Q2: on the backward
        // side only the first leaf of a pair adds), row 1 timeLen of this direction, row 2 orbitLen_
        double P1;
        {
          const bool upd = (i == 0) || first_k || fw;
          const double x = row == 0 ? (upd ? lw_k : 0.0) : h_k;
          const double carry = row == 0 ? ((i == 0) ? 0.0 : (fw ? lwtSumf : lwtSumb)) : row == 1 ? (fw ? timeLenF_ : timeLenB_) : orbitLen_;
          double s = inb ? x : 0.0;
          s = (k == k0) ? carry + s : s;
          const double xs = s;
          s = scan_steps(s, xs, nbat);
          P1 = s;
        }
        const double Wn = exp(-Hn_k + H0 + P1);  // Wnew = exp(-H + H0 + lwtSum)  (row 0)
        double WS;
        {
          double s = inb ? Wn : 0.0;
          const double carry = (i == 0) ? 0.0 : WnewSum;
          s = (k == k0) ? carry + s : s;
          const double xs = s;
          s = scan_steps(s, xs, nbat);
          WS = s;
        }
        WN_STAMP(4);
        const bool inb0 = inb && lane < 16;
        const bool drawn_k = (i != 0) && (WS > prm.wtSumThresh);
        // :426  take this state with probability Wnew / WnewSum; i == 0 (:322-328): unconditional, no uniform
        const bool pick_k = (i == 0) ? true : (drawn_k && up_k < Wn / WS);
        const uint64_t m_in = __ballot(inb0);
        const uint64_t m_nf = __ballot(inb0 && !isfinite(Hn_k));
        const uint64_t m_dr = __ballot(inb0 && drawn_k);
        const uint64_t m_pk = __ballot(inb0 && pick_k);
        const bool pred = fw ? predF : predB;
        const uint64_t m_bad = (i != 0) ? ((pred ? ~m_dr : m_dr) & m_in) : 0ull;

        // ---- U-turn checks, one lane each (stopCondition, WALNUTS.py:95-97, with the older state on the side opposite to xi) ----
        uint64_t m_fire = 0ull;  // leaves at which a sub-tree check fired
        bool f0_now = false;
        if constexpr (WIDE) {
          // wave-wide dots (the one-stream engine's uturn_dots: products summed over the lane's registers in order, then over the
          // wave).  The checks of leaves r = k0 .. k1-1 in order -- the full-orbit check at the doubling's last leaf, the sub-tree
          // checks at odd leaves: levels 1 .. ctz(leaf + 1) -- as a scalar walk; the loads of the NEXT check are in flight while this
          // one is summed (a lone wave waits ~900 cycles for a load from L2; loads return in order).
          struct Chk { int r, lvl; bool ok; };
          auto nlv_of = [&](int r) -> int { const int nr = nb0 + r; return (i >= 1 && (nr & 1)) ? __builtin_ctz((unsigned)(nr + 1)) : 0; };
          auto first_at = [&](int r) -> Chk {  // first check at or behind leaf r
            Chk c; c.r = r; c.lvl = 0; c.ok = false;
#pragma unroll 1
            for (; c.r < k1; ++c.r) {
              const bool full = nb0 + c.r == nleaf - 1;
              if (full || nlv_of(c.r) > 0) { c.lvl = full ? 0 : 1; c.ok = true; break; }
            }
            return c;
          };
          auto advance = [&](const Chk& c) -> Chk {
            if (c.lvl < nlv_of(c.r)) { Chk d = c; d.lvl = c.lvl + 1; return d; }
            return first_at(c.r + 1);
          };
          auto issue = [&](const Chk& c, double* qr, double* vr, double* ql, double* vl) {
            int lq, lv;
            if (c.lvl == 0) { lq = W_COLDQ; lv = W_COLDV; }
            else if (c.lvl <= 4) { lq = W_RQ + (c.r + 1 - (1 << c.lvl)); lv = lq + (W_RV - W_RQ); }
            else {
              const int nl = nb0 + c.r + 1 - (1 << c.lvl);  // left leaf of the sub-tree
              const int sl = nl == 0 ? 0 : __builtin_ctz((unsigned)nl);
              lq = W_SLOT + 2 * sl; lv = lq + 1;
            }
            vld(o_ringq(c.r), qr); vld(o_ringv(c.r), vr); vld(lq, ql); vld(lv, vl);
          };
          auto sum_up = [&](const Chk& c, const double* qr, const double* vr, const double* ql, const double* vl) {
            double a1 = 0.0, a2 = 0.0;
            WN_OV(e) {
              const double dq = qr[e] - ql[e];
              a1 += vr[e] * dq;
              a2 += vl[e] * dq;
            }
            reduce2(a1, a2);
            const bool fire = ((double)xi * a1 < 0.0) || ((double)xi * a2 < 0.0);
            if (uni_b(fire)) {
              if (c.lvl == 0) f0_now = true; else m_fire |= 1ull << c.r;
            }
          };
          Chk ca = first_at(k0);
          if (ca.ok) {
            double aqr[NE], avr[NE], aql[NE], avl[NE], bqr[NE], bvr[NE], bql[NE], bvl[NE];
            issue(ca, aqr, avr, aql, avl);
#pragma unroll 1
            for (;;) {
              Chk cb = advance(ca);
              if (cb.ok) issue(cb, bqr, bvr, bql, bvl);
              sum_up(ca, aqr, avr, aql, avl);
              if (!cb.ok) break;
              ca = advance(cb);
              if (ca.ok) issue(ca, aqr, avr, aql, avl);
              sum_up(cb, bqr, bvr, bql, bvl);
              if (!ca.ok) break;
            }
          }
        } else {
#pragma clang fp contract(off)
          int r = -1, lq = 0, lv = 0, lvl = 0;  // right leaf (ring entry), LDS offsets of the left state
          if (lane < 8) { r = 2 * lane + 1; lvl = 1; }
          else if (lane < 12) { r = 4 * (lane - 8) + 3; lvl = 2; }
          else if (lane < 14) { r = 8 * (lane - 12) + 7; lvl = 3; }
          else if (lane == 14) { r = 15; lvl = 4; }
          else if (lane == 15) { r = (nleaf - 1) & 15; lvl = 0; }
          else if (lane - 16 + 5 <= M) { r = 15; lvl = lane - 16 + 5; }
          bool act = r >= k0 && r < k1;
          const int nr = nb0 + r;
          if (lvl == 0) {
            act = act && nr == nleaf - 1;
            lq = COLDQ; lv = COLDV;
          } else if (lvl <= 4) {
            act = act && i >= 1;
            lq = RQ + RS * (r + 1 - (1 << lvl)); lv = lq + (RV - RQ);
          } else {
            act = act && i >= 1 && lvl <= __builtin_ctz((unsigned)(nr + 1));
            const int nl = nr + 1 - (1 << lvl);  // left leaf of the sub-tree
            const int sl = nl == 0 ? 0 : __builtin_ctz((unsigned)nl);
            lq = SLOT + 32 * sl; lv = lq + 16;
          }
          bool fire = false;
          if (act) {
            const int rq = RQ + RS * r, rv = RV + RS * r;
            double s1[4], s2[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              double p1[4], p2[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int d = 4 * g4 + e;
                const double dq = L[rq + d] - L[lq + d];
                p1[e] = L[rv + d] * dq;
                p2[e] = L[lv + d] * dq;
              }
              s1[g4] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
              s2[g4] = (p2[0] + p2[1]) + (p2[2] + p2[3]);
            }
            const double d1 = ((s1[0] + s1[1]) + (s1[2] + s1[3])) + 0.0, d2 = ((s2[0] + s2[1]) + (s2[2] + s2[3])) + 0.0;
            fire = ((double)xi * d1 < 0.0) || ((double)xi * d2 < 0.0);
          }
          const uint64_t m_chk = __ballot(act && fire);
          if (m_chk) {
            uint64_t m = m_chk;
            while (m) {
              const int c = __builtin_ctzll(m);
              m &= m - 1;
              if (c == 15) { f0_now = true; continue; }
              const int rr = c < 8 ? 2 * c + 1 : c < 12 ? 4 * (c - 8) + 3 : c < 14 ? 8 * (c - 12) + 7 : 15;
              m_fire |= 1ull << rr;
            }
          }
        }
        WN_STAMP(5);
        // ---- the first event decides how many leaves of the batch stand ----
        const int t_nf = m_nf ? __builtin_ctzll(m_nf) : 64;
        const int t_sub = m_fire ? __builtin_ctzll(m_fire) : 64;
        const int t_bad = m_bad ? __builtin_ctzll(m_bad) : 64;  // leaves behind it used wrong draw indices
        const int t_ev = t_nf < t_sub ? t_nf : t_sub;
        bool end_nf = false, end_sub = false, redo = false;
        int Lfull;  // last leaf whose bookkeeping is complete
        if (t_ev <= t_bad && t_ev < k1) {
          end_nf = t_nf <= t_sub;  // the non-finite test comes first (:316-319 before any U-turn test)
          end_sub = !end_nf;
          Lfull = end_nf ? t_ev - 1 : t_ev;
        } else if (t_bad < k1 - 1) {
          redo = true;
          Lfull = t_bad;
        } else {
          Lfull = k1 - 1;
        }
        const int A_end = end_nf ? t_ev : Lfull;  // last leaf that reached the accumulators at the top of the reference's step
        // ---- accumulators over leaves k0 .. A_end ----
        {
          const bool macc = inb && k <= A_end;
          // nevalF, nevalB, #(If != Ib), #(If == 0): four rows of one register
          int v4 = row == 0 ? ef_k : row == 1 ? eb_k : row == 2 ? (int)(If_k != Ib_k) : (int)(If_k == 0);
          v4 = rowsum_i(macc ? v4 : 0);
          nevalF += rl_i(v4, 0); nevalB += rl_i(v4, 16); nneq += rl_i(v4, 32); nzero += rl_i(v4, 48);
          nused += A_end - k0 + 1;
          if constexpr (WARM) if (iter <= prm.warmupIter && prm.adaptH) {  // igrConstQ.push(np.log(intOut.igrConst)), leaf by leaf (:313)
#pragma unroll 1
            for (int j = k0; j <= A_end; ++j) wn_p2_push(prm, p2q, p2m, p2n, wn_readlane(lgV, j));
          }
          if (INTEG != I_FIXED) {
            int m4 = row == 0 ? If_k : row == 1 ? c_k : row == 2 ? -If_k : -c_k;
            m4 = rowmin_i(macc ? m4 : (1 << 30));
            const int a0 = rl_i(m4, 0), a1 = rl_i(m4, 16), a2 = -rl_i(m4, 32), a3 = -rl_i(m4, 48);
            minIf = a0 < minIf ? a0 : minIf; minc = a1 < minc ? a1 : minc; maxIf = a2 > maxIf ? a2 : maxIf; maxc = a3 > maxc ? a3 : maxc;
          } else {
            minIf = minc = maxIf = maxc = 0;
          }
          double d4 = row == 0 ? lw_k : row == 1 ? -lw_k : row == 2 ? Hn_k : -Hn_k;
          d4 = rowmin_d(macc ? d4 : INFINITY);
          const double b0 = wn_readlane(d4, 0), b1 = -wn_readlane(d4, 16), b2 = wn_readlane(d4, 32), b3 = -wn_readlane(d4, 48);
          minlw = b0 < minlw ? b0 : minlw; maxlw = b1 > maxlw ? b1 : maxlw;
          Hmin = b2 < Hmin ? b2 : Hmin; Hmax = b3 > Hmax ? b3 : Hmax;
          if (__ballot(macc && lane < 16 && isnan(Hn_k))) Hnan = true;
          const double tl = wn_readlane(P1, 16 + A_end);
          const int idxA = fw ? b + 1 + (nb0 + A_end) : a - 1 - (nb0 + A_end);
          if (fw) { timeLenF_ = tl; maxFint = idxA; } else { timeLenB_ = tl; maxBint = idxA; }
          // orbitLen_ += h: i == 0 before the integrator call (:300), else behind the non-finite test (:432 / 471 / 519 / 561)
          const int o_end = (end_nf && i != 0) ? t_ev - 1 : A_end;
          if (o_end >= k0) orbitLen_ = wn_readlane(P1, 32 + o_end);
          if (Lfull >= k0) {
            const double ls = wn_readlane(P1, Lfull);
            if (fw) lwtSumf = ls; else lwtSumb = ls;
            WnewSum = wn_readlane(WS, Lfull);
            const uint64_t mp = m_pk & ((2ull << Lfull) - 1ull);
            if (mp) {  // the last leaf that took the multinomial draw is the proposal
              const int Pk = 63 - __builtin_clzll(mp);
              if (ip == ipl) ip ^= 1;  // keep qPropLast intact
              double qp[NE];
              vld(o_ringq(Pk), qp);
              vst(o_prop(ip), qp);
              lpP = wn_readlane(lp_k, Pk);
              L_ = fw ? b + 1 + (nb0 + Pk) : a - 1 - (nb0 + Pk);
              const double tp = wn_readlane(P1, 16 + Pk);
              indexStat_ = fw ? tp : -tp;
            }
            if (i != 0) {  // the draws these leaves consumed, and what the next leaf is expected to do
              const uint64_t rng = m_in & ((2ull << Lfull) - 1ull);
              // leaves with an even index drew the pair's two step sizes
              const uint64_t evn = (nb0 & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
              ku += 2u * (uint32_t)__builtin_popcountll(rng & evn) + (INTEG == I_ADAPT_R2P ? (uint32_t)__builtin_popcountll(rng) : 0u) +
                    (uint32_t)__builtin_popcountll(rng & m_dr);
              const bool dl = (m_dr >> Lfull) & 1ull;
              if (fw) predF = dl; else predB = dl;
              // the second step size of a pair was drawn with the first (:395): its index, for a batch that starts at the pair's second leaf
              // (wide mode's one-leaf batches; a restart behind a mispredicted draw sets it again below)
              if constexpr (WIDE) khc = ku - 1u - (INTEG == I_ADAPT_R2P ? 1u : 0u) - (uint32_t)dl;
            } else {
              ku += 1u + (INTEG == I_ADAPT_R2P ? 1u : 0u);
            }
            Ha = wn_readlane(Hn_k, Lfull);
          }
        }
        st_[10] += 1; st_[12] += redo ? 1 : 0;
        WN_STAMP(7);
        // recordOrbitStats: np.minimum / np.maximum over generated(q) of the leaves that passed the non-finite test, k0 .. Lfull (:331-333, 364-366,
        // 434-436, ..; a NaN stays: numpy's minimum / maximum propagate it) -- lane = element, one read-modify-write of the row per batch
        if constexpr (!WIDE) if (prm.omin && Lfull >= k0) {
          const int fk = (int)(prm.generated == G_FIRSTK);
          const int kk = fk * prm.gen_k + (1 - fk) * D;
          const size_t orow = ((size_t)clc * prm.numIter + (iter - 1)) * prm.dg;
          const bool mine = lane < 16 && lane < kk && lane < D;
          double mn = mine ? prm.omin[orow + lane] : 0.0, mx = mine ? prm.omax[orow + lane] : 0.0;
#pragma unroll 1
          for (int j = k0; j <= Lfull; ++j) {
            const double x = L[RQ + RS * j + k16];
            mn = (isnan(x) || isnan(mn)) ? NAN : (x < mn ? x : mn);
            mx = (isnan(x) || isnan(mx)) ? NAN : (x > mx ? x : mx);
          }
          if (mine) { prm.omin[orow + lane] = mn; prm.omax[orow + lane] = mx; }
        }
        if (f0_now && !end_nf && !end_sub && !redo) fire0 = true;
        n = nb0 + Lfull + 1;
        if (end_nf) {  // :316-319, 414-417, 457-459 (Q3, Q4)
          const bool first_e = (i == 0) || !((nb0 + t_ev) & 1);
          if (first_e) stopCode = 999;
          endIter = true;
          break;
        }
        if (end_sub) { sub = true; break; }
        if (redo) {  // restart behind the mispredicted leaf: its state is in the ring
          vld(o_ringq(Lfull), qa); vld(o_ringv(Lfull), va);
          double lpp = tgt.grad(qa, ga, lane);
          (void)lpp;
          // the second step size of a pair was drawn with the first (:395): its index, for a restart at the pair's second leaf
          khc = ku - 1u - (INTEG == I_ADAPT_R2P ? 1u : 0u) - (uint32_t)((m_dr >> Lfull) & 1ull);
        }
      }
      if (endIter) break;
      // ---- end of the doubling (:590-648) ----
      indexStat_ = indexStat_ / (timeLenF_ + timeLenB_);  // :595 (Q6)
      if (sub) {  // :597-605
        ip = ipl; lpP = lpPL; L_ = Lold_; indexStat_ = indexStatOld_;
        NdS = i; NdC = i + 1; stopCode = 5;
        endIter = true;
        break;
      }
      rng_ensure((uint32_t)iter, ku);
      if (!__any(uni(ku++) < WnewSum / WoldSum)) { ip = ipl; lpP = lpPL; L_ = Lold_; indexStat_ = indexStatOld_; }  // :613-616
      bothEndsPassive = (lwtSumb < WN_LOGZERO + 1.0) && (lwtSumf < WN_LOGZERO + 1.0);  // :624
      if (__any(fire0 || bothEndsPassive)) {  // :625-634
        stopCode = fire0 ? 4 : -4;
        NdS = NdC = i + 1;
        orbitLenSam_ = orbitLen_;
        endIter = true;
        break;
      }
      WoldSum += WnewSum;  // :641-648
      orbitLenSam_ = orbitLen_;
      NdS = NdC = i + 1;
      if (fw) b += nleaf; else a -= nleaf;
      i++;
      if (i == M) { endIter = true; break; }
    }
    // ---- end of the iteration (:653-695) ----
    {
      double qP[NE];
      vld(o_prop(ip), qP);
      emit_generated(iter, qP);  // samples[:,iterN] = generated(qc), qc = qProp
      double d[NDIAG];
      d[0] = L_; d[1] = NdS; d[2] = orbitLen_; d[3] = orbitLenSam_; d[4] = maxFint; d[5] = maxBint;
      d[6] = nevalF; d[7] = nevalB;
      const bool AD = INTEG != I_FIXED;
      d[8] = AD ? minIf : 0; d[9] = AD ? maxIf : 0; d[10] = AD ? minlw : 0.0; d[11] = AD ? maxlw : 0.0;
      d[12] = bothEndsPassive ? 1.0 : 0.0;
      d[13] = ((lwtSumb < WN_LOGZERO + 1.0) || (lwtSumf < WN_LOGZERO + 1.0)) ? 1.0 : 0.0;
      d[14] = AD ? (double)nneq / (double)nused : 0.0;
      d[15] = Hc;
      d[16] = AD ? (double)nzero / (double)nused : 1.0;
      d[17] = Hnan ? NAN : (Hmax - Hmin);
      d[18] = dlt; d[19] = (double)stopCode; d[20] = NdC; d[21] = AD ? minc : 0; d[22] = AD ? maxc : 0;
      d[23] = indexStat_;
      double vv = 0.0;
#pragma unroll
      for (int kk = 0; kk < NDIAG; ++kk) vv = lane == kk ? d[kk] : vv;
      if (lane < NDIAG) prm.diag[((size_t)clc * prm.numIter + (iter - 1)) * NDIAG + lane] = vv;
    }
    if constexpr (WARM) if (iter <= prm.warmupIter) {  // tuning parameter adaptation (:701-712), behind the diagnostics row: it holds the values this iteration ran with
      double* const fc = prm.facs ? prm.facs + (size_t)clc * 2 * (size_t)prm.warmupIter : nullptr;
      const double err = Hnan ? NAN : (Hmax - Hmin);
      wn_adapt_update(prm, iter, err, fc, lane == 0, dlt, Hc, p2q, p2n);
    }
    // longest job first: a launch is as long as its busiest chain, and a chain that has needed many evaluations per iteration so far goes on
    // needing them (a funnel-neck / long-orbit chain stays one): such a wave takes its SIMD's issue slots ahead of the wave it shares it with
    // (BASELINE config 3, six launches: 171 / 135 / 234 / 203 / 213 / 165 ms -> 172 / 132 / 235 / 186 / 208 / 162; scheduling only: the same draws)
    {
      evals_total += nevalF + nevalB;
      const int per_it = evals_total / iter;
      const int pr = per_it > 2048 ? 3 : per_it > 512 ? 2 : per_it > 128 ? 1 : 0;
      if (pr != prio_now) { prio_now = pr; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    }
    iter++;
    if (iter > prm.numIter) live = false;
  }
  (void)gen_sum;
#if defined(WN_STAMPS)
  WN_STAMP(9);
  if (prm.stamps && lane == 0) {
    unsigned long long* o = prm.stamps + (size_t)cl * 16;
    for (int k = 0; k < 16; ++k) o[k] = st_[k];
  }
#endif
  if constexpr (QUEUE) {
    if (!prm.dynq) break;  // (no queue head: the launch dealt one chain per wave)
    int nxt = 0;
    if (lane == 0) nxt = atomicAdd(prm.dynq, 1);
    cl_ = (int)gridDim.x * K + __builtin_amdgcn_readfirstlane(nxt);
    if (cl_ >= prm.C) break;
  } else break;
  }  // for (;;): the wave's chains
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// plain sampling runs (no trace, no warm-up, no orbit statistics) of the three leapfrog integrators on the <= 16-D targets of the
// one-chain-per-wave layout; WN_ORBIT_LEGACY=1 keeps the one-stream engine (A/B tests)
// wide-mode geometries: StockWatson (12 registers per lane), the dense Gaussian of 65 .. 128 dimensions with its matrix in LDS (2 registers)
static bool orbit_dense_geom(const KernelChoice& kc) { return kc.layout == 2 && kc.p0 == 2 && kc.W == 1 && kc.G == 1 && kc.p1 == T_DENSEGAUSS; }
// ... and std / diagonal Gaussians of 17 .. 128 dimensions, whose adaptive runs otherwise sit in the cooperative layout (coop: layout 1, CK_DIAG = 1)
static bool orbit_diag_geom(const KernelChoice& kc) { return kc.layout == 1 && kc.p1 == 1 && kc.W == 4 && kc.G == 1 && (kc.p0 == 16 || kc.p0 == 32); }
// ... and of 129 .. 512 dimensions, whose adaptive runs otherwise sit on the one-stream engine of the wave layout (layout 2, 8 registers per lane)
static bool orbit_diag8_geom(const KernelChoice& kc) { return kc.layout == 2 && kc.p0 == 8 && kc.W == 1 && kc.G == 1 && kc.p1 == T_DIAGGAUSS; }
// ... and fixedLeapFrog on 129 .. 256 dimensions (cooperative layout with four tiles per wave: layout 1, 64 k-steps)
static bool orbit_diag64_geom(const KernelChoice& kc) { return kc.layout == 1 && kc.p1 == 1 && kc.W == 4 && kc.G == 1 && kc.p0 == 64; }
bool orbit_wide_geom(const KernelChoice& kc) {
  return (kc.layout == 2 && kc.p0 == 12 && kc.W == 1 && kc.G == 1 && kc.p1 == T_STOCKWATSON) || orbit_dense_geom(kc) || orbit_diag_geom(kc) || orbit_diag8_geom(kc) ||
         orbit_diag64_geom(kc);
}
int orbit_wide_ne(const KernelChoice& kc) { return orbit_diag_geom(kc) ? 2 : orbit_diag64_geom(kc) ? 4 : kc.p0; }
size_t orbit_wide_work_doubles(int NE, int C, int M) { return (size_t)C * orbit::wide_doubles(NE, M); }
// Which plain sampling runs of that geometry take the wide mode: fixedLeapFrog (mainSW.py:56-62: a leaf is ONE evaluation, the one-stream
// engine spends 40 % of it on bookkeeping).  A macro step of the adaptive integrators is 8+ evaluations (minC = 3): its bookkeeping is a
// twentieth of it in either kernel, and the one-stream engine's is the cheaper one (no ring, no batch draws); WN_ROUTE_ORBIT_WIDE_ADAPT sends
// them here all the same (tests/test_gpu_orbit.py: the generic kernel against that engine, bit for bit).
bool orbit_wide_run(const KernelChoice& kc, int integrator) {
  if (!orbit_wide_geom(kc) || (kc.route & (WN_ROUTE_ORBIT_LEGACY | WN_ROUTE_ORBIT_NOWIDE))) return false;
  // the dense Gaussian is here for its adaptive runs (wn_k_wavelds.hip): a macro step is an evaluation or two, batches of 16 leaves pay
  if (orbit_dense_geom(kc)) return integrator == I_ADAPT_D || integrator == I_ADAPT_R2P;
  // std / diagonal Gaussian, 17 <= D <= 128: the same argument with a trivial gradient (the cooperative layout's tick machine is what costs)
  if (orbit_diag_geom(kc)) return (integrator == I_ADAPT_D || integrator == I_ADAPT_R2P) && !(kc.route & WN_ROUTE_DIAG_COOP);
  // fixedLeapFrog on 129 .. 256 dimensions (3.7e8 -> 5.9e8 steps/s at D = 256; up to 112 dimensions the role-split kernels are faster, beyond 256 the
  // wave layout's engine as fast: tools/fixed_diag8_ab.py)
  if (orbit_diag64_geom(kc)) return integrator == I_FIXED && !(kc.route & WN_ROUTE_DIAG_COOP);
  if (orbit_diag8_geom(kc)) return (integrator == I_ADAPT_D || integrator == I_ADAPT_R2P) && !(kc.route & WN_ROUTE_DIAG_COOP);
  return integrator == I_FIXED || ((integrator == I_ADAPT_D || integrator == I_ADAPT_R2P) && (kc.route & WN_ROUTE_ORBIT_WIDE_ADAPT));
}

bool orbit_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm) {
  const bool wide = orbit_wide_geom(kc);
  if (!wide && (kc.layout != 2 || kc.p0 != 1 || kc.W != 1 || kc.G != 1)) return false;
  // (warm-up iterations run here as well, except StockWatson's: its WARM instantiations are not built)
  if (warm && kc.layout == 2 && kc.p1 == T_STOCKWATSON) return false;
  if (prm.mode != MODE_SAMPLE || prm.trace_n || prm.trace) return false;
  // orbit statistics: the one-register form (funnel10, the 2-D targets), generated = identity / first k
  if (prm.omin && (wide || prm.generated == G_Q0_SUMSQ)) return false;
  if (adapt == 2) return false;
  if (prm.integrator != I_FIXED && prm.integrator != I_ADAPT_D && prm.integrator != I_ADAPT_R2P) return false;
  if (prm.M < 1 || prm.M > MAX_M || (!wide && prm.D > 16) || prm.maxC > 12 || prm.minC < 0 || prm.minC > prm.maxC) return false;
  if (prm.generated != G_IDENTITY && prm.generated != G_FIRSTK && prm.generated != G_Q0_SUMSQ) return false;
  if (kc.route & WN_ROUTE_ORBIT_LEGACY) return false;
  if (wide) return orbit_wide_run(kc, prm.integrator) && (!orbit_diag_geom(kc) || prm.D <= 128) && (!orbit_diag64_geom(kc) || (prm.D > 128 && prm.D <= 256)) && (!orbit_diag8_geom(kc) || (prm.D > 128 && prm.D <= 512));
  switch (kc.p1) {
    case T_FUNNEL10: case T_CORRGAUSS: case T_SMILE: case T_FUNNEL1: case T_MODFUNNEL: return true;
  }
  return false;
}

template <class T_, class = void> struct has_fused_step { static constexpr bool value = false; };
template <class T_> struct has_fused_step<T_, typename std::enable_if<T_::FUSED_STEP>::type> { static constexpr bool value = true; };

template <class Tgt>
static hipError_t go_orbit(const Params& prm, hipStream_t st, bool exact) {
  const size_t dyn = (size_t)orbit::lds_doubles(prm.M) * sizeof(double);
  dim3 grid(prm.C), block(64);
  // a target with a micro step of its own (one reduction instead of two) runs it unless WN_ROUTE_ORBIT_EXACT asks for the arithmetic of
  // the one-stream engine (the bit-for-bit comparisons of tests/test_gpu_orbit.py)
  const bool warm = prm.warmupIter > 0;  // (the warm-up iterations run here too: WARM instantiations, each target's default arithmetic)
  if (warm) {
    if constexpr (has_fused_step<Tgt>::value) if (!exact) {
      if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, true, 1, 1, true>), grid, block, dyn, st, prm);
      else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, true, 1, 1, true>), grid, block, dyn, st, prm);
      else hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, true, 1, 1, true>), grid, block, dyn, st, prm);
      return hipGetLastError();
    }
    if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, false, 1, 1, true>), grid, block, dyn, st, prm);
    else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, false, 1, 1, true>), grid, block, dyn, st, prm);
    else hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, false, 1, 1, true>), grid, block, dyn, st, prm);
    return hipGetLastError();
  }
  if constexpr (has_fused_step<Tgt>::value) if (!exact) {
    if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, true>), grid, block, dyn, st, prm);
    else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, true>), grid, block, dyn, st, prm);
    else hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, true>), grid, block, dyn, st, prm);
    return hipGetLastError();
  }
  if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P>), grid, block, dyn, st, prm);
  else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D>), grid, block, dyn, st, prm);
  else hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED>), grid, block, dyn, st, prm);
  return hipGetLastError();
}

// wide mode: vectors in the chain's workspace, no LDS
template <class Tgt, int NE>
static hipError_t go_orbit_wide(const Params& prm, hipStream_t st) {
  dim3 grid(prm.C), block(64);
  if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, false, NE>), grid, block, 0, st, prm);
  else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, false, NE>), grid, block, 0, st, prm);
  else hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, false, NE>), grid, block, 0, st, prm);
  return hipGetLastError();
}

// ... several chains per workgroup: the target stages its data in LDS once for all of them
template <class Tgt, int NE, int K, bool FIXED_TOO = false>
static hipError_t go_orbit_wide_k(const Params& prm, hipStream_t st) {
  dim3 grid((prm.C + K - 1) / K), block(64 * K);
  if constexpr (tgt_staged<Tgt>::value) {
    // persistent grid + chain queue (see the kernel): as many workgroups as are resident at once (the staged matrix takes most of a CU's LDS:
    // one per CU); the first gridDim.x * K chains are dealt statically, the rest through prm.dynq (zeroed by the caller in stream order)
    static int resident = 0;
    if (!resident) {
      int dev = 0, cus = 0, per = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)wn_orbit_kernel<Tgt, I_ADAPT_D, false, NE, K>, 64 * K, 0) == hipSuccess && cus > 0 && per > 0)
        resident = cus * per;
      else resident = 256;
    }
    if (prm.dynq && prm.warmupIter == 0 && (int)grid.x > resident) grid.x = (unsigned)resident;  // (warm-up runs: one chain per wave, no queue)
  }
  if (prm.warmupIter > 0) {
    if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, false, NE, K, true>), grid, block, 0, st, prm);
    else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, false, NE, K, true>), grid, block, 0, st, prm);
    else if (FIXED_TOO) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, false, NE, K, true>), grid, block, 0, st, prm);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_R2P, false, NE, K>), grid, block, 0, st, prm);
  else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_ADAPT_D, false, NE, K>), grid, block, 0, st, prm);
  else if (FIXED_TOO) hipLaunchKernelGGL((wn_orbit_kernel<Tgt, I_FIXED, false, NE, K>), grid, block, 0, st, prm);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_orbit(const KernelChoice& kc, const Params& prm, hipStream_t st) {
  const bool exact = (kc.route & WN_ROUTE_ORBIT_EXACT) != 0;
  if (orbit_dense_geom(kc)) return go_orbit_wide_k<WTgtDenseLDS<2>, 2, 8>(prm, st);
  if (orbit_diag_geom(kc)) return prm.D <= 16 ? go_orbit<WTgtDiagRow>(prm, st, exact) : go_orbit_wide_k<WTgtDiag<2>, 2, 4>(prm, st);
  if (orbit_diag64_geom(kc)) return go_orbit_wide_k<WTgtDiag<4>, 4, 4, true>(prm, st);
  if (orbit_diag8_geom(kc)) return prm.D <= 256 ? go_orbit_wide_k<WTgtDiag<4>, 4, 4>(prm, st) : go_orbit_wide_k<WTgtDiag<8>, 8, 4>(prm, st);
  if (orbit_wide_geom(kc)) return go_orbit_wide<WTgtStockWatson<12>, 12>(prm, st);
  switch (kc.p1) {
    case T_FUNNEL10: return go_orbit<WTgtFunnel10>(prm, st, exact);
    case T_CORRGAUSS: return go_orbit<WTgtPair<TgtCorrGauss>>(prm, st, exact);
    case T_SMILE: return go_orbit<WTgtPair<TgtSmile>>(prm, st, exact);
    case T_FUNNEL1: return go_orbit<WTgtPair<TgtFunnel1>>(prm, st, exact);
    case T_MODFUNNEL: return go_orbit<WTgtPair<TgtModFunnel>>(prm, st, exact);
  }
  return hipErrorInvalidValue;
}

}  // namespace wn
