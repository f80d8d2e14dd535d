// wn_fast.h -- the role-split form of the fixedLeapFrog / Gaussian headline path (wn_k_fast.hip).
//
// wn_chain.h runs one chain's whole tick -- leapfrog + gradient, energies, U-turn reductions, weights,
// multinomial pick, doubling / iteration ends -- as ONE instruction stream, so in the 16-chain MFMA
// group kernel the 50 MFMAs of a tick (3.2 k cycles) wait for ~11 k cycles of scalar bookkeeping.  For
// fixedLeapFrog the trajectory of an iteration does not depend on that bookkeeping at all until the orbit
// stops (adaptiveIntegrators.py:49-59: one step, lwt = 0; WALNUTS.py:392-587: weights and picks never feed
// back into (q, v)), so the work is split between two kinds of wavefront of one workgroup:
//
//   T waves (4, one per SIMD)  own the D-vectors of the 16 chains of the group in MFMA layout and do, per
//       tick, exactly what a command record tells them: leapfrog step with signed step size hs = xi*h,
//       the batched gradient G = -A Q (v_mfma_f64_16x16x4_f64), per-wave partial sums of lp, kinetic energy
//       and the U-turn dot products of the sub-tree checks due at this leaf (subTreePlan, WALNUTS.py:22-41),
//       sub-tree endpoint stores, and -- on command -- proposal stores, orbit-end swaps, iteration starts.
//   B wave (1, the bookkeeper)  owns every per-chain scalar of WALNUTS.py:189-695 (this file, FastBook):
//       it RETIRES the partial sums of tick t-1 while the T waves run tick t and ISSUES the command of tick
//       t+1.  Decisions therefore reach the vectors two ticks late; nothing ever has to be rolled back:
//         * a multinomial pick (WALNUTS.py:426,464,512,554) copies the picked position out of the
//           double-buffered Q exchange in LDS, which still holds the position of tick t-1 at tick t+1;
//         * at the last leaf of a doubling the chain idles until the verdict of WALNUTS.py:597-634 is in
//           (two ticks; the end swap the next doubling may need is done in the first of them);
//         * a sub-U-turn or a non-finite energy found at leaf n ends the iteration; the one leaf already in
//           flight is discarded (nothing it did is visible: proposals are stored only on command).
//       The draw order of the reference is kept exactly: the step sizes of a leaf pair are
//       issued ahead of the retirement of the previous leaf only when that leaf's multinomial draw is
//       certain to happen (WnewSum is non-decreasing inside a doubling, WALNUTS.py:424-426); otherwise the
//       chain idles until it is known.
//
// FastBook is plain scalar code, host + device: a CPU test harness under tests/ runs it against the oracle on
// the CPU with a scalar stand-in for the T waves (test tool only).
#pragma once
#include <math.h>
#include <stdint.h>

#include "wn_chain.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define WN_OPAQUE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#else
#define WN_OPAQUE4(a, b, c, d) do {} while (0)
#endif

namespace wn {
namespace fast {

// ---- command record, B -> T: one per chain and tick (16 bytes) ---------------------------------
enum {
  OP_STEP = 1u,      // leapfrog step with hs (otherwise hs = 0: the chain idles, its gradient is refreshed)
  OP_NEWITER = 2u,   // before the step: q <- proposal[ipnew], v <- fresh momentum, passive end <- (q, v);
                     // emits generated(q) as draw row `n` (WALNUTS.py:164,695) and the partials keinit / gen
  OP_INIT = 4u,      // with OP_NEWITER at tick 0: q <- q0, proposal[0] <- q, no momentum (draw row 0)
  OP_LAST = 8u,      // with OP_NEWITER: only the emit (the run's last draw); the chain is finished
  OP_SWAP = 16u,     // before the step: exchange (q, v) with the passive orbit end (WALNUTS.py:283-286)
  OP_PICK = 32u,     // copy the position of tick t-2 into proposal[ippick]
  OP_IPPICK = 64u,   // bit: ippick
  OP_IPNEW = 128u,   // bit: ipnew
  OP_I_SHIFT = 8,    // bits 8..12: doubling index i of an OP_STEP (n = leaf index inside the doubling)
  OP_UNDO = 1u << 14, // traced runs: the leaf that was in flight when the iteration ended is not an integrator call of the
                     // reference -- the T wave takes its trace row back
  OP_DONE = 1u << 15 // every chain of the group has finished: all waves leave the tick loop
};
struct alignas(16) Cmd {
  double hs;
  uint32_t op;
  uint32_t n;
};

// partial-sum quantities, T -> B
enum { Q_LP = 0, Q_KE = 1, Q_KEINIT = 2, Q_GEN = 3, Q_DOT0 = 4 };  // dots of check j: Q_DOT0 + 2j, + 2j + 1

// which U-turn checks does leaf n of doubling i need (wn_chain.h nchecks / need_check): check 0 = full orbit
// against the passive end at the last leaf, check j >= 1 = sub-tree of 2**j leaves ending at odd leaf n
WN_HD int nchecks(int i, int n) {
  int m = 0;
  if (n == (1 << i) - 1) m = 1;
  if (i >= 1 && (n & 1)) m = __builtin_ctz((unsigned)(n + 1)) + 1;
  return m;
}
WN_HD bool need_check(int i, int n, int j) {
  if (j == 0) return n == (1 << i) - 1;
  return i >= 1 && (n & 1) && j <= __builtin_ctz((unsigned)(n + 1));
}
WN_HD int slot_of_leaf(int m) { return m == 0 ? 0 : __builtin_ctz((unsigned)m); }

// U-turn verdicts of leaf n of doubling i from the summed dot products: bit j set when check j fired --
// stopCondition (WALNUTS.py:95-97) with the older state on the side opposite to xi.  `src.dots(j, d1, d2)`.
template <class Src>
WN_HD uint32_t fire_of(int i, int n, int xi, Src& src) {
  uint32_t firemask = 0;
  const int nchk = nchecks(i, n);
  for (int j = 0; j < nchk; ++j) {
    if (!need_check(i, n, j)) continue;
    double d1, d2;
    src.dots(j, d1, d2);
    if (((double)xi * d1 < 0.0) || ((double)xi * d2 < 0.0)) firemask |= (1u << j);
  }
  return firemask;
}

// ---- the bookkeeper: every per-chain scalar of one chain ---------------------------------------
struct FastBook {
  enum { ST_NEEDITER = 0, ST_FIRST, ST_RUN, ST_DRAIN, ST_DEAD };
  enum { K_NONE = 0, K_STEP, K_NEWITER };
  struct Slot {  // what the T waves were told to do in one tick
    int kind, n, row;
    bool discard, last, init;
    double h;
    // member-wise copy: an aggregate copy (memcpy) keeps the whole bookkeeper in scratch memory on the device
    WN_HD void set(const Slot& o) { kind = o.kind; n = o.n; row = o.row; discard = o.discard; last = o.last; init = o.init; h = o.h; }
    WN_HD void clear() { kind = 0; n = 0; row = 0; discard = false; last = false; init = false; h = 0.0; }
  };
  // WALNUTS.py scalars (names as in wn_chain.h ChainScalars)
  double H0, Ha, Hb, lpP, lpPL, WoldSum, WnewSum, h2;
  double orbitLen_, orbitLenSam_, indexStat_, indexStatOld_, timeLenF_, timeLenB_, Hmin, Hmax;
  int iter, i, xi, a, b, nleaf, n_rt, n_is;
  uint32_t ku;
  int nevalF, L_, Lold_, NdS, NdC, maxFint, maxBint, stopCode, ip, ipl;
  bool Hnan, pend_pick, swap_next, undo;
  int pick_ip, st;
  int zwant, zready;  // momentum normals: iteration wanted next / iteration the buffer holds
  Slot s0, s1;
  uint32_t cblkI, cblkR;  // cached Philox block of the uniform stream, issue side / retire side
  double cuI0, cuI1, cuR0, cuR1;
  uint32_t chain;  // global chain id
  double pre_lpn, pre_Hn, pre_Wsum, pre_ratio;  // retire_pre -> retire_step

  WN_HD void init(uint32_t chain_global, bool inrange, int numIter) {
    chain = chain_global;
    H0 = Ha = Hb = lpP = lpPL = 0.0; WoldSum = 1.0; WnewSum = 0.0; h2 = 0.0;
    orbitLen_ = orbitLenSam_ = indexStat_ = indexStatOld_ = timeLenF_ = timeLenB_ = 0.0;
    Hmin = INFINITY; Hmax = -INFINITY; Hnan = false;
    iter = 1; i = 0; xi = 1; a = b = 0; nleaf = 1; n_rt = n_is = 0; ku = 0;
    nevalF = L_ = Lold_ = NdS = NdC = maxFint = maxBint = stopCode = 0; ip = ipl = 0;
    pend_pick = false; swap_next = false; undo = false; pick_ip = 0;
    st = inrange ? ST_NEEDITER : ST_DEAD;
    zwant = (inrange && numIter >= 1) ? 1 : 0; zready = 0;
    pre_lpn = pre_Hn = pre_Wsum = pre_ratio = 0.0;
    s0.clear();
    s1.clear();
    if (inrange) { s1.kind = K_NEWITER; s1.init = true; }  // tick 0: every chain loads q0 (OP_INIT)
    cblkI = cblkR = 0xFFFFFFFFu;
    cuI0 = cuI1 = cuR0 = cuR1 = 0.0;
  }

  // k-th uniform of (chain, iteration) -- wn_rng.h stream 0.  Two one-block caches chosen at the call site: the
  // issue side reads ahead (step sizes, direction bits) while the retire side follows (picks, accepts), one
  // Philox block each per leaf pair.  (One cache indexed by block parity -- as an array or as selects between
  // members -- kept the whole bookkeeper in scratch memory: SROA gives up on addresses chosen at run time.)
  // A miss first asks `io.uniforms(side, iter, block, ...)`: on the device another wave computes the blocks this
  // chain will need next (wn_k_fast.hip), so the bookkeeper rarely runs Philox itself.
  template <class IO>
  WN_HD double Ui(const Params& prm, IO& io, uint32_t k) {
    const uint32_t blk = k >> 1;
    if (cblkI != blk) {
      double ua, ub;
      if (!io.uniforms(0, iter, blk, ua, ub)) {
        Philox4 r = philox4x32_10(blk, (uint32_t)iter, chain, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
        ua = u53(r.w[0], r.w[1]); ub = u53(r.w[2], r.w[3]);
      }
      cuI0 = ua; cuI1 = ub; cblkI = blk;
    }
    return (k & 1u) ? cuI1 : cuI0;
  }
  template <class IO>
  WN_HD double Ur(const Params& prm, IO& io, uint32_t k) {
    const uint32_t blk = k >> 1;
    if (cblkR != blk) {
      double ua, ub;
      if (!io.uniforms(1, iter, blk, ua, ub)) {
        Philox4 r = philox4x32_10(blk, (uint32_t)iter, chain, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
        ua = u53(r.w[0], r.w[1]); ub = u53(r.w[2], r.w[3]);
      }
      cuR0 = ua; cuR1 = ub; cblkR = blk;
    }
    return (k & 1u) ? cuR1 : cuR0;
  }
  template <class IO>
  WN_HD int dir(const Params& prm, IO& io, int j) {  // WALNUTS.py:216,283: B = floor(uniform(0,2,size=M)); xi = (-1)**B[j]
    const double u0 = 0.0 + (2.0 - 0.0) * Ui(prm, io, (uint32_t)j);
    return ((int)floor(u0) != 0) ? -1 : 1;
  }
  WN_HD static double h_from(const Params& prm, double U) {  // numpy legacy uniform(low, high) = low + (high-low)*U
    const double low = prm.H * (1 - prm.s), high = prm.H * (1 + prm.s);
    return low + (high - low) * U;
  }
  template <class IO>
  WN_HD double h_of(const Params& prm, IO& io, uint32_t k) { return h_from(prm, Ui(prm, io, k)); }

  WN_HD void start_doubling_common() {  // WALNUTS.py:283-294
    n_rt = 0; n_is = 0;
    nleaf = 1 << i;
    ipl = ip; lpPL = lpP; Lold_ = L_; indexStatOld_ = indexStat_;
    if (i > 0) WnewSum = 0.0;  // :390
  }
  WN_HD void pick_state(int idx, double lpn) {
    if (ip == ipl) ip ^= 1;  // keep qPropLast intact
    pend_pick = true; pick_ip = ip;
    lpP = lpn; L_ = idx;
    indexStat_ = xi > 0 ? timeLenF_ : -timeLenB_;
  }
  WN_HD void revert_prop() {  // qProp = qPropLast (:599-601, 614-616)
    ip = ipl; lpP = lpPL; L_ = Lold_; indexStat_ = indexStatOld_;
  }

  template <class IO>
  WN_HD void end_iteration(const Params& prm, IO& io) {  // WALNUTS.py:653-695 (the draw itself: next OP_NEWITER)
    double d[NDIAG];
    d[0] = L_; d[1] = NdS; d[2] = orbitLen_; d[3] = orbitLenSam_; d[4] = maxFint; d[5] = maxBint;
    d[6] = nevalF; d[7] = 0.0;
    d[8] = 0; d[9] = 0; d[10] = 0.0; d[11] = 0.0;  // fixedLeapFrog: If = Ib = c = 0, lwt = 0 (adaptiveIntegrators.py:57-59)
    d[12] = 0.0; d[13] = 0.0; d[14] = 0.0;
    d[15] = prm.H; d[16] = 1.0;
    d[17] = Hnan ? NAN : (Hmax - Hmin);
    d[18] = prm.delta; d[19] = (double)stopCode; d[20] = NdC; d[21] = 0; d[22] = 0;
    d[23] = indexStat_;
    io.write_diag((size_t)(iter - 1), d);
    iter++;
    if (s1.kind == K_STEP) { s1.discard = true; undo = true; }  // the leaf in flight belongs to the orbit that just ended
    swap_next = false;
    st = ST_NEEDITER;
  }

  template <class IO>
  WN_HD void doubling_end(const Params& prm, IO& io, bool sub, uint32_t firemask) {
    indexStat_ = indexStat_ / (timeLenF_ + timeLenB_);  // :595 (Q6)
    if (sub) {  // :597-605
      revert_prop();
      NdS = i; NdC = i + 1; stopCode = 5;
      end_iteration(prm, io);
      return;
    }
    if (!(Ur(prm, io, ku++) < WnewSum / WoldSum)) revert_prop();  // :613-616
    const bool joined = (firemask & 1u) != 0;                // :622
    if (joined) {                                            // :625-634 (lwtSum = 0: never "both ends passive")
      stopCode = 4;
      NdS = NdC = i + 1;
      orbitLenSam_ = orbitLen_;
      end_iteration(prm, io);
      return;
    }
    WoldSum += WnewSum;  // :641-648
    orbitLenSam_ = orbitLen_;
    NdS = NdC = i + 1;
    if (xi > 0) b += nleaf; else a -= nleaf;
    i++;
    if (i == prm.M) { end_iteration(prm, io); return; }
    const int nxi = dir(prm, io, i);  // :283
    if (nxi != xi) { const double th = Ha; Ha = Hb; Hb = th; xi = nxi; }  // the T waves swapped on OP_SWAP
    start_doubling_common();
    st = ST_RUN;
  }

  // retire the leaf the T waves integrated in the tick whose sums `io` holds (wn_chain.h after_micro + step_done)
  // The arithmetic of retiring a leaf that needs nothing but the leaf's lp / KE sums and the state the previous tick left:
  // energy, multinomial weight and the acceptance ratio of the new state.  Split off so that the device's bookkeeper
  // can do it while the T waves are still in front of their barrier (wn_k_fast.hip); retire_step() takes it from here.
  template <class IO>
  WN_HD void retire_pre(const Params& prm, IO& io) {
    if (!(s0.kind == K_STEP && !s0.discard)) return;
    pre_lpn = 0.5 * io.sum(Q_LP);              // lp = 1/2 q.g for the Gaussian targets
    pre_Hn = -pre_lpn + 0.5 * io.sum(Q_KE);    // adaptiveIntegrators.py:55
#if defined(WN4_NOEXP)  // timing diagnostic only (wrong weights): what the FP64 exp costs the bookkeeper's tick
    const double Wnew = 1.0 + 1e-9 * (-pre_Hn + H0 + 0.0);
#else
    const double Wnew = exp(-pre_Hn + H0 + 0.0);   // lwtSum = 0
#endif
    pre_Wsum = i == 0 ? Wnew : WnewSum + Wnew;     // :322 / :424
#if defined(WN4_NODIV)  // timing diagnostic only
    pre_ratio = Wnew * 0.5;
#else
    pre_ratio = Wnew / pre_Wsum;                   // :426
#endif
  }

  template <class IO>
  WN_HD void retire_step(const Params& prm, IO& io, const int n, const double h) {
    const double lpn = pre_lpn, Hn = pre_Hn;
    const uint32_t firemask = io.firemask(i, n, xi);  // bit j: U-turn check j of this leaf fired (fire_of)
    const bool first = (i == 0) || ((n & 1) == 0);
    const int idx = xi > 0 ? b + 1 + n : a - 1 - n;
    io.trace_step(iter, i, xi, h, Ha, lpn);  // one row per integrator call (traced runs; Ham0 = the neighbouring state's Hamiltonian)
    if (i == 0) ku += 1;                 // :298  the step size was drawn before the step
    else if ((n & 1) == 0) ku += 2;      // :395  size=2, before the first step of the pair
    nevalF += 1;
    if (isnan(Hn)) Hnan = true;
    Hmin = Hn < Hmin ? Hn : Hmin; Hmax = Hn > Hmax ? Hn : Hmax;
    {  // select form: `if (xi > 0) timeLenF_ += h; else timeLenB_ += h;` became a store through a phi of the two
       // members' addresses, which kept the whole bookkeeper in scratch memory
      const bool fw = xi > 0;
      const double tf = timeLenF_ + h, tb = timeLenB_ + h;
      maxFint = fw ? idx : maxFint; maxBint = fw ? maxBint : idx;
      timeLenF_ = fw ? tf : timeLenF_; timeLenB_ = fw ? timeLenB_ : tb;
    }
    if (i == 0) orbitLen_ += h;  // :300
    n_rt = n + 1;
    if (!isfinite(Hn)) {         // :316-319, 414-417, 457-459 (Q3, Q4)
      if (first) stopCode = 999;
      end_iteration(prm, io);
      return;
    }
    WnewSum = pre_Wsum;
    if (i == 0) {  // :322-328, 368-369: unconditional, no uniform
      pick_state(idx, lpn);
    } else {
      if (WnewSum > prm.wtSumThresh && Ur(prm, io, ku++) < pre_ratio) pick_state(idx, lpn);  // :426
      orbitLen_ += h;
    }
    Ha = Hn;
    bool sub = false;
    if (i >= 1 && (n & 1)) {
      const uint32_t m = ((2u << __builtin_ctz((unsigned)(n + 1))) - 1u) & ~1u;  // checks 1..ctz(n+1)
      sub = (firemask & m) != 0;
    }
    if (sub) doubling_end(prm, io, true, firemask);
    else if (n + 1 == nleaf) doubling_end(prm, io, false, firemask);
  }

  template <class IO>
  WN_HD void retire_newiter(const Params& prm, IO& io, const int row, const bool init_, const bool last_) {
    if (prm.generated == G_Q0_SUMSQ) io.write_gen((size_t)row, io.sum(Q_GEN));
    if (init_) {  // f0 = lpFun(q0)  (WALNUTS.py:177).  Iteration 1 was issued one tick ago, before this was
      lpP = 0.5 * io.sum(Q_LP);  // known: its copy of qPropLast's log density (start_doubling_common) is set here
      lpPL = lpP;
      return;
    }
    if (last_) return;
    // Hs[I0] = -f0 + 0.5*sum(v**2) (:256); multinomialLscale (:259)
    H0 = -lpP + 0.5 * io.sum(Q_KEINIT);
    Ha = Hb = H0;
    Hnan = isnan(H0); Hmin = Hmax = H0;
    if (iter + 1 <= prm.numIter) zwant = iter + 1;  // the momentum buffer is free again
  }

  // One bookkeeper tick: retire tick t-1, issue tick t+1.  `tn` = t+1.
  template <class IO>
  WN_HD Cmd tick(const Params& prm, IO& io, uint32_t tn) {
    retire_pre(prm, io);
    return tick_post(prm, io, tn);
  }
  template <class IO>
  WN_HD Cmd tick_post(const Params& prm, IO& io, uint32_t tn) {
    if (s0.kind == K_NEWITER) retire_newiter(prm, io, s0.row, s0.init, s0.last);
    else if (s0.kind == K_STEP && !s0.discard) retire_step(prm, io, s0.n, s0.h);
    s0.set(s1);
    Slot ns;
    ns.clear();
    Cmd c;
    c.hs = 0.0; c.op = 0u; c.n = 0u;
    if (pend_pick) { c.op |= OP_PICK | (pick_ip ? OP_IPPICK : 0u); pend_pick = false; }
    if (undo) { c.op |= OP_UNDO; undo = false; }
    const bool odd = (tn & 1u) != 0;
    if (st == ST_NEEDITER) {
      if (odd && (iter > prm.numIter || zready == iter)) {
        const bool last = iter > prm.numIter;
        c.op |= OP_NEWITER | (ip ? OP_IPNEW : 0u) | (last ? OP_LAST : 0u);
        c.n = (uint32_t)(iter - 1);
        ns.kind = K_NEWITER; ns.row = iter - 1; ns.last = last;
        if (last) st = ST_DEAD;
        else {  // WALNUTS.py:196-276, scalar part
          a = 0; b = 0; maxFint = 0; maxBint = 0;
          WoldSum = 1.0; WnewSum = 0.0;
          stopCode = 0; nevalF = 0; Lold_ = L_ = 0; orbitLen_ = orbitLenSam_ = 0.0; NdS = NdC = 0;
          indexStat_ = indexStatOld_ = 0.0; timeLenF_ = timeLenB_ = 0.0;
          cblkI = cblkR = 0xFFFFFFFFu;  // the uniform caches are keyed by block only
          i = 0;
          xi = dir(prm, io, 0);
          ku = (uint32_t)prm.M;
          start_doubling_common();
          st = ST_FIRST;
        }
      }
    } else if (st == ST_FIRST) {  // the single leaf of doubling 0 (:296-369)
      const double h = h_of(prm, io, ku);
      c.op |= OP_STEP;  // i = 0, n = 0
      c.hs = (double)xi * h;
      ns.kind = K_STEP; ns.n = 0; ns.h = h;
      n_is = 1;
      swap_next = (1 < prm.M) && dir(prm, io, 1) != xi;
      st = ST_DRAIN;
    } else if (st == ST_RUN) {
      bool go = true;
      double h = 0.0;
      if ((n_is & 1) == 0) {
        if (odd) go = false;  // leaf pairs start on even ticks: all chains of a group check / store in step
        else {
          const int inflight = n_is - n_rt;  // leaves issued and not yet retired (0 or 1)
          uint32_t kh = ku;
          if (inflight > 0) {
            if (WnewSum > prm.wtSumThresh) kh = ku + (uint32_t)inflight;  // their multinomial draws are certain
            else go = false;                                             // not known yet: wait for them
          }
          if (go) { h = h_of(prm, io, kh); h2 = h_of(prm, io, kh + 1u); }        // :395
        }
      } else {
        h = h2;
      }
      if (go) {
        c.op |= OP_STEP | ((uint32_t)i << OP_I_SHIFT);
        c.n = (uint32_t)n_is;
        c.hs = (double)xi * h;
        ns.kind = K_STEP; ns.n = n_is; ns.h = h;
        n_is++;
        if (n_is == nleaf) {
          swap_next = (i + 1 < prm.M) && dir(prm, io, i + 1) != xi;
          st = ST_DRAIN;
        }
      }
    } else if (st == ST_DRAIN) {
      if (swap_next) { c.op |= OP_SWAP; swap_next = false; }
    }
    s1.set(ns);
    return c;
  }

  // ---- the ordinary leaf: fast path of tick() ----------------------------------------------------------------
  // Most ticks of a chain are the same event: in the middle of a doubling (i >= 1) the leaf of tick t-1 is retired -- a
  // finite energy, no sub-tree U-turn, its multinomial draw taken -- and leaf n+2 is issued while leaf n+1 is in flight.
  // tick() handles it through the same branches as every rare event; on the device every branch is an exec-mask save, a
  // vote and a jump, and the bookkeeper's ~600 instructions per tick compete with the T wave's MFMAs for the SIMD (FP64
  // MFMAs and vector instructions do not overlap on this chip: tools/mfma_valu_coissue.hip).  fast_prepare() decides
  // (after retire_pre(), WITHOUT touching the state) whether this tick is that event and computes what it changes, straight-line;
  // fast_commit() applies it.  State, command and draws are bit for bit those of tick() (tests/test_fast_book.py runs
  // both on every tick and compares every member).  A wave takes the fast path when all its chains are ordinary.
  struct Fast {  // (the leaf's energy, weight sum and acceptance ratio are retire_pre()'s: pre_lpn, pre_Hn, pre_Wsum, pre_ratio)
    double h, hnext2;
    uint32_t ku_new;
    int n, idx;
    bool pick, even_issue;
  };
  // u = k-th uniform of this iteration if it is at hand without running Philox (either one-block cache, or the io's window)
  template <class IO>
  WN_HD bool peek(const IO& io, uint32_t k, double& u) const {
    const uint32_t bl = k >> 1;
    double w0, w1;
    const bool hw = io.uniforms(2, iter, bl, w0, w1);
    // (select form over register copies: a select between two loaded members becomes a load from a selected ADDRESS, and a
    // struct whose members' addresses are selected at run time stays in scratch memory)
    double r0 = cuR0, r1 = cuR1, i0 = cuI0, i1 = cuI1;
    WN_OPAQUE4(r0, r1, i0, i1);
    const bool cr = bl == cblkR, ci = bl == cblkI;
    const double x0 = cr ? r0 : ci ? i0 : w0, x1 = cr ? r1 : ci ? i1 : w1;
    u = (k & 1u) ? x1 : x0;
    return cr || ci || hw;
  }
  // first uniform index an ordinary retirement of this tick reads (the io prefetches its block and the next two)
  WN_HD uint32_t fast_first_uniform() const { return ku + (((s0.n & 1) == 0) ? 2u : 0u); }
  template <class IO>
  WN_HD bool fast_prepare(const Params& prm, const IO& io, uint32_t tn, Fast& f) const {
    bool ok = s0.kind == K_STEP && !s0.discard && s1.kind == K_STEP && !s1.discard && st == ST_RUN && i >= 1 && !undo;
    const int n = s0.n;
    f.n = n;
    ok = ok && isfinite(pre_Hn);  // retire_pre() has run
    // no sub-tree U-turn at an odd leaf; the last leaf of the doubling is not ordinary (s1 is leaf n+1 < nleaf)
    if (n & 1) {
      const uint32_t m = ((2u << __builtin_ctz((unsigned)(n + 1))) - 1u) & ~1u;
      ok = ok && (io.firemask(i, n, xi) & m) == 0;
    }
    ok = ok && s1.n == n + 1 && n_is == n + 2 && n_rt == n;
    f.idx = xi > 0 ? b + 1 + n : a - 1 - n;
    uint32_t k = ku + (((n & 1) == 0) ? 2u : 0u);     // :395 size=2, before the first step of the pair
    // the leaf's multinomial draw (:426) must happen (then leaf n+1's will too: WnewSum does not decrease)
    ok = ok && pre_Wsum > prm.wtSumThresh;
    double u;
    ok = peek(io, k, u) && ok;
    f.pick = u < pre_ratio;
    k += 1u;
    // issue leaf n+2 (tick_post, ST_RUN): not the last leaf of the doubling (its issue draws the next direction)
    ok = ok && n + 3 < nleaf;
    f.even_issue = (n & 1) == 0;
    ok = ok && (!f.even_issue || (tn & 1u) == 0);  // leaf pairs start on even ticks
    double ua = 0.0, ub = 0.0;
    if (f.even_issue) {  // h, h2 = uniforms k+1, k+2 (one leaf in flight: its draw comes first)
      ok = peek(io, k + 1u, ua) && ok;
      ok = peek(io, k + 2u, ub) && ok;
      f.h = h_from(prm, ua);
      f.hnext2 = h_from(prm, ub);
    } else {
      f.h = h2;
      f.hnext2 = h2;
    }
    f.ku_new = k;
    return ok;
  }
  template <class IO>
  WN_HD Cmd fast_commit(const Params& prm, IO& io, const Fast& f) {
    const int n = f.n;
    const double h = s0.h;
    io.trace_step(iter, i, xi, h, Ha, pre_lpn);
    nevalF += 1;
    Hmin = pre_Hn < Hmin ? pre_Hn : Hmin; Hmax = pre_Hn > Hmax ? pre_Hn : Hmax;
    {
      const bool fw = xi > 0;
      const double tf = timeLenF_ + h, tb = timeLenB_ + h;
      maxFint = fw ? f.idx : maxFint; maxBint = fw ? maxBint : f.idx;
      timeLenF_ = fw ? tf : timeLenF_; timeLenB_ = fw ? timeLenB_ : tb;
    }
    n_rt = n + 1;
    WnewSum = pre_Wsum;
    ku = f.ku_new;
    // (the one-block uniform caches keep what they hold: still valid blocks of this iteration)
    {  // pick_state(idx, lpn), select form
      const bool p = f.pick;
      const int nip = (ip == ipl) ? (ip ^ 1) : ip;
      ip = p ? nip : ip;
      pick_ip = p ? ip : pick_ip;
      pend_pick = pend_pick || p;
      lpP = p ? pre_lpn : lpP;
      L_ = p ? f.idx : L_;
      const double is = xi > 0 ? timeLenF_ : -timeLenB_;
      indexStat_ = p ? is : indexStat_;
    }
    orbitLen_ += h;
    Ha = pre_Hn;
    // tick_post
    s0.kind = K_STEP; s0.n = n + 1; s0.row = 0; s0.discard = false; s0.last = false; s0.init = false; s0.h = s1.h;
    Cmd c;
    c.op = OP_STEP | ((uint32_t)i << OP_I_SHIFT) | (pend_pick ? (OP_PICK | (pick_ip ? OP_IPPICK : 0u)) : 0u);
    pend_pick = false;
    c.n = (uint32_t)(n + 2);
    c.hs = (double)xi * f.h;
    h2 = f.hnext2;
    s1.kind = K_STEP; s1.n = n + 2; s1.row = 0; s1.discard = false; s1.last = false; s1.init = false; s1.h = f.h;
    n_is = n + 3;
    return c;
  }
  WN_HD bool finished() const { return st == ST_DEAD && s0.kind == K_NONE && s1.kind == K_NONE && !pend_pick; }
};

}  // namespace fast
}  // namespace wn
