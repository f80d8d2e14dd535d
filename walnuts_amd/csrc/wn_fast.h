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
// The lag is a template parameter.  LAG = 2 is the protocol above (wn_k_fast.hip).  LAG = 3 (wn_k_fast4.hip) retires tick t-2
// while tick t runs: the bookkeeper then has TWO vector ticks for one of its own, so neither wave waits for the other on an
// ordinary leaf (with LAG = 2 every slow tick of either side stalled the other one).  Two leaves are in flight instead of one:
// a pick copies the position of tick t-3 (the T wave keeps it), an iteration end discards up to two leaves, and the step sizes of
// a leaf pair are issued ahead of up to two retirements (their draws counted as above).
//
// FastBook is plain scalar code, host + device: a CPU test harness under tests/ runs it against the oracle on
// the CPU with a scalar stand-in for the T waves (test tool only).
#pragma once
#include <math.h>
#include <stdint.h>

#include "wn_chain.h"

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
  OP_PICK = 32u,     // copy the position of tick t-LAG into proposal[ippick]
  OP_IPPICK = 64u,   // bit: ippick
  OP_IPNEW = 128u,   // bit: ipnew
  OP_I_SHIFT = 8,    // bits 8..12: doubling index i of an OP_STEP (n = leaf index inside the doubling)
  OP_UNDO2 = 1u << 13, // (LAG = 3) a second one
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
template <int LAG>
struct FastBookT {
  static_assert(LAG == 2 || LAG == 3, "ticks between a leaf and its retirement");
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
  bool Hnan, pend_pick, swap_next;
  int undo;  // trace rows the T wave has to take back (leaves discarded at an iteration end)
  int pick_ip, st;
  int zwant, zready;  // momentum normals: iteration wanted next / iteration the buffer holds
  Slot s0, s1, s2;  // ticks t-LAG+1 (retired next) .. t+1 (s2: LAG = 3 only)
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
    pend_pick = false; swap_next = false; undo = 0; pick_ip = 0;
    st = inrange ? ST_NEEDITER : ST_DEAD;
    zwant = (inrange && numIter >= 1) ? 1 : 0; zready = 0;
    pre_lpn = pre_Hn = pre_Wsum = pre_ratio = 0.0;
    s0.clear();
    s1.clear();
    s2.clear();
    if (inrange) {  // tick 0: every chain loads q0 (OP_INIT)
      Slot& f = LAG == 3 ? s2 : s1;
      f.kind = K_NEWITER; f.init = true;
    }
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
    // the leaves in flight belong to the orbit that just ended
    if (s1.kind == K_STEP) { s1.discard = true; undo += 1; }
    if (LAG == 3) { if (s2.kind == K_STEP) { s2.discard = true; undo += 1; } }
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

  // One bookkeeper tick: retire tick t+1-LAG, issue tick t+1.  `tn` = t+1.
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
    if (LAG == 3) s1.set(s2);
    Slot ns;
    ns.clear();
    Cmd c;
    c.hs = 0.0; c.op = 0u; c.n = 0u;
    if (pend_pick) { c.op |= OP_PICK | (pick_ip ? OP_IPPICK : 0u); pend_pick = false; }
    if (undo > 0) { c.op |= OP_UNDO | (undo > 1 ? OP_UNDO2 : 0u); undo = 0; }
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
          // leaves issued and not yet retired (at most LAG - 1): retiring an even one advances the draw count by its pair's two
          // step sizes (:395) -- drawn when the pair was issued -- and every one by its multinomial draw (:426)
          uint32_t kh = ku;
          if (n_is > n_rt) {
            if (WnewSum > prm.wtSumThresh) kh = first_issue_uniform();  // their multinomial draws are certain
            else go = false;                                            // not known yet: wait for them
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
    if (LAG == 3) s2.set(ns); else s1.set(ns);
    return c;
  }
  // first uniform index an ordinary retirement of this tick reads, and the first one the issue of an ordinary leaf pair behind it
  // reads (the io prefetches their blocks)
  WN_HD uint32_t first_retire_uniform() const { return ku + (((s0.n & 1) == 0) ? 2u : 0u); }
  WN_HD uint32_t first_issue_uniform() const {  // ku + sum over the leaves m in flight of (m even ? 3 : 1)
    return ku + (uint32_t)(n_is - n_rt) + 2u * (uint32_t)(((n_is + 1) >> 1) - ((n_rt + 1) >> 1));
  }

  WN_HD bool finished() const { return st == ST_DEAD && s0.kind == K_NONE && s1.kind == K_NONE && (LAG == 2 || s2.kind == K_NONE) && !pend_pick; }
};
using FastBook = FastBookT<2>;

// ================================================================================================
// The same bookkeeper (LAG = 3) as TWO instruction streams, wn_k_fast4.hip: a RETIRE wave and an ISSUE wave, each
// serving the chains of two T waves.  Why: FP64 matrix and vector instructions share a SIMD's issue slots on this
// chip (tools/mfma_valu_coissue.hip), so a tick's period is the T wave's instructions PLUS the bookkeeper's; one
// bookkeeper stream per T wave (~600 instructions for four chains in sixteen redundant replicas) cost 2.9 k of a
// 7.5 k-cycle tick.  Eight chains per stream halve the streams, and cutting the stream in two keeps the SIMDs even.
//   RetireBook  (R wave)  everything FastBookT does when it retires a tick -- energies, weights, multinomial and
//       acceptance draws, U-turn verdicts, doubling / iteration ends, diagnostics -- told WHAT the tick was by the
//       command record the T wave executed (kind, leaf index n, |hs| = h), not by the issue side;
//   IssueBook   (I wave)  everything it does when it issues one: the automaton NEEDITER / FIRST / RUN / DRAIN, step
//       sizes and direction bits, momentum normals and the uniform window (wn_k_fast4.hip).
// One message per chain and tick goes R -> I (RMsg): the draw count and retired-leaf count the issue side needs to
// index its step-size draws, whether the multinomial draws of the leaves in flight are certain, doubling / iteration
// end events, the proposal index, the pick of the retired leaf.  I(t) applies R(t)'s message and then does exactly
// what FastBookT<3>::tick_post does behind its retirement: results are those of FastBookT<3>, tick for tick
// (tests/test_fast_book.py runs both against the oracle).
struct alignas(16) RMsg {
  uint32_t ku, w, iter, zwant;
};
enum {
  RM_NRT_MASK = 0xFFFFu,  // leaves of the running doubling retired so far
  RM_CERTAIN = 1u << 16,  // WnewSum > wtSumThresh: the leaves in flight will take their multinomial draws
  RM_CONT = 1u << 17,     // this tick's retirement ended a doubling; the orbit goes on with doubling i+1
  RM_END = 1u << 18,      // ... ended the iteration (`iter` is the next one's index)
  RM_IP = 1u << 19,       // proposal buffer that holds the state the next iteration starts from
  RM_PICK = 1u << 20,     // the retired leaf was picked: the T wave copies its position into proposal[RM_PICKIP]
  RM_PICKIP = 1u << 21,
  RM_DEAD = 1u << 22      // the run's last draw has been retired
};

struct BookUniforms {  // the chain's uniform stream (wn_rng.h stream 0) behind a one-block cache
  uint32_t chain;
  int iter;
  uint32_t cblk;
  double cu0, cu1;
  template <class IO>
  WN_HD double U(const Params& prm, IO& io, int side, uint32_t k) {
    const uint32_t blk = k >> 1;
    if (cblk != blk) {
      double ua, ub;
      if (!io.uniforms(side, iter, blk, ua, ub)) {
        Philox4 r = philox4x32_10(blk, (uint32_t)iter, chain, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
        ua = u53(r.w[0], r.w[1]); ub = u53(r.w[2], r.w[3]);
      }
      cu0 = ua; cu1 = ub; cblk = blk;
    }
    return (k & 1u) ? cu1 : cu0;
  }
  template <class IO>
  WN_HD int dir(const Params& prm, IO& io, int side, int j) {  // WALNUTS.py:216,283
    const double u0 = 0.0 + (2.0 - 0.0) * U(prm, io, side, (uint32_t)j);
    return ((int)floor(u0) != 0) ? -1 : 1;
  }
};

// WARM: the instantiation that also runs warm-up adaptation (a compile-time switch: the plain kernels carry none of it)
template <bool WARM>
struct RetireBookT : BookUniforms {
  double H0, Ha, Hb, lpP, lpPL, WoldSum, WnewSum;
  double orbitLen_, orbitLenSam_, indexStat_, indexStatOld_, timeLenF_, timeLenB_, Hmin, Hmax;
  int i, xi, a, b, nleaf, n_rt;
  uint32_t ku;
  int nevalF, L_, Lold_, NdS, NdC, maxFint, maxBint, stopCode, ip, ipl;
  bool Hnan, pend_pick, discarding, dead;
  int pick_ip, zwant;
  uint32_t ev;
  double pre_lpn, pre_Hn, pre_Wsum, pre_ratio;
  // warm-up adaptation (WALNUTS.py:120-127, 313, 701-712; round 5): the chain's step size H and tolerance delta, the P2 markers of
  // log(igrConst) -- wn_chain.h wn_p2_push / wn_adapt_update, the engine's own functions.  Runs without warm-up never touch them
  // (prm.warmupIter = 0: Hc stays prm.H bit for bit).
  double Hc, dc, p2q[5];
  int p2m[5], p2n;

  WN_HD void init_warm(const Params& prm) {
    Hc = prm.H; dc = prm.delta; p2n = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { p2q[k] = 0.0; p2m[k] = k + 1; }
  }
  WN_HD void init(uint32_t chain_global, bool inrange, int numIter) {
    chain = chain_global; iter = 1; cblk = 0xFFFFFFFFu; cu0 = cu1 = 0.0;
    H0 = Ha = Hb = lpP = lpPL = 0.0; WoldSum = 1.0; WnewSum = 0.0;
    orbitLen_ = orbitLenSam_ = indexStat_ = indexStatOld_ = timeLenF_ = timeLenB_ = 0.0;
    Hmin = INFINITY; Hmax = -INFINITY; Hnan = false;
    i = 0; xi = 1; a = b = 0; nleaf = 1; n_rt = 0; ku = 0;
    nevalF = L_ = Lold_ = NdS = NdC = maxFint = maxBint = stopCode = 0; ip = ipl = 0;
    pend_pick = false; discarding = false; dead = !inrange; pick_ip = 0;
    zwant = (inrange && numIter >= 1) ? 1 : 0;
    ev = 0u;
    pre_lpn = pre_Hn = pre_Wsum = pre_ratio = 0.0;
  }
  WN_HD void start_doubling_common() {  // WALNUTS.py:283-294
    n_rt = 0;
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
  WN_HD void revert_prop() { ip = ipl; lpP = lpPL; L_ = Lold_; indexStat_ = indexStatOld_; }  // qProp = qPropLast (:599-601, 614-616)

  template <class IO>
  WN_HD void end_iteration(const Params& prm, IO& io) {  // WALNUTS.py:653-695 (the draw itself: next OP_NEWITER)
    double d[NDIAG];
    d[0] = L_; d[1] = NdS; d[2] = orbitLen_; d[3] = orbitLenSam_; d[4] = maxFint; d[5] = maxBint;
    d[6] = nevalF; d[7] = 0.0;
    d[8] = 0; d[9] = 0; d[10] = 0.0; d[11] = 0.0;  // fixedLeapFrog: If = Ib = c = 0, lwt = 0 (adaptiveIntegrators.py:57-59)
    d[12] = 0.0; d[13] = 0.0; d[14] = 0.0;
    d[15] = WARM ? Hc : prm.H; d[16] = 1.0;
    d[17] = Hnan ? NAN : (Hmax - Hmin);
    d[18] = WARM ? dc : prm.delta; d[19] = (double)stopCode; d[20] = NdC; d[21] = 0; d[22] = 0;
    d[23] = indexStat_;
    if constexpr (WARM) if (iter <= prm.warmupIter) {  // tuning parameter adaptation (:701-712) -- after the row is filled: it records the values this iteration ran with
      wn_adapt_update(prm, iter, d[17], io.facs(), io.facs_writer(), dc, Hc, p2q, p2n);
      io.write_hc(Hc);  // (the issue side draws the next iteration's step sizes around the new H; it reads this behind the RM_END message)
    }
    io.write_diag((size_t)(iter - 1), d);
    iter++;
    discarding = true;  // the leaves in flight belong to the orbit that just ended: every step up to the next iteration start
    ev = RM_END;
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
    if (!(U(prm, io, 1, ku++) < WnewSum / WoldSum)) revert_prop();  // :613-616
    const bool joined = (firemask & 1u) != 0;                   // :622
    if (joined) {                                               // :625-634 (lwtSum = 0: never "both ends passive")
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
    const int nxi = dir(prm, io, 1, i);  // :283
    if (nxi != xi) { const double th = Ha; Ha = Hb; Hb = th; xi = nxi; }  // the T waves swapped on OP_SWAP
    start_doubling_common();
    ev = RM_CONT;
  }
  template <class IO>
  WN_HD void retire_step(const Params& prm, IO& io, const int n, const double h) {
    const double lpn = pre_lpn, Hn = pre_Hn;
    const uint32_t firemask = io.firemask(i, n, xi);  // bit j: U-turn check j of this leaf fired (fire_of)
    const bool first = (i == 0) || ((n & 1) == 0);
    const int idx = xi > 0 ? b + 1 + n : a - 1 - n;
    io.trace_step(iter, i, xi, h, Ha, lpn);
    if constexpr (WARM) if (iter <= prm.warmupIter && prm.adaptH)  // igrConstQ.push(np.log(intOut.igrConst)) :313; igrConst of fixedLeapFrog: adaptiveIntegrators.py:59
      wn_p2_push(prm, p2q, p2m, p2n, log(h * pow(fmax(1.0e-10, fabs(Ha - Hn)), -1.0 / 3.0)));
    if (i == 0) ku += 1;                 // :298  the step size was drawn before the step
    else if ((n & 1) == 0) ku += 2;      // :395  size=2, before the first step of the pair
    nevalF += 1;
    if (isnan(Hn)) Hnan = true;
    Hmin = Hn < Hmin ? Hn : Hmin; Hmax = Hn > Hmax ? Hn : Hmax;
    {
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
      if (WnewSum > prm.wtSumThresh && U(prm, io, 1, ku++) < pre_ratio) pick_state(idx, lpn);  // :426
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
  WN_HD void retire_newiter(const Params& prm, IO& io, const Cmd& c) {
    const bool init_ = (c.op & OP_INIT) != 0, last_ = (c.op & OP_LAST) != 0;
    discarding = false;
    if (!init_ && !last_) {  // WALNUTS.py:196-276, scalar part (FastBookT does it when it issues the iteration start)
      a = 0; b = 0; maxFint = 0; maxBint = 0;
      WoldSum = 1.0; WnewSum = 0.0;
      stopCode = 0; nevalF = 0; Lold_ = L_ = 0; orbitLen_ = orbitLenSam_ = 0.0; NdS = NdC = 0;
      indexStat_ = indexStatOld_ = 0.0; timeLenF_ = timeLenB_ = 0.0;
      cblk = 0xFFFFFFFFu;  // the uniform cache is keyed by block only
      i = 0;
      xi = dir(prm, io, 1, 0);
      ku = (uint32_t)prm.M;
      start_doubling_common();
    }
    if (prm.generated == G_Q0_SUMSQ) io.write_gen((size_t)c.n, io.sum(Q_GEN));
    if (init_) {  // f0 = lpFun(q0)  (WALNUTS.py:177)
      lpP = 0.5 * io.sum(Q_LP);
      lpPL = lpP;
      return;
    }
    if (last_) { dead = true; return; }
    H0 = -lpP + 0.5 * io.sum(Q_KEINIT);  // Hs[I0] = -f0 + 0.5*sum(v**2) (:256)
    Ha = Hb = H0;
    Hnan = isnan(H0); Hmin = Hmax = H0;
    io.write_h0(H0);  // (the issue side certifies the first leaf of a doubling itself: IssueBook::early_certain)
    if (iter + 1 <= prm.numIter) zwant = iter + 1;  // the momentum buffer is free again
  }
  // does the command `c` (the one the T wave executed in the tick whose sums `io` holds) retire a leaf?
  WN_HD bool retires_step(const Cmd& c) const { return (c.op & OP_STEP) != 0 && (c.op & OP_NEWITER) == 0 && !discarding; }
  template <class IO>
  WN_HD RMsg tick(const Params& prm, IO& io, const Cmd& c) {
    ev = 0u;
    if (c.op & OP_NEWITER) retire_newiter(prm, io, c);
    else if (retires_step(c)) {
      pre_lpn = 0.5 * io.sum(Q_LP);              // lp = 1/2 q.g for the Gaussian targets
      pre_Hn = -pre_lpn + 0.5 * io.sum(Q_KE);    // adaptiveIntegrators.py:55
      const double Wnew = exp(-pre_Hn + H0 + 0.0);   // lwtSum = 0
      pre_Wsum = i == 0 ? Wnew : WnewSum + Wnew;     // :322 / :424
      pre_ratio = Wnew / pre_Wsum;                   // :426
      retire_step(prm, io, (int)c.n, fabs(c.hs));    // hs = xi * h
    }
    RMsg m;
    m.ku = ku; m.iter = (uint32_t)iter; m.zwant = (uint32_t)zwant;
    m.w = ((uint32_t)n_rt & RM_NRT_MASK) | (WnewSum > prm.wtSumThresh ? RM_CERTAIN : 0u) | ev | (ip ? RM_IP : 0u) |
          (pend_pick ? (RM_PICK | (pick_ip ? RM_PICKIP : 0u)) : 0u) | (dead ? RM_DEAD : 0u);
    pend_pick = false;
    return m;
  }
};

using RetireBook = RetireBookT<false>;

template <bool WARM>
struct IssueBookT : BookUniforms {
  enum { ST_NEEDITER = 0, ST_FIRST, ST_RUN, ST_DRAIN, ST_DEAD };
  int st, i, xi, nleaf, n_is, n_rt;
  uint32_t ku;
  bool certain, swap_next, c1step, c2step, rdead;
  double h2;
  int zwant, zready, ip;
  double Hc;  // the chain's step size H (warm-up adapts it on the retire side: read behind every RM_END of a warm-up iteration)

  WN_HD void init_warm(const Params& prm) { Hc = prm.H; }
  WN_HD void init(uint32_t chain_global, bool inrange, int numIter) {
    chain = chain_global; iter = 1; cblk = 0xFFFFFFFFu; cu0 = cu1 = 0.0;
    st = inrange ? ST_NEEDITER : ST_DEAD;
    i = 0; xi = 1; nleaf = 1; n_is = n_rt = 0; ku = 0u;
    certain = false; swap_next = false; c1step = c2step = false; rdead = !inrange;
    h2 = 0.0;
    zwant = (inrange && numIter >= 1) ? 1 : 0; zready = 0; ip = 0;
  }
  WN_HD double h_from(const Params& prm, double Uu) const {  // numpy legacy uniform(low, high) = low + (high-low)*U
    const double Hx = WARM ? Hc : prm.H;
    const double low = Hx * (1 - prm.s), high = Hx * (1 + prm.s);
    return low + (high - low) * Uu;
  }
  template <class IO>
  WN_HD double h_of(const Params& prm, IO& io, uint32_t k) { return h_from(prm, U(prm, io, 0, k)); }
  WN_HD uint32_t first_issue_uniform() const {  // ku + sum over the leaves m in flight of (m even ? 3 : 1)
    return ku + (uint32_t)(n_is - n_rt) + 2u * (uint32_t)(((n_is + 1) >> 1) - ((n_rt + 1) >> 1));
  }
  // The second leaf pair of a doubling is due while its first pair is still in flight, and whether those two leaves take their
  // multinomial draws (WALNUTS.py:426: only if WnewSum > wtSumThresh) decides where in the stream the pair's step sizes are.  The
  // retire side knows one tick too late (two idle ticks per doubling).  The issue side can know in time: the sub-tree's weight sum
  // starts from zero (:390), so the first leaf's own weight decides, and its sums are in LDS when the T wave has finished that tick.
  // wants_early(): this tick's issue is that case; early_certain(): the retire side's arithmetic (RetireBook::tick), bit for bit.
  WN_HD bool wants_early(uint32_t tn, const RMsg& m) const {
    return st == ST_RUN && n_is == 2 && (tn & 1u) == 0 && (m.w & (RM_END | RM_CONT | RM_CERTAIN | RM_NRT_MASK)) == 0;
  }
  WN_HD static bool early_certain(const Params& prm, double lp_sum, double ke_sum, double H0) {
    const double lpn = 0.5 * lp_sum;
    const double Hn = -lpn + 0.5 * ke_sum;
    const double Wnew = exp(-Hn + H0 + 0.0);
    return (0.0 + Wnew) > prm.wtSumThresh;
  }
  // ---- the ordinary tick, straight-line (round 5) -------------------------------------------------------------------------------
  // The issue wave's pass is what the kernel's tick is made of (wn_k_fast4.hip, -DWN4_PAD_I: every instruction added to it costs the
  // tick ~1.7 quad-cycles; the T and retire waves' cost a third / a half of that), and most of its ticks are the same thing for all its
  // chains: a chain in the middle of a doubling gets the next leaf of its pair.  fast_ok(): this tick is that for this chain --
  // state RUN, the message carries no doubling / iteration end, an even tick starts a pair whose step sizes can be drawn (the draws of
  // the leaves in flight are certain), an odd tick issues the second leaf and it is not the doubling's last.  tick_fast(): what tick()
  // does in that case, bit for bit, in select form and without the uniform cache's branches (fast_kh() says which uniforms an even
  // tick reads: the caller checks that its window holds them).  The kernel takes it when EVERY chain of the wave qualifies; the CPU
  // harness takes it whenever the chain does (the scalar T-wave stand-in under tests/), so both paths run against the oracle.
  WN_HD bool fast_ok(uint32_t tn, const RMsg& m) const {
    const int nrt = (int)(m.w & RM_NRT_MASK);
    const bool ev = (n_is & 1) == 0 && (n_is <= nrt || (m.w & RM_CERTAIN) != 0);
    const bool od = (n_is & 1) != 0 && n_is + 1 < nleaf;
    return st == ST_RUN && (m.w & (RM_END | RM_CONT)) == 0 && ((tn & 1u) ? od : ev);
  }
  WN_HD uint32_t fast_kh(const RMsg& m) const {  // first uniform of the pair's two step sizes (even ticks), with the message applied
    const int nrt = (int)(m.w & RM_NRT_MASK);
    const uint32_t ahead = m.ku + (uint32_t)(n_is - nrt) + 2u * (uint32_t)(((n_is + 1) >> 1) - ((nrt + 1) >> 1));
    return n_is > nrt ? ahead : m.ku;
  }
  template <class IO>
  WN_HD Cmd tick_fast(const Params& prm, IO& io, uint32_t tn, const RMsg& m) {
    ku = m.ku; n_rt = (int)(m.w & RM_NRT_MASK); certain = (m.w & RM_CERTAIN) != 0;
    zwant = (int)m.zwant; rdead = (m.w & RM_DEAD) != 0;
    Cmd c;
    c.op = OP_STEP | ((uint32_t)i << OP_I_SHIFT) | ((m.w & RM_PICK) ? (OP_PICK | ((m.w & RM_PICKIP) ? OP_IPPICK : 0u)) : 0u);
    c.n = (uint32_t)n_is;
    double h = h2;
    if ((tn & 1u) == 0) {  // (the same for every chain of the wave: a scalar branch)
      const uint32_t kh = fast_kh(m);
      h = h_from(prm, io.u_direct(iter, kh));
      h2 = h_from(prm, io.u_direct(iter, kh + 1u));  // :395
    }
    c.hs = (double)xi * h;
    n_is++;
    c2step = c1step;
    c1step = true;
    return c;
  }

  // One tick: apply the retire side's message of this tick, issue tick `tn`.
  template <class IO>
  WN_HD Cmd tick(const Params& prm, IO& io, uint32_t tn, const RMsg& m, bool early = false) {
    Cmd c;
    c.hs = 0.0; c.op = 0u; c.n = 0u;
    ku = m.ku; n_rt = (int)(m.w & RM_NRT_MASK); certain = (m.w & RM_CERTAIN) != 0 || early;
    zwant = (int)m.zwant; rdead = (m.w & RM_DEAD) != 0;
    if (m.w & RM_PICK) c.op |= OP_PICK | ((m.w & RM_PICKIP) ? OP_IPPICK : 0u);
    if (m.w & RM_END) {  // the leaves in flight (the commands of ticks tn-1, tn-2) belong to the orbit that just ended
      const int u = (c1step ? 1 : 0) + (c2step ? 1 : 0);
      if (u > 0) c.op |= OP_UNDO | (u > 1 ? OP_UNDO2 : 0u);
      c1step = c2step = false;
      if constexpr (WARM) if (iter <= prm.warmupIter) Hc = io.read_hc();  // (`iter` is still the iteration that just ended: warm-up adapted its H, :711-712)
      st = ST_NEEDITER; iter = (int)m.iter; ip = (m.w & RM_IP) ? 1 : 0;
      swap_next = false;
      cblk = 0xFFFFFFFFu;  // the uniform cache is keyed by block only
    }
    if (m.w & RM_CONT) {  // doubling i+1 (WALNUTS.py:283-294)
      i++;
      xi = dir(prm, io, 0, i);
      n_is = 0; nleaf = 1 << i;
      st = ST_RUN;
    }
    const bool odd = (tn & 1u) != 0;
    if (st == ST_NEEDITER) {
      if (odd && (iter > prm.numIter || zready == iter)) {
        const bool last = iter > prm.numIter;
        c.op |= OP_NEWITER | (ip ? OP_IPNEW : 0u) | (last ? OP_LAST : 0u);
        c.n = (uint32_t)(iter - 1);
        if (last) st = ST_DEAD;
        else {
          i = 0;
          xi = dir(prm, io, 0, 0);
          n_is = 0; nleaf = 1;
          st = ST_FIRST;
        }
      }
    } else if (st == ST_FIRST) {  // the single leaf of doubling 0 (:296-369)
      const double h = h_of(prm, io, (uint32_t)prm.M);
      c.op |= OP_STEP;  // i = 0, n = 0
      c.hs = (double)xi * h;
      n_is = 1;
      swap_next = (1 < prm.M) && dir(prm, io, 0, 1) != xi;
      st = ST_DRAIN;
    } else if (st == ST_RUN) {
      bool go = true;
      double h = 0.0;
      if ((n_is & 1) == 0) {
        if (odd) go = false;  // leaf pairs start on even ticks: all chains of a group check / store in step
        else {
          uint32_t kh = ku;
          if (n_is > n_rt) {
            if (certain) kh = first_issue_uniform();  // the multinomial draws of the leaves in flight are certain
            else go = false;                          // not known yet: wait for them
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
        n_is++;
        if (n_is == nleaf) {
          swap_next = (i + 1 < prm.M) && dir(prm, io, 0, i + 1) != xi;
          st = ST_DRAIN;
        }
      }
    } else if (st == ST_DRAIN) {
      if (swap_next) { c.op |= OP_SWAP; swap_next = false; }
    }
    c2step = c1step;
    c1step = (c.op & OP_STEP) != 0;
    return c;
  }
  WN_HD bool finished() const { return st == ST_DEAD && rdead; }
};
using IssueBook = IssueBookT<false>;

}  // namespace fast
}  // namespace wn
