// fastsim.cpp -- CPU run of the role-split headline path's bookkeeper (walnuts_amd/csrc/wn_fast.h,
// FastBookT<LAG>) with a scalar stand-in for the T waves of wn_k_fast.hip (LAG = 2) / wn_k_fast4.hip (LAG = 3): same
// command / partial-sum protocol, same lag, one chain at a time, plain loops for the vectors.
//
// TEST TOOL ONLY (see hostsim.cpp): checks the issue / retire automaton against the oracle in a
// container without a GPU.  Never loaded by the walnuts_amd package.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "wn_fast.h"

using namespace wn;
using namespace wn::fast;

namespace {

struct TSim {  // what the four T waves hold for ONE chain
  int D, M, NLC;
  const Params* prm;
  int cl;
  std::vector<double> q, v, g, qb, vb, VL, zn;
  std::vector<double> qhist[3];  // position of tick t in qhist[t % LAG]
  int LAG = 2;
  std::vector<double> prop[2];
  std::vector<std::vector<double>> lq, lv, sq, sv;  // LDS-cached levels, memory slots
  double sums[4 + 2 * (MAX_M + 2)];
  int ntr = 0;  // trace rows written (the T wave's own count: wn_k_fast4.hip)

  void grad() {
    if (prm->tpar == nullptr) { for (int r = 0; r < D; ++r) g[r] = -q[r]; return; }
    if (dense) {
      for (int r = 0; r < D; ++r) {
        double a = 0.0;
        for (int c = 0; c < D; ++c) a += -prm->tpar[(size_t)r * D + c] * q[c];
        g[r] = a;
      }
    } else {
      for (int r = 0; r < D; ++r) g[r] = -(prm->tpar[r] * q[r]);
    }
  }
  bool dense;

  void dots(const std::vector<double>& qo, const std::vector<double>& vo, double& d1, double& d2) {
    double a1 = 0.0, a2 = 0.0;
    for (int e = 0; e < D; ++e) {
      const double dq = q[e] - qo[e];
      a1 += v[e] * dq;
      a2 += vo[e] * dq;
    }
    d1 = a1; d2 = a2;
  }

  void tick(uint32_t t, const Cmd& c) {
    const int NFAST = 2 + NLC;
    for (auto& s : sums) s = 0.0;
    if (c.op & OP_PICK) prop[(c.op & OP_IPPICK) ? 1 : 0] = qhist[t % LAG];  // the position of tick t - LAG
    if (c.op & OP_UNDO) ntr -= 1;  // the leaves in flight when the iteration ended are not integrator calls of the reference
    if (c.op & OP_UNDO2) ntr -= 1;
    if (c.op & OP_NEWITER) {
      if (c.op & OP_INIT) {
        for (int e = 0; e < D; ++e) q[e] = prm->q0[(size_t)cl * D + e];
        prop[0] = q;
      } else {
        q = prop[(c.op & OP_IPNEW) ? 1 : 0];
      }
      double* row = prm->samples + ((size_t)cl * (prm->numIter + 1) + c.n) * prm->dg;
      if (prm->generated == G_Q0_SUMSQ) {
        row[0] = q[0];
        double s = 0.0;
        for (int e = 0; e < D; ++e) s += q[e] * q[e];
        sums[Q_GEN] = s;
      } else {
        const int k = prm->generated == G_FIRSTK ? prm->gen_k : D;
        for (int e = 0; e < k; ++e) row[e] = q[e];
      }
      if (!(c.op & (OP_INIT | OP_LAST))) {
        double kp = 0.0;
        for (int e = 0; e < D; ++e) { v[e] = zn[e]; kp += v[e] * v[e]; }
        sums[Q_KEINIT] = kp;
        qb = q; vb = v;
      }
    }
    if (c.op & OP_SWAP) { q.swap(qb); v.swap(vb); }
    const double hs = (c.op & OP_STEP) ? c.hs : 0.0, hs2 = 0.5 * hs;
    for (int e = 0; e < D; ++e) { v[e] = v[e] + hs2 * g[e]; q[e] = q[e] + hs * v[e]; }
    qhist[t % LAG] = q;
    grad();
    double lp = 0.0, ke = 0.0;
    for (int e = 0; e < D; ++e) { v[e] = v[e] + hs2 * g[e]; lp += q[e] * g[e]; ke += v[e] * v[e]; }
    sums[Q_LP] = lp; sums[Q_KE] = ke;
    if (c.op & OP_STEP) {
      if (prm->trace && ntr >= 0 && ntr < prm->trace_cap) {
        double* tr = prm->trace + ((size_t)cl * prm->trace_cap + ntr) * (TRACE_NSCAL + 2 * D) + TRACE_NSCAL;
        for (int e = 0; e < D; ++e) { tr[e] = q[e]; tr[D + e] = v[e]; }
      }
      ntr += 1;
      const int i = (int)((c.op >> OP_I_SHIFT) & 31u), n = (int)c.n;
      const int nchk = nchecks(i, n);
      for (int j = 0; j < nchk; ++j) {
        if (!need_check(i, n, j)) continue;
        double d1, d2;
        if (j == 0) dots(qb, vb, d1, d2);
        else if (j == 1) dots(qhist[(t - 1) % LAG], VL, d1, d2);
        else if (j < NFAST) dots(lq[j - 2], lv[j - 2], d1, d2);
        else { const int sl = slot_of_leaf(n + 1 - (1 << j)); dots(sq[sl], sv[sl], d1, d2); }
        sums[Q_DOT0 + 2 * j] = d1; sums[Q_DOT0 + 2 * j + 1] = d2;
      }
      if (i >= 1 && (n & 1) == 0) {
        VL = v;
        if (i >= 2 && (n & 3) == 0) {
          for (int j = 2; j < NFAST; ++j)
            if (j <= i && (n & ((1 << j) - 1)) == 0) { lq[j - 2] = q; lv[j - 2] = v; }
          if (n == 0 ? (i >= NFAST) : (__builtin_ctz((unsigned)n) >= NFAST)) { const int sl = slot_of_leaf(n); sq[sl] = q; sv[sl] = v; }
        }
      }
    }
  }
};

struct HostIO {
  const Params* prm;
  int cl;
  const double* sums;
  int* ntrace;
  double* h0 = nullptr;  // (two-stream form) the iteration's initial Hamiltonian, retire side -> issue side
  void write_h0(double v) const { if (h0) *h0 = v; }
  double* hc = nullptr;  // (two-stream form, warm-up) the adapted step size H, retire side -> issue side
  void write_hc(double v) const { if (hc) *hc = v; }
  double read_hc() const { return *hc; }
  double* facs() const { return prm->facs ? prm->facs + (size_t)cl * 2 * (size_t)prm->warmupIter : nullptr; }
  bool facs_writer() const { return true; }
  void trace_step(int iter, int i, int xi, double h, double ham0, double lpn) {
    const int row = (*ntrace)++;
    if (prm->trace && row < prm->trace_cap) {
      double* r = prm->trace + ((size_t)cl * prm->trace_cap + row) * (TRACE_NSCAL + 2 * prm->D);
      r[0] = iter; r[1] = i; r[2] = xi; r[3] = h; r[4] = ham0; r[5] = lpn; r[6] = 0; r[7] = 0; r[8] = 0; r[9] = 0.0; r[10] = 1; r[11] = 0;
    }
  }
  double sum(int k) const { return sums[k]; }
  void dots(int j, double& d1, double& d2) const { d1 = sums[Q_DOT0 + 2 * j]; d2 = sums[Q_DOT0 + 2 * j + 1]; }
  uint32_t firemask(int i, int n, int xi) const { return fire_of(i, n, xi, *this); }
  // the uniform window of the device (wn_k_fast4.hip) always has the block: here it is computed on the spot
  bool uniforms(int, int iter, uint32_t blk, double& ua, double& ub) const {
    Philox4 r = philox4x32_10(blk, (uint32_t)iter, prm->chain0 + (uint32_t)cl, 0u, (uint32_t)prm->seed, (uint32_t)(prm->seed >> 32));
    ua = u53(r.w[0], r.w[1]); ub = u53(r.w[2], r.w[3]);
    return true;
  }
  double u_direct(int iter, uint32_t k) const {  // (the device reads its window of uniforms without the tag checks: IssueBookT::tick_fast)
    Philox4 r = philox4x32_10(k >> 1, (uint32_t)iter, prm->chain0 + (uint32_t)cl, 0u, (uint32_t)prm->seed, (uint32_t)(prm->seed >> 32));
    return (k & 1u) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
  }
  void write_gen(size_t row, double v) const { prm->samples[((size_t)cl * (prm->numIter + 1) + row) * prm->dg + 1] = v; }
  void write_diag(size_t row, const double* d) const {
    double* o = prm->diag + ((size_t)cl * prm->numIter + row) * NDIAG;
    for (int k = 0; k < NDIAG; ++k) o[k] = d[k];
  }
};

}  // namespace

// returns the number of ticks the slowest chain took (>= 0) or -1 when a chain exceeded its bound
template <int LAG>
static long run_lag(Params* prm, int target, int nlc, double* idle_frac) {
  const int D = prm->D;
  long worst = 0;
  double steps = 0, ticks = 0;
  constexpr int NSUM = 4 + 2 * (MAX_M + 2);
  for (int cl = 0; cl < prm->C; ++cl) {
    TSim T;
    T.D = D; T.M = prm->M; T.NLC = nlc; T.prm = prm; T.cl = cl; T.dense = target == T_DENSEGAUSS; T.LAG = LAG;
    for (auto* x : {&T.q, &T.v, &T.g, &T.qb, &T.vb, &T.VL, &T.zn, &T.qhist[0], &T.qhist[1], &T.qhist[2], &T.prop[0], &T.prop[1]}) x->assign(D, 0.0);
    T.lq.assign(nlc + 1, std::vector<double>(D, 0.0)); T.lv = T.lq;
    T.sq.assign(prm->M + 2, std::vector<double>(D, 0.0)); T.sv = T.sq;
    FastBookT<LAG> B;
    B.init(prm->chain0 + (uint32_t)cl, true, prm->numIter);
    // commands issued and not yet executed: the bookkeeper's tick t issues tick t+1; sums written and not yet retired: its tick t
    // retires tick t+1-LAG
    Cmd cur, nxt;
    cur.hs = 0.0; cur.n = 0; cur.op = OP_NEWITER | OP_INIT;
    double hist[3][NSUM] = {{0}};  // sums of tick t in hist[t % 3]
    double psums[NSUM] = {0};
    const double cap = 64.0 + 1.3 * (double)prm->numIter * ((double)(1u << prm->M) + 6.0 * prm->M + 16.0);
    uint32_t t = 0;
    int ntrace = 0;
    for (;; ++t) {
      if ((double)t > cap) return -1;
      // momentum normals: the B wave refills the buffer of a chain whose next iteration has none yet
      if (B.zwant != B.zready && B.zwant >= 1) {
        for (int e = 0; e < D; ++e) T.zn[e] = rng_normal(prm->seed, B.chain, (uint32_t)B.zwant, (uint32_t)e);
        B.zready = B.zwant;
      }
      if (t + 1 >= (uint32_t)LAG) memcpy(psums, hist[(t + 1 - LAG) % 3], sizeof(psums));  // the sums of tick t+1-LAG
      else memset(psums, 0, sizeof(psums));
      HostIO io{prm, cl, psums, &ntrace};
      nxt = B.tick(*prm, io, t + 1);  // retires tick t+1-LAG, issues tick t+1
      T.tick(t, cur);                 // tick t
      memcpy(hist[t % 3], T.sums, sizeof(psums));
      if (cur.op & OP_STEP) steps += 1;
      cur = nxt;
      if (B.finished()) break;
    }
    if (prm->trace_n) prm->trace_n[cl] = ntrace;
    ticks += t + 1;
    if ((long)t + 1 > worst) worst = t + 1;
  }
  if (idle_frac) *idle_frac = 1.0 - steps / ticks;
  return worst;
}
// the bookkeeper as two instruction streams (wn_fast.h RetireBook / IssueBook, wn_k_fast4.hip): per tick the retire side retires tick
// t-2 from the command the T wave executed then and posts its message, the issue side applies it and issues tick t+1
template <bool WARM>
static long run_split(Params* prm, int target, int nlc, double* idle_frac) {
  constexpr int LAG = 3;
  const int D = prm->D;
  long worst = 0;
  double steps = 0, ticks = 0;
  constexpr int NSUM = 4 + 2 * (MAX_M + 2);
  for (int cl = 0; cl < prm->C; ++cl) {
    TSim T;
    T.D = D; T.M = prm->M; T.NLC = nlc; T.prm = prm; T.cl = cl; T.dense = target == T_DENSEGAUSS; T.LAG = LAG;
    for (auto* x : {&T.q, &T.v, &T.g, &T.qb, &T.vb, &T.VL, &T.zn, &T.qhist[0], &T.qhist[1], &T.qhist[2], &T.prop[0], &T.prop[1]}) x->assign(D, 0.0);
    T.lq.assign(nlc + 1, std::vector<double>(D, 0.0)); T.lv = T.lq;
    T.sq.assign(prm->M + 2, std::vector<double>(D, 0.0)); T.sv = T.sq;
    RetireBookT<WARM> R;  // (WARM: the instantiation of the kernels that run warm-up iterations, wn_k_fast4.hip -DWN4_WARM_TU)
    IssueBookT<WARM> I;
    R.init(prm->chain0 + (uint32_t)cl, true, prm->numIter);
    I.init(prm->chain0 + (uint32_t)cl, true, prm->numIter);
    R.init_warm(*prm);
    I.init_warm(*prm);
    Cmd ring[4];  // the command of tick t in ring[t % 4]
    for (auto& c : ring) { c.hs = 0.0; c.n = 0; c.op = 0; }
    ring[0].op = OP_NEWITER | OP_INIT;
    double hist[3][NSUM] = {{0}};  // sums of tick t in hist[t % 3]
    double psums[NSUM] = {0};
    const double cap = 64.0 + 1.3 * (double)prm->numIter * ((double)(1u << prm->M) + 6.0 * prm->M + 16.0);
    uint32_t t = 0;
    int ntrace = 0;
    double h0_shared = 0.0, hc_shared = prm->H;
    for (;; ++t) {
      if ((double)t > cap) return -1;
      if (I.zwant != I.zready && I.zwant >= 1) {  // momentum normals (the I wave)
        for (int e = 0; e < D; ++e) T.zn[e] = rng_normal(prm->seed, I.chain, (uint32_t)I.zwant, (uint32_t)e);
        I.zready = I.zwant;
      }
      Cmd old;
      old.hs = 0.0; old.n = 0; old.op = 0;
      if (t + 1 >= (uint32_t)LAG) { memcpy(psums, hist[(t + 1 - LAG) % 3], sizeof(psums)); old = ring[(t + 1 - LAG) % 4]; }
      else memset(psums, 0, sizeof(psums));
      HostIO io{prm, cl, psums, &ntrace, &h0_shared, &hc_shared};
      const RMsg m = R.tick(*prm, io, old);           // retires tick t-2
      bool early = false;  // the first leaf of a doubling (tick t-1), certified by the issue side from the T wave's sums
      if (I.wants_early(t + 1, m)) early = IssueBook::early_certain(*prm, hist[(t - 1) % 3][Q_LP], hist[(t - 1) % 3][Q_KE], h0_shared);
      // (the straight-line form of the ordinary tick whenever the chain qualifies, as the kernel does when all its chains do)
      const Cmd nxt = I.fast_ok(t + 1, m) ? I.tick_fast(*prm, io, t + 1, m) : I.tick(*prm, io, t + 1, m, early);  // issues tick t+1
      T.tick(t, ring[t % 4]);                         // tick t
      memcpy(hist[t % 3], T.sums, sizeof(psums));
      if (ring[t % 4].op & OP_STEP) steps += 1;
      ring[(t + 1) % 4] = nxt;
      if (I.finished()) break;
    }
    if (prm->trace_n) prm->trace_n[cl] = ntrace;
    ticks += t + 1;
    if ((long)t + 1 > worst) worst = t + 1;
  }
  if (idle_frac) *idle_frac = 1.0 - steps / ticks;
  return worst;
}
// lag: 2, 3 = FastBookT<lag>; 4 = the two-stream form of lag 3
extern "C" long fastsim_run2(Params* prm, int target, int nlc, double* idle_frac, int lag) {
  if (lag == 4) return prm->warmupIter > 0 ? run_split<true>(prm, target, nlc, idle_frac) : run_split<false>(prm, target, nlc, idle_frac);
  return lag == 3 ? run_lag<3>(prm, target, nlc, idle_frac) : run_lag<2>(prm, target, nlc, idle_frac);
}
extern "C" long fastsim_run(Params* prm, int target, int nlc, double* idle_frac) { return fastsim_run2(prm, target, nlc, idle_frac, 2); }
extern "C" int fastsim_params_size() { return (int)sizeof(Params); }
