// wn_k_orbit2.hip -- the trajectory-first kernel's wide mode (wn_k_orbit.hip: a chain whose vectors fill the wave, StockWatson D = 756 in
// 12 registers per lane) on TWO wavefronts per chain, fixedLeapFrog runs (mainSW.py:56-62; BASELINE config 5's fixedLeapFrog run).
//
// Config 5 is 256 chains on 256 CUs: ONE wave per CU on a machine with four SIMDs per CU, and that wave's time per leaf is
// 4.3 k cycles of leapfrog step + gradient (one instruction per ~4.8 cycles: a lone wave) plus 2.4-3 k cycles of NUTS bookkeeping in every
// form tried so far -- replicated scalar code (one-stream engine), lane-parallel in batches of 16 (wn_k_orbit.hip's wide mode: the
// batch form saves the scalar code but pays its U-turn dots with loads from L2, ~900 cycles of latency each for a lone wave).  The
// bookkeeping does not feed the trajectory: for fixedLeapFrog the next leaf needs the previous leaf's state and a step size, nothing the
// weights, the multinomial draws or the U-turn checks produce -- until the orbit stops.  So a SECOND wave on another SIMD does it:
//
//   T wave   one command -> one RUN: the leaves n0 .. 2**i - 1 of doubling i, back to back (step sizes from the wave's own lane-parallel
//            draws, under the assumption "every leaf takes its multinomial draw like the one before": WALNUTS.py:395,426);
//            each leaf's (q, v) to a ring of 32 entries in the chain's workspace, its (H, lp) to a record in LDS, the count of published
//            leaves to an LDS word -- one leaf late, so that the release fence in front of it finds the stores long complete.
//   R wave   retires whatever has been published, in sub-batches inside a block of 16, lane k = leaf k, exactly as wn_k_orbit.hip's R
//            phase does (serial prefix sums in the reference's order, U-turn checks as wave-wide dots with the next check's loads in
//            flight, ballots for the first event); it owns every scalar of WALNUTS.py:189-695, the proposals, the slots, the draws it
//            verifies the T wave's assumption with.  On an event inside a doubling (sub-tree U-turn, non-finite energy, a draw that was
//            not taken as assumed) it raises the abort word and sends the next command: NEWITER (proposal, lp), GO (next doubling: which
//            end is active), REDO (restart behind leaf n with the corrected draw index; the state is in the ring), EXIT.
// Leaves the T wave ran ahead of an event are never counted: draws, nevalF / nevalB and every diagnostic are the reference's.  The
// arithmetic of a micro step, of every sum and of every dot is wn_k_orbit.hip's wide mode = the one-stream engine's (tests/test_gpu_orbit.py).
// Every wait is bounded: a wave that gives up sets the DEAD word (and the engine's abort flag), both waves leave.
#include "wn_launch.h"
#include "wn_policy_wave.h"
#include "wn_orbit.h"

namespace wn {

namespace orbit2 {
// the chain's workspace, in vectors of NE * 64 doubles: ring of 32 leaves (two blocks of 16)
enum { V_RQ = 0, V_RV = 32, V_COLDQ = 64, V_COLDV = 65, V_COLDG = 66, V_PROP = 67, V_SLOT = 69 };
__host__ __device__ constexpr size_t wide_doubles(int NE, int M) { return (size_t)(V_SLOT + 2 * (M + 1)) * NE * 64; }
// LDS, in doubles: records of the ring's leaves [entry][Hn, lp, If | Ib << 8 | c << 16 | (nEvalF | nEvalB << 16) << 32, -], then the words of
// the hand-off
enum { L_REC = 0, L_PROD = 128, L_CONS = 129, L_ABORT = 130, L_DEAD = 131, L_ACK = 132, L_KP = 133, L_CMD = 136, L_PART = 144, L_TOTAL = L_PART + 32 * 64 * 2 };
// L_PART (fixedLeapFrog): [entry][lane][lp share, KE share] -- the T wave leaves the two wave sums of a leaf to the R wave (the same
// wn_wavesum on the same 64 shares: the same bits); nothing the T wave does next needs the energy
enum { K_NEWITER = 1, K_GO = 2, K_REDO = 3, K_EXIT = 4 };
typedef __attribute__((address_space(3))) double lds2_d;
typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64;
// the R wave retires a sub-batch when this many leaves are there (or the block / the doubling / the T wave's run has ended): a sub-batch
// has a fixed cost of ~2.5 k cycles (draws, scans, ballots) that a leaf at a time would make the R wave the slower of the two
#ifndef WN_O2_BATCH
#define WN_O2_BATCH 4
#endif
// diagnostic builds (-DWN_STAMPS): cycles per segment, summed per wave (no vmcnt wait: stores stay in flight as in the plain build)
#if defined(WN_STAMPS)
#define WN_O2STAMP(k)                                                                                   \
  do {                                                                                                  \
    unsigned long long _t;                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                          \
    o2st_[k] += _t - o2last_;                                                                           \
    o2last_ = _t;                                                                                       \
  } while (0)
#else
#define WN_O2STAMP(k) do {} while (0)
#endif
// Every inter-wave wait is bounded by WALL-CLOCK time, not by an iteration count (2^24 spins were 1-2 s at today's clocks, and
// a slow but healthy peer -- clock throttling, a debugger, a maxC = 12 macro step of tens of milliseconds -- would have become a hard abort):
// s_memrealtime counts at a constant 100 MHz; the clock is read once per 1024 spins.
#ifndef WN_O2_DEADLINE_TICKS
#define WN_O2_DEADLINE_TICKS 3000000000ull  // 30 s
#endif
// (the clock is first read after 1024 spins: a wait that ends at once -- the common case -- never touches it)
__device__ __forceinline__ bool wn_o2_expired(unsigned& spins, unsigned long long& t0, unsigned mult) {
  if ((++spins & 0x3FFu) != 0u) return false;
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (t0 == 0ull) { t0 = now | 1ull; return false; }
  return now - t0 > (unsigned long long)mult * WN_O2_DEADLINE_TICKS;
}
#define WN_O2_EXPIRED(spins, t0, mult) wn_o2_expired(spins, t0, mult)
#ifndef WN_O2_SPINS
#define WN_O2_SPINS (1u << 24)
#endif
}  // namespace orbit2

template <class Tgt, int NE, int INTEG>
__global__ __launch_bounds__(INTEG == I_FIXED ? 128 : 256) void wn_orbit2_kernel(Params prm, int* kabort) {
  using namespace orbit;
  using namespace orbit2;
  static_assert(NE % 2 == 0, "stored vectors are register pairs");
  constexpr int DP = NE * 64;
  // adaptive integrators: two HELPER waves (roles 2, 3) run the refinement levels c+1, c+2 of a search while the T wave runs level c (see
  // the T wave's macro step); their hand-off: a job word, the start state of the search (two buffers), each helper's end state and verdict
  constexpr bool ADAPT = INTEG != I_FIXED;
  constexpr int NH = 2;
  constexpr int NWAVES = ADAPT ? 2 + NH : 2;
  constexpr int L_SURE = L_CMD + 7;
  constexpr int L_S = L_TOTAL, L_E = L_S + 2 * 3 * DP, L_JOB = L_E + NH * 3 * DP, L_JCANCEL = L_JOB + 1, L_JPAR = L_JOB + 2, L_HDONE = L_JPAR + 4,
                L_HRES = L_HDONE + NH, LDS_N = ADAPT ? L_HRES + 2 * NH : L_TOTAL;
  static_assert(LDS_N * 8 <= 160 * 1024, "LDS budget (160 KiB per CU)");
  __shared__ double wn_o2_lds[LDS_N];
  lds2_d* const L = (lds2_d*)wn_o2_lds;
  lds_u64* const LU = (lds_u64*)wn_o2_lds;
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;  // 0: T wave, 1: R wave, 2 ..: helper waves (adaptive integrators)
  const int k16 = lane & 15;
  const int row = lane >> 4;
  const int cl = blockIdx.x;
  const uint32_t cg = prm.chain0 + (uint32_t)cl;
  const uint32_t seedlo = (uint32_t)prm.seed, seedhi = (uint32_t)(prm.seed >> 32);
  const int D = prm.D, M = prm.M, minC = prm.minC, maxC = prm.maxC;
  const double dlt = prm.delta;
  constexpr uint32_t r_ = INTEG == I_ADAPT_R2P ? 1u : 0u;  // adaptLeapFrogR2P draws one uniform per macro step (adaptiveIntegrators.py:392)
  const bool inrange = cl < prm.C;
  const int clc = inrange ? cl : 0;
  const double Hc = prm.H;
  Tgt tgt;
  tgt.init(prm, lane, nullptr);
  for (int k = threadIdx.x; k < LDS_N; k += 64 * NWAVES) L[k] = 0.0;
  __syncthreads();

#define WN_OV(e) _Pragma("unroll") for (int e = 0; e < NE; ++e)
  auto reduce2 = [&](double& a, double& b) { a = wn_wavesum(a); b = wn_wavesum(b); };
  auto reduce1 = [&](double a) -> double { return wn_wavesum(a); };
  auto gidx = [&](int e) -> int {
    if constexpr (tgt_mapped<Tgt>::value) return tgt.map_gidx(e, lane);
    else return e * 64 + lane < D ? e * 64 + lane : -1;
  };
  // a stored vector is [register pair][lane][2]: one 16-byte access per lane and pair of registers
  typedef double d2_t __attribute__((ext_vector_type(2)));
  d2_t* const Wk = reinterpret_cast<d2_t*>(prm.work + (size_t)clc * orbit2::wide_doubles(NE, M)) + lane;
  auto vst = [&](int off, const double* x) {
#pragma unroll
    for (int j = 0; j < NE / 2; ++j) { d2_t t; t[0] = x[2 * j]; t[1] = x[2 * j + 1]; Wk[(size_t)off * (DP / 2) + j * 64] = t; }
  };
  auto vld = [&](int off, double* x) {
#pragma unroll
    for (int j = 0; j < NE / 2; ++j) { const d2_t t = Wk[(size_t)off * (DP / 2) + j * 64]; x[2 * j] = t[0]; x[2 * j + 1] = t[1]; }
  };
  // ---- the hand-off words ----
  auto dead = [&]() -> bool { return LU[L_DEAD] != 0ull; };
  auto give_up = [&]() { LU[L_DEAD] = 1ull; if (lane == 0 && kabort) *kabort = 1; };
  auto release = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); };
  auto acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };

  // ---- the draws of leaves k0 .. 15 of a block, lane-parallel (wn_k_orbit.hip): lane k (of every row) gets leaf k's step size and its
  // multinomial uniform.  `ku` = the stream index of the first draw of leaf k0, `khc` = the index of the second step size of a pair when
  // k0 is the pair's second leaf, `pred` = "every leaf takes its multinomial draw" ----
  auto draws = [&](int i, int iter, int k0, uint32_t ku, uint32_t khc, bool pred, double& hV, double& urV, double& upV) {
    const uint32_t p_ = (pred && i != 0) ? 1u : 0u;
    const uint32_t ce = (i != 0 ? 2u : 1u) + r_ + p_, co = r_ + p_;
    const uint32_t klow = (k0 & 1) ? khc : ku;
    const uint32_t base = (uint32_t)uni_i((int)(klow >> 1));
    Philox4 r = philox4x32_10(base + (uint32_t)(lane >> 1), (uint32_t)iter, cg, 0u, seedlo, seedhi);
    const double ru0 = (lane & 1) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
    const uint32_t j = (uint32_t)k16;
    const uint32_t nE = ((j + 1u) >> 1) - (((uint32_t)k0 + 1u) >> 1), nO = (j >> 1) - ((uint32_t)k0 >> 1);  // even / odd leaves in [k0, j)
    const uint32_t kuS = ku + ce * nE + co * nO;  // the leaf's first draw
    const bool evn = (i == 0) || !(j & 1u);
    uint32_t ih = evn ? kuS : kuS - ce + 1u;  // the second step size of the pair was drawn with the first
    ih = (j == (uint32_t)k0 && (k0 & 1)) ? khc : ih;
    const uint32_t ir = evn ? kuS + (i != 0 ? 2u : 1u) : kuS;
    const uint32_t ipk = ir + r_;
    const uint32_t off = 2u * base;
    const double uh = __shfl(ru0, (int)((ih - off) & 63u)), up = __shfl(ru0, (int)((ipk - off) & 63u));
    urV = r_ ? __shfl(ru0, (int)((ir - off) & 63u)) : 0.0;
    const double low = Hc * (1 - prm.s), high = Hc * (1 + prm.s);
    hV = low + (high - low) * uh;  // numpy legacy uniform(low, high) = low + (high-low)*U  (WALNUTS.py:298,395)
    upV = up;
  };
  // the stream index behind the leaves [k0, k1) of a block when every one of them does what `pred` says
  auto ku_behind = [&](int i, int k0, int k1, uint32_t ku, bool pred) -> uint32_t {
    if (i == 0) return ku + 1u + r_;
    const uint32_t p_ = pred ? 1u : 0u;
    const uint32_t nE = (((uint32_t)k1 + 1u) >> 1) - (((uint32_t)k0 + 1u) >> 1), nO = ((uint32_t)k1 >> 1) - ((uint32_t)k0 >> 1);
    return ku + (2u + r_ + p_) * nE + (r_ + p_) * nO;
  };

  // one micro pass of 2**cc leapfrog steps (adaptiveIntegrators.py:72-94) on (q, v, g) in place; 1: accepted, 0: not, -1: `cancel()` said stop
  auto pass_core = [&](double* q, double* v, double* g, double HamStart, double hcur, int cc, double& lpC, double& HC, auto cancel) -> int {
    const double hh = ldexp(hcur, -cc);  // h / 2**c (exact scaling either way)
    bool allfin = isfinite(HamStart);
    const double hh2 = 0.5 * hh, hq = hh;
    int st = 1 << cc;
    lpC = 0.0; HC = HamStart;
#pragma unroll 1
    do {  // one branch per micro step
      if (cancel()) return -1;
      WN_OV(e) { v[e] = v[e] + hh2 * g[e]; q[e] = q[e] + hq * v[e]; }
      double lpp = tgt.grad(q, g, lane);
      double ke = 0.0;
      WN_OV(e) { v[e] = v[e] + hh2 * g[e]; ke += v[e] * v[e]; }
      reduce2(lpp, ke);
      lpp = Tgt::lp_finish(lpp);
      const double Hi = -lpp + 0.5 * ke;  // Hams[i] = -fnew + 0.5*sum(vv*vv)
      lpC = lpp; HC = Hi;
      allfin = allfin && isfinite(Hi);
    } while (--st);
    return uni_b(allfin && (fabs(HamStart - HC) < dlt)) ? 1 : 0;  // all(isfinite(Hams)) and abs(Hams[0]-Hams[-1]) < delta  (:87-94)
  };
  // a state (q, v, g) in LDS: [vector][register pair][lane][2]
  typedef __attribute__((address_space(3))) d2_t lds_d2;
  auto lst = [&](int off, const double* q, const double* v, const double* g) {
    lds_d2* b = reinterpret_cast<lds_d2*>(L + off) + lane;
#pragma unroll
    for (int j = 0; j < NE / 2; ++j) {
      d2_t t; t[0] = q[2 * j]; t[1] = q[2 * j + 1]; b[j * 64] = t;
      t[0] = v[2 * j]; t[1] = v[2 * j + 1]; b[(NE / 2 + j) * 64] = t;
      t[0] = g[2 * j]; t[1] = g[2 * j + 1]; b[(NE + j) * 64] = t;
    }
  };
  auto lld = [&](int off, double* q, double* v, double* g) {
    const lds_d2* b = reinterpret_cast<const lds_d2*>(L + off) + lane;
#pragma unroll
    for (int j = 0; j < NE / 2; ++j) {
      d2_t t = b[j * 64]; q[2 * j] = t[0]; q[2 * j + 1] = t[1];
      t = b[(NE / 2 + j) * 64]; v[2 * j] = t[0]; v[2 * j + 1] = t[1];
      t = b[(NE + j) * 64]; g[2 * j] = t[0]; g[2 * j + 1] = t[1];
    }
  };
  unsigned long long o2st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o2last_ = __builtin_readcyclecounter();
  (void)o2st_; (void)o2last_;
  auto stamps_out = [&]() {
#if defined(WN_STAMPS)
    if (prm.stamps && lane == 0) {
      unsigned long long* o = prm.stamps + ((size_t)blockIdx.x * 2 + role) * 8;
      for (int k = 0; k < 8; ++k) o[k] = o2st_[k];
    }
#endif
  };
  if constexpr (ADAPT) if (role >= 2) {
    // =====================================================================================================================
    // helper wave hj: level base + 1 + hj of every search job the T wave posts
    // =====================================================================================================================
    const int hj = role - 2;
    uint32_t last = 0;
    for (;;) {
      unsigned long long w = 0;
      unsigned spins = 0; unsigned long long spin_t0 = 0ull;
      bool out = false;
      for (;;) {
        w = LU[L_JOB];
        const uint32_t js = (uint32_t)(w >> 32);
        if (js == 0xFFFFFFFFu || dead()) { out = true; break; }
        if (js != last) break;
        if (WN_O2_EXPIRED(spins, spin_t0, 8)) { out = true; break; }  // (no job for seconds: the T wave is gone)
        __builtin_amdgcn_s_sleep(1);
      }
      if (out) break;
      acquire();
      const uint32_t js = (uint32_t)(w >> 32);
      last = js;
      const int base = (int)((uint32_t)w & 0xFFu), jmax = (int)(((uint32_t)w >> 8) & 0xFFu);
      const bool need_state = (((uint32_t)w >> 16) & 1u) != 0u;
      const int level = base + 1 + hj;
      if (level > jmax) { LU[L_HDONE + hj] = ((unsigned long long)js << 32) | 8ull; continue; }  // not asked for
      const int buf = (int)(js & 1u);
      double q[NE], v[NE], g[NE], lpC, HC;
      lld(L_S + buf * 3 * DP, q, v, g);
      const double Ham = L[L_JPAR + 2 * buf], hcur = L[L_JPAR + 2 * buf + 1];
      const int r = pass_core(q, v, g, Ham, hcur, level, lpC, HC, [&]() -> bool { return (uint32_t)LU[L_JCANCEL] == js || (uint32_t)(LU[L_JOB] >> 32) != js; });
      if (r >= 0) {
        if (need_state) lst(L_E + hj * 3 * DP, q, v, g);
        L[L_HRES + 2 * hj] = lpC; L[L_HRES + 2 * hj + 1] = HC;
      }
      release();
      LU[L_HDONE + hj] = ((unsigned long long)js << 32) | (r > 0 ? 1ull : 0ull) | 2ull | (r < 0 ? 4ull : 0ull);
    }
    return;
  }
  if (role == 0) {
    // =====================================================================================================================
    // T wave: the trajectory
    // =====================================================================================================================
    // active end of the orbit: q, g, and vi = xi * v (the integrator's orientation: xi is the same for every leaf of a run)
    double qa[NE], vi[NE], ga[NE];
    WN_OV(e) { qa[e] = 0.0; vi[e] = 0.0; ga[e] = 0.0; }
    int xi_run = 1;  // the orientation vi is in
    double Ha = 0.0;  // energy at the active end (the adaptive integrators' Ham0)
    uint32_t seq = 0, jseq = 0;
    int last_level = 0;
    (void)jseq; (void)last_level;
    for (;;) {
      ++seq;
      WN_O2STAMP(4);
      // ---- wait for command `seq` ----
      unsigned long long w0 = 0;
      {
        unsigned spins = 0; unsigned long long spin_t0 = 0ull;
        for (;;) {
          w0 = LU[L_CMD];
          if ((uint32_t)(w0 >> 32) == seq) break;
          if (dead()) break;
          if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); break; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (dead()) break;
      acquire();
      WN_O2STAMP(2);
      const int kind = (int)(uint32_t)w0;
      if (kind == K_EXIT) break;
      const unsigned long long w1 = LU[L_CMD + 1], w2 = LU[L_CMD + 2], w3 = LU[L_CMD + 3], w4 = LU[L_CMD + 4];
      const int ip = (int)(uint32_t)w1, iter = (int)(uint32_t)(w1 >> 32);
      int n = uni_i((int)(uint32_t)w2);
      const int i = uni_i((int)(uint32_t)(w2 >> 32));
      uint32_t ku = (uint32_t)w3, khc = (uint32_t)(w3 >> 32);
      const bool pred = ((uint32_t)w4 & 1u) != 0u, swap = ((uint32_t)w4 & 2u) != 0u;
      const int xi = (int)(uint32_t)(w4 >> 32);
      const double lpP_c = L[L_CMD + 5];
      Ha = L[L_CMD + 6];
      if (kind == K_NEWITER) {
        vld(V_PROP + ip, qa);
        double lpp = tgt.grad(qa, ga, lane);  // f0, grad0 = lpFun(qc)  (WALNUTS.py:249); the R wave holds its value
        (void)lpp;
        // :236  v = normal(size=d)  (rolled: one copy of Philox + Box-Muller in the code; the register is picked with selects)
        double va[NE];
        WN_OV(e) va[e] = 0.0;
#pragma unroll 1
        for (int e1 = 0; e1 < NE; ++e1) {
          int gi = -1;
          WN_OV(e) gi = e == e1 ? gidx(e) : gi;
          const double z = gi >= 0 ? rng_normal(prm.seed, cg, (uint32_t)iter, (uint32_t)gi) : 0.0;
          WN_OV(e) va[e] = e == e1 ? z : va[e];
        }
        vst(V_COLDQ, qa); vst(V_COLDV, va); vst(V_COLDG, ga);
        double kp = 0.0;
        WN_OV(e) kp += va[e] * va[e];
        WN_OV(e) vi[e] = (double)xi * va[e];
        xi_run = xi;
        double gp = 0.0;
        reduce2(kp, gp);
        Ha = -lpP_c + 0.5 * kp;  // Hs[I0] = -f0 + 0.5*sum(v**2)  (:256): the R wave forms the same expression from the same two numbers
        L[L_KP] = kp;
        release();
        LU[L_ACK] = (unsigned long long)seq;
      } else if (kind == K_GO) {
        if (swap) {  // the other end of the orbit becomes the active one (:283-294); stored velocities are in forward-time orientation
          double tq[NE], tv[NE], tg[NE], va[NE];
          vld(V_COLDQ, tq); vld(V_COLDV, tv); vld(V_COLDG, tg);
          WN_OV(e) va[e] = (double)xi_run * vi[e];
          vst(V_COLDQ, qa); vst(V_COLDV, va); vst(V_COLDG, ga);
          WN_OV(e) { qa[e] = tq[e]; vi[e] = (double)xi * tv[e]; ga[e] = tg[e]; }
          xi_run = xi;
        }
      } else {  // K_REDO: restart behind leaf n - 1, whose state is in the ring
        vld(V_RQ + ((n - 1) & 31), qa); vld(V_RV + ((n - 1) & 31), vi);  // (the ring holds xi * v; a restart stays inside its doubling)
        xi_run = xi;
        double lpp = tgt.grad(qa, ga, lane);
        (void)lpp;
      }
      // ---- the run: leaves n .. nleaf-1 of doubling i ----
      const int nleaf = 1 << i;
      bool stop = false;
      while (n < nleaf && !stop) {
        const int nb0 = n & ~15, k0 = n & 15;
        const int k1 = nleaf - nb0 < 16 ? nleaf - nb0 : 16;
        WN_O2STAMP(4);
        if (k0 == 0 && n >= 32) {  // the ring entries of block (n >> 4) - 2 are about to be overwritten: the R wave must be through with it
          release();
          LU[L_PROD] = ((unsigned long long)seq << 32) | (unsigned)n;
          unsigned spins = 0; unsigned long long spin_t0 = 0ull;
          while ((int)(uint32_t)LU[L_CONS] < n - 16) {
            if ((uint32_t)LU[L_ABORT] == seq || dead()) { stop = true; break; }
            if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
          }
          if (stop) break;
        }
        WN_O2STAMP(3);
        double hV, urV, upV;
        draws(i, iter, k0, ku, khc, pred, hV, urV, upV);
        (void)upV; (void)urV;
        WN_O2STAMP(5);
        if constexpr (INTEG != I_FIXED) {
          // ---- adaptLeapFrogD / adaptLeapFrogR2P (adaptiveIntegrators.py:65-137 / 361-475).  A macro step is a SEARCH: passes of
          // 2**c micro steps at c = minC, minC+1, .. from the same start state until one is accepted (here: 67 % of the macro steps at
          // c = 3, 17 % at 4, 6 % at 5, ..), then the same search backwards from the result over c = minC .. If-1.  The passes of a
          // search do not depend on each other, and this CU has idle SIMDs: the T wave posts the start state as a JOB, runs level c
          // itself while the helper waves run c+1 and c+2, and takes the first accepted level in the reference's order -- the levels
          // the reference would have tried are the ones counted (nEvalF / nEvalB), their arithmetic is the same whoever runs them.
          // A leaf is published at once (the fence waits for its stores: ~1 k cycles of ~100 k).  While the R wave has not yet seen a
          // positive weight in this doubling (the first leaves often have If != Ib: weight zero, no multinomial draw, WALNUTS.py:426),
          // the T wave waits for its verdict on leaf n-1 before it starts leaf n (~3 k cycles): the stream index of leaf n's draws
          // depends on it, and a macro step run on the wrong step size is 100 k cycles thrown away.
          bool aborted = false;
          double hcur = 0.0;
          auto never = [&]() -> bool { return false; };
          auto post_job = [&](const double* q, const double* v, const double* g, double Ham, int base, int jmax, bool need_state) -> uint32_t {
            ++jseq;
            const int buf = (int)(jseq & 1u);
            lst(L_S + buf * 3 * DP, q, v, g);
            L[L_JPAR + 2 * buf] = Ham; L[L_JPAR + 2 * buf + 1] = hcur;
            release();
            LU[L_JOB] = ((unsigned long long)jseq << 32) | (need_state ? 0x10000u : 0u) | ((uint32_t)jmax << 8) | (uint32_t)base;
            return jseq;
          };
          auto wait_helper = [&](int hj, uint32_t js) -> uint32_t {  // the helper's verdict on job js: 1 accepted, 2 done, 4 abandoned, 8 not asked
            unsigned spins = 0; unsigned long long spin_t0 = 0ull;
            unsigned long long w;
            for (;;) {
              w = LU[L_HDONE + hj];
              if ((uint32_t)(w >> 32) == js) break;
              if (dead()) { aborted = true; return 4u; }
              if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); aborted = true; return 4u; }
              __builtin_amdgcn_s_sleep(1);
            }
            acquire();
            return (uint32_t)w;
          };
          // levels lo .. hi of a search from (qs, vs, gs), in the reference's order: the first accepted level (or -1); `last` = the last level
          // computed; its end state in (q, v, g) when want_state
          auto search = [&](const double* qs, const double* vs, const double* gs, double Ham, int lo, int hi, bool want_state, bool also_next, double* q,
                            double* v, double* g, double& lpC, double& HC, int& next_hj) -> int {
            next_hj = -1;  // (also_next: the helper whose end-state buffer holds the pass at the accepted level + 1, if one ran it)
            int base = lo;
#pragma unroll 1
            for (;;) {
              const int top = base + NH < hi ? base + NH : hi;
              const int jtop = (also_next && top - base < NH) ? top + 1 : top;  // (adaptLeapFrogR2P: the level behind the accepted one rides along when a helper is free)
              const uint32_t js = jtop > base ? post_job(qs, vs, gs, Ham, base, jtop, want_state) : 0u;
              WN_OV(e) { q[e] = qs[e]; v[e] = vs[e]; g[e] = gs[e]; }
              if ((uint32_t)LU[L_ABORT] == seq) { aborted = true; return -1; }
              int r = pass_core(q, v, g, Ham, hcur, base, lpC, HC, never);
              int lvl = base, found = r > 0 ? base : -1;
#pragma unroll 1
              for (int hj = 0; found < 0 && base + 1 + hj <= top; ++hj) {
                const uint32_t fl = wait_helper(hj, js);
                if (aborted) return -1;
                lvl = base + 1 + hj;
                if ((fl & 1u) || lvl == hi) {  // accepted -- or the last level of the search: its state is used silently (Q5)
                  if (want_state) lld(L_E + hj * 3 * DP, q, v, g);
                  lpC = L[L_HRES + 2 * hj]; HC = L[L_HRES + 2 * hj + 1];
                  if (fl & 1u) found = lvl;
                }
              }
              if (found >= 0 && also_next) {  // the pass at found + 1 from the same start state (:400-424): a helper may have it
                const int hn = found + 1 - base - 1;
                if (hn >= 0 && hn < NH && found + 1 <= jtop) {
                  const uint32_t fl = wait_helper(hn, js);
                  if (aborted) return -1;
                  if ((fl & 2u) && !(fl & 4u)) next_hj = hn;
                }
              }
              if (js) LU[L_JCANCEL] = (unsigned long long)js;  // whoever is still running a level of this job may stop
              if (found >= 0 || top >= hi) { last_level = found >= 0 ? found : hi; return found; }
              base = top + 1;
            }
          };
#pragma unroll 1
          for (int k = k0; k < k1; ++k) {
            if ((uint32_t)LU[L_SURE] != seq) {  // the R wave's verdict on the leaves in front of this one
              unsigned spins = 0; unsigned long long spin_t0 = 0ull;
              while ((int)(uint32_t)LU[L_CONS] < nb0 + k && (uint32_t)LU[L_SURE] != seq) {
                if ((uint32_t)LU[L_ABORT] == seq || dead()) { aborted = true; break; }
                if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); aborted = true; break; }
                __builtin_amdgcn_s_sleep(1);
              }
              if (aborted || (uint32_t)LU[L_ABORT] == seq) { stop = true; break; }
            }
            hcur = wn_readlane(hV, k);
            const double Ham0 = Ha;  // Hs[] of the neighbouring stored index = energy at the active end (WALNUTS.py:302,400,443,487,530)
            int nEvF = 0, nEvB = 0, If, Ib, cret;
            double lpn, Hn;
            double qw[NE], vw[NE], gw[NE], lpC = 0.0, HC = 0.0;
            int next_hj = -1;  // adaptLeapFrogR2P: the helper that has the pass at If + 1
            // adaptLeapFrogR2P :392: with probability R2Pprob0 keep c = If, else re-integrate at c = If+1
            const bool keep = INTEG != I_ADAPT_R2P || uni_b(wn_readlane(urV, k) < prm.R2Pprob0);
            // forward search (:69-94 / 366-389)
            const int fnd = search(qa, vi, ga, Ham0, minC, maxC, true, !keep, qw, vw, gw, lpC, HC, next_hj);
            if (aborted) { stop = true; break; }
            If = fnd >= 0 ? fnd : maxC;  // Q5: the last computed state is used silently
            nEvF = (2 << If) - (1 << minC);  // the levels minC .. If the reference tried: sum of 2**c
            int cSim = If, maxTry = If - 1;
            Ib = If;
            lpn = lpC; Hn = HC;
            if (!keep) {  // :400-424, 434-437
              if (next_hj >= 0) {  // (a cancelled helper writes nothing: its buffer holds that pass until its next job is through)
                lld(L_E + next_hj * 3 * DP, qw, vw, gw);
                lpC = L[L_HRES + 2 * next_hj]; HC = L[L_HRES + 2 * next_hj + 1];
              } else {
                WN_OV(e) { qw[e] = qa[e]; vw[e] = vi[e]; gw[e] = ga[e]; }
                pass_core(qw, vw, gw, Ham0, hcur, If + 1, lpC, HC, never);
              }
              nEvF += 2 << If;
              cSim = If + 1; maxTry = maxC; Ib = maxC;
              lpn = lpC; Hn = HC;
            }
            if (maxTry >= minC) {  // backward search from (qOut, -vOut, gOut)  (:107-132 / 440-464): only the verdicts count
              double vneg[NE], qb[NE], vb[NE], gb[NE], lpB, HB;
              WN_OV(e) vneg[e] = -vw[e];
              int hn_;
              const int fb = search(qw, vneg, gw, Hn, minC, maxTry, false, false, qb, vb, gb, lpB, HB, hn_);
              if (aborted) { stop = true; break; }
              if (fb >= 0) Ib = fb;
              nEvB = (2 << (fb >= 0 ? fb : maxTry)) - (1 << minC);
            }
            cret = INTEG == I_ADAPT_R2P ? cSim : If;  // :135-137 / 466-475 (lwt follows from If, Ib, c: the R wave forms it)
            WN_OV(e) { qa[e] = qw[e]; vi[e] = vw[e]; ga[e] = gw[e]; }
            Ha = Hn;
            WN_O2STAMP(0);
            const int ent = (nb0 + k) & 31;
            vst(V_RQ + ent, qa); vst(V_RV + ent, vi);  // (the ring holds xi * v: intOut.v (:135, :473) is formed by the R wave)
            L[L_REC + 4 * ent] = Hn; L[L_REC + 4 * ent + 1] = lpn;
            LU[L_REC + 4 * ent + 2] = ((unsigned long long)(uint32_t)(nEvF | (nEvB << 16)) << 32) | (uint32_t)(If | (Ib << 8) | (cret << 16));
            n = nb0 + k + 1;
            o2st_[6] += 1;
            release();
            LU[L_PROD] = ((unsigned long long)seq << 32) | (unsigned)n;
            if (!uni_b(isfinite(Hn))) { stop = true; break; }  // forced reject (:316-319)
            WN_O2STAMP(1);
          }
        } else
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
          const double hh = wn_readlane(hV, k);
          const double hh2 = 0.5 * hh, hq = hh;
          // fixedLeapFrog (adaptiveIntegrators.py:49-59) from (q, xi*v, g): `vi` IS xi*v for the whole run
          WN_OV(e) { vi[e] = vi[e] + hh2 * ga[e]; qa[e] = qa[e] + hq * vi[e]; }
          // the new position goes to the ring HERE, a whole gradient in front of the velocity: two bursts of 6 KB instead of one of 12
          // (one burst: 490 cycles per leaf, two: tools/sw_nostore_ab.sh); the count word is published behind the NEXT leaf's gradient
          const int ent = (nb0 + k) & 31;
          vst(V_RQ + ent, qa);
          const double lpp = tgt.grad(qa, ga, lane);
          double ke = 0.0;
          WN_OV(e) { vi[e] = vi[e] + hh2 * ga[e]; ke += vi[e] * vi[e]; }
          {  // this lane's shares of lp and of the kinetic energy: summed over the wave by the R wave (36 instructions of a dependent chain less per leaf here)
            d2_t pr; pr[0] = lpp; pr[1] = ke;
            *reinterpret_cast<__attribute__((address_space(3))) d2_t*>(L + L_PART + ((size_t)ent * 64 + lane) * 2) = pr;
          }
          WN_O2STAMP(0);
          // every fourth leaf: publish the leaves in front of this one (their stores were issued at least a whole evaluation ago: the
          // release fence does not wait), and look at the abort word -- the R wave has seen an event inside the doubling
          if ((k & 3) == 0) {
            release();
            LU[L_PROD] = ((unsigned long long)seq << 32) | (unsigned)(nb0 + k);
            if ((uint32_t)LU[L_ABORT] == seq) { stop = true; break; }
          }
          // (the ring holds xi * v, the integrator's orientation: intOut.v = xi * vv (adaptiveIntegrators.py:57) is formed by the R wave)
          vst(V_RV + ent, vi);
          n = nb0 + k + 1;
          o2st_[6] += 1;
          // (a non-finite energy -- forced reject, :316-319 -- is the R wave's to see: it raises the abort word)
          WN_O2STAMP(1);
        }
        ku = ku_behind(i, k0, k1, ku, pred);
      }
      release();
      LU[L_PROD] = ((unsigned long long)seq << 32) | 0x80000000u | (unsigned)n;  // (bit 31: the run has ended)
    }
    if constexpr (ADAPT) LU[L_JOB] = 0xFFFFFFFF00000000ull;  // the helpers leave
    stamps_out();
    return;
  }

  // =======================================================================================================================
  // R wave: the bookkeeping (WALNUTS.py:189-695 minus the integrator calls)
  // =======================================================================================================================
  // the wave's window of uniforms: lane l holds uniform 2 * rb_base + l of iteration rb_iter (wn_rng.h stream 0)
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

  double H0 = 0.0, Ha = 0.0, Hb = 0.0, lp0, WoldSum = 1.0, WnewSum = 0.0, lwtSumf = 0.0, lwtSumb = 0.0;
  double orbitLen_ = 0.0, orbitLenSam_ = 0.0, indexStat_ = 0.0, indexStatOld_ = 0.0, timeLenF_ = 0.0, timeLenB_ = 0.0;
  double minlw = INFINITY, maxlw = -INFINITY, Hmin = INFINITY, Hmax = -INFINITY, lpP = 0.0, lpPL = 0.0;
  int iter = 1, i = 0, xi = 1, a = 0, b = 0, n = 0, nleaf = 1;
  uint32_t Bbits = 0, ku = 0, khc = 0;
  int nevalF = 0, nevalB = 0, L_ = 0, Lold_ = 0, NdS = 0, NdC = 0, maxFint = 0, maxBint = 0, stopCode = 0;
  int minIf = 1 << 30, maxIf = -(1 << 30), minc = 1 << 30, maxc = -(1 << 30), nneq = 0, nzero = 0, nused = 0;
  int ip = 0, ipl = 0;
  bool Hnan = false, bothEndsPassive = false;
  bool predF = true, predB = true;  // "the next leaf in this direction takes its multinomial draw"
  uint32_t seq = 0;
  bool gone = false;  // a wait gave up

  auto send = [&](int kind, int n_, bool pred_, bool swap_) {
    ++seq;
    LU[L_CONS] = (unsigned long long)(unsigned)n_;  // (leaves of the new run retired so far: the T wave's ring flow control)
    // "every leaf of the rest of this doubling takes its multinomial draw": true once the doubling's weights sum to something (:426)
    LU[L_SURE] = (i == 0 || WnewSum > prm.wtSumThresh) ? (unsigned long long)(seq) : 0ull;
    LU[L_CMD + 1] = ((unsigned long long)(uint32_t)iter << 32) | (uint32_t)ip;
    LU[L_CMD + 2] = ((unsigned long long)(uint32_t)i << 32) | (uint32_t)n_;
    LU[L_CMD + 3] = ((unsigned long long)khc << 32) | ku;
    LU[L_CMD + 4] = ((unsigned long long)(uint32_t)xi << 32) | (pred_ ? 1u : 0u) | (swap_ ? 2u : 0u);
    L[L_CMD + 5] = lpP; L[L_CMD + 6] = Ha;
    release();
    LU[L_CMD] = ((unsigned long long)seq << 32) | (uint32_t)kind;
  };
  auto emit_generated = [&](int it, const double* qP) {
    double* rowp = prm.samples + ((size_t)clc * (prm.numIter + 1) + it) * prm.dg;
    // (arithmetic, not a select of the two kernel arguments: hipcc turned the select into a two-entry array in scratch)
    const int fk = (int)(prm.generated == G_FIRSTK), sq = (int)(prm.generated == G_Q0_SUMSQ);
    const int kk = sq ? 1 : fk * prm.gen_k + (1 - fk) * D;
    WN_OV(e) { const int gi = gidx(e); if (gi >= 0 && gi < kk) rowp[gi] = qP[e]; }
    if (sq) {
      double s = 0.0;
      WN_OV(e) s += qP[e] * qP[e];
      s = reduce1(s);
      if (lane == 0) rowp[1] = s;
    }
  };

  // ---- q_c = q0; f0, grad0 = lpFun(q0)  (WALNUTS.py:177,249) ----
  {
    double q0v[NE], g0v[NE];
    WN_OV(e) { const int gi = gidx(e); q0v[e] = (gi >= 0 && inrange) ? prm.q0[(size_t)clc * D + gi] : 0.0; }
    double lp_part = tgt.grad(q0v, g0v, lane);
    lpP = Tgt::lp_finish(reduce1(lp_part));
    vst(V_PROP + ip, q0v);
    if (inrange) emit_generated(0, q0v);  // samples[:,0] = generated(q0)  (:164)
  }
  bool live = inrange && prm.numIter >= 1;

  while (__any(live) && !gone) {  // ---- iterations (WALNUTS.py:189) ----
    // ---- iteration start (:196-276) ----
    rng_ensure((uint32_t)iter, 0u);  // uniforms 0..63: the M <= 24 direction bits and the first steps
    Bbits = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)M; ++k)  // :216  B = floor(uniform(0,2,size=M))
      if ((int)floor(0.0 + (2.0 - 0.0) * uni(k)) != 0) Bbits |= (1u << k);
    Bbits = (uint32_t)uni_i((int)Bbits);
    ku = (uint32_t)M;
    a = 0; b = 0; maxFint = 0; maxBint = 0;
    predF = predB = true;
    i = 0;
    xi = (Bbits & 1u) ? -1 : 1;
    send(K_NEWITER, 0, xi > 0 ? predF : predB, false);
    WoldSum = 1.0; WnewSum = 0.0; lwtSumb = 0.0; lwtSumf = 0.0;  // :260-264
    stopCode = 0; bothEndsPassive = false;
    nevalF = nevalB = 0; Lold_ = L_ = 0; orbitLen_ = orbitLenSam_ = 0.0; NdS = NdC = 0;
    indexStat_ = indexStatOld_ = 0.0; timeLenF_ = timeLenB_ = 0.0;
    minIf = minc = 1 << 30; maxIf = maxc = -(1 << 30); nneq = nzero = nused = 0;
    minlw = INFINITY; maxlw = -INFINITY;
    {  // the T wave's momentum: Hs[I0] = -f0 + 0.5*sum(v**2)  (:256); multinomialLscale (:259)
      unsigned spins = 0; unsigned long long spin_t0 = 0ull;
      while ((uint32_t)LU[L_ACK] != seq) {
        if (dead()) { gone = true; break; }
        if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); gone = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (gone) break;
      acquire();
      const double kp = L[L_KP];
      lp0 = lpP;
      H0 = -lp0 + 0.5 * kp;
      Ha = Hb = H0;
      Hnan = isnan(H0); Hmin = Hmax = H0;
    }
    bool endIter = false;

    while (!endIter) {  // ---- doublings (:281) ----
      {  // start of doubling i (:283-294)
        const int nxi = ((Bbits >> i) & 1u) ? -1 : 1;
        const bool swap = nxi != xi;
        if (swap) {  // the other end of the orbit becomes the active one (the T wave swaps the vectors)
          const double th = Ha; Ha = Hb; Hb = th;
          xi = nxi;
        }
        n = 0; nleaf = 1 << i;
        ipl = ip; lpPL = lpP; Lold_ = L_; indexStatOld_ = indexStat_;
        if (i > 0) WnewSum = 0.0;  // :390
        if (i > 0) send(K_GO, 0, xi > 0 ? predF : predB, swap);
      }
      const bool fw = xi > 0;
      bool predRun = fw ? predF : predB;  // what the T wave's running command assumes
      bool fire0 = false;  // full-orbit U-turn (:622), due at the last leaf of the doubling
      bool sub = false;

      while (n < nleaf) {  // ---- sub-batches of the doubling's blocks ----
        const int nb0 = n & ~15;
        const int k0 = n & 15;
        int k1;
        WN_O2STAMP(4);
        {  // what the T wave has published of this run
          unsigned spins = 0; unsigned long long spin_t0 = 0ull;
          int npub = 0;
          const int blk_end = nleaf - nb0 < 16 ? nleaf : nb0 + 16;
          constexpr int BATCH = ADAPT ? 1 : WN_O2_BATCH;  // (a macro step of the adaptive integrators is ~100 k cycles: retire it at once)
          const int want = n + BATCH < blk_end ? n + BATCH : blk_end;
          for (;;) {
            const unsigned long long w = LU[L_PROD];
            const bool mine = (uint32_t)(w >> 32) == seq;
            npub = mine ? (int)((uint32_t)w & 0x7FFFFFFFu) : 0;
            if (npub >= want || (mine && ((uint32_t)w & 0x80000000u) && npub > n)) break;
            if (dead()) { gone = true; break; }
            if (WN_O2_EXPIRED(spins, spin_t0, 1)) { give_up(); gone = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (gone) break;
          acquire();
          npub = uni_i(npub);
          k1 = npub - nb0 < 16 ? npub - nb0 : 16;
        }
        WN_O2STAMP(2);
        const int rb = nb0 & 16;  // the block's half of the ring
        // the draws of leaves k0 .. k1-1 as the reference makes them, from the stream index the bookkeeping has reached
        double hV, urV, upV;
        draws(i, iter, k0, ku, khc, predRun, hV, urV, upV);
        (void)urV;
        // left endpoint of the sub-trees of 2**j leaves, j >= 5, that start at this block's first leaf (levels 1..4 stay inside the ring)
        if (k0 == 0 && (nb0 == 0 ? i >= 5 : __builtin_ctz((unsigned)nb0) >= 5)) {
          const int sl = nb0 == 0 ? 0 : __builtin_ctz((unsigned)nb0);
          double tq[NE], tv[NE];
          vld(V_RQ + rb, tq); vld(V_RV + rb, tv);
          WN_OV(e) tv[e] = (double)xi * tv[e];  // (ring: xi * v; slots and the passive end: v in forward-time orientation)
          vst(V_SLOT + 2 * sl, tq); vst(V_SLOT + 2 * sl + 1, tv);
        }
        // =========================== bookkeeping of leaves k0 .. k1-1, lane k = leaf k ===========================
        const int k = k16;
        const bool inb = k >= k0 && k < k1;
        double Hn_k, lp_k;
        if constexpr (INTEG == I_FIXED) {  // the energies of leaves k0 .. k1-1 from the T wave's shares: Hams[i] = -fnew + 0.5*sum(vv*vv)
          Hn_k = 0.0; lp_k = 0.0;
#pragma unroll 1
          for (int j = k0; j < k1; ++j) {
            const d2_t pr = *reinterpret_cast<__attribute__((address_space(3))) d2_t*>(L + L_PART + ((size_t)(rb + j) * 64 + lane) * 2);
            double a_ = pr[0], b_ = pr[1];
            reduce2(a_, b_);
            a_ = Tgt::lp_finish(a_);
            const double H_ = -a_ + 0.5 * b_;
            Hn_k = k == j ? H_ : Hn_k; lp_k = k == j ? a_ : lp_k;
          }
        } else {
          Hn_k = L[L_REC + 4 * (rb + k)]; lp_k = L[L_REC + 4 * (rb + k) + 1];
        }
        const double h_k = hV, up_k = upV;
        int iA = 0, iB = 1;  // If | Ib << 8 | c << 16;  nEvalF | nEvalB << 16  (fixedLeapFrog: 0, one evaluation)
        if constexpr (INTEG != I_FIXED) {
          const unsigned long long rr = LU[L_REC + 4 * (rb + k) + 2];
          iA = (int)(uint32_t)rr; iB = (int)(uint32_t)(rr >> 32);
        }
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
        const bool inb0 = inb && lane < 16;
        const bool drawn_k = (i != 0) && (WS > prm.wtSumThresh);
        // :426  take this state with probability Wnew / WnewSum; i == 0 (:322-328): unconditional, no uniform
        const bool pick_k = (i == 0) ? true : (drawn_k && up_k < Wn / WS);
        const uint64_t m_in = __ballot(inb0);
        const uint64_t m_nf = __ballot(inb0 && !isfinite(Hn_k));
        const uint64_t m_dr = __ballot(inb0 && drawn_k);
        const uint64_t m_pk = __ballot(inb0 && pick_k);
        const uint64_t m_bad = (i != 0) ? ((predRun ? ~m_dr : m_dr) & m_in) : 0ull;

        WN_O2STAMP(0);
        // ---- U-turn checks (stopCondition, WALNUTS.py:95-97, with the older state on the side opposite to xi): wave-wide dots, the
        // loads of the next check in flight while this one is summed (wn_k_orbit.hip, wide mode) ----
        uint64_t m_fire = 0ull;  // leaves at which a sub-tree check fired
        bool f0_now = false;
        {
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
            if (c.lvl == 0) { lq = V_COLDQ; lv = V_COLDV; }
            else if (c.lvl <= 4) { lq = V_RQ + rb + (c.r + 1 - (1 << c.lvl)); lv = lq + (V_RV - V_RQ); }
            else {
              const int nl = nb0 + c.r + 1 - (1 << c.lvl);  // left leaf of the sub-tree
              const int sl = nl == 0 ? 0 : __builtin_ctz((unsigned)nl);
              lq = V_SLOT + 2 * sl; lv = lq + 1;
            }
            vld(V_RQ + rb + c.r, qr); vld(V_RV + rb + c.r, vr); vld(lq, ql); vld(lv, vl);
          };
          auto sum_up = [&](const Chk& c, const double* qr, const double* vr, const double* ql, const double* vl) {
            // (ring entries hold xi * v: the right state always, the left one of levels 1..4; +-1 is exact, the products are the one-stream engine's)
            const double sr = (double)xi, sl_ = (c.lvl >= 1 && c.lvl <= 4) ? (double)xi : 1.0;
            double a1 = 0.0, a2 = 0.0;
            WN_OV(e) {
              const double dq = qr[e] - ql[e];
              const double vrf = sr * vr[e], vlf = sl_ * vl[e];
              a1 += vrf * dq;
              a2 += vlf * dq;
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
        }
        WN_O2STAMP(1);
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
        } else if (t_bad < k1 && nb0 + t_bad < nleaf - 1) {  // (the T wave runs on beyond this sub-batch: any leaf behind it is wrong)
          redo = true;
          Lfull = t_bad;
        } else {
          Lfull = k1 - 1;
        }
        const int A_end = end_nf ? t_ev : Lfull;  // last leaf that reached the accumulators at the top of the reference's step
        // ---- accumulators over leaves k0 .. A_end ----
        {
          const bool macc = inb && k <= A_end;
          if constexpr (INTEG == I_FIXED) {
            nevalF += A_end - k0 + 1;  // one evaluation per leaf
          } else {
            // nevalF, nevalB, #(If != Ib), #(If == 0): four rows of one register
            int v4 = row == 0 ? ef_k : row == 1 ? eb_k : row == 2 ? (int)(If_k != Ib_k) : (int)(If_k == 0);
            v4 = rowsum_i(macc ? v4 : 0);
            nevalF += rl_i(v4, 0); nevalB += rl_i(v4, 16); nneq += rl_i(v4, 32); nzero += rl_i(v4, 48);
            nused += A_end - k0 + 1;
            int m4 = row == 0 ? If_k : row == 1 ? c_k : row == 2 ? -If_k : -c_k;
            m4 = rowmin_i(macc ? m4 : (1 << 30));
            const int a0 = rl_i(m4, 0), a1 = rl_i(m4, 16), a2 = -rl_i(m4, 32), a3 = -rl_i(m4, 48);
            minIf = a0 < minIf ? a0 : minIf; minc = a1 < minc ? a1 : minc; maxIf = a2 > maxIf ? a2 : maxIf; maxc = a3 > maxc ? a3 : maxc;
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
              vld(V_RQ + rb + Pk, qp);
              vst(V_PROP + ip, qp);
              lpP = wn_readlane(lp_k, Pk);
              L_ = fw ? b + 1 + (nb0 + Pk) : a - 1 - (nb0 + Pk);
              const double tp = wn_readlane(P1, 16 + Pk);
              indexStat_ = fw ? tp : -tp;
            }
            if (i != 0) {  // the draws these leaves consumed, and what the next leaf is expected to do
              const uint64_t rng = m_in & ((2ull << Lfull) - 1ull);
              // leaves with an even index drew the pair's two step sizes
              const uint64_t evn = (nb0 & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
              ku += 2u * (uint32_t)__builtin_popcountll(rng & evn) + r_ * (uint32_t)__builtin_popcountll(rng) + (uint32_t)__builtin_popcountll(rng & m_dr);
              const bool dl = (m_dr >> Lfull) & 1ull;
              if (fw) predF = dl; else predB = dl;
              // the second step size of a pair was drawn with the first (:395): its index, for a sub-batch that starts at the pair's second leaf
              khc = ku - 1u - r_ - (uint32_t)dl;
            } else {
              ku += 1u + r_;
            }
            Ha = wn_readlane(Hn_k, Lfull);
          }
        }
        if (f0_now && !end_nf && !end_sub && !redo) fire0 = true;
        o2st_[6] += (unsigned long long)(Lfull + 1 - k0); o2st_[7] += 1;
        n = nb0 + Lfull + 1;
        // (the abort word in front of the count: the T wave must not take "leaf n-1 retired" for "go on" when the verdict is a restart)
        if (end_nf || end_sub || redo) LU[L_ABORT] = (unsigned long long)seq;
        if (i == 0 || WnewSum > prm.wtSumThresh) LU[L_SURE] = (unsigned long long)seq;
        LU[L_CONS] = (unsigned long long)(unsigned)n;
        WN_O2STAMP(3);
        if (end_nf) {  // :316-319, 414-417, 457-459 (Q3, Q4)
          const bool first_e = (i == 0) || !((nb0 + t_ev) & 1);
          if (first_e) stopCode = 999;
          endIter = true;
          LU[L_ABORT] = (unsigned long long)seq;
          break;
        }
        if (end_sub) { sub = true; LU[L_ABORT] = (unsigned long long)seq; break; }
        if (redo) {  // restart behind the mispredicted leaf: its state is in the ring, the T wave re-evaluates the gradient
          LU[L_ABORT] = (unsigned long long)seq;
          predRun = fw ? predF : predB;
          send(K_REDO, n, predRun, false);
        }
      }
      if (gone || endIter) break;
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
    if (gone) break;
    // ---- end of the iteration (:653-695) ----
    {
      double qP[NE];
      vld(V_PROP + ip, qP);
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
    iter++;
    if (iter > prm.numIter) live = false;
  }
  if (!gone) { i = 0; send(K_EXIT, 0, false, false); }
  stamps_out();
}

// ---- host side ----------------------------------------------------------------------------------------------------------
size_t orbit2_work_doubles(int NE, int C, int M) { return (size_t)C * orbit2::wide_doubles(NE, M); }

// Plain sampling runs of the three leapfrog integrators on the wide geometry (StockWatson) WHEN THE CHAINS LEAVE SIMDs IDLE.  A chain here
// is 2 waves (fixedLeapFrog) or 4 (adaptive: 107 KB of LDS, one chain per CU), and what it buys is latency: with more chains than that the
// one-stream engine's one wave per chain fills the SIMDs itself and wins on throughput (tools/sw_chains_ab.sh, evaluations/s, this kernel
// against the engine: adaptLeapFrogD 256 chains 9.4e7 / 7.2e7, 512: 1.1e8 / 1.4e8, 2048: 1.5e8 / 3.3e8; fixedLeapFrog 512: 2.1e8 / 1.4e8,
// 1024: 2.2e8 / 2.5e8).  So: adaptive integrators up to one chain per CU, fixedLeapFrog up to two (WN_ORBIT2_ANYC=1 lifts the bound).
// WN_ORBIT_ONEWAVE=1 keeps the one-wave forms (fixedLeapFrog: wn_k_orbit.hip's wide mode; the adaptive integrators: the one-stream
// engine, or that wide mode with WN_ORBIT_WIDE_ADAPT=1), WN_ORBIT_LEGACY=1 the one-stream engine, WN_ORBIT2_NOADAPT=1 this kernel for
// fixedLeapFrog only.
static int device_cus() {  // (asked on every wn_set_target / launch: looked up once per device)
  static int cached[16] = {0};
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 16 && cached[dev]) return cached[dev];
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 16) cached[dev] = n;
  return n;
}
bool orbit2_run(const KernelChoice& kc, int integrator, int n_chains) {
  if (!orbit_wide_geom(kc) || kc.p1 != T_STOCKWATSON || (kc.route & (WN_ROUTE_ORBIT_LEGACY | WN_ROUTE_ORBIT_NOWIDE | WN_ROUTE_ORBIT_ONEWAVE))) return false;
  const bool adaptive = integrator == I_ADAPT_D || integrator == I_ADAPT_R2P;
  if (integrator != I_FIXED && !(adaptive && !(kc.route & WN_ROUTE_ORBIT2_NOADAPT))) return false;
  return (kc.route & WN_ROUTE_ORBIT2_ANYC) || n_chains <= device_cus() * (adaptive ? 1 : 2);
}
bool orbit2_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm) {
  if (!orbit2_run(kc, prm.integrator, prm.C)) return false;
  if (warm || prm.mode != MODE_SAMPLE || prm.trace_n || prm.trace || prm.omin || adapt == 2) return false;
  if (prm.M < 1 || prm.M > MAX_M || prm.maxC > 12 || prm.minC < 0 || prm.minC > prm.maxC) return false;
  return prm.generated == G_IDENTITY || prm.generated == G_FIRSTK || prm.generated == G_Q0_SUMSQ;
}

hipError_t launch_orbit2(const KernelChoice& kc, const Params& prm, int* kabort, hipStream_t st) {
  (void)kc;
  dim3 grid(prm.C), block(128), block4(256);  // adaptive integrators: T, R and two helper waves
  if (prm.integrator == I_ADAPT_R2P) hipLaunchKernelGGL((wn_orbit2_kernel<WTgtStockWatson<12>, 12, I_ADAPT_R2P>), grid, block4, 0, st, prm, kabort);
  else if (prm.integrator == I_ADAPT_D) hipLaunchKernelGGL((wn_orbit2_kernel<WTgtStockWatson<12>, 12, I_ADAPT_D>), grid, block4, 0, st, prm, kabort);
  else hipLaunchKernelGGL((wn_orbit2_kernel<WTgtStockWatson<12>, 12, I_FIXED>), grid, block, 0, st, prm, kabort);
  return hipGetLastError();
}

}  // namespace wn
