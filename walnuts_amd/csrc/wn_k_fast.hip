// wn_k_fast.hip -- role-split kernel of the headline path: fixedLeapFrog on a dense / diagonal Gaussian,
// 16 chains per workgroup in the MFMA layout of wn_policy_coop.h.  Eight waves per workgroup, two per SIMD:
// seven T waves, each owning ONE 16-row tile of every D-vector (4 doubles per lane and vector, 25 resident
// A fragments) and one B wave (the bookkeeper of wn_fast.h).  See wn_fast.h for the protocol.  Two waves per
// SIMD hide each other's LDS / cross-lane latencies, and a T wave needs ~150 of its 256 registers, which
// leaves room to keep every on-chip U-turn check of a tick in flight at once and to prefetch the left
// endpoints of the deep sub-tree levels from memory under the MFMAs.
//
// One tick of the workgroup:
//
//   T waves                                               B wave
//   ------------------------------------------------     ---------------------------------------------
//   read command C[t]; proposal store / iteration        read + sum the partials P[t-1] (lp, KE, U-turn
//   start / end swap on command; first half of the       dots -> fire mask) of the leaf being retired
//   leapfrog step; publish Q[t] to LDS
//   ---------------------------- barrier A (LDS only) -----------------------------------------------
//   G = -A Q (25 x v_mfma_f64_16x16x4_f64 per wave),     FastBook::tick: retire tick t-1 (energies, weights,
//   second half step, per-wave partials P[t] of lp,      multinomial pick, doubling / iteration ends,
//   KE and the U-turn dots due at this leaf, sub-tree    diagnostics row), issue command C[t+1]; momentum
//   endpoint stores                                      normals of one chain's next iteration
//   ---------------------------- barrier B (LDS only) -----------------------------------------------
//
// LDS (doubles): Q exchange 2 x [k-step][lane] (double-buffered: buffer t&1 still holds the position of
// tick t-2 when a pick of that leaf is commanded, buffer (t-1)&1 is the first state of the leaf pair),
// passive orbit end (q, v), NLC cached sub-tree levels (q, v), partials [quantity][tp][chain], commands.
// Memory: deeper sub-tree levels, the two proposal buffers (positions only: the gradient of a restart
// state comes out of the idle tick's MFMA) and the momentum normals, in the engine's workspace.
#include "wn_launch.h"  // hip_runtime first

#include "wn_fast.h"
#include "wn_policy_coop.h"  // d4_t, CK_DENSE / CK_DIAG

namespace wn {
using namespace fast;

#define WN_EV(e) _Pragma("unroll") for (int e = 0; e < NE; ++e) if (e < nev)
#define WN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// diagnostic builds (-DWN_STAMPS): cycles per segment of the tick, summed per wave
#if defined(WN_STAMPS)
#define WN_FSTAMP(k)                                                                         \
  do {                                                                                       \
    unsigned long long _t;                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
    fst_[k] += _t - fst_last_;                                                               \
    fst_last_ = _t;                                                                          \
  } while (0)
#define WN_FSTAMP_DECL unsigned long long fst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fst_last_ = __builtin_readcyclecounter();
#define WN_FSTAMP_OUT(prm)                                                                             \
  if ((prm).stamps && (threadIdx.x & 63) == 0) {                                                       \
    unsigned long long* o_ = (prm).stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16;        \
    for (int k = 0; k < 8; ++k) o_[k] = fst_[k];                                                       \
  }
#else
#define WN_FSTAMP(k) do {} while (0)
#define WN_FSTAMP_DECL
#define WN_FSTAMP_OUT(prm)
#endif

template <int KS, int TK, int NLC>
struct FastGeom {
  static constexpr int NW = 8;                // tile slots = waves (slot 7 is the B wave: no tile)
  static constexpr int NE = 4;                // doubles per lane and vector: one 16-row tile
  static constexpr int VECD = NW * NE * 64;   // one D-vector of the 16-chain group in memory (padded to 128 rows)
  static constexpr int REG = KS * 64;         // ... in LDS: [s][lane], s = 4*tile + e < KS (k-steps that hold coordinates)
  static constexpr int DPAD = 16 * NW;
  static constexpr int MP = 11;               // largest M (sizes the partials)
  static constexpr int NQ = Q_DOT0 + 2 * (MP + 1);
  static constexpr int NFAST = 2 + NLC;
  static_assert(KS <= 28, "seven tiles of 16 rows");
  // LDS map, in doubles
  static constexpr int OFF_Q = 0;                          // 2 x REG
  static constexpr int OFF_COLD = OFF_Q + 2 * REG;         // qb, vb
  static constexpr int OFF_LC = OFF_COLD + 2 * REG;        // NLC x (q, v)
  static constexpr int OFF_PART = OFF_LC + NLC * 2 * REG;  // NQ x [tile pair][chain][2]: 16-byte stride between a wave's chains, two tiles per 16-byte read of the bookkeeper
  static constexpr int OFF_CMD = OFF_PART + NQ * 16 * NW;  // 2 x 16 commands of 2 doubles
  static constexpr int OFF_DUMP = OFF_CMD + 2 * 16 * 2;    // 64 doubles nobody reads (see put() of the T waves)
  static constexpr int OFF_RED = OFF_DUMP + 64;            // service wave -> bookkeeper: [chain][lp, KE, initial KE, generated sum, fire words 0..3 (2 doubles)]
  static constexpr int OFF_RQ = OFF_RED + 16 * 6;          // bookkeeper -> service wave, 2 x [chain][iteration | block I, block R]
  static constexpr int OFF_SVC = OFF_RQ + 2 * 16 * 2;      // service wave -> bookkeeper, 2 x [chain][4] x {iteration | block, u0, u1}
  static constexpr int LDS_DOUBLES = OFF_SVC + 2 * 16 * 4 * 3;
  static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget (160 KiB per CU)");
  // workspace of one group, in doubles: (M+1) slots x (q, v), 2 proposals, momentum normals [chain][DPAD]
  static constexpr size_t group_doubles(int M) { return (size_t)((M + 1) * 2 + 2) * VECD + (size_t)16 * DPAD; }
};

// x[l] + x[l^16] + x[l^32] + x[l^48] in every lane (CoopPolicy::rowsum4)
__device__ __forceinline__ double fast_rowsum4(double x) {
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

// ------------------------------------------------------------------------------------------------
// Service duties of T wave 3 (the wave that shares its SIMD with the bookkeeper and, for D = 100, owns the 4-row tile):
// what the bookkeeper needs from a tick but any wave can compute.
//   sums:     before barrier A of tick t, the sums over the tile slots of tick t-1's partials (lp, KE, initial KE,
//             generated sum: one per lane group) and the U-turn verdicts of that tick's leaf (level j in lane group
//             j & 3) -- wn_fast.h fire_of.
//   uniforms: behind the MFMAs of tick t, the Philox blocks the bookkeeper asked for at tick t-1 (the next two of its
//             issue-side and of its retire-side stream: one block per lane), which it will look up at tick t+1.
// ------------------------------------------------------------------------------------------------
template <class G>
__device__ __forceinline__ void fast_service_sums(double* lds, uint32_t pop, uint32_t pn, double phs) {
  const int lane = threadIdx.x & 63, cn = lane & 15, lg = lane >> 4;
  const double* part = lds + G::OFF_PART;
  auto sum8 = [&](int k) {  // sum of the eight tile partials of one quantity, in tile order
    typedef double d2r_t __attribute__((ext_vector_type(2)));
    const d2r_t* r = reinterpret_cast<const d2r_t*>(part + k * 16 * G::NW + cn * 2);  // + tile pair * 16
    const d2r_t a = r[0], b = r[16], c = r[32], d = r[48];
    return ((a[0] + a[1]) + (b[0] + b[1])) + ((c[0] + c[1]) + (d[0] + d[1]));
  };
  double* red = lds + G::OFF_RED + cn * 6;
  if (__any((pop & OP_NEWITER) != 0)) red[lg] = sum8(lg);  // lg 2, 3 are read: Q_KEINIT, Q_GEN (the bookkeeper sums Q_LP, Q_KE itself)
  const bool step = (pop & OP_STEP) != 0;
  const int i = (int)((pop >> OP_I_SHIFT) & 31u), n = (int)pn;
  const int nchk = step ? nchecks(i, n) : 0;
  const double xi = phs < 0.0 ? -1.0 : 1.0;  // hs = xi * h
  uint32_t fire = 0u;
  for (int j0 = 0; __any(nchk > j0); j0 += 4) {
    const int j = j0 + lg;
    const int jr = j <= G::MP ? j : G::MP;  // rows beyond the deepest level are not read
    const double d1 = sum8(Q_DOT0 + 2 * jr), d2 = sum8(Q_DOT0 + 2 * jr + 1);
    if (nchk > j && need_check(i, n, j) && ((xi * d1 < 0.0) || (xi * d2 < 0.0))) fire |= (1u << j);
  }
  reinterpret_cast<uint32_t*>(red + 4)[lg] = fire;
}
template <class G>
__device__ __forceinline__ void fast_service_uniforms(const Params& prm, double* lds, uint32_t t) {
  const int lane = threadIdx.x & 63, cn = lane & 15, lg = lane >> 4;
  const uint32_t* rq = reinterpret_cast<const uint32_t*>(lds + G::OFF_RQ + (t & 1u) * 32 + cn * 2);  // written at tick t-1
  const uint32_t iter = rq[0], blk = ((lg & 1) ? rq[2] : rq[1]) + (uint32_t)(lg >> 1);
  const uint32_t chain = prm.chain0 + blockIdx.x * 16 + cn;
  Philox4 r = philox4x32_10(blk, iter, chain, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
  double* e = lds + G::OFF_SVC + (t & 1u) * 192 + (cn * 4 + lg) * 3;
  reinterpret_cast<uint32_t*>(e)[0] = iter;
  reinterpret_cast<uint32_t*>(e)[1] = blk;
  e[1] = u53(r.w[0], r.w[1]);
  e[2] = u53(r.w[2], r.w[3]);
}

// the two halves of fast_rowsum4, for work that is spread over MFMA gaps
__device__ __forceinline__ double fast_rowsum4_a(double x) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const u2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const u2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double fast_rowsum4_b(double y) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const unsigned lo = __double2loint(y), hi = __double2hiint(y);
  const u2_t a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const u2_t b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// ------------------------------------------------------------------------------------------------
// T waves
// ------------------------------------------------------------------------------------------------
// NEV: registers e < NEV of this wave hold coordinates (k-steps s = 4*tile + e < KS), the rest stay 0.  A template
// parameter, not a run-time bound: every `e < nev` below is then resolved at compile time, and the phase-2 block of
// the tick loop stays free of control flow (see there).
template <int KS, int TK, int NLC, int NEV, bool SVC>
__device__ __forceinline__ void fast_t_wave(const Params& prm, double* lds, const int tile) {
  using G = FastGeom<KS, TK, NLC>;
  constexpr int nev = NEV;
  constexpr int NE = G::NE, NFAST = G::NFAST;
  const int lane = threadIdx.x & 63, cn = lane & 15, lg = lane >> 4;
  const int cl = blockIdx.x * 16 + cn;
  const bool inrange = cl < prm.C;
  const int D = prm.D;
  auto gidx = [&](int e) { return 16 * tile + 4 * e + lg; };

  double* const own = lds + (size_t)(4 * tile) * 64 + lane;  // + region offset + e*64
  double* part = lds + G::OFF_PART;
  const Cmd* cmdbuf = reinterpret_cast<const Cmd*>(lds + G::OFF_CMD);
  double* wbase = prm.work + (size_t)blockIdx.x * G::group_doubles(prm.M);
  auto gvec = [&](int vec) { return wbase + (size_t)vec * G::VECD + ((size_t)(cn * G::NW + tile) * NE) * 4 + lg; };  // + e*4
  const int vprop = (prm.M + 1) * 2;
  const double* znext = wbase + (size_t)((prm.M + 1) * 2 + 2) * G::VECD + (size_t)cn * G::DPAD;

  // -A fragments of this wave's row tile, resident for the whole run
  constexpr int NA = (TK == CK_DENSE) ? KS : 1;
  double afrag[NA];
  double adiag[NE];
  if (TK == CK_DENSE) {
    const int row = 16 * tile + (lane & 15);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int col = 4 * s + (lane >> 4);
      afrag[s] = (row < D && col < D) ? -prm.tpar[(size_t)row * D + col] : 0.0;
    }
  } else {
    afrag[0] = 0.0;
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) adiag[e] = (TK == CK_DIAG && gidx(e) < D) ? prm.tpar[gidx(e)] : 0.0;

  double q[NE], v[NE], g[NE], VL[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) { q[e] = v[e] = g[e] = VL[e] = 0.0; }

  // partials for the bookkeeper: [quantity][chain][tile slot]
  // (every lane stores: lanes of lane groups 1..3 into a dump row.  A store under `if (lg == 0)` is control flow, and
  // control flow between the MFMAs would cut the block in which the check work is to issue in their gaps)
  double* const dump = lds + G::OFF_DUMP + lane;
  double* const prow = part + ((tile >> 1) * 16 + cn) * 2 + (tile & 1);  // + k * 16 * NW
  auto put = [&](int k, double x) {
    x = fast_rowsum4(x);
    double* dst = lg == 0 ? prow + k * 16 * G::NW : dump;
    *dst = x;
  };

  // Row sums of FOUR quantities at once, as a butterfly: lane groups 0/1 exchange so that even rows end with a and b
  // summed over the pair and odd rows with c and d, then the halves exchange: lane group 0 ends with the sum of a,
  // 1 with c, 2 with b, 3 with d -- 6 permlane swaps and 3 additions instead of 16 and 8 -- and each lane group
  // stores its quantity (same association as fast_rowsum4: (x0 + x1) + (x2 + x3)).
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  auto put4 = [&](int ka, int kb, int kc, int kd, double a, double b, double c, double d) {
    auto pair16 = [](double P, double Q) {  // even rows: P_even + P_odd, odd rows: Q_even + Q_odd
      const u2_t l = __builtin_amdgcn_permlane16_swap(__double2loint(P), __double2loint(Q), false, false);
      const u2_t h = __builtin_amdgcn_permlane16_swap(__double2hiint(P), __double2hiint(Q), false, false);
      return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    };
    const double r1 = pair16(a, c), r2 = pair16(b, d);
    const u2_t l = __builtin_amdgcn_permlane32_swap(__double2loint(r1), __double2loint(r2), false, false);
    const u2_t h = __builtin_amdgcn_permlane32_swap(__double2hiint(r1), __double2hiint(r2), false, false);
    const double z = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    const int k = lg == 0 ? ka : lg == 1 ? kc : lg == 2 ? kb : kd;
    prow[k * 16 * G::NW] = z;
  };

  uint32_t pop = 0u, pn = 0u;  // last tick's command (service wave)
  double phs = 0.0;
  double pq0[NE], pv0[NE], pq1[NE], pv1[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) { pq0[e] = pv0[e] = pq1[e] = pv1[e] = 0.0; }
  WN_FSTAMP_DECL
  for (uint32_t t = 0;; ++t) {
    const int bq_ = G::OFF_Q + (int)(t & 1u) * G::REG, bp_ = G::OFF_Q + (int)((t + 1u) & 1u) * G::REG;  // this tick's / the previous tick's Q buffer
    Cmd c;
    if (t == 0) { c.hs = 0.0; c.op = inrange ? (OP_NEWITER | OP_INIT) : 0u; c.n = 0u; }
    else c = cmdbuf[(t & 1u) * 16 + cn];
    const uint32_t op = c.op;
    if (__any((op & OP_DONE) != 0)) break;
    if (SVC && t > 0) fast_service_sums<G>(lds, pop, pn, phs);
    const bool step = (op & OP_STEP) != 0;
    const int i = (int)((op >> OP_I_SHIFT) & 31u), n = (int)c.n;
    const int nchk = step ? nchecks(i, n) : 0;
    double keinit_part = 0.0, gen_part = 0.0;
    // ---- phase 1: commands ----
    // left endpoints of the first two deep sub-tree levels this leaf closes: requested from memory now, used
    // after the MFMAs (the slots were stored by these same lanes)
    // (the registers live across ticks and are not cleared: clearing them here costs an s_waitcnt vmcnt(0) at the
    // top of EVERY tick -- a possibly pending load into them -- which also waits for the previous tick's endpoint
    // stores.  A lane that loads nothing computes dots from stale values, which its bookkeeper never looks at.)
    const bool deep0 = nchk > NFAST, deep1 = nchk > NFAST + 1;
    if (__any(deep0)) {
      if (deep0) {
        const int sl = slot_of_leaf(n + 1 - (1 << NFAST));
        const double* sq = gvec(sl * 2);
        const double* sv = gvec(sl * 2 + 1);
        WN_EV(e) { pq0[e] = sq[e * 4]; pv0[e] = sv[e * 4]; }
      }
      if (deep1) {
        const int sl = slot_of_leaf(n + 1 - (2 << NFAST));
        const double* sq = gvec(sl * 2);
        const double* sv = gvec(sl * 2 + 1);
        WN_EV(e) { pq1[e] = sq[e * 4]; pv1[e] = sv[e * 4]; }
      }
    }
    if (__any((op & OP_PICK) != 0)) {  // the position of tick t-2 still sits in this tick's Q buffer
      if (op & OP_PICK) {
        double* dst = gvec(vprop + ((op & OP_IPPICK) ? 1 : 0));
        WN_EV(e) dst[e * 4] = own[bq_ + e * 64];
      }
    }
    if (__any((op & OP_NEWITER) != 0)) {
      if (op & OP_NEWITER) {
        if (op & OP_INIT) {
          double* dst = gvec(vprop);
          WN_EV(e) {
            q[e] = gidx(e) < D ? prm.q0[(size_t)cl * D + gidx(e)] : 0.0;
            dst[e * 4] = q[e];
          }
        } else {
          const double* src = gvec(vprop + ((op & OP_IPNEW) ? 1 : 0));
          WN_EV(e) q[e] = src[e * 4];
        }
        // samples[:, row] = generated(q)  (WALNUTS.py:164,695)
        double* row = prm.samples + ((size_t)cl * (prm.numIter + 1) + c.n) * prm.dg;
        if (prm.generated == G_Q0_SUMSQ) {
          if (tile == 0 && lg == 0) row[0] = q[0];
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < NE; ++e) s += q[e] * q[e];
          gen_part = s;
        } else {
          const int k = prm.generated == G_FIRSTK ? prm.gen_k : D;
          WN_EV(e) if (gidx(e) < k) row[gidx(e)] = q[e];
        }
        if (!(op & (OP_INIT | OP_LAST))) {  // v = normal(size=d); both ends start at (qc, v)  (:236-253)
          double kp = 0.0;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const double z = e < nev ? znext[gidx(e)] : 0.0;
            v[e] = z;
            kp += z * z;
          }
          keinit_part = kp;
          WN_EV(e) { own[G::OFF_COLD + e * 64] = q[e]; own[G::OFF_COLD + G::REG + e * 64] = v[e]; }
        }
        // q and v were loaded from memory in here: wait for them HERE.  Left to the compiler, the wait lands on the
        // common path in front of the half step as s_waitcnt vmcnt(0), which also waits for this tick's prefetches
        // and for every store still in flight -- on every tick.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      }
    }
    if (__any((op & OP_SWAP) != 0)) {  // the other end of the orbit becomes the active one (:283-286)
      if (op & OP_SWAP) {
        WN_EV(e) {
          const double tq = own[G::OFF_COLD + e * 64], tv = own[G::OFF_COLD + G::REG + e * 64];
          own[G::OFF_COLD + e * 64] = q[e]; own[G::OFF_COLD + G::REG + e * 64] = v[e];
          q[e] = tq; v[e] = tv;
        }
      }
    }
    // ---- first half of the leapfrog step (adaptiveIntegrators.py:51-52 with hs = xi*h; v is kept in
    // forward-time orientation: xi*(xi*v + h/2 g) = v + (xi h/2) g exactly) ----
    const double hs = step ? c.hs : 0.0, hs2 = 0.5 * hs;
#pragma unroll
    for (int e = 0; e < NE; ++e) { v[e] = v[e] + hs2 * g[e]; q[e] = q[e] + hs * v[e]; }
    WN_EV(e) own[bq_ + e * 64] = q[e];
    WN_FSTAMP(0);
    WN_LDS_BARRIER();  // A
    WN_FSTAMP(1);
    // ---- phase 2: gradient, second half step, partials ----
    // (On this chip the FP64 matrix and vector instructions share their arithmetic units: vector work placed in the
    // gaps of the MFMA chain -- tried, in source order and with sched_group_barrier -- lengthened the chain by about
    // what it saved behind it.  What counts is the NUMBER of vector instructions per tick.)
    if (TK == CK_DENSE) {
      d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};  // (a wave whose tile is beyond D multiplies zeros: no branch here)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double bq = lds[bq_ + s * 64 + lane];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[s], bq, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = acc[r];
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) g[e] = -(adiag[e] * q[e]);
    }
    double pl = 0.0, pk = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) pl += q[e] * g[e];
#pragma unroll
    for (int e = 0; e < NE; ++e) { v[e] = v[e] + hs2 * g[e]; pk += v[e] * v[e]; }
    WN_FSTAMP(2);
    // ---- U-turn dots of the sub-tree checks due at this leaf: dq = q' - qo, d1 = v'.dq, d2 = vo.dq (stopCondition,
    // WALNUTS.py:95-97; velocities forward-time).  All on-chip levels of the group's chains, needed by this lane's
    // chain or not (the bookkeeper reads only what its leaf needs): passive orbit end, leaf pair, cached sub-tree
    // levels, the two prefetched deep levels.  Registers e >= nev are zero on both sides of every product. ----
    constexpr int NCK = NFAST + 2;
    const bool anychk = __any(nchk > 0);
    if (anychk) {
      auto level = [&](int j, double& d1, double& d2) {  // j is a compile-time constant at every call
        double qo[NE], vo[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          if (j == 0) { qo[e] = e < nev ? own[G::OFF_COLD + e * 64] : 0.0; vo[e] = e < nev ? own[G::OFF_COLD + G::REG + e * 64] : 0.0; }
          else if (j == 1) { qo[e] = e < nev ? own[bp_ + e * 64] : 0.0; vo[e] = VL[e]; }
          else if (j < NFAST) {
            qo[e] = e < nev ? own[G::OFF_LC + (j - 2) * 2 * G::REG + e * 64] : 0.0;
            vo[e] = e < nev ? own[G::OFF_LC + ((j - 2) * 2 + 1) * G::REG + e * 64] : 0.0;
          } else if (j == NFAST) { qo[e] = pq0[e]; vo[e] = pv0[e]; }
          else { qo[e] = pq1[e]; vo[e] = pv1[e]; }
        }
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const double dqe = q[e] - qo[e];
          a1 += v[e] * dqe;
          a2 += vo[e] * dqe;
        }
        d1 = a1; d2 = a2;
      };
      // four quantities per butterfly (put4): (lp, KE, level 0), then two levels each -- a pair only when some chain of
      // the group checks that deep at this leaf (the deeper a level, the rarer: the bookkeeper never reads the stale rows)
      double a1, a2, b1, b2;
      level(0, a1, a2);
      put4(Q_LP, Q_KE, Q_DOT0, Q_DOT0 + 1, pl, pk, a1, a2);
#pragma unroll
      for (int j = 1; j + 1 < NCK; j += 2) {
        if (j == 1 || __any(nchk > j)) {
          level(j, a1, a2); level(j + 1, b1, b2);
          put4(Q_DOT0 + 2 * j, Q_DOT0 + 2 * j + 1, Q_DOT0 + 2 * j + 2, Q_DOT0 + 2 * j + 3, a1, a2, b1, b2);
        }
      }
      if ((NCK & 1) == 0 && __any(nchk > NCK - 1)) {
        level(NCK - 1, a1, a2);
        put(Q_DOT0 + 2 * (NCK - 1), a1); put(Q_DOT0 + 2 * (NCK - 1) + 1, a2);
      }
    } else {
      put(Q_LP, pl); put(Q_KE, pk);
    }
    if (__any((op & OP_NEWITER) != 0)) { put(Q_KEINIT, keinit_part); put(Q_GEN, gen_part); }
    WN_FSTAMP(3);
    for (int j = NCK; __any(nchk > j); ++j) {  // still deeper levels (one leaf in 2**NCK): fetched here
      const bool need = nchk > j;
      double e1 = 0.0, e2 = 0.0;
      if (need) {
        const int sl = slot_of_leaf(n + 1 - (1 << j));
        const double* sq = gvec(sl * 2);
        const double* sv = gvec(sl * 2 + 1);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const double dqe = q[e] - (e < nev ? sq[e * 4] : 0.0);
          e1 += v[e] * dqe;
          e2 += (e < nev ? sv[e * 4] : 0.0) * dqe;
        }
      }
      put(Q_DOT0 + 2 * j, e1); put(Q_DOT0 + 2 * j + 1, e2);
    }
    WN_FSTAMP(7);
    // ---- statePush (:431,518): only sub-tree left endpoints are ever read back ----
    const bool even = step && i >= 1 && (n & 1) == 0;
    if (even) {
#pragma unroll
      for (int e = 0; e < NE; ++e) VL[e] = v[e];
    }
    const bool ep = even && i >= 2 && (n & 3) == 0;
    if (__any(ep)) {
      if (ep) {
#pragma unroll
        for (int j = 2; j < NFAST; ++j)
          if (j <= i && (n & ((1 << j) - 1)) == 0) {
            WN_EV(e) {
              own[G::OFF_LC + (j - 2) * 2 * G::REG + e * 64] = q[e];
              own[G::OFF_LC + ((j - 2) * 2 + 1) * G::REG + e * 64] = v[e];
            }
          }
        if (n == 0 ? (i >= NFAST) : (__builtin_ctz((unsigned)n) >= NFAST)) {
          const int sl = slot_of_leaf(n);
          double* sq = gvec(sl * 2);
          double* sv = gvec(sl * 2 + 1);
          WN_EV(e) { sq[e * 4] = q[e]; sv[e * 4] = v[e]; }
        }
      }
    }
    if (SVC) { if (t > 0) fast_service_uniforms<G>(prm, lds, t); pop = op; pn = c.n; phs = step ? c.hs : 0.0; }
    WN_FSTAMP(4);
    WN_LDS_BARRIER();  // B
    WN_FSTAMP(5);
#if defined(WN_STAMPS)
    fst_[6] += 1;
#endif
  }
  WN_FSTAMP_OUT(prm)
}

// ------------------------------------------------------------------------------------------------
// B wave
// ------------------------------------------------------------------------------------------------
struct FastDevIO {
  const Params* prm;
  int cl, lg;
  double lp, ke, keinit, gen;
  uint32_t fire;
  __device__ __forceinline__ double sum(int k) const { return k == Q_LP ? lp : k == Q_KE ? ke : k == Q_KEINIT ? keinit : gen; }
  __device__ __forceinline__ uint32_t firemask(int, int, int) const { return fire; }
  __device__ __forceinline__ void trace_step(int, int, int, double, double, double) const {}  // traced runs go to wn_k_fast4.hip / the one-stream kernel
  // the service wave's Philox blocks: entries side, side + 2 = the two next blocks of that stream when they were asked for
  uint32_t s_it[4], s_blk[4];
  double s_u0[4], s_u1[4];
  __device__ __forceinline__ bool uniforms(int side, int iter, uint32_t blk, double& ua, double& ub) const {
    const bool h0 = side ? (s_it[1] == (uint32_t)iter && s_blk[1] == blk) : (s_it[0] == (uint32_t)iter && s_blk[0] == blk);
    const bool h1 = side ? (s_it[3] == (uint32_t)iter && s_blk[3] == blk) : (s_it[2] == (uint32_t)iter && s_blk[2] == blk);
    ua = side ? (h0 ? s_u0[1] : s_u0[3]) : (h0 ? s_u0[0] : s_u0[2]);
    ub = side ? (h0 ? s_u1[1] : s_u1[3]) : (h0 ? s_u1[0] : s_u1[2]);
    return h0 || h1;
  }
  __device__ __forceinline__ void write_gen(size_t row, double v) const {
    if (lg == 0) prm->samples[((size_t)cl * (prm->numIter + 1) + row) * prm->dg + 1] = v;
  }
  // one diagnostics row = three 64-B lines; the four replicas of a chain write 16 B of each (CoopPolicy::diag_store)
  __device__ __forceinline__ void write_diag(size_t row, const double* d) const {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    double* o = prm->diag + ((size_t)cl * prm->numIter + row) * NDIAG;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d2_t v;
      v[0] = lg == 0 ? d[8 * k] : lg == 1 ? d[8 * k + 2] : lg == 2 ? d[8 * k + 4] : d[8 * k + 6];
      v[1] = lg == 0 ? d[8 * k + 1] : lg == 1 ? d[8 * k + 3] : lg == 2 ? d[8 * k + 5] : d[8 * k + 7];
      *reinterpret_cast<d2_t*>(o + 8 * k + 2 * lg) = v;
    }
  }
};

template <int KS, int TK, int NLC>
__device__ __forceinline__ void fast_b_wave(const Params& prm, double* lds, int* abort_flag) {
  using G = FastGeom<KS, TK, NLC>;
  const int lane = threadIdx.x & 63, cn = lane & 15, lg = lane >> 4;
  const int cl = blockIdx.x * 16 + cn;
  const bool inrange = cl < prm.C;
  double* part = lds + G::OFF_PART;
  Cmd* cmdbuf = reinterpret_cast<Cmd*>(lds + G::OFF_CMD);
  double* znext = prm.work + (size_t)blockIdx.x * G::group_doubles(prm.M) + (size_t)((prm.M + 1) * 2 + 2) * G::VECD;

  FastBook bk;
  bk.init(prm.chain0 + (uint32_t)cl, inrange, prm.numIter);
  FastDevIO io;
  io.prm = &prm; io.cl = cl; io.lg = lg;
  io.lp = io.ke = io.keinit = io.gen = 0.0; io.fire = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) { io.s_it[k] = 0u; io.s_blk[k] = 0xFFFFFFFFu; io.s_u0[k] = io.s_u1[k] = 0.0; }
  // tile slots without coordinates (slot 7 = this wave, and the tiles beyond the dimension, whose waves only keep the
  // barriers company) are never written by a T wave: their partials are zero for the whole run.  The T waves write
  // their first partials behind barrier A of tick 0.
  for (int k = lane; k < G::NQ * 16 * G::NW; k += 64) part[k] = 0.0;
  // commands, sums, requests and the service wave's uniform entries: LDS keeps what the PREVIOUS kernel on this CU left
  // there, and an entry {iteration, block, u0, u1} of an earlier launch of this very kernel is well-formed -- the
  // bookkeeper reads the entries of tick 0 (never written) at tick 1 and would take another run's uniforms for its own
  // whenever (iteration, block) happened to match (seen once in ~15 full test-suite runs, after a numIter = 1 launch)
  for (int k = G::OFF_CMD + lane; k < G::LDS_DOUBLES; k += 64) lds[k] = 0.0;
  // every chain needs at most numIter * (2^M - 1 + 2 idle ticks per doubling + start) ticks; twice that is an internal error
  const double cap = 256.0 + 2.0 * (double)prm.numIter * ((double)(1u << prm.M) + 4.0 * (double)prm.M + 16.0);

  WN_FSTAMP_DECL
  for (uint32_t t = 0;; ++t) {
    // ---- (before barrier A) momentum normals v = normal(size=d) (WALNUTS.py:236, wn_rng.h stream 1) of ONE chain's next
    // iteration per tick: lane L makes elements 2L, 2L+1 (one Philox block, one Box-Muller pair) ----
    {
      const unsigned long long want = __ballot(lg == 0 && bk.zwant != bk.zready && bk.zwant >= 1);
      if (want) {
        const int cs = __builtin_ctzll(want);  // lanes 0..15 are chains 0..15
        const int it = __shfl(bk.zwant, cs);
        const uint32_t ch = (uint32_t)__shfl((int)bk.chain, cs);
        {
          Philox4 p = philox4x32_10((uint32_t)lane, (uint32_t)it, ch, 1u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
          const uint64_t x0 = ((uint64_t)p.w[1] << 32) | p.w[0];
          const double u1 = ((double)(x0 >> 12) + 0.5) * 0x1.0p-52;
          const double u2 = u53(p.w[2], p.w[3]);
          const double r = sqrt(-2.0 * log(u1));
          const double th = 6.283185307179586476925286766559 * u2;
          typedef double d2_t __attribute__((ext_vector_type(2)));
          d2_t z;
          z[0] = (2 * lane < prm.D) ? r * cos(th) : 0.0;
          z[1] = (2 * lane + 1 < prm.D) ? r * sin(th) : 0.0;
          *reinterpret_cast<d2_t*>(znext + (size_t)cs * G::DPAD + 2 * lane) = z;
        }
        if (cn == cs) bk.zready = it;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // visible to the T waves behind the barrier
      }
    }
    // ---- still before barrier A (the T waves are busy with their commands and the first half step): the lp and KE
    // sums over the tile slots of tick t-1's partials (in tile order, as fast_service_sums), and the arithmetic of
    // retiring that leaf that needs nothing else (energy, exp, acceptance ratio: FastBook::retire_pre) ----
    if (t > 0) {
      typedef double d2r_t __attribute__((ext_vector_type(2)));
      const d2r_t* r = reinterpret_cast<const d2r_t*>(part + cn * 2);  // + quantity * 16 * NW + tile pair * 16
      const d2r_t a = r[0], b = r[16], c_ = r[32], d = r[48];
      const d2r_t* r2 = r + (Q_KE * 16 * G::NW) / 2;
      const d2r_t a2 = r2[0], b2 = r2[16], c2 = r2[32], d2 = r2[48];
      io.lp = ((a[0] + a[1]) + (b[0] + b[1])) + ((c_[0] + c_[1]) + (d[0] + d[1]));
      io.ke = ((a2[0] + a2[1]) + (b2[0] + b2[1])) + ((c2[0] + c2[1]) + (d2[0] + d2[1]));
      bk.retire_pre(prm, io);
    }
    WN_FSTAMP(0);
    WN_LDS_BARRIER();  // A
    WN_FSTAMP(1);
    if (t > 0) {  // the service wave's sums / verdicts of tick t-1 and its uniforms of tick t-1
      const double* red = lds + G::OFF_RED + cn * 6;
      io.keinit = red[2]; io.gen = red[3];
      const uint32_t* fw = reinterpret_cast<const uint32_t*>(red + 4);
      io.fire = (fw[0] | fw[1]) | (fw[2] | fw[3]);
      const double* sv = lds + G::OFF_SVC + ((t - 1u) & 1u) * 192 + cn * 12;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        io.s_it[k] = reinterpret_cast<const uint32_t*>(sv + 3 * k)[0]; io.s_blk[k] = reinterpret_cast<const uint32_t*>(sv + 3 * k)[1];
        io.s_u0[k] = sv[3 * k + 1]; io.s_u1[k] = sv[3 * k + 2];
      }
    }
    Cmd c = bk.tick_post(prm, io, t + 1u);
    WN_FSTAMP(2);
    bool done = __all(bk.finished());
    if ((double)t > cap) {  // cannot happen; never hang the GPU
      if (lane == 0) atomicExch(abort_flag, 1);
      done = true;
    }
    if (done) c.op |= OP_DONE;
    if (lg == 0) cmdbuf[((t + 1u) & 1u) * 16 + cn] = c;
    // what the service wave is to compute next tick: the next two Philox blocks of both uniform streams
    if (lg == 0) {
      uint32_t* rq = reinterpret_cast<uint32_t*>(lds + G::OFF_RQ + ((t + 1u) & 1u) * 32 + cn * 2);
      rq[0] = (uint32_t)bk.iter; rq[1] = bk.cblkI + 1u; rq[2] = bk.cblkR + 1u;
    }
    WN_FSTAMP(4);
    WN_LDS_BARRIER();  // B
    WN_FSTAMP(5);
#if defined(WN_STAMPS)
    fst_[6] += 1;
#endif
    if (done) break;
  }
  WN_FSTAMP_OUT(prm)
}

// a T wave whose tile lies beyond the dimension: it only keeps the barriers company (wave 3: and does the service duties)
template <int KS, int TK, int NLC, bool SVC>
__device__ __forceinline__ void fast_idle_wave(const Params& prm, double* lds) {
  using G = FastGeom<KS, TK, NLC>;
  const Cmd* cmdbuf = reinterpret_cast<const Cmd*>(lds + G::OFF_CMD);
  const int cn = threadIdx.x & 15;
  const bool inrange = (int)(blockIdx.x * 16 + cn) < prm.C;
  uint32_t pop = 0u, pn = 0u;
  double phs = 0.0;
  for (uint32_t t = 0;; ++t) {
    Cmd c;
    if (t == 0) { c.hs = 0.0; c.op = inrange ? (OP_NEWITER | OP_INIT) : 0u; c.n = 0u; }
    else c = cmdbuf[(t & 1u) * 16 + cn];
    if (__any((c.op & OP_DONE) != 0)) break;
    if (SVC && t > 0) fast_service_sums<G>(lds, pop, pn, phs);
    WN_LDS_BARRIER();  // A
    if (SVC) { if (t > 0) fast_service_uniforms<G>(prm, lds, t); pop = c.op; pn = c.n; phs = (c.op & OP_STEP) ? c.hs : 0.0; }
    WN_LDS_BARRIER();  // B
  }
}

template <int KS, int TK, int NLC>
__global__ __launch_bounds__(512) void wn_fast_kernel(Params prm, int* abort_flag) {
  __shared__ double lds[FastGeom<KS, TK, NLC>::LDS_DOUBLES];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform for the compiler too: no exec masking around MFMAs
  // tile of T wave w.  Waves w and w+4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically: speed only),
  // so the B wave (7) sits beside wave 3, which gets the last tile -- for D = 100 the one with only 4 rows
  if (wave < 7) {
    const int tile = wave < 3 ? wave : (wave == 3 ? 6 : wave - 1);
    constexpr int NEV_LAST = (KS % 4) ? (KS % 4) : 4;  // registers of the tile that holds the last k-steps
    const int nev = KS - 4 * tile;
    if (wave == 3) {  // tile 6, and the service duties (fast_service_*)
      constexpr int NEV3 = KS - 24 >= 4 ? 4 : (KS - 24 > 0 ? KS - 24 : 0);
      if (NEV3 > 0) fast_t_wave<KS, TK, NLC, (NEV3 > 0 ? NEV3 : 1), true>(prm, lds, tile);
      else fast_idle_wave<KS, TK, NLC, true>(prm, lds);
    } else if (nev >= 4) fast_t_wave<KS, TK, NLC, 4, false>(prm, lds, tile);
    else if (nev > 0) fast_t_wave<KS, TK, NLC, NEV_LAST, false>(prm, lds, tile);
    else fast_idle_wave<KS, TK, NLC, false>(prm, lds);
  } else {
    fast_b_wave<KS, TK, NLC>(prm, lds, abort_flag);
  }
}

// can the role-split kernel run this launch?  fixedLeapFrog sampling without trace / warm-up / orbit statistics
bool fast_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm) {
  if (kc.layout != 1 || adapt != 0 || warm || prm.mode != MODE_SAMPLE) return false;
  if (prm.trace_n || prm.omin || prm.M > 11) return false;
  if (prm.D > 112) return false;  // seven 16-row tiles
  if (kc.p1 == CK_DENSE) return kc.p0 == 25 || kc.p0 == 16;
  return kc.p0 == 16 || kc.p0 == 32;
}

hipError_t launch_fast(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st) {
  dim3 grid((prm.C + 15) / 16), block(512);
#define WN_FAST_GO(KS, TK) \
  { hipLaunchKernelGGL((wn_fast_kernel<KS, TK, (KS <= 25 ? 3 : 2)>), grid, block, 0, st, prm, abort_flag); return hipGetLastError(); }
  if (kc.p1 == CK_DENSE) {
    if (kc.p0 == 16) WN_FAST_GO(16, CK_DENSE)
    if (kc.p0 == 25) WN_FAST_GO(25, CK_DENSE)
  } else {
    if (kc.p0 == 16) WN_FAST_GO(16, CK_DIAG)
    if (kc.p0 == 32) WN_FAST_GO(28, CK_DIAG)  // D <= 112
  }
#undef WN_FAST_GO
  return hipErrorInvalidValue;
}

}  // namespace wn
