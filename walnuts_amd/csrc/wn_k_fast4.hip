// wn_k_fast4.hip -- the headline path, fixedLeapFrog on a dense Gaussian with D <= 100 (BASELINE configs[1]): one T wave owns
// ALL coordinates of a group of FOUR chains (a "pair": the name is from the first form, one bookkeeper wave per T wave); the
// chains' scalars (wn_fast.h) are kept by a RETIRE wave and an ISSUE wave per QUAD = two T waves.  A workgroup holds 1, 2 or 4
// T waves (16 chains per CU at 4096 chains, 4 at 1024): [R I] T, [R I] T T, or [R I R I] T T T T -- every SIMD then runs one T wave
// and one half of a bookkeeper.
//
// Why four chains per wave.  wn_k_fast.hip spreads a 16-chain group over seven T waves (one 16-row tile each): every
// tick they exchange the positions through LDS, sum their partial dot products through LDS and a service wave, and
// meet at two barriers -- ~4.5 k of a tick's 7.7 k cycles.  v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks,
// 16 cycles: the same flop per cycle as 16x16x4, tools/mfma_f64_4x4_probe.hip) makes a 4-chain group small enough
// for ONE wave: 7 doubles per lane and vector.  Nothing crosses waves any more: no position exchange, every dot
// product is complete inside the wave, no barrier in the tick, and the -A fragments stream from one LDS image shared by
// the workgroup's T waves.
//
// Lane l = 16 r + 4 b + x.  Measured operand layout of the instruction (probe): A[block b][i][k] sits in lane
// 16 k + 4 b + i, B[b][k][j] in lane 16 k + 4 b + j, D[b][i][j] in lane 16 i + 4 b + j.  Register s of a vector holds
// coordinate 16 s + 4 b + r of chain x -- the D layout of the 16-row set s when block b computes rows 16 s + 4 b ..
// + 3.  So the ROWS of a set are split over the blocks and every block needs the same four coordinates of Q as its B
// operand: k-step kappa (coordinates 4 kappa .. 4 kappa + 3) is block kappa & 3 of register kappa >> 2, read back
// block-replicated from a copy of the position in LDS.  G comes out complete and where Q lives: nothing is reduced across
// lanes on the dependent path.  (The first form of this kernel split K over the blocks instead: no replication, but a
// 15-instruction reduce-scatter with three FP64 additions behind every 24 MFMAs -- measured 3.5 k cycles for the 157 MFMAs of
// a tick, whose pipe time is 2.5 k; FP64 vector instructions wait for the FP64 matrix pipe.)  For D = 100 rows 96..99 would be
// a seventh set with 12 of 16 rows padding: they are done K-split (six MFMAs on the registers as they are, one block sum, one
// MFMA for columns 96..99): 157 MFMAs per tick instead of 175.
//
// One tick (the protocol and commands are wn_fast.h's, THREE-tick lag: RetireBook / IssueBook = FastBookT<3> in two streams):
//   T wave (tick t)                          R wave (tick t)                          I wave (tick t)
//   poll command record C[t]; pick /         wait for T's tick t-2; read C[t-2]       momentum normals, uniform window;
//   iteration start / end swap; first        and the sums P[t-2]; U-turn verdicts;    wait for R's tick t; read its messages;
//   half step; G = -A Q; second half step;   RetireBook::tick; message to I           IssueBook::tick; publish C[t+1]; touch the
//   sums P[t] of lp, KE and the U-turn                                                lines T will fetch in tick t+1
//   dots due at this leaf; endpoint stores
// Hand-off through counters in LDS and the tick tag inside the command record; sums / messages triple-, commands
// quadruple-buffered.  The bookkeeper has two vector ticks for one of its own, and the T waves run at a higher priority: on an
// ordinary leaf the T wave does not wait (with the two-tick lag of wn_k_fast.hip every slow tick of either side stalled the
// other: T waited 21 % of its time, the bookkeeper 10 %).
#include "wn_launch.h"  // hip_runtime first

#include <type_traits>

#include "wn_fast.h"

namespace wn {
using namespace fast;

template <int NS_, bool TAIL_, bool DENSE_ = true>
struct Fast4Geom {
  static constexpr bool DENSE = DENSE_;          // dense precision matrix (MFMA gradient); false: diagonal / standard Gaussian, g = -a .* q
  static constexpr int NS = NS_;                 // registers per lane and vector: coordinate 16 s + 4 b + r
  static constexpr bool TAIL = TAIL_;            // the last k-step holds <= 4 coordinates (block 0 only): tail MFMA
  static constexpr int KF = TAIL ? NS - 1 : NS;  // full k-steps of the main loop
  static constexpr int VEC = NS * 64;            // one vector of a 4-chain group, [s][lane]
  static constexpr int DPAD = 128;               // momentum normals per chain
  static constexpr int MP = 11;                  // largest M
  static constexpr int NB = 8;                   // batches of four sums: (lp, KE, level 0), levels (1,2) .. (11,-), (keinit, gen)
  static constexpr int NFAST = 3;                // check levels 0 (passive end), 1 (registers), 2 (LDS) are on chip
  // -A fragments (64 doubles each).  Main: k-step kappa (coordinates 4 kappa .. 4 kappa + 3) x 16-row set S, usage order
  // kappa-major; with TAIL the last four rows (row group 4 (NS-1)) are done K-split: NS-1 fragments + one for the tail columns
  static constexpr int NSR = TAIL ? NS - 1 : NS;                  // 16-row sets computed with replicated operands
  static constexpr int NK = TAIL ? 4 * (NS - 1) + 1 : 4 * NS;     // k-steps of four coordinates
  static constexpr int NMAIN = NK * NSR;
  static constexpr int NFR = DENSE ? NMAIN + (TAIL ? NS : 0) : 0;
  static_assert(!DENSE || NSR % 2 == 0, "two fragments per 16-byte LDS read");
  // LDS, in doubles: the fragment image, then one region per pair
  static constexpr int OFF_A = 0;
  static constexpr int P_LC = 0;                        // level-2 left endpoint: q, v
  static constexpr int P_STAGE = P_LC + 2 * VEC;        // this tick's position, from where the MFMAs' B operands are read block-replicated
  static constexpr int P_PK = P_STAGE + VEC;            // the position of tick t-2, kept one more tick: a pick commanded at tick t+1 copies it
  static constexpr int P_SUMS = P_PK + VEC;             // 3 x [batch][block b][chain x]  (tick t in slot t % 3)
  static constexpr int P_CMD = P_SUMS + 3 * NB * 16;    // 4 x 4 commands of 2 doubles    (tick t in slot t & 3)
  static constexpr int P_MSG = P_CMD + 32;              // 3 x 4 messages R -> I of 2 doubles (bookkeeper tick t in slot t % 3)
  static constexpr int P_USTAT = P_MSG + 24;            // uniform blocks 0..7 of the running iteration: [chain][8] x {tag, u0, u1}
  static constexpr int P_URING = P_USTAT + 4 * 8 * 3;   // sliding window of 16 blocks: [chain][blk & 15] x {tag, u0, u1}
  // hand-off words (32 bit): [0] ticks the T wave has finished, [2] commands the I wave has issued, [4] ticks the R wave has
  // finished, [6] the run is over (the last two in the FIRST pair of a quad only)
  static constexpr int P_SEQ = P_URING + 4 * 16 * 3;
  static constexpr int P_H0 = P_SEQ + 8;               // the running iteration's initial Hamiltonian per chain (R wave -> I wave)
  static constexpr int P_HC = P_H0 + 4;                // warm-up: the chain's adapted step size H (R wave -> I wave, behind an RM_END message)
  static constexpr int PAIR_DOUBLES = P_HC + 4;
  // "halves" (one 4-chain group on TWO T waves, configs[1]'s 1024 chains: fast4_halves): a second position stage (the two waves of a group
  // are a tick apart at most), the second wave's sums, two hand-off words
  static constexpr int P_STAGE2 = PAIR_DOUBLES;
  static constexpr int P_SUMS2 = P_STAGE2 + VEC;
  static constexpr int P_XSEQ = P_SUMS2 + 3 * NB * 16;
  static constexpr int PAIR_DOUBLES_H = P_XSEQ + 2;
  static constexpr int NSPLIT = NSR / 2;  // register sets 0 .. NSPLIT-1 belong to the first wave, the rest (and the tail rows) to the second
  static constexpr int OFF_PAIR = NFR * 64;
  static_assert(((size_t)OFF_PAIR + 4 * (size_t)PAIR_DOUBLES) * sizeof(double) <= 160 * 1024, "four pairs and the fragment image in one CU's LDS");
  __host__ __device__ static constexpr size_t lds_bytes(int pairs) { return ((size_t)OFF_PAIR + (size_t)pairs * PAIR_DOUBLES) * sizeof(double); }
  __host__ __device__ static constexpr size_t lds_bytes_halves() { return ((size_t)OFF_PAIR + (size_t)PAIR_DOUBLES_H) * sizeof(double); }
  // workspace of one 4-chain group, in doubles: (M+1) slots x (q, v), 2 proposals, the passive orbit end (q, v), momentum normals [chain][DPAD]
  __host__ __device__ static constexpr size_t group_doubles(int M) { return (size_t)((M + 1) * 2 + 4) * VEC + (size_t)4 * DPAD; }
};

// diagnostic builds (tools/build_variant.sh stamps -DWN_STAMPS): cycles per segment of the tick, summed per wave
#if defined(WN_STAMPS)
#define WN4_STAMP(k)                                                                         \
  do {                                                                                       \
    unsigned long long _t;                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
    fst_[k] += _t - fst_last_;                                                               \
    fst_last_ = _t;                                                                          \
  } while (0)
#define WN4_STAMP_DECL unsigned long long fst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fst_last_ = __builtin_readcyclecounter();
#define WN4_STAMP_OUT(prm)                                                                             \
  if ((prm).stamps && (threadIdx.x & 63) == 0) {                                                       \
    unsigned long long* o_ = (prm).stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16;        \
    for (int k = 0; k < 8; ++k) o_[k] = fst_[k];                                                       \
  }
#else
#define WN4_STAMP(k) do {} while (0)
#define WN4_STAMP_DECL
#define WN4_STAMP_OUT(prm)
#endif

// Control-flow switches of the tick (A/B builds: tools/variant_fast4_many.sh name:-DWN4_OPT=<mask> ...).  Round 5 measured each one alone
// on the bench workload (gpurun_out/r05/ab2.txt, ab3.txt; 1.378e9 steps/s with none of them): the T wave's branches are NOT what its tick
// is made of -- ten fewer branches and exec-mask regions per tick change nothing (1: 1.373e9, 4: 1.377e9, 8: 1.375e9) -- while 14-28 more
// vector-memory instructions per odd tick cost 4 % (the levels 3 / 4 prefetched for every chain of the group behind one vote instead of
// per chain under exec masks: 1.322e9, removed).  All four below together: 1.381e9.
#if !defined(WN4_OPT)
#define WN4_OPT (1 + 4 + 8 + 16)
#endif
constexpr bool OPT_POLL = (WN4_OPT & 1) != 0;    // the spin loop out of line, tick 0 no special case
constexpr bool OPT_L12 = (WN4_OPT & 4) != 0;     // levels 1 / 2 on every odd tick without votes
constexpr bool OPT_EP = (WN4_OPT & 8) != 0;      // endpoint pushes as predicated stores (no vote, no branch)
// (measured and dropped, round 5, gpurun_out/r05/ab13.txt: an even tick's endpoint pushes issued in the shadow of the NEXT tick's MFMAs -- the
//  state pushed is that tick's (qc, vi), which it only reads -- : -0.7 %; the level-2 endpoint read from LDS in the shadow of the tick's last MFMAs:
//  -19 %, its fourteen registers spill.  The MFMA shadow is not free: with the bookkeepers above the T waves it is where THEY run.)
// (measured and dropped, gpurun_out/r05/ab14.txt: workspace vectors with two registers of a lane side by side in memory -- 16-byte accesses, 54 + 85
//  dwordx4 in the code object, half the vector-memory instructions: +0.4 % (1.400e9 -> 1.406e9) for 4 VGPR spills and 20 B of scratch in the headline
//  instantiation; even switched off its address helper cost the default build 2 spills.  Removed.)
constexpr bool OPT_PRE = (WN4_OPT & 64) != 0;    // T wave (whole-group form too): the next tick's command record requested behind the MFMA phase
constexpr bool OPT_IFAST = (WN4_OPT & 32) != 0;  // I wave: the ordinary tick as straight-line code when every chain of the wave qualifies
constexpr bool OPT_SLIDE = (WN4_OPT & 16) != 0;  // I wave: a window refill slides the windows of ALL its chains (the Philox passes are paid per wave, not per chain)

// sensitivity diagnostic (tools/variant_fast4_many.sh padT:-DWN4_PAD_T=40 ...): N dependent vector instructions more per tick in one
// kind of wave -- whose pass is the tick made of?
template <int N>
__device__ __forceinline__ void wn4_pad(double& sink) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(sink));
}
#if !defined(WN4_PAD_T)
#define WN4_PAD_T 0
#endif
#if !defined(WN4_PAD_R)
#define WN4_PAD_R 0
#endif
#if !defined(WN4_PAD_I)
#define WN4_PAD_I 0
#endif

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// DPP move of a double: lanes whose bank (= block b = (lane >> 2) & 3) is in BANKS take src moved by CTL inside their row of
// 16 lanes, the others keep old
template <int CTL, int BANKS>
__device__ __forceinline__ double dpp_sel(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTL, 0xF, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTL, 0xF, BANKS, false);
  return __hiloint2double(hi, lo);
}
enum { DPP_SHL4 = 0x104, DPP_SHL8 = 0x108, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118, DPP_ROR8 = 0x128 };  // row_shl:n lane l <- l+n; row_shr:n lane l <- l-n

// block b' ends with x_{b'} summed over the four blocks: (x[0] + x[2]) + (x[1] + x[3])
__device__ __forceinline__ double blocks_reduce_scatter(double x0, double x1, double x2, double x3) {
  // distance 8: blocks 0,1 keep x0 / x1, blocks 2,3 keep x2 / x3
  const double u = dpp_sel<DPP_ROR8, 0xC>(x0, x2) + dpp_sel<DPP_ROR8, 0x3>(x2, x0);  // [x0(0+2), x0(1+3), x2(0+2), x2(1+3)]
  const double w = dpp_sel<DPP_ROR8, 0xC>(x1, x3) + dpp_sel<DPP_ROR8, 0x3>(x3, x1);  // the same of x1 / x3
  // distance 4: even blocks keep u, odd blocks keep w
  return dpp_sel<DPP_SHR4, 0xA>(u, w) + dpp_sel<DPP_SHL4, 0x5>(w, u);  // [x0, x1, x2, x3] summed
}
__device__ __forceinline__ double blocks_sum(double x) {  // every block ends with the sum over the four blocks
  x = x + dpp_sel<DPP_ROR8, 0xF>(x, x);
  return dpp_sel<DPP_SHR4, 0xA>(x, x) + dpp_sel<DPP_SHL4, 0x5>(x, x);
}
__device__ __forceinline__ double block0_bcast(double x) {  // block 0 of every row to the row's other blocks
  x = dpp_sel<DPP_SHR4, 0x2>(x, x);
  return dpp_sel<DPP_SHR8, 0xC>(x, x);
}
// x[l] + x[l^16] + x[l^32] + x[l^48] in every lane (the sum over r)
__device__ __forceinline__ double rows_sum(double x) {
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
// Predicated stores WITHOUT a branch.  `if (lane predicate) { stores }` compiles to s_and_saveexec + s_cbranch_execz + the stores +
// s_or exec, and the compiler always keeps the skip branch around LDS / memory instructions.  A T wave pays 25-55 cycles for every
// branch, taken or not (tools/branch_cost.hip), and the endpoint pushes / picks below are due for SOME chain of the group on most
// ticks: here exec is narrowed around the stores inside one asm block -- no vote, no branch; lanes outside `mask` issue nothing.
// (The compiler does not count these stores in its s_waitcnt bookkeeping.  LDS operations and, on gfx9, vector memory operations
// complete in issue order, so the extra outstanding operations can only make a later s_waitcnt N wait longer, never shorter.)
// q[s] -> base[s * 64], v[s] -> base[(NSET + s) * 64]  (doubles; the level-2 endpoint in LDS)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ uint32_t lds_addr(const double* p) { return (uint32_t)(uintptr_t)(const lds_f64*)p; }
template <int NSET, int A, int B>
__device__ __forceinline__ void lds_store_qv_masked(uint64_t mask, uint32_t addr, const double (&q)[NSET], const double (&v)[NSET]) {
  constexpr int N = B - A;
  static_assert(N >= 1 && N <= 4, "one asm block stores up to four register sets");
  uint64_t sv;
  if constexpr (N == 1)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2st64_b64 %2, %3, %4 offset0:%5 offset1:%6\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(addr), "v"(q[A]), "v"(v[A]), "n"(A), "n"(NSET + A) : "memory", "scc");
  else if constexpr (N == 2)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2st64_b64 %2, %3, %4 offset0:%7 offset1:%8\n\tds_write2st64_b64 %2, %5, %6 offset0:%9 offset1:%10\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(addr), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "n"(A), "n"(NSET + A), "n"(A + 1), "n"(NSET + A + 1) : "memory", "scc");
  else if constexpr (N == 3)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2st64_b64 %2, %3, %4 offset0:%9 offset1:%10\n\tds_write2st64_b64 %2, %5, %6 offset0:%11 offset1:%12\n\t"
                 "ds_write2st64_b64 %2, %7, %8 offset0:%13 offset1:%14\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(addr), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "v"(q[A + 2]), "v"(v[A + 2]),
                   "n"(A), "n"(NSET + A), "n"(A + 1), "n"(NSET + A + 1), "n"(A + 2), "n"(NSET + A + 2) : "memory", "scc");
  else
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2st64_b64 %2, %3, %4 offset0:%11 offset1:%12\n\tds_write2st64_b64 %2, %5, %6 offset0:%13 offset1:%14\n\t"
                 "ds_write2st64_b64 %2, %7, %8 offset0:%15 offset1:%16\n\tds_write2st64_b64 %2, %9, %10 offset0:%17 offset1:%18\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(addr), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "v"(q[A + 2]), "v"(v[A + 2]), "v"(q[A + 3]), "v"(v[A + 3]),
                   "n"(A), "n"(NSET + A), "n"(A + 1), "n"(NSET + A + 1), "n"(A + 2), "n"(NSET + A + 2), "n"(A + 3), "n"(NSET + A + 3) : "memory", "scc");
}
template <int NSET, int S0, int S1>
__device__ __forceinline__ void lds_store_qv_masked_all(uint64_t mask, const double* base, const double (&q)[NSET], const double (&v)[NSET]) {
  const uint32_t addr = lds_addr(base);
  constexpr int MID = S0 + ((S1 - S0) > 4 ? 4 : (S1 - S0));
  lds_store_qv_masked<NSET, S0, MID>(mask, addr, q, v);
  if constexpr (MID < S1) lds_store_qv_masked<NSET, MID, S1>(mask, addr, q, v);
}
// the same to memory: q[s] -> pq[s * 16], v[s] -> pv[s * 16]  (doubles; a chain-major vector of the workspace)
template <int NSET, int A, int B>
__device__ __forceinline__ void glb_store_qv_masked(uint64_t mask, double* pq, double* pv, const double (&q)[NSET], const double (&v)[NSET]) {
  constexpr int N = B - A;
  static_assert(N >= 1 && N <= 4, "one asm block stores up to four register sets");
  uint64_t sv;
  if constexpr (N == 1)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %4, off offset:%6\n\tglobal_store_dwordx2 %3, %5, off offset:%6\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(pq), "v"(pv), "v"(q[A]), "v"(v[A]), "n"(A * 128) : "memory", "scc");
  else if constexpr (N == 2)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %4, off offset:%8\n\tglobal_store_dwordx2 %3, %5, off offset:%8\n\t"
                 "global_store_dwordx2 %2, %6, off offset:%9\n\tglobal_store_dwordx2 %3, %7, off offset:%9\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(pq), "v"(pv), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "n"(A * 128), "n"(A * 128 + 128) : "memory", "scc");
  else if constexpr (N == 3)
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %4, off offset:%10\n\tglobal_store_dwordx2 %3, %5, off offset:%10\n\t"
                 "global_store_dwordx2 %2, %6, off offset:%11\n\tglobal_store_dwordx2 %3, %7, off offset:%11\n\t"
                 "global_store_dwordx2 %2, %8, off offset:%12\n\tglobal_store_dwordx2 %3, %9, off offset:%12\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(pq), "v"(pv), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "v"(q[A + 2]), "v"(v[A + 2]),
                   "n"(A * 128), "n"(A * 128 + 128), "n"(A * 128 + 256) : "memory", "scc");
  else
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %4, off offset:%12\n\tglobal_store_dwordx2 %3, %5, off offset:%12\n\t"
                 "global_store_dwordx2 %2, %6, off offset:%13\n\tglobal_store_dwordx2 %3, %7, off offset:%13\n\t"
                 "global_store_dwordx2 %2, %8, off offset:%14\n\tglobal_store_dwordx2 %3, %9, off offset:%14\n\t"
                 "global_store_dwordx2 %2, %10, off offset:%15\n\tglobal_store_dwordx2 %3, %11, off offset:%15\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(pq), "v"(pv), "v"(q[A]), "v"(v[A]), "v"(q[A + 1]), "v"(v[A + 1]), "v"(q[A + 2]), "v"(v[A + 2]), "v"(q[A + 3]), "v"(v[A + 3]),
                   "n"(A * 128), "n"(A * 128 + 128), "n"(A * 128 + 256), "n"(A * 128 + 384) : "memory", "scc");
}
template <int NSET, int S0, int S1>
__device__ __forceinline__ void glb_store_qv_masked_all(uint64_t mask, double* pq, double* pv, const double (&q)[NSET], const double (&v)[NSET]) {
  constexpr int MID = S0 + ((S1 - S0) > 4 ? 4 : (S1 - S0));
  glb_store_qv_masked<NSET, S0, MID>(mask, pq, pv, q, v);
  if constexpr (MID < S1) glb_store_qv_masked<NSET, MID, S1>(mask, pq, pv, q, v);
}

// ------------------------------------------------------------------------------------------------
// T wave
// ------------------------------------------------------------------------------------------------
// the pair's hand-off: a word counts up in LDS; the partner polls it (LDS operations of one wave execute in order: data first, then
// the word).  Bounded: a protocol error must not hang the GPU.
// The hand-off words are read and written through LDS-address-space pointers: through a generic `volatile uint32_t*` the compiler
// emits FLAT loads with system scope, and every poll then waits for vmcnt(0) -- i.e. for every global store the wave still has in
// flight (the endpoint stores of the tick before: a full store latency at the top of the next tick).
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64;
typedef __attribute__((address_space(3))) volatile double lds_vf64;
__device__ __forceinline__ lds_vu32* lds_words(const double* p) { return (lds_vu32*)p; }
// A command record in LDS = {hs, op | n << 32}; bits 16..31 of op carry the tick it is for (CMD_TAG), so the T wave polls the record
// itself -- one LDS round trip per tick instead of a counter and then the record.  The writer stores hs first, the reader reads it second
// (LDS operations of a wave execute in order).
enum { CMD_TAG_SHIFT = 16 };
__device__ __forceinline__ void cmd_publish(double* rec, const Cmd& c, uint32_t tick) {
  *(lds_vf64*)rec = c.hs;
  *(lds_vu64*)(rec + 1) = (uint64_t)(c.op | ((tick & 0xFFFFu) << CMD_TAG_SHIFT)) | ((uint64_t)c.n << 32);
}
__device__ __forceinline__ void cmd_read(const double* rec, uint64_t& w, double& hs) {
  w = *(const lds_vu64*)(rec + 1);
  hs = *(const lds_vf64*)rec;
}
__device__ __forceinline__ bool cmd_is_for(uint64_t w, uint32_t tick) { return (((uint32_t)w >> CMD_TAG_SHIFT) & 0xFFFFu) == (tick & 0xFFFFu); }
// HALF: 0 = the wave owns the whole group (every register set); 1 / 2 = the group's two T waves of the halves mode: register sets
// [0, NSPLIT) / [NSPLIT, NS).  Each computes the gradient rows of its own sets -- the B operands (ALL coordinates) come from the LDS
// stage, which both fill: one LDS hand-off per tick -- and the partial sums of its own coordinates; the R wave adds the two.
template <class G, bool TRACE, int HALF = 0>
__device__ __forceinline__ void fast4_t_wave(const Params& prm, const double* lds, double* pr, const int grp, int* abort_flag) {
  constexpr int NS = G::NS, NFAST = G::NFAST;
  constexpr int S0 = HALF == 2 ? G::NSPLIT : 0, S1 = HALF == 1 ? G::NSPLIT : NS;          // own register sets
  constexpr int R0 = HALF == 2 ? G::NSPLIT : 0, R1 = HALF == 1 ? G::NSPLIT : G::NSR;      // own 16-row sets of the MFMA phase
  constexpr bool OWN_TAIL = G::TAIL && HALF != 1;
  static_assert(HALF == 0 || G::DENSE, "halves: dense targets");
  const int lane = threadIdx.x & 63, x = lane & 3, b = (lane >> 2) & 3, r = lane >> 4;
  const int cl = grp * 4 + x;
  const bool inrange = cl < prm.C;
  const int D = prm.D;
  auto coord = [&](int s) { return 16 * s + 4 * b + r; };
  double* const wbase = prm.work + (size_t)grp * G::group_doubles(prm.M);
  // vectors in memory are chain-major: [chain x][register s][16 r-b lanes] -- a store by ONE chain of the group (an endpoint push, a pick)
  // then writes whole 128-byte lines (in [s][lane] order it scattered 8-byte pieces at a 32-byte stride over four times as many lines:
  // 178 GB of L2 write-backs per launch measured, profiles/r03_traffic_first.json)
  auto gvec = [&](int vec) { return wbase + (size_t)vec * G::VEC + x * (NS * 16) + 4 * r + b; };  // + s * 16
  const int vprop = (prm.M + 1) * 2, vcold = vprop + 2;
  const double* const znext = wbase + (size_t)((prm.M + 1) * 2 + 4) * G::VEC + (size_t)x * G::DPAD;
  double* const lc = pr + G::P_LC + lane;
  double* const stage0 = pr + G::P_STAGE;
  double* const pk = pr + G::P_PK + lane;
  double* const sums = pr + (HALF == 2 ? G::P_SUMS2 : G::P_SUMS);
  lds_vu32* const seq = lds_words(pr + G::P_SEQ) + (HALF == 2 ? 1 : 0);  // [0]: ticks finished by T ([1]: by the group's second T wave)
  lds_vu32* const xseq = lds_words(pr + G::P_XSEQ);  // halves: [w] = ticks whose position wave w has staged

  // P(t) = the position after the drift of tick t lives in X[t & 1]: at the start of tick t, X[(t-1) & 1] is the current position and
  // X[t & 1] still holds P(t-2); the drift overwrites it with P(t), and P(t-1) stays for the level-1 check behind the gradient.  No
  // register copies.  A multinomial pick commanded at tick t copies P(t-3) (wn_fast.h, LAG = 3): P(t-2) goes to LDS at the top of every
  // tick (`pk`) and is read from there one tick later.
  // ... and the velocity alternates the same way: the first half step of tick t writes V[t & 1] from V[(t-1) & 1], which therefore still
  // holds the velocity the previous tick ended with -- the left endpoint's of a leaf pair at its second leaf (level-1 check).  No copy.
  double X0[NS], X1[NS], V0[NS], V1[NS], g[NS];
  double pq0[NS], pv0[NS], pq1[NS], pv1[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { X0[s] = X1[s] = V0[s] = V1[s] = g[s] = 0.0; pq0[s] = pv0[s] = pq1[s] = pv1[s] = 0.0; }
  int ntr = 0;  // trace rows written (TRACE)
  double adiag[NS];  // (diagonal / standard Gaussian) the precisions of this lane's coordinates; tpar is all ones for the standard Gaussian
#pragma unroll
  for (int s = 0; s < NS; ++s) adiag[s] = (!G::DENSE && coord(s) < D) ? prm.tpar[coord(s)] : 0.0;

  // Halves: a half wave is alone on its SIMD (all 512 registers of a lane are its own) and needs 75 + 7 fragments: they are read from
  // the image ONCE and stay in registers -- no fragment traffic in the tick at all (the whole-group wave needs 157 and re-reads them
  // from LDS every tick, three k-steps ahead of their use)
  constexpr bool RESIDENT = G::DENSE && HALF != 0;
  constexpr int NOWN = RESIDENT ? R1 - R0 : 1, NKR = RESIDENT ? G::NK : 1;
  double frag[NKR][NOWN];
  if constexpr (RESIDENT) {
    const double* const img0 = lds + G::OFF_A + 2 * lane;
#pragma unroll
    for (int k = 0; k < G::NK; ++k)
#pragma unroll
      for (int S = R0; S < R1; ++S) frag[k][S - R0] = img0[(k * (G::NSR / 2) + (S >> 1)) * 128 + (S & 1)];
  }
  double fxr[NS > 1 ? NS - 1 : 1], ftailr = 0.0;  // ... and the tail rows' (K split over the register sets + columns 16 (NS-1) ..)
  if constexpr (RESIDENT && OWN_TAIL) {
    const double* const img0 = lds + G::OFF_A + 2 * lane;
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) fxr[j] = img0[G::NMAIN * 64 + (j >> 1) * 128 + (j & 1)];
    ftailr = lds[G::OFF_A + (G::NMAIN + NS - 1) * 64 + lane];
  }
  uint64_t cw_pre = 0;   // the command record of the NEXT tick, read early (tag 0 never matches a tick >= 1 that has not been issued)
  double chs_pre = 0.0;
  // one batch of four per-chain sums: lanes r = 0 hold [block b'][chain x] = 16 consecutive doubles
  auto put4 = [&](double* dst, double a0, double a1, double a2, double a3) {
    // the sum over the four rows r is one MFMA with an all-ones A operand: D[b][i][j] = sum_k B[b][k][j], i.e. lane (i, b, x) gets the
    // sum over r of lanes (r, b, x) -- in every row, which is what the four permlane swaps and two additions of rows_sum() produce
    const double z = mfma4(1.0, blocks_reduce_scatter(a0, a1, a2, a3), 0.0);
    dst[lane & 15] = z;  // (every row holds the same sums: all lanes store -- an exec-mask region costs a wave ~37 cycles, tools/branch_cost.hip)
  };
  WN4_STAMP_DECL

  // One tick.  PAR = t & 1 is a compile-time constant (the loop below is unrolled by two): the bookkeeper starts leaf pairs on even
  // ticks, so a stepping chain's leaf index has the tick's parity -- sub-tree checks (odd leaves), their prefetches, iteration starts
  // (odd ticks) and endpoint stores (even leaves) each live in one of the two bodies only.  Returns false when the run is over.
  auto tick = [&](auto par_tag, double (&qc)[NS], double (&qn)[NS], double (&vi)[NS], double (&v)[NS], const uint32_t t, const int t3) -> bool {  // t3 = t % 3
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int NSR = G::NSR, NK = G::NK, NP = NSR / 2;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const double* const img = lds + G::OFF_A + 2 * lane;
    // the command record is asked for first, the fragments of the first k-steps (which do not depend on it) right behind: LDS returns in
    // order, so an ordinary tick has its command after one round trip and the fragments arrive while it is decoded
    const double* const rec = pr + G::P_CMD + ((t & 3u) * 4 + x) * 2;
    // (the record of this tick was asked for behind the MFMA phase of the tick before -- the issue wave is a tick ahead: it is
    // usually there, and an ordinary tick starts without an LDS round trip; the poll below catches the other case)
    // (halves only: with a second wave on the SIMD the round trip is hidden and the extra read costs 1.5 %)
    uint64_t cw = cw_pre;
    double chs = chs_pre;
    if (HALF == 0 && !OPT_PRE && (OPT_POLL || t != 0)) cmd_read(rec, cw, chs);  // (tick 0: the record the kernel's prologue wrote -- tag 0 -- so tick 0 is no special case)
    // fragments in flight: 3 k-steps ahead (six MFMAs = 96 cycles each: an LDS round trip is covered); a half wave issues three MFMAs
    // per k-step and has the register file of a SIMD to itself: 6 ahead, and only the pairs of its own row sets
    constexpr int AHEAD = HALF ? 6 : 3, NF = HALF ? 8 : 4, QAHEAD = HALF ? 4 : 2, NQ = HALF ? 8 : 3;
    constexpr int P0 = R0 >> 1, P1 = (R1 + 1) >> 1;
    // halves: a wave's MFMA chain starts with the k-steps whose B operand is its OWN part of the position (its own stage writes, no
    // hand-off needed) and meets the partner's part IWAIT k-steps later -- by then the partner's stage is there and nobody waits
    constexpr int KB = HALF == 2 ? 4 * S0 : 0;                                      // the second wave starts at its first own k-step
    constexpr int IWAIT = HALF == 0 ? -1 : (HALF == 1 ? 4 * S1 : NK - KB) - QAHEAD;  // iteration whose B-operand prefetch is the first of the partner's
    auto kmap = [](int i) { return HALF == 2 ? (i + KB) % NK : i; };
    d2_t fa[NF][NP];
    auto load_k = [&](int kap, d2_t* f) {
#pragma unroll
      for (int j = P0; j < P1; ++j) f[j] = *reinterpret_cast<const d2_t*>(img + (kap * NP + j) * 128);
    };
    if (G::DENSE && !RESIDENT) {
#pragma unroll
      for (int k0 = 0; k0 < AHEAD; ++k0) if (k0 < NK) load_k(kmap(k0), fa[k0]);
    }
    d2_t fx[(NS - 1) / 2 > 0 ? (NS - 1) / 2 : 1];
    double ftail = 0.0;
    if (G::DENSE && OWN_TAIL) {
      if constexpr (RESIDENT) {
#pragma unroll
        for (int j = 0; j < (NS - 1) / 2; ++j) { fx[j][0] = fxr[2 * j]; fx[j][1] = fxr[2 * j + 1]; }
        ftail = ftailr;
      } else {
#pragma unroll
        for (int j = 0; j < (NS - 1) / 2; ++j) fx[j] = *reinterpret_cast<const d2_t*>(img + G::NMAIN * 64 + j * 128);
        ftail = lds[G::OFF_A + (G::NMAIN + NS - 1) * 64 + lane];
      }
    }
    WN4_STAMP(7);
    Cmd c;
    bool gaveup = false;  // (a bounded spin gave up: the tick runs to its end on whatever it has -- nothing it stores is read -- and leaves there:
                          //  ONE exit from the tick body, at its end, instead of a return in the middle with its merge branches)
    if constexpr (OPT_POLL) {
      // the ordinary tick finds its record: ONE branch; the spin loop is out of the way
      if (__builtin_expect(!__all(cmd_is_for(cw, t)), 0)) {
        uint32_t spins = 0;
        do {  // (bounded: a protocol error must not hang the GPU)
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 23)) { if (lane == 0) atomicExch(abort_flag, 1); gaveup = true; cw = 0; chs = 0.0; break; }
          cmd_read(rec, cw, chs);
        } while (!__all(cmd_is_for(cw, t)));
      }
      c.hs = chs; c.op = (uint32_t)cw & 0xFFFFu; c.n = (uint32_t)(cw >> 32);
    } else {
      if (t == 0) { c.hs = 0.0; c.op = inrange ? (OP_NEWITER | OP_INIT) : 0u; c.n = 0u; }
      else {
        uint32_t spins = 0;
        while (!__all(cmd_is_for(cw, t))) {  // (bounded: a protocol error must not hang the GPU)
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 23)) { if (lane == 0) atomicExch(abort_flag, 1); return false; }
          cmd_read(rec, cw, chs);
        }
        c.hs = chs; c.op = (uint32_t)cw & 0xFFFFu; c.n = (uint32_t)(cw >> 32);
      }
    }
    WN4_STAMP(5);
    const uint32_t op = c.op;
    double* const sp = sums + t3 * (G::NB * 16);
    const bool step = (op & OP_STEP) != 0;
    const int i = (int)((op >> OP_I_SHIFT) & 31u), n = (int)c.n;
    const int nchk = (PAR == 1 && step) ? nchecks(i, n) : 0;   // checks of levels >= 1 close at odd leaves
    const bool l0due = step && n == (1 << i) - 1;               // the whole orbit against its passive end: last leaf of a doubling
    double keinit_part = 0.0, gen_part = 0.0;
    // ---- commands ----
    const bool deep0 = nchk > NFAST, deep1 = nchk > NFAST + 1;
    // ONE vote for everything that does not happen on an ordinary leaf; `anydeep`: the sub-tree checks of levels >= 3 (SOME chain of the
    // group closes such a sub-tree on two odd ticks in three)
    // (no __builtin_expect on these votes: with `anyrare` / `anyl0` marked unlikely hipcc 7.2 produced a kernel that ran 15 % slower AND
    //  returned wrong draws -- 70 mismatching chains in 120 fuzz configurations, gpurun_out/r05)
    const bool anyrare = __any((op & (OP_PICK | OP_NEWITER | OP_SWAP | OP_UNDO | OP_DONE)) != 0 || deep0);
    const bool anydeep = PAR == 1 && anyrare && __any(deep0);
    if (anydeep) {  // (per chain, under exec masks: fetching both levels for every chain of the group cost 4 % -- vector-memory instructions are what a tick pays for)
      if (deep0) {
        const int sl = slot_of_leaf(n + 1 - (1 << NFAST));
        const double *sq = gvec(sl * 2), *sv = gvec(sl * 2 + 1);
#pragma unroll
        for (int s = S0; s < S1; ++s) { pq0[s] = sq[s * 16]; pv0[s] = sv[s * 16]; }
      }
      if (deep1) {
        const int sl = slot_of_leaf(n + 1 - (2 << NFAST));
        const double *sq = gvec(sl * 2), *sv = gvec(sl * 2 + 1);
#pragma unroll
        for (int s = S0; s < S1; ++s) { pq1[s] = sq[s * 16]; pv1[s] = sv[s * 16]; }
      }
    }
    if (anyrare) {
      if (__any((op & OP_PICK) != 0)) {  // `pk` holds the position of tick t-3
        if (op & OP_PICK) {
          double* dst = gvec(vprop + ((op & OP_IPPICK) ? 1 : 0));
#pragma unroll
          for (int s = S0; s < S1; ++s) dst[s * 16] = pk[s * 64];
        }
      }
      if (TRACE) { ntr -= ((op & OP_UNDO) ? 1 : 0) + ((op & OP_UNDO2) ? 1 : 0); }  // the leaves in flight when the iteration ended
    }
#pragma unroll
    for (int s = S0; s < S1; ++s) pk[s * 64] = qn[s];  // P(t-2): a pick commanded at the next tick copies it (same wave: after the read above)
    // ---- first half of the leapfrog step (adaptiveIntegrators.py:51-52 with hs = xi*h; v stays forward-time) ----
    const double hs = step ? c.hs : 0.0, hs2 = 0.5 * hs;
#pragma unroll
    for (int s = S0; s < S1; ++s) { v[s] = vi[s] + hs2 * g[s]; qn[s] = qc[s] + hs * v[s]; }
    // iteration starts and end swaps replace the position this tick produces (qn; such ticks do not step: hs = 0).  qc is left alone: it
    // is the position of tick t-1, which a pick commanded at tick t+2 copies (the last leaf of a doubling can be picked while the end swap
    // of the next doubling is already under way)
    if (anyrare) {
      if ((PAR == 1 || t == 0) && __any((op & OP_NEWITER) != 0)) {  // (iterations start on odd ticks; tick 0 loads q0)
        if (op & OP_NEWITER) {
          if (op & OP_INIT) {
            double* dst = gvec(vprop);
#pragma unroll
            for (int s = S0; s < S1; ++s) {
              qn[s] = coord(s) < D ? prm.q0[(size_t)cl * D + coord(s)] : 0.0;
              dst[s * 16] = qn[s];
            }
          } else {
            const double* src = gvec(vprop + ((op & OP_IPNEW) ? 1 : 0));
#pragma unroll
            for (int s = S0; s < S1; ++s) qn[s] = src[s * 16];
          }
          // samples[:, row] = generated(q)  (WALNUTS.py:164,695)
          double* row = prm.samples + ((size_t)cl * (prm.numIter + 1) + c.n) * prm.dg;
          if (prm.generated == G_Q0_SUMSQ) {
            if (S0 == 0 && lane < 4) row[0] = qn[0];  // coordinate 0: register 0 of lane (r = 0, b = 0, x)
            double s2 = 0.0;
#pragma unroll
            for (int s = S0; s < S1; ++s) s2 += qn[s] * qn[s];
            gen_part = s2;
          } else {
            const int k = prm.generated == G_FIRSTK ? prm.gen_k : D;
#pragma unroll
            for (int s = S0; s < S1; ++s) if (coord(s) < k) row[coord(s)] = qn[s];
          }
          if (!(op & (OP_INIT | OP_LAST))) {  // v = normal(size=d); both ends start at (qc, v)  (:236-253)
            double kp = 0.0;
#pragma unroll
            for (int s = S0; s < S1; ++s) {
              const double z = znext[coord(s)];
              v[s] = z;
              kp += z * z;
            }
            keinit_part = kp;
            double *dq_ = gvec(vcold), *dv_ = gvec(vcold + 1);
#pragma unroll
            for (int s = S0; s < S1; ++s) { dq_[s * 16] = qn[s]; dv_[s * 16] = v[s]; }
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): waited for HERE, not on the common path (wn_k_fast.hip)
        }
      }
      if (__any((op & OP_SWAP) != 0)) {  // the other end of the orbit becomes the active one (:283-286); the passive end lives in memory
        if (op & OP_SWAP) {
          double *dq_ = gvec(vcold), *dv_ = gvec(vcold + 1);
          double tq[NS], tv[NS];
#pragma unroll
          for (int s = S0; s < S1; ++s) { tq[s] = dq_[s * 16]; tv[s] = dv_[s * 16]; }
#pragma unroll
          for (int s = S0; s < S1; ++s) { dq_[s * 16] = qc[s]; dv_[s * 16] = v[s]; qn[s] = tq[s]; v[s] = tv[s]; }  // (no step on this tick)
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
      }
    }
    if (!G::DENSE) {  // diagonal / standard Gaussian: g = -(a .* q)  (targetDistr.py stdGauss; SURVEY 8d "diagonal variant")
#pragma unroll
      for (int s = S0; s < S1; ++s) g[s] = -(adiag[s] * qn[s]);
      WN4_STAMP(0);
    }
    // the position goes to LDS: k-step kappa's B operand (coordinates 4 kappa .. + 3 in EVERY block) is read back from there with the
    // block index replaced -- one LDS read per k-step instead of six DPP moves (FP64 MFMAs and vector instructions share the SIMD)
    if (G::DENSE) {
    double* const stage = HALF == 0 ? stage0 : (PAR ? pr + G::P_STAGE2 : stage0);  // (halves: two buffers -- the partner may still read tick t-1's)
#pragma unroll
    for (int s = S0; s < S1; ++s) stage[s * 64 + lane] = qn[s];
    if (HALF != 0) xseq[HALF - 1] = t + 1u;  // (all lanes, one word: no exec-mask region)  // (same wave, in order: behind the stage writes; the partner waits for it IWAIT k-steps into its chain)
    WN4_STAMP(0);
    // ---- G = -A Q ----
    {
      const double* const st0 = stage + 16 * r + x;  // + s * 64 + 4 * b'
      double accx = 0.0, gx = 0.0, qlast = 0.0;
      double acc[NSR];  // (set by the first k-step, whose accumulator operand is the literal 0: no register zeroing)
      const double* const stl = stage + lane;  // a register set of the PARTNER, in its own layout (tail rows: K split over the register sets)
      double qks[NQ];  // the B operands of the k-steps in flight / in use
      // the tail rows' K split over the register sets: own sets first, the partner's once its stage is known to be there
      constexpr int NOWN_T = HALF == 2 ? (NS - 1) - S0 : NS - 1;
      auto tail_iter = [](int j) { return (HALF == 2 && j >= NOWN_T) ? IWAIT + (j - NOWN_T) : j; };
      auto tail_reg = [](int j) { return HALF == 2 ? (j < NOWN_T ? S0 + j : j - NOWN_T) : j; };
      constexpr int ISUM = HALF == 2 ? IWAIT + (NS - 2 - NOWN_T) + 3 : NS + 1;  // blocks_sum: behind the last of those MFMAs
      static_assert(!OWN_TAIL || ISUM < NK - 1, "the tail's block sum must precede its last MFMA");
#pragma unroll
      for (int k0 = 0; k0 < QAHEAD; ++k0) if (k0 < NK) qks[k0] = st0[(kmap(k0) >> 2) * 64 + 4 * (kmap(k0) & 3)];
#pragma unroll
      for (int it = 0; it < NK; ++it) {
        const int kap = kmap(it);
        if (HALF != 0 && it == IWAIT) {  // from here on the partner's coordinates are read: it must have staged tick t
          uint32_t spins = 0;
          while ((int32_t)(xseq[2 - HALF] - (t + 1u)) < 0) {
            if (++spins > (1u << 24)) { if (lane == 0) atomicExch(abort_flag, 1); gaveup = true; break; }
          }
          asm volatile("" ::: "memory");
        }
        if (!RESIDENT) if (it + AHEAD < NK) load_k(kmap(it + AHEAD), fa[(it + AHEAD) % NF]);
        if (it + QAHEAD < NK) qks[(it + QAHEAD) % NQ] = st0[(kmap(it + QAHEAD) >> 2) * 64 + 4 * (kmap(it + QAHEAD) & 3)];
        const double qk = qks[it % NQ];
        const d2_t* f = fa[it % NF];
#pragma unroll
        for (int S = R0; S < R1; ++S) {
          const double fs = RESIDENT ? frag[RESIDENT ? kap : 0][RESIDENT ? S - R0 : 0] : f[S >> 1][S & 1];
          acc[S] = it == 0 ? mfma4(fs, qk, 0.0) : mfma4(fs, qk, acc[S]);
        }
        if (OWN_TAIL) {
          if (kap == NK - 1) qlast = qk;
#pragma unroll
          for (int j = 0; j < NS - 1; ++j) {
            if (tail_iter(j) == it) {
              const int sr = tail_reg(j);
              const double qx = (sr >= S0 && sr < S1) ? qn[sr] : stl[sr * 64];
              accx = j == 0 ? mfma4(fx[sr >> 1][sr & 1], qx, 0.0) : mfma4(fx[sr >> 1][sr & 1], qx, accx);
            }
          }
          if (it == ISUM) gx = blocks_sum(accx);            // (two k-steps behind its last MFMA)
          if (it == NK - 1) gx = mfma4(ftail, qlast, gx);   // columns 16 (NS-1) .. + 3
        }
      }
#pragma unroll
      for (int S = R0; S < R1; ++S) g[S] = acc[S];
      if (OWN_TAIL) g[NS - 1] = b == 0 ? gx : 0.0;  // blocks 1..3 of the last register are padding
    }
    }
    WN4_STAMP(1);
    if (HALF != 0 || OPT_PRE) cmd_read(pr + G::P_CMD + (((t + 1u) & 3u) * 4 + x) * 2, cw_pre, chs_pre);
    // the passive orbit end lives in memory (once per doubling: the whole-orbit check at its last leaf, and the end swaps): requested
    // here, in the registers the fragments have left, used after the half step
    const bool anyl0 = __any(l0due);
    double cq[NS], cv[NS];
#pragma unroll
    for (int s = S0; s < S1; ++s) asm volatile("" : "=v"(cq[s]), "=v"(cv[s]));  // (whatever the registers hold: the level-0 sums of a chain that is not at its last leaf are never read)
    if (anyl0) {
      if (l0due) {
        const double *sq = gvec(vcold), *sv = gvec(vcold + 1);
#pragma unroll
        for (int s = S0; s < S1; ++s) { cq[s] = sq[s * 16]; cv[s] = sv[s * 16]; }
      }
    }
    // ---- second half step; lp = 1/2 q.g, KE = 1/2 v.v ----
    double pl = 0.0, pk = 0.0;
#pragma unroll
    for (int s = S0; s < S1; ++s) pl += qn[s] * g[s];
#pragma unroll
    for (int s = S0; s < S1; ++s) { v[s] = v[s] + hs2 * g[s]; pk += v[s] * v[s]; }
    if (TRACE) {  // one row per integrator call: the state (q', v') goes here, the scalars come from the bookkeeper
      if (step && prm.trace && ntr < prm.trace_cap && ntr >= 0) {
        double* tr = prm.trace + ((size_t)cl * prm.trace_cap + ntr) * (TRACE_NSCAL + 2 * D) + TRACE_NSCAL;
#pragma unroll
        for (int s = S0; s < S1; ++s) if (coord(s) < D) { tr[coord(s)] = qn[s]; tr[D + coord(s)] = v[s]; }
      }
      if (step) ntr += 1;
    }
    // ---- U-turn dots of the sub-tree checks due at this leaf (stopCondition, WALNUTS.py:95-97) ----
    auto dots = [&](const double* qo, const double* vo, double& d1, double& d2) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int s = S0; s < S1; ++s) {
        const double dq = qn[s] - qo[s];
        a1 += v[s] * dq;
        a2 += vo[s] * dq;
      }
      d1 = a1; d2 = a2;
    };
    {
      double a1 = 0.0, a2 = 0.0;
      if (anyl0) dots(cq, cv, a1, a2);  // level 0
      put4(sp, pl, pk, a1, a2);
    }
    if (PAR == 1) {
      if constexpr (OPT_L12) {  // levels 1 (the leaf pair: registers) and 2 (LDS): on every odd tick, no vote (some chain of the group needs them on 15 odd ticks in 16;
         // the sums of a chain that does not are never read)
        double a1, a2, b1, b2;
        dots(qc, vi, a1, a2);
        double qo[NS], vo[NS];
#pragma unroll
        for (int s = S0; s < S1; ++s) { qo[s] = lc[s * 64]; vo[s] = lc[G::VEC + s * 64]; }
        dots(qo, vo, b1, b2);
        put4(sp + 16, a1, a2, b1, b2);
      } else if (__any(nchk > 1)) {
        double a1, a2, b1 = 0.0, b2 = 0.0;
        dots(qc, vi, a1, a2);
        if (__any(nchk > 2)) {
          double qo[NS], vo[NS];
#pragma unroll
          for (int s = S0; s < S1; ++s) { qo[s] = lc[s * 64]; vo[s] = lc[G::VEC + s * 64]; }
          dots(qo, vo, b1, b2);
        }
        put4(sp + 16, a1, a2, b1, b2);
      }
      if (anydeep) {
        {  // levels 3, 4: prefetched
          double a1, a2, b1, b2;
          dots(pq0, pv0, a1, a2);
          dots(pq1, pv1, b1, b2);
          put4(sp + 32, a1, a2, b1, b2);
        }
        for (int j = 5; __any(nchk > j); j += 2) {  // still deeper levels (one leaf in 32): fetched here
          double e[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (nchk > j + h) {
              const int sl = slot_of_leaf(n + 1 - (1 << (j + h)));
              const double *sq = gvec(sl * 2), *sv = gvec(sl * 2 + 1);
              double qo[NS], vo[NS];
#pragma unroll
              for (int s = S0; s < S1; ++s) { qo[s] = sq[s * 16]; vo[s] = sv[s * 16]; }
              dots(qo, vo, e[2 * h], e[2 * h + 1]);
            }
          }
          put4(sp + 16 * ((j + 1) >> 1), e[0], e[1], e[2], e[3]);
        }
      }
    }
    if ((PAR == 1 || t == 0) && anyrare && __any((op & OP_NEWITER) != 0)) put4(sp + 16 * (G::NB - 1), keinit_part, gen_part, 0.0, 0.0);
    if constexpr (WN4_PAD_T > 0) { double sink_ = pl; wn4_pad<WN4_PAD_T>(sink_); asm volatile("" ::"v"(sink_)); }
    seq[0] = t + 1u;  // the sums of tick t are in LDS (same wave, in order; all lanes store the one word)
    WN4_STAMP(2);
    // ---- statePush (:431,518): only sub-tree left endpoints are ever read back; they are even leaves ----
    if (PAR == 0) {
      // the level-2 endpoint (LDS) is due for SOME chain of the group on 15 even ticks in 16: predicated stores, no vote, no branch
      const bool ep = step && i >= 2 && (n & 3) == 0;
      const bool epg = ep && (n == 0 ? (i >= NFAST) : (__builtin_ctz((unsigned)n) >= NFAST));  // ... levels >= 3 (memory) on two in three
      if constexpr (OPT_EP) {
        lds_store_qv_masked_all<NS, S0, S1>(__ballot(ep), lc, qn, v);
        if (__any(epg)) {
          const int sl = epg ? slot_of_leaf(n) : 0;
          glb_store_qv_masked_all<NS, S0, S1>(__ballot(epg), gvec(sl * 2), gvec(sl * 2 + 1), qn, v);
        }
      } else if (__any(ep)) {
        if (ep) {
#pragma unroll
          for (int s = S0; s < S1; ++s) { lc[s * 64] = qn[s]; lc[G::VEC + s * 64] = v[s]; }
          if (epg) {
            const int sl = slot_of_leaf(n);
            double *sq = gvec(sl * 2), *sv = gvec(sl * 2 + 1);
#pragma unroll
            for (int s = S0; s < S1; ++s) { sq[s * 16] = qn[s]; sv[s * 16] = v[s]; }
          }
        }
      }
    }
    WN4_STAMP(3);
    WN4_STAMP(4);
#if defined(WN_STAMPS)
    fst_[6] += 1;
#endif
    // every chain of the group has finished (OP_DONE is issued to all four at once), or a spin gave up: the one way out
    return !(gaveup || (anyrare && __any((op & OP_DONE) != 0)));
  };
  if ((HALF != 0 || OPT_PRE) && OPT_POLL) cmd_read(pr + G::P_CMD + x * 2, cw_pre, chs_pre);  // tick 0's record (the kernel's prologue wrote it)
  int t3 = 0;
  for (uint32_t t = 0;; t += 2) {
    if (!tick(std::integral_constant<int, 0>{}, X1, X0, V1, V0, t, t3)) break;
    t3 = t3 == 2 ? 0 : t3 + 1;
    if (!tick(std::integral_constant<int, 1>{}, X0, X1, V0, V1, t + 1u, t3)) break;
    t3 = t3 == 2 ? 0 : t3 + 1;
  }
  WN4_STAMP_OUT(prm)
}

// ------------------------------------------------------------------------------------------------
// The bookkeeper: an R wave (RetireBook) and an I wave (IssueBook) per QUAD = two pairs (wn_fast.h).  Lane l serves chain slot
// l & 7 (pair (l >> 2) & 1 of the quad, chain x = l & 3) as replica l >> 3: eight replicas run the identical scalar code on
// identical inputs and stay bit-identical; in the R wave replica j turns check levels j and j + 8 into their verdict bits.
// ------------------------------------------------------------------------------------------------
template <class G, bool TRACE>
struct Fast4IO {
  const Params* prm;
  double* pr;  // the LDS region of this lane's pair
  int cl, x, rep;
  double lp, ke, keinit, gen;
  uint32_t fire;
  int ntrace;
  __device__ __forceinline__ double sum(int k) const { return k == Q_LP ? lp : k == Q_KE ? ke : k == Q_KEINIT ? keinit : gen; }
  __device__ __forceinline__ uint32_t firemask(int, int, int) const { return fire; }
  // a block of the chain's uniform stream from the window the I wave keeps in LDS (a miss: the caller runs Philox itself)
  __device__ __forceinline__ bool uniforms(int, int iter, uint32_t blk, double& ua, double& ub) const {
    const double* e = pr + (blk < 8u ? G::P_USTAT + (x * 8 + (int)blk) * 3 : G::P_URING + (x * 16 + (int)(blk & 15u)) * 3);
    const lds_vu32* tag = lds_words(e);
    const uint32_t t0 = tag[0], t1 = tag[1];  // (the writer stores the values first, then the tag)
    ua = *(const lds_vf64*)(e + 1); ub = *(const lds_vf64*)(e + 2);
    return t0 == (uint32_t)iter && t1 == blk;
  }
  // (issue wave) its own window of uniforms, read without the tag checks: the wave wrote the entries itself and knows what they hold --
  // blocks 0..7 and [wbase, wbase + 16) of iteration witer (IssueBookT::tick_fast)
  int witer;
  uint32_t wbase;
  __device__ __forceinline__ bool win_has(int iter, uint32_t k) const { const uint32_t blk = k >> 1; return iter == witer && (blk < 8u || blk - wbase < 16u); }
  __device__ __forceinline__ double u_direct(int, uint32_t k) const {
    const uint32_t blk = k >> 1;
    const double* e = pr + (blk < 8u ? G::P_USTAT + (x * 8 + (int)blk) * 3 : G::P_URING + (x * 16 + (int)(blk & 15u)) * 3);
    return *(const lds_vf64*)(e + 1 + (k & 1u));
  }
  __device__ __forceinline__ void write_h0(double v) const { if (rep == 0) *(lds_vf64*)(pr + G::P_H0 + x) = v; }
  // warm-up (WALNUTS.py:701-712): the adapted H goes to the issue side through LDS, written before the message that ends the iteration
  // (same wave, in order); the sorted energy-error factors of the delta quantile live in memory, [chain][2][warmupIter], one writer
  __device__ __forceinline__ void write_hc(double v) const { if (rep == 0) *(lds_vf64*)(pr + G::P_HC + x) = v; }
  __device__ __forceinline__ double read_hc() const { return *(const lds_vf64*)(pr + G::P_HC + x); }
  __device__ __forceinline__ double* facs() const { return prm->facs ? prm->facs + (size_t)cl * 2 * (size_t)prm->warmupIter : nullptr; }
  __device__ __forceinline__ bool facs_writer() const { return rep == 0; }
  __device__ __forceinline__ void write_gen(size_t row, double v) const {
    if (rep == 0) prm->samples[((size_t)cl * (prm->numIter + 1) + row) * prm->dg + 1] = v;
  }
  __device__ __forceinline__ void write_diag(size_t row, const double* d) const {  // once per chain and iteration
    if (rep == 0) {
      double* o = prm->diag + ((size_t)cl * prm->numIter + row) * NDIAG;
#pragma unroll
      for (int k = 0; k < NDIAG; ++k) o[k] = d[k];
    }
  }
  __device__ __forceinline__ void trace_step(int iter, int i, int xi, double h, double ham0, double lpn) {
    if (!TRACE) return;
    const int row = ntrace++;
    if (prm->trace && row < prm->trace_cap && rep == 0) {
      double* r = prm->trace + ((size_t)cl * prm->trace_cap + row) * (TRACE_NSCAL + 2 * prm->D);
      r[0] = iter; r[1] = i; r[2] = xi; r[3] = h; r[4] = ham0; r[5] = lpn; r[6] = 0; r[7] = 0;
      r[8] = 0; r[9] = 0.0; r[10] = 1; r[11] = 0;  // fixedLeapFrog: If = Ib = c = 0, lwt = 0, one gradient (adaptiveIntegrators.py:57-59)
    }
  }
};

#if !defined(WN4_BSLEEP)
#define WN4_BSLEEP 1
#endif
// wait until *p >= want (every lane its own word) or the quad's run-is-over word is set; false: neither within the bound
__device__ __forceinline__ bool seq_wait2(const lds_vu32* p, uint32_t want, const lds_vu32* over) {
  uint32_t spins = 0;
  while (__any((int32_t)(*p - want) < 0) && *over == 0u) {
    __builtin_amdgcn_s_sleep(WN4_BSLEEP);  // (coarser polling, up to s_sleep 16, changes nothing: gpurun_out/r03/ab30.txt)
    if (++spins > (1u << 23)) return false;
  }
  asm volatile("" ::: "memory");
  return true;
}

// HALVES: the group's vectors are on two T waves -- the R wave waits for both (odd replicas watch the second wave's word) and adds their sums
template <class G, bool TRACE, bool HALVES = false, bool WARM = false>
__device__ __forceinline__ void fast4_r_wave(const Params& prm, double* quad, const int grp0, const int npairs, int* abort_flag) {
  const int lane = threadIdx.x & 63, c8 = lane & 7, pq = c8 >> 2, x = c8 & 3, rep = lane >> 3;
  constexpr int H2 = G::P_SUMS2 - G::P_SUMS;  // from a sum of the first T wave to the same sum of the second
  const int cl = (grp0 + pq) * 4 + x;
  const bool inrange = pq < npairs && cl < prm.C;
  double* const pr = quad + (size_t)(pq < npairs ? pq : 0) * G::PAIR_DOUBLES;  // (lanes without a pair look at pair 0 and do nothing)
  const Cmd* const cmdbuf = reinterpret_cast<const Cmd*>(pr + G::P_CMD);
  RMsg* const msgbuf = reinterpret_cast<RMsg*>(pr + G::P_MSG);
  const lds_vu32* const seqT = lds_words(pr + G::P_SEQ) + (HALVES ? (rep & 1) : 0);
  lds_vu32* const seqQ = lds_words(quad + G::P_SEQ);  // [4]: R ticks finished, [6]: over
  RetireBookT<WARM> bk;
  bk.init(prm.chain0 + (uint32_t)cl, inrange, prm.numIter);
  bk.init_warm(prm);
  Fast4IO<G, TRACE> io;
  io.prm = &prm; io.pr = pr; io.cl = cl; io.x = x; io.rep = rep;
  io.lp = io.ke = io.keinit = io.gen = 0.0; io.fire = 0u; io.ntrace = 0; io.witer = 0; io.wbase = 0u;
  WN4_STAMP_DECL
  int t3 = 1;  // (t - 2) % 3: the slot of the sums this tick retires
  for (uint32_t t = 0;; ++t, t3 = t3 == 2 ? 0 : t3 + 1) {
    // ---- the T waves of the quad have finished tick t-2: its sums are in LDS ----
    if (t > 1 && !seq_wait2(seqT, t - 1u, seqQ + 6)) { if (lane == 0) atomicExch(abort_flag, 1); break; }
    if (seqQ[6] != 0u) break;
#if defined(WN4_TONLY)  // diagnostic: nothing is retired
    if (lane == 0) seqQ[4] = t + 1u;
    continue;
#endif
    WN4_STAMP(5);
    Cmd c;
    c.hs = 0.0; c.op = 0u; c.n = 0u;
    if (t > 1 && inrange) { c = cmdbuf[((t - 2u) & 3u) * 4 + x]; c.op &= 0xFFFFu; }  // what the T wave did in that tick (without the record's tick tag)
    const double* sp = pr + G::P_SUMS + t3 * (G::NB * 16);
    const bool retiring = bk.retires_step(c);
    const int cn = (int)c.n;
    const bool haschk = retiring && ((cn & 1) != 0 || cn == (1 << bk.i) - 1);  // odd leaves close sub-trees, the last leaf the orbit
    const unsigned long long vote = __ballot(haschk) | (__ballot((c.op & OP_NEWITER) != 0) ? 1ull << 63 : 0ull);  // (one branch for both)
    io.lp = sp[x]; io.ke = sp[4 + x];
    if (HALVES) { io.lp += sp[H2 + x]; io.ke += sp[H2 + 4 + x]; }
    io.fire = 0u;
    if (vote) {
      if (vote >> 63) {
        io.keinit = sp[16 * (G::NB - 1) + x]; io.gen = sp[16 * (G::NB - 1) + 4 + x];
        if (HALVES) { io.keinit += sp[H2 + 16 * (G::NB - 1) + x]; io.gen += sp[H2 + 16 * (G::NB - 1) + 4 + x]; }
      }
      if (vote << 1) {  // verdict of check level j (wn_fast.h fire_of) by replica j & 7, then OR over the replicas of the chain
        uint32_t f = 0u;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int j = rep + 8 * pass;
          const int jl = j <= G::MP ? j : G::MP;
          const double* dp = sp + (jl == 0 ? 8 : 16 * ((jl + 1) >> 1) + ((jl & 1) ? 0 : 8)) + x;
          double d1 = dp[0], d2 = dp[4];
          if (HALVES) { d1 += dp[H2]; d2 += dp[H2 + 4]; }
          if (retiring && j <= G::MP && need_check(bk.i, cn, j)) {
            const double xi = (double)bk.xi;
            if ((xi * d1 < 0.0) || (xi * d2 < 0.0)) f |= 1u << j;
          }
        }
        f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x128, 0xF, 0xF, false);  // row_ror:8: lanes l and l ^ 8
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        u2_t w = __builtin_amdgcn_permlane16_swap(f, f, false, false);
        f = w[0] | w[1];
        w = __builtin_amdgcn_permlane32_swap(f, f, false, false);
        io.fire = w[0] | w[1];
      }
    }
    WN4_STAMP(1);
    const RMsg m = bk.tick(prm, io, c);
    if constexpr (WN4_PAD_R > 0) { double sink_ = io.lp; wn4_pad<WN4_PAD_R>(sink_); asm volatile("" ::"v"(sink_)); }
    WN4_STAMP(2);
    if (rep == 0 && pq < npairs) msgbuf[t3 == 0 ? 2 * 4 + x : (t3 - 1) * 4 + x] = m;  // slot t % 3 = (t3 + 2) % 3
    if (lane == 0) seqQ[4] = t + 1u;  // message t is in LDS (same wave, in order)
    WN4_STAMP(3);
#if defined(WN_STAMPS)
    fst_[6] += 1;
#endif
  }
  WN4_STAMP_OUT(prm)
  if (TRACE && prm.trace_n && inrange && rep == 0) prm.trace_n[cl] = io.ntrace;
}

template <class G, bool TRACE, bool HALVES = false, bool WARM = false>
__device__ __forceinline__ void fast4_i_wave(const Params& prm, double* quad, const int grp0, const int npairs, int* abort_flag) {
  const int lane = threadIdx.x & 63, c8 = lane & 7, pq = c8 >> 2, x = c8 & 3, rep = lane >> 3;
  const int cl = (grp0 + pq) * 4 + x;
  const bool inrange = pq < npairs && cl < prm.C;
  double* const pr = quad + (size_t)(pq < npairs ? pq : 0) * G::PAIR_DOUBLES;
  const RMsg* const msgbuf = reinterpret_cast<const RMsg*>(pr + G::P_MSG);
  lds_vu32* const seqQ = lds_words(quad + G::P_SEQ);    // [4]: R ticks finished, [6]: over
  IssueBookT<WARM> bk;
  bk.init(prm.chain0 + (uint32_t)cl, inrange, prm.numIter);
  bk.init_warm(prm);
  Fast4IO<G, TRACE> io;
  io.prm = &prm; io.pr = pr; io.cl = cl; io.x = x; io.rep = rep;
  io.lp = io.ke = io.keinit = io.gen = 0.0; io.fire = 0u; io.ntrace = 0;
  int& witer = io.witer;          // iteration and first block of the uniform window
  uint32_t& wbase = io.wbase;
  witer = 0; wbase = 0u;
  // every chain needs at most numIter * (2^M - 1 + idle ticks per doubling + start) ticks; twice that is an internal error
  const double cap = 256.0 + 2.0 * (double)prm.numIter * ((double)(1u << prm.M) + 6.0 * (double)prm.M + 16.0);
  WN4_STAMP_DECL
  uint32_t tch[16];       // registers of the cache-touching loads in flight (below)
#pragma unroll
  for (int k = 0; k < 16; ++k) tch[k] = 0u;
  bool touched = false;
  int t3 = 0;  // t % 3
  for (uint32_t t = 0;; ++t, t3 = t3 == 2 ? 0 : t3 + 1) {
#if defined(WN4_TONLY)  // diagnostic: the T waves alone, on a fixed stream of ordinary leaves
    {
      Cmd c;
#if defined(WN4_TONLY_MIS)  // ... with the four chains of a wave at different places of their orbits (as in a real run)
      c.hs = 1.0e-4; c.op = OP_STEP | (10u << OP_I_SHIFT); c.n = (t + 26u * (uint32_t)x + 14u * (uint32_t)pq) & 1023u;
#else
      c.hs = 1.0e-4; c.op = OP_STEP | (10u << OP_I_SHIFT); c.n = t & 1023u;
#endif
      const bool done = t >= (uint32_t)(WN4_TONLY);
      if (done) { c.op |= OP_DONE; if (lane == 0) seqQ[6] = 1u; }
      if (rep == 0 && pq < npairs) cmd_publish(pr + G::P_CMD + (((t + 1u) & 3u) * 4 + x) * 2, c, t + 1u);
      if (done) break;
      const lds_vu32* sT = lds_words(pr + G::P_SEQ);
      if (t > 1 && !seq_wait2(sT, t - 1u, seqQ + 6)) break;
      continue;
    }
#endif
    // ---- momentum normals v = normal(size=d) (WALNUTS.py:236, wn_rng.h stream 1) of ONE chain's next iteration per tick:
    // lane L makes elements 2L, 2L+1 (one Philox block, one Box-Muller pair) ----
    // (ONE vote per tick for the three things that are rarely due: momentum normals, a window refill, the end of the run)
    const bool zw = bk.zwant != bk.zready && bk.zwant >= 1;
    const bool fresh = bk.iter != witer;
    const bool need = inrange && bk.iter <= prm.numIter && (fresh || (bk.ku >> 1) + 6u > wbase + 15u);
    const bool anyrare = __any(zw || need || bk.st == IssueBook::ST_DEAD);
    if (anyrare) {
      const unsigned long long want = __ballot(rep == 0 && zw);
      if (want) {
        const int cs = __builtin_ctzll(want);  // lanes 0..7 are chain slots 0..7
        const int it = __shfl(bk.zwant, cs);
        const uint32_t ch = (uint32_t)__shfl((int)bk.chain, cs);
        double* const znext = prm.work + (size_t)(grp0 + (cs >> 2)) * G::group_doubles(prm.M) + (size_t)((prm.M + 1) * 2 + 4) * G::VEC + (size_t)(cs & 3) * G::DPAD;
        {
          Philox4 p = philox4x32_10((uint32_t)lane, (uint32_t)it, ch, 1u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
          const uint64_t x0 = ((uint64_t)p.w[1] << 32) | p.w[0];
          const double u1 = ((double)(x0 >> 12) + 0.5) * 0x1.0p-52;
          const double u2 = u53(p.w[2], p.w[3]);
          const double rr = sqrt(-2.0 * log(u1));
          const double th = 6.283185307179586476925286766559 * u2;
          typedef double d2_t __attribute__((ext_vector_type(2)));
          d2_t z;
          z[0] = (2 * lane < prm.D) ? rr * cos(th) : 0.0;
          z[1] = (2 * lane + 1 < prm.D) ? rr * sin(th) : 0.0;
          *reinterpret_cast<d2_t*>(znext + 2 * lane) = z;
        }
        if (c8 == cs) bk.zready = it;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // visible to the T wave before the command that reads it is
      }
      // ---- the window of uniforms (wn_rng.h stream 0): when a chain starts an iteration or nears the end of its sliding
      // window, two Philox passes make sixteen blocks for each of the eight chains (lane = chain slot, block base + replica,
      // + 8).  Blocks 0..7 of the iteration stay (the direction bits of WALNUTS.py:216 are read until the last doubling).
      if (__any(need)) {
        const uint32_t nb = fresh ? 0u : (bk.ku >> 1);
        // the two Philox passes below cost the WAVE the same whether one chain stores its sixteen blocks or all eight do: when one chain
        // needs a refill, every chain slides its window to its own draw count (refills per wave: one in ~12 ticks instead of one in two)
        const bool slide = OPT_SLIDE ? (inrange && bk.iter <= prm.numIter && (fresh || (bk.ku >> 1) > wbase + 1u)) : need;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const uint32_t blk = nb + (uint32_t)(rep + 8 * pass);
          const Philox4 p = philox4x32_10(blk, (uint32_t)bk.iter, bk.chain, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
          const double u0 = u53(p.w[0], p.w[1]), u1 = u53(p.w[2], p.w[3]);
          if (slide) {
            double* e = pr + G::P_URING + (x * 16 + (int)(blk & 15u)) * 3;
            e[1] = u0; e[2] = u1;
            asm volatile("" ::: "memory");  // values first, then the tag (the R wave reads the tag first)
            lds_words(e)[0] = (uint32_t)bk.iter; lds_words(e)[1] = blk;
            if (blk < 8u) {
              double* es = pr + G::P_USTAT + (x * 8 + (int)blk) * 3;
              es[1] = u0; es[2] = u1;
              asm volatile("" ::: "memory");
              lds_words(es)[0] = (uint32_t)bk.iter; lds_words(es)[1] = blk;
            }
          }
        }
        if (slide) { witer = bk.iter; wbase = nb; }
      }
    }
    WN4_STAMP(0);
    // ---- the R wave has finished its tick t: its messages are in LDS ----
    if (!seq_wait2(seqQ + 4, t + 1u, seqQ + 6)) { if (lane == 0) atomicExch(abort_flag, 1); break; }
    WN4_STAMP(5);
    RMsg m;
    m.ku = 0u; m.w = RM_DEAD; m.iter = 1u; m.zwant = 0u;  // (lanes without a pair: nothing to do, finished)
    if (pq < npairs) m = msgbuf[t3 * 4 + x];
    // the second leaf pair of a doubling: its first leaf (tick t-1) is certified here, from the T wave's sums, one tick before the retire
    // side would (wn_fast.h IssueBook::wants_early: two idle ticks per doubling otherwise).  Rare: one tick per doubling and chain.
    // the ordinary tick -- every chain of the wave in the middle of a doubling, its uniforms in the window -- is straight-line code
    // (wn_fast.h IssueBookT::tick_fast): no early-certification vote, no state machine, no tag checks
    bool fl = bk.fast_ok(t + 1u, m);
    if (((t + 1u) & 1u) == 0) { const uint32_t kh = bk.fast_kh(m); fl = fl && io.win_has(bk.iter, kh) && io.win_has(bk.iter, kh + 1u); }
    const bool isdead = bk.st == IssueBookT<WARM>::ST_DEAD;
    const bool allfast = OPT_IFAST && __all(fl || isdead);
    bool early = false;
    if (!allfast) {
      const bool we = inrange && bk.wants_early(t + 1u, m);
      if (__any(we)) {
        if (!seq_wait2(lds_words(pr + G::P_SEQ) + (HALVES ? (rep & 1) : 0), t, seqQ + 6)) { if (lane == 0) atomicExch(abort_flag, 1); break; }  // the T wave(s) finished tick t-1
        const double* spe = pr + G::P_SUMS + (t3 == 0 ? 2 : t3 - 1) * (G::NB * 16);                                  // slot (t - 1) % 3
        double elp = *(const lds_vf64*)(spe + x), eke = *(const lds_vf64*)(spe + 4 + x);
        if (HALVES) { elp += *(const lds_vf64*)(spe + (G::P_SUMS2 - G::P_SUMS) + x); eke += *(const lds_vf64*)(spe + (G::P_SUMS2 - G::P_SUMS) + 4 + x); }
        const double eh0 = *(const lds_vf64*)(pr + G::P_H0 + x);
        early = we && IssueBook::early_certain(prm, elp, eke, eh0);
      }
    }
    WN4_STAMP(1);
    Cmd c;
    if (allfast) {
      c = bk.tick_fast(prm, io, t + 1u, m);
      c.op = isdead ? 0u : c.op; c.hs = isdead ? 0.0 : c.hs; c.n = isdead ? 0u : c.n;  // (a lane without a chain, or a finished one: tick() issues nothing)
    } else {
      c = bk.tick(prm, io, t + 1u, m, early);
    }
    if constexpr (WN4_PAD_I > 0) { double sink_ = c.hs; wn4_pad<WN4_PAD_I>(sink_); c.hs = sink_ == 12345.678 ? 0.0 : c.hs; }
    WN4_STAMP(2);
    // ---- every chain of the quad has finished: all its waves leave (the quads of a workgroup do not wait for each other) ----
    bool done = anyrare && __all(bk.finished());  // (finished() needs ST_DEAD, which was part of this tick's one vote, or shows at the next)
    if ((double)t > cap) {  // cannot happen; never hang the GPU
      if (lane == 0) atomicExch(abort_flag, 1);
      done = true;
    }
    if (done) { c.op |= OP_DONE; if (lane == 0) seqQ[6] = 1u; }
    if (rep == 0 && pq < npairs) cmd_publish(pr + G::P_CMD + (((t + 1u) & 3u) * 4 + x) * 2, c, t + 1u);
#if !defined(WN4_NOTOUCH)
    // (the values these loads return are never used, but the registers they land in must stay reserved until they have arrived: they are
    //  "read" HERE, a whole tick after the loads were issued -- round 4 read them right behind the loads, and the issue wave, whose pass
    //  is what the tick is made of (tools: -DWN4_PAD_I), stood still for a memory round trip on every tick that touched something)
    if (touched) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { asm volatile("" ::"v"(tch[k])); tch[k] = 0u; }
      touched = false;
    }
    // ---- the vectors the T wave will fetch from memory in tick t+1 with nothing to do meanwhile -- the passive orbit end (last leaf of a
    // doubling, end swap) and the sub-tree endpoints of levels >= 5 (fetched when due) -- are pulled into the CU's vector cache now: one
    // 4-byte load per 128-byte line (replica s touches register s of the chain's vector); the values are not used.
    {
      constexpr int NS = G::NS;
      const bool step = (c.op & OP_STEP) != 0;
      const int ci = (int)((c.op >> OP_I_SHIFT) & 31u), cn = (int)c.n;
      const bool cold = inrange && ((step && cn == (1 << ci) - 1) || (c.op & OP_SWAP) != 0);
      const int nchk = (inrange && step && (cn & 1)) ? nchecks(ci, cn) : 0;
      if (__any(cold || nchk > 5)) {
        const uint32_t* const wb = reinterpret_cast<const uint32_t*>(prm.work + (size_t)(grp0 + (pq < npairs ? pq : 0)) * G::group_doubles(prm.M) +
                                                                     x * (NS * 16) + (rep < NS ? rep : 0) * 16);
        const size_t vstride = (size_t)G::VEC * 2;  // (in 4-byte words)
        const int vcold = (prm.M + 1) * 2 + 2;
        if (cold && rep < NS) { tch[0] = wb[vcold * vstride]; tch[1] = wb[(vcold + 1) * vstride]; }
#pragma unroll
        for (int j = 5; j <= G::MP; ++j) {
          if (__any(nchk > j)) {
            if (nchk > j && rep < NS) {
              const int sl = slot_of_leaf(cn + 1 - (1 << j));
              tch[2 * (j - 4)] = wb[(size_t)(sl * 2) * vstride];
              tch[2 * (j - 4) + 1] = wb[(size_t)(sl * 2 + 1) * vstride];
            }
          }
        }
        touched = true;
      }
    }
#endif
    WN4_STAMP(3);
#if defined(WN_STAMPS)
    fst_[6] += 1;
#endif
    if (done) break;
  }
  WN4_STAMP_OUT(prm)
}

// WARM: the retire / issue waves also run warm-up adaptation of H and delta (wn_fast.h RetireBookT<true> / IssueBookT<true>); these
// instantiations are compiled in a translation unit of their own (-DWN4_WARM_TU, csrc/Makefile: wn_k_fast4w.o)
template <int NS, bool TAIL, bool TRACE, bool DENSE, bool HALVES = false, bool WARM = false>
__global__ __launch_bounds__(HALVES ? 256 : 512) void wn_fast4_kernel(Params prm, int* abort_flag, int pairs) {
  using G = Fast4Geom<NS, TAIL, DENSE>;
  extern __shared__ double lds[];
  // the -A fragment image (see the header): fragment f, lane (r, b, x)
  const int D = prm.D;
  for (int idx = threadIdx.x; idx < G::NFR * 64; idx += blockDim.x) {
    const int f = idx >> 6, l = idx & 63, x = l & 3, b = (l >> 2) & 3, r = l >> 4;  // A operand: lane 16 k + 4 b + i
    int row, col, dst;
    if (f < G::NMAIN) {  // k-step kappa x row set S: block b holds rows 16 S + 4 b + i against columns 4 kappa + k
      const int kap = f / G::NSR, S = f - kap * G::NSR;
      row = 16 * S + 4 * b + x; col = 4 * kap + r;
      dst = (f >> 1) * 128 + 2 * l + (f & 1);
    } else {             // TAIL: rows 16 (NS-1) + i, K split over the blocks (columns 16 s + 4 b + k); last: the tail columns
      const int sx = f - G::NMAIN;
      row = 16 * (NS - 1) + x;
      if (sx < NS - 1) { col = 16 * sx + 4 * b + r; dst = G::NMAIN * 64 + (sx >> 1) * 128 + 2 * l + (sx & 1); }
      else { col = 16 * (NS - 1) + r; dst = (G::NMAIN + NS - 1) * 64 + l; }
    }
    lds[G::OFF_A + dst] = (row < D && col < D) ? -prm.tpar[(size_t)row * D + col] : 0.0;
  }
  // LDS keeps what the previous kernel on this CU left there: every control area starts from zero
  for (int idx = threadIdx.x; idx < (HALVES ? G::PAIR_DOUBLES_H : pairs * G::PAIR_DOUBLES); idx += blockDim.x) lds[G::OFF_PAIR + idx] = 0.0;
  __syncthreads();
  // tick 0 is not commanded by anybody: every chain loads q0 (the R wave reads the record when it retires the tick)
  for (int idx = threadIdx.x; idx < pairs * 4; idx += blockDim.x) {
    Cmd* c0 = reinterpret_cast<Cmd*>(lds + G::OFF_PAIR + (size_t)(idx >> 2) * G::PAIR_DOUBLES + G::P_CMD) + (idx & 3);
    const bool in = (blockIdx.x * pairs + (idx >> 2)) * 4 + (idx & 3) < prm.C;
    c0->hs = 0.0; c0->op = in ? (OP_NEWITER | OP_INIT) : 0u; c0->n = 0u;
  }
  __syncthreads();
  // Waves: quads = ceil(pairs / 2) bookkeeper pairs (R, I) first, then the T waves.  With four pairs: R0 I0 R1 I1 T0 T1 T2 T3, and waves
  // w, w + 4 share a SIMD (speed only): every SIMD runs one T wave and one half of a bookkeeper, which serves the quad's two T waves.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if constexpr (HALVES) {  // one group per workgroup: R I Ta Tb, a SIMD each
    double* pr = lds + G::OFF_PAIR;
    if (wave == 0) fast4_r_wave<G, TRACE, true, WARM>(prm, pr, (int)blockIdx.x, 1, abort_flag);
    else if (wave == 1) fast4_i_wave<G, TRACE, true, WARM>(prm, pr, (int)blockIdx.x, 1, abort_flag);
    else if (wave == 2) fast4_t_wave<G, TRACE, 1>(prm, lds, pr, (int)blockIdx.x, abort_flag);
    else fast4_t_wave<G, TRACE, 2>(prm, lds, pr, (int)blockIdx.x, abort_flag);
    return;
  }
  const int quads = (pairs + 1) >> 1;
#if defined(WN4_TFIRST)  // diagnostic: the T waves are the older ones
  const bool is_t = wave < pairs;
  const int tw = wave, bw = wave - pairs;
#else
  const bool is_t = wave >= 2 * quads;
  const int tw = wave - 2 * quads, bw = wave;
#endif
  if (is_t) {
    // Priorities.  Round 3: T waves at priority 1 over bookkeepers at 0 (1.32e9 steps/s against 1.27e9 with equal priorities; raising
    // the R or I wave was slower THEN).  Round 5 measured whose pass the tick is made of -- N dependent vector instructions more per tick in
    // one kind of wave (-DWN4_PAD_T / _R / _I = 40): T +1.5 %, R +2.1 %, I +4.1 % of the tick (gpurun_out/r05/ab4.txt) -- the ISSUE wave's,
    // whose vector instructions get no slot during the 650 quad-cycles of its SIMD's MFMA phase.  With the bookkeepers ABOVE the T waves
    // (I 3, R 2, T 1) their instructions slip in between the MFMAs: 1.375e9 -> 1.412e9 (ab7.txt / ab8.txt; every assignment with both
    // bookkeepers above T gives 1.407-1.414e9, coarser polling changes nothing).
#if !defined(WN4_TPRIO)
#define WN4_TPRIO 1
#endif
    __builtin_amdgcn_s_setprio(WN4_TPRIO);
    double* pr = lds + G::OFF_PAIR + (size_t)tw * G::PAIR_DOUBLES;
    fast4_t_wave<G, TRACE>(prm, lds, pr, (int)blockIdx.x * pairs + tw, abort_flag);
  } else {
#if !defined(WN4_RPRIO)
#define WN4_RPRIO 2
#endif
#if !defined(WN4_IPRIO)
#define WN4_IPRIO 3
#endif
    if (!(bw & 1)) __builtin_amdgcn_s_setprio(WN4_RPRIO);
    if (bw & 1) __builtin_amdgcn_s_setprio(WN4_IPRIO);
    const int quad = bw >> 1;
    double* qd = lds + G::OFF_PAIR + (size_t)(2 * quad) * G::PAIR_DOUBLES;
    const int np = pairs - 2 * quad >= 2 ? 2 : 1;
    if (bw & 1) fast4_i_wave<G, TRACE, false, WARM>(prm, qd, (int)blockIdx.x * pairs + 2 * quad, np, abort_flag);
    else fast4_r_wave<G, TRACE, false, WARM>(prm, qd, (int)blockIdx.x * pairs + 2 * quad, np, abort_flag);
  }
}

// can this kernel run the launch?  fixedLeapFrog sampling (traced or not; untraced also with warm-up adaptation of H and delta -- the
// reference's default call, WALNUTS.py:111-125: integrator=fixedLeapFrog, warmupIter=1000 --, no orbit statistics) on a dense Gaussian
// with D <= 100 or a diagonal / standard one with D <= 112
#if !defined(WN4_WARM_TU)
bool fast4_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm) {
  if (kc.layout != 1 || adapt != 0 || prm.mode != MODE_SAMPLE) return false;
  if (warm && (prm.trace || prm.trace_n)) return false;  // (traced warm-up runs: the one-stream kernel)
  if (prm.omin || prm.M > 11) return false;
  return kc.p1 == 0 /* CK_DENSE */ ? prm.D <= 100 : prm.D <= 112;
}

int fast4_pairs(int C) {  // T waves per workgroup: fill the chip's 256 CUs before stacking them on a CU
  const int groups = (C + 3) / 4;
  return groups >= 4 * 256 ? 4 : groups >= 2 * 256 ? 2 : 1;
}
#else
int fast4_pairs(int C);
#endif

template <int NS, bool TAIL, bool DENSE, bool WARM>
static hipError_t fast4_go(const Params& prm, int* abort_flag, hipStream_t st, bool whole) {
  using G = Fast4Geom<NS, TAIL, DENSE>;
  const int pairs = fast4_pairs(prm.C), groups = (prm.C + 3) / 4;
  hipError_t r;
  constexpr bool CAN_TRACE = !WARM;  // (WARM instantiations exist untraced only)
  // fewer groups than two per CU (BASELINE configs[1]'s literal 1024 chains = 256 groups): the group's vectors on TWO T waves (halves), so
  // that a CU's four SIMDs run R, I and two half-tick T waves instead of R, I, one T wave and nothing
  if constexpr (DENSE && NS >= 6) if (pairs == 1 && !whole) {
    dim3 grid(groups), block(256);
    const size_t shm = G::lds_bytes_halves();
    if (CAN_TRACE && prm.trace_n) {
      r = hipFuncSetAttribute((const void*)wn_fast4_kernel<NS, TAIL, CAN_TRACE, DENSE, true, WARM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (r != hipSuccess) return r;
      hipLaunchKernelGGL((wn_fast4_kernel<NS, TAIL, CAN_TRACE, DENSE, true, WARM>), grid, block, shm, st, prm, abort_flag, 1);
    } else {
      r = hipFuncSetAttribute((const void*)wn_fast4_kernel<NS, TAIL, false, DENSE, true, WARM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (r != hipSuccess) return r;
      hipLaunchKernelGGL((wn_fast4_kernel<NS, TAIL, false, DENSE, true, WARM>), grid, block, shm, st, prm, abort_flag, 1);
    }
    return hipGetLastError();
  }
  dim3 grid((groups + pairs - 1) / pairs), block(64 * (pairs + 2 * ((pairs + 1) / 2)));
  const size_t shm = G::lds_bytes(pairs);
  if (CAN_TRACE && prm.trace_n) {
    r = hipFuncSetAttribute((const void*)wn_fast4_kernel<NS, TAIL, CAN_TRACE, DENSE, false, WARM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (r != hipSuccess) return r;
    hipLaunchKernelGGL((wn_fast4_kernel<NS, TAIL, CAN_TRACE, DENSE, false, WARM>), grid, block, shm, st, prm, abort_flag, pairs);
  } else {
    r = hipFuncSetAttribute((const void*)wn_fast4_kernel<NS, TAIL, false, DENSE, false, WARM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (r != hipSuccess) return r;
    hipLaunchKernelGGL((wn_fast4_kernel<NS, TAIL, false, DENSE, false, WARM>), grid, block, shm, st, prm, abort_flag, pairs);
  }
  return hipGetLastError();
}

template <bool WARM>
static hipError_t fast4_dispatch(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st) {
  const int D = prm.D;
  const bool whole = (kc.route & WN_ROUTE_FAST4_WHOLE) != 0;
  if (kc.p1 == 0 /* CK_DENSE */) {
    if (D <= 32) return fast4_go<2, false, true, WARM>(prm, abort_flag, st, whole);
    if (D <= 64) return fast4_go<4, false, true, WARM>(prm, abort_flag, st, whole);
    if (D <= 96) return fast4_go<6, false, true, WARM>(prm, abort_flag, st, whole);
    if (D <= 100) return fast4_go<7, true, true, WARM>(prm, abort_flag, st, whole);
    return hipErrorInvalidValue;
  }
  if (D <= 32) return fast4_go<2, false, false, WARM>(prm, abort_flag, st, whole);
  if (D <= 64) return fast4_go<4, false, false, WARM>(prm, abort_flag, st, whole);
  if (D <= 96) return fast4_go<6, false, false, WARM>(prm, abort_flag, st, whole);
  if (D <= 112) return fast4_go<7, false, false, WARM>(prm, abort_flag, st, whole);
  return hipErrorInvalidValue;
}

#if defined(WN4_WARM_TU)
// (this translation unit: the warm-up instantiations only)
hipError_t launch_fast4_warm(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st) { return fast4_dispatch<true>(kc, prm, abort_flag, st); }
#else
hipError_t launch_fast4_warm(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st);  // wn_k_fast4w.o
hipError_t launch_fast4(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st) {
  if (prm.warmupIter > 0) return launch_fast4_warm(kc, prm, abort_flag, st);
  return fast4_dispatch<false>(kc, prm, abort_flag, st);
}

// doubles of workspace per chain group, for wn_capi.hip's allocation (every geometry fits the role-split kernel's share)
size_t fast4_work_doubles(int C, int M) { return (size_t)((C + 3) / 4) * Fast4Geom<7, true>::group_doubles(M); }
#endif

}  // namespace wn
