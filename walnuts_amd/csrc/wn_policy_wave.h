// wn_policy_wave.h -- layout policy "one chain per wavefront": the D-vectors of a chain are striped
// over the 64 lanes of one wave (element e*64 + lane in register e, D <= 64*NE), every lane keeps a
// replica of the chain's scalars, dot products are DPP / permlane all-reduces that return the same
// bits in every lane, and nothing needs LDS or a barrier.  This is the layout for large D
// (100 < D <= 1024) where the state of a chain no longer fits one lane or one MFMA group:
//
//   * fused, HBM-bound targets (std / diagonal Gaussian, StockWatson): Engine::run() in one
//     persistent kernel, Tgt::grad evaluated in the wave;
//   * split mode (Engine::split_tick): the gradient comes from outside the chain kernel -- the
//     batched FP64-MFMA GEMM of wn_k_wave.hip for dense Gaussians whose precision matrix cannot
//     stay on chip (D = 1000, SURVEY.md 8d cfg 4), or the caller's own batched (logp, grad) on
//     device tensors (the reference's lpFun plugin).  A chain is parked in memory between launches.
//
// Every memory-resident vector of a chain is [vector][DP] with DP = 64*NE, element e*64+lane, so a
// wave load/store of one register touches 512 contiguous bytes.
#pragma once
#include "wn_chain.h"
#include "wn_targets.h"

namespace wn {

struct EmptyVecW {};

#ifndef WN_CL_SLEEP
#define WN_CL_SLEEP 8  // 64-cycle units between polls of a cluster's arrival counter
#endif

// (every CTRL used with wn_dpp is a permutation inside the row -- quad_perm, row_mirror, row_half_mirror: no lane reads out of
//  bounds.  bound_ctrl = true tells the compiler so: with `false` it materialised the unused `old` operand, two v_mov_b32 per step.)
template <int CTRL>
__device__ __forceinline__ double wn_dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, same bits everywhere: xor-1, xor-2 inside quads, mirrored halves and rows
// (every exchange is symmetric, so both partners add the same two numbers), then across rows
__device__ __forceinline__ double wn_wavesum(double x) {
  x += wn_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += wn_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += wn_dpp<0x141>(x);  // row_half_mirror
  x += wn_dpp<0x140>(x);  // row_mirror
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

// ---- sums of quantities that live in lanes 0..15 only (the <= 16-D targets: WavePolicy::VLDS) ----
// sum over the row of 16 lanes, in the order of wn_wavesum's first four steps; every lane of a row gets its row's sum
__device__ __forceinline__ double wn_rowsum16(double x) {
  x += wn_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += wn_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += wn_dpp<0x141>(x);  // row_half_mirror
  x += wn_dpp<0x140>(x);  // row_mirror
  return x;
}
// [a rows 0,1 | b rows 0,1]
__device__ __forceinline__ double wn_pack32(double a, double b) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const u2_t l = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const u2_t h = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(h[0], l[0]);
}
// [a row 0 | b row 0 | a row 2 | b row 2]
__device__ __forceinline__ double wn_pack16(double a, double b) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const u2_t l = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const u2_t h = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(h[0], l[0]);
}
// the wave sum of a quantity whose rows 1..3 are zero, with wn_wavesum's bits ((s + 0) + (0 + 0)): one row reduction
__device__ __forceinline__ double wn_rowsum0(double x) {
  x = wn_rowsum16(x);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 0), __builtin_amdgcn_readlane(__double2loint(x), 0)) + 0.0;
}


// exp(x) with the device library's own algorithm and constants (dn = rint(x log2 e), t = x - dn ln2 in two pieces, a
// degree-11 polynomial, ldexp; ocml expD) -- but with the Horner steps as THREE-operand v_fma_f64: the compiler turns
// `fma(t, p, c)` into v_mov_b64 acc, c + v_fmac_f64 (two instructions per step: 11 extra moves per exp, and the wave
// kernels are bound by the instruction count of one wave).  The steps are opaque to the compiler (inline asm), so an
// exp of a uniform argument is still computed per lane -- what the callers want.
__device__ __forceinline__ double wn_fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double wn_exp(double x) {
  const double dn = rint(x * 0x1.71547652b82fep+0);
  double t = wn_fma3(dn, -0x1.62e42fefa39efp-1, x);
  t = wn_fma3(dn, -0x1.abc9e3b39803fp-56, t);
  double p = wn_fma3(t, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = wn_fma3(t, p, 0x1.71dee623fde64p-19);
  p = wn_fma3(t, p, 0x1.a01997c89e6b0p-16);
  p = wn_fma3(t, p, 0x1.a01a014761f6ep-13);
  p = wn_fma3(t, p, 0x1.6c16c1852b7b0p-10);
  p = wn_fma3(t, p, 0x1.1111111122322p-7);
  p = wn_fma3(t, p, 0x1.55555555502a1p-5);
  p = wn_fma3(t, p, 0x1.5555555555511p-3);
  p = wn_fma3(t, p, 0x1.000000000000bp-1);
  p = wn_fma3(t, p, 1.0);
  p = wn_fma3(t, p, 1.0);
  double z = ldexp(p, (int)dn);
  z = x > 0x1.62e42fefa39efp+9 ? INFINITY : z;
  z = x < -0x1.74910d52d3051p+9 ? 0.0 : z;
  return z;  // (a NaN argument gives NaN: both comparisons are false)
}

template <int CTRL>
__device__ __forceinline__ double wn_dpp0(double x) {  // out-of-row sources read as zero
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wn_readlane(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// inclusive prefix sum over the 64 lanes: row_shr 1,2,4,8 inside each row of 16 (DPP: no LDS round trip per
// step, unlike a shuffle-based Hillis-Steele scan), then the totals of the lower rows through v_readlane
__device__ __forceinline__ double wn_scan_up(double x, int lane) {
  x += wn_dpp0<0x111>(x);
  x += wn_dpp0<0x112>(x);
  x += wn_dpp0<0x114>(x);
  x += wn_dpp0<0x118>(x);
  const double t0 = wn_readlane(x, 15), t1 = wn_readlane(x, 31), t2 = wn_readlane(x, 47);
  const int row = lane >> 4;
  double s0 = t0, s01 = t0 + t1, s012 = (t0 + t1) + t2;  // (select form: see wn_scan_down)
  asm volatile("" : "+v"(s0), "+v"(s01), "+v"(s012));
  double add = s012;
  add = row == 2 ? s01 : add;
  add = row == 1 ? s0 : add;
  add = row == 0 ? 0.0 : add;
  return x + add;
}
// inclusive suffix sum (lane l gets the sum over lanes >= l)
__device__ __forceinline__ double wn_scan_down(double x, int lane) {
  x += wn_dpp0<0x101>(x);  // row_shl:1
  x += wn_dpp0<0x102>(x);
  x += wn_dpp0<0x104>(x);
  x += wn_dpp0<0x108>(x);
  const double t1 = wn_readlane(x, 16), t2 = wn_readlane(x, 32), t3 = wn_readlane(x, 48);
  const int row = lane >> 4;
  // (select form over sums every lane computes: as nested conditional expressions the compiler built divergent if / else regions --
  //  five exec-mask regions and two branches per scan, ~170 cycles for a wave alone on its SIMD, tools/branch_cost.hip)
  double s3 = t3, s23 = t2 + t3, s123 = t1 + (t2 + t3);
  asm volatile("" : "+v"(s3), "+v"(s23), "+v"(s123));
  double add = s123;
  add = row == 1 ? s23 : add;
  add = row == 2 ? s3 : add;
  add = row == 3 ? 0.0 : add;
  return x + add;
}


enum { EXT_NONE = 0, EXT_HALF_QG = 1, EXT_LP = 2 };

// lp = -1/2 sum a_i q_i^2, grad = -a_i q_i  (stdGauss, targetDistr.py:18-21, is a == 1)
template <int NE, int ES = 64>
struct WTgtDiag {
  static constexpr int EXT = EXT_NONE;
  double a[NE];
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 0;
  __device__ __forceinline__ void init(const Params& prm, int l0, double*) {
#pragma unroll
    for (int e = 0; e < NE; ++e) a[e] = (e * ES + l0) < prm.D ? prm.tpar[e * ES + l0] : 0.0;
  }
  __device__ __forceinline__ double grad(const double* q, double* g, int) const {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) { g[e] = -(a[e] * q[e]); s += q[e] * g[e]; }
    return s;
  }
  __device__ __forceinline__ static double lp_finish(double r) { return 0.5 * r; }
};

// gradient supplied from outside the chain kernel; lp = 1/2 q.g (Gaussian) or supplied too
template <int NE, int EXT_>
struct WTgtExt {
  static constexpr int EXT = EXT_;
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 0;
  __device__ __forceinline__ void init(const Params&, int, double*) {}
  __device__ __forceinline__ double grad(const double*, double*, int) const { return 0.0; }
  __device__ __forceinline__ static double lp_finish(double r) { return EXT_ == EXT_HALF_QG ? 0.5 * r : r; }
};


// ---- wave scans on DPP only (GFX9 DPP: wave_shr / wave_shl / row_bcast exist on gfx950) ----
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double wn_dppm(double x) {  // lanes outside ROWMASK, and out-of-range sources, read as zero
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// lane l gets lane l-1's value (lane 0: zero) / lane l+1's value (lane 63: zero): one DPP move per half, no LDS round trip
__device__ __forceinline__ double wn_wave_shr1(double x) { return wn_dpp0<0x138>(x); }
__device__ __forceinline__ double wn_wave_shl1(double x) { return wn_dpp0<0x130>(x); }
// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 (six DPP steps, no v_readlane)
__device__ __forceinline__ double wn_scan_up_b(double x) {
  x += wn_dpp0<0x111>(x);
  x += wn_dpp0<0x112>(x);
  x += wn_dpp0<0x114>(x);
  x += wn_dpp0<0x118>(x);
  x += wn_dppm<0x142, 0xA>(x);  // row_bcast:15
  x += wn_dppm<0x143, 0xC>(x);  // row_bcast:31
  return x;
}

// StockWatson stochastic volatility (BASELINE.json config 5; WALNUTSpy_examples/StockWatson/
// sw_innov.stan:7-51 as bridgestan's log_density_gradient would evaluate it: unconstrained
// parameters, constants dropped).  D = 3T, parameter order tSigma, z1, zinn[T-2], x1, xinn[T-1],
// tau1, tauinn[T-1]; tpar = [T, y[1..T]], T <= 256.
//
// The three latent series are cumulative sums over time, so the chain's vectors live in a TIME-MAJOR register layout
// (MAPPED: the policy asks the target where an element sits): lane l owns t = 4l+1 .. 4l+4, register 4s+k holds series s
// (0 = the z block, 1 = x, 2 = tau) at t = 4l+k+1; tSigma, which belongs to no series, sits in the z block's unused slot
// t = T.  The engine's vector arithmetic is element-wise and its sums are over all lanes, so it does not care; the
// gradient needs no transpose (round 3 moved q and g through a 6 KB LDS transpose and back on every evaluation: 48 LDS
// accesses and three fences), forward (z, x, tau) and backward (adjoint) sums are lane-local runs of four plus wave scans
// of the lane totals on DPP alone (wn_scan_up_b, wn_wave_shr1: round 3 used ds_bpermute shifts and v_readlane row
// totals), every conditional is a select (a lone wave pays 25-55 cycles per branch or exec-mask region), and exp() is
// wn_exp.
template <int NE>
struct WTgtStockWatson {
  static_assert(NE == 12, "three series x four time points per lane");
  static constexpr int EXT = EXT_NONE;
  static constexpr int TPL = 4;              // time points per lane
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 0;
  static constexpr bool SPEC_ENGINE = true;  // tick-latency bound: plain sampling runs get the specialised engine (wn_k_wave.hip go())
  static constexpr bool MAPPED = true;
  int T, laneT, kT;
  double y[TPL];
  __device__ __forceinline__ void init(const Params& prm, int lane, double*) {
    T = (int)prm.tpar[0];
    laneT = (T - 1) >> 2; kT = (T - 1) & 3;
#pragma unroll
    for (int k = 0; k < TPL; ++k) { const int t = TPL * lane + k + 1; y[k] = t <= T ? prm.tpar[t] : 0.0; }
  }
  // element held by register e of lane `lane` (parameter index), or -1
  __device__ __forceinline__ int map_gidx(int e, int lane) const {
    const int s = e >> 2, t = TPL * lane + (e & 3) + 1;
    if (t > T) return -1;
    if (s == 0) return t == T ? 0 : t;
    return s * T + t - 1;
  }
  __device__ __forceinline__ static double lp_finish(double r) { return r; }
  __device__ __forceinline__ static double sel4(const double* a, int k) { return k == 0 ? a[0] : k == 1 ? a[1] : k == 2 ? a[2] : a[3]; }
  // The transformed parameters of sw_innov.stan:17-35 for the orbit statistics (generated = param_constrain, mainSW.py:28-29): the three
  // series at this lane's four time points -- the forward sums of grad() below, same code -- and sigma, gpri.  `put(index in the
  // (6T+1)-wide row of generated(), value)`: sigma, gpri at 3T, 3T+1; z[t], t = 1..T-1, at 3T+1+t; x[t] at 4T+t; tau[t] at 5T+t.
  static constexpr bool HAS_CONSTRAIN = true;
  template <class F>
  __device__ __forceinline__ void constrain(const double* q, int lane, F put) const {
    const double tS = wn_readlane(sel4(q, kT), laneT);
    const double sigma = wn_exp(-0.5 * tS), gpri = wn_exp(tS);
    const int t0 = TPL * lane + 1;
    double uz[TPL], ux[TPL], tin[TPL], z[TPL], x[TPL], tau[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      const double rz = t <= T - 1 ? q[k] : 0.0, rx = q[4 + k];
      tin[k] = q[8 + k];
      uz[k] = t == 1 ? rz : sigma * rz;
      ux[k] = t == 1 ? rx : sigma * rx;
    }
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) { run += uz[k]; z[k] = run; }
    double base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) z[k] += base;
    run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) { run += ux[k]; x[k] = run; }
    base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) x[k] += base;
    double zprev[TPL];
    zprev[0] = wn_wave_shr1(z[TPL - 1]);
#pragma unroll
    for (int k = 1; k < TPL; ++k) zprev[k] = z[k - 1];
    run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      const double e_ = wn_exp(0.5 * zprev[k]);
      run += t == 1 ? tin[k] : ((t >= 2 && t <= T) ? e_ : 0.0) * tin[k];
      tau[k] = run;
    }
    base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) tau[k] += base;
    if (lane == 0) { put(3 * T, sigma); put(3 * T + 1, gpri); }
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      if (t <= T - 1) put(3 * T + 1 + t, z[k]);
      if (t <= T) { put(4 * T + t, x[k]); put(5 * T + t, tau[k]); }
    }
  }
  __device__ __forceinline__ double grad(const double* q, double* g, int lane) const {
    const double tS = wn_readlane(sel4(q, kT), laneT);
    const double sigma = wn_exp(-0.5 * tS), gpri = wn_exp(tS);
    const int t0 = TPL * lane + 1;  // first time point of this lane (1-based)
    // (t = 4 lane + k + 1: t == 1 is register k = 0 of lane 0 and nothing else, t >= 2 holds for every k >= 1 -- written so that the
    //  compiler sees it: 24 of the evaluation's 73 selects go away, round 5)
    const bool first = t0 == 1;  // this lane's register 0 is t = 1
    double rz[TPL], rx[TPL], tin[TPL], uz[TPL], ux[TPL], z[TPL], x[TPL], tau[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      // raw parameters of the three blocks at time t (first entry of a block = initial state; slots beyond the series hold zeros,
      // the z slot t = T holds tSigma)
      rz[k] = t <= T - 1 ? q[k] : 0.0;       // z1 | zinn[t-1]
      rx[k] = q[4 + k];                      // x1 | xinn[t-1]
      tin[k] = q[8 + k];                     // tau1 | tauinn[t-1]
      uz[k] = (k == 0 && first) ? rz[k] : sigma * rz[k];
      ux[k] = (k == 0 && first) ? rx[k] : sigma * rx[k];
    }
    // z[t], x[t]: inclusive sums over time (sw_innov.stan:27-31)
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) { run += uz[k]; z[k] = run; }
    double base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) z[k] += base;
    run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) { run += ux[k]; x[k] = run; }
    base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) x[k] += base;
    // tau[t] = tau[t-1] + exp(z[t-1]/2) * tauinn[t-1]  (:33-34): needs z one step back
    double zprev[TPL], ezp[TPL];
    zprev[0] = wn_wave_shr1(z[TPL - 1]);
#pragma unroll
    for (int k = 1; k < TPL; ++k) zprev[k] = z[k - 1];
    run = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      const double e_ = wn_exp(0.5 * zprev[k]);
      ezp[k] = ((k > 0 || !first) && t <= T) ? e_ : 0.0;
      run += (k == 0 && first) ? tin[k] : ezp[k] * tin[k];
      tau[k] = run;
    }
    base = wn_scan_up_b(wn_wave_shr1(run));
#pragma unroll
    for (int k = 0; k < TPL; ++k) tau[k] += base;
    // log density (:40-49) and the adjoints of tau[t], x[t]
    double lp = 0.0, a[TPL], b[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      const bool on = t <= T;
      const double w = wn_exp(-x[k]), res = y[k] - tau[k];
      const double rt = ((k > 0 || !first) && on) ? tin[k] : 0.0;
      lp += on ? (-0.5 * x[k] - 0.5 * (res * res) * w) : 0.0;
      lp += -0.5 * (rt * rt);
      a[k] = on ? res * w : 0.0;
      b[k] = on ? (-0.5 + 0.5 * (res * res) * w) : 0.0;
    }
    // innovation priors of z and x (:44-45) on the raw parameters
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const double pz = (k > 0 || !first) ? rz[k] : 0.0;
      const double px = (k > 0 || !first) ? rx[k] : 0.0;
      lp += -0.5 * (pz * pz) - 0.5 * (px * px);
    }
    lp += lane == 0 ? 5.0 * tS - 0.5 * gpri : 0.0;
    // backward: A[t] = sum_{t' >= t} a[t'], B[t] = sum_{t' >= t} b[t']
    double A[TPL], B[TPL];
    run = 0.0;
#pragma unroll
    for (int k = TPL - 1; k >= 0; --k) { run += a[k]; A[k] = run; }
    base = wn_scan_down(wn_wave_shl1(run), lane);
#pragma unroll
    for (int k = 0; k < TPL; ++k) A[k] += base;
    run = 0.0;
#pragma unroll
    for (int k = TPL - 1; k >= 0; --k) { run += b[k]; B[k] = run; }
    base = wn_scan_down(wn_wave_shl1(run), lane);
#pragma unroll
    for (int k = 0; k < TPL; ++k) B[k] += base;
    // cc[t] = d/dz[t-1] = 1/2 exp(z[t-1]/2) tauinn[t-1] A[t]  (t = 2..T); CS[t] = sum_{t' > t} cc[t']
    double cc[TPL], CS[TPL], dsig = 0.0;
#pragma unroll
    for (int k = 0; k < TPL; ++k) cc[k] = 0.5 * ezp[k] * tin[k] * A[k];  // (ezp = 0 outside t = 2..T)
    run = 0.0;
#pragma unroll
    for (int k = TPL - 1; k >= 0; --k) { CS[k] = run; run += cc[k]; }
    base = wn_scan_down(wn_wave_shl1(run), lane);
#pragma unroll
    for (int k = 0; k < TPL; ++k) CS[k] += base;
    const double z1 = wn_readlane(z[0], 0), x1 = wn_readlane(x[0], 0);
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      dsig += cc[k] * (zprev[k] - z1);      // (cc = 0 outside t = 2..T)
      dsig += b[k] * (x[k] - x1);           // (b = 0 beyond T)
    }
    dsig = wn_wavesum(dsig);
    // gradients, where the parameters are
    const double gS = 5.0 - 0.5 * gpri - 0.5 * dsig;
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
      const int t = t0 + k;
      const double gz = (k == 0 && first) ? CS[k] : sigma * CS[k] - rz[k];
      g[k] = t <= T - 1 ? gz : (t == T ? gS : 0.0);
      const double gx = (k == 0 && first) ? B[k] : sigma * B[k] - rx[k];
      g[4 + k] = t <= T ? gx : 0.0;
      const double gt = (k == 0 && first) ? A[k] : ezp[k] * A[k] - tin[k];
      g[8 + k] = t <= T ? gt : 0.0;
    }
    return lp;
  }
};

// funnel10 (targetDistr.py:74-78) for one chain per wave: lane i < 11 holds q_i.  Chains of a funnel
// run need wildly different numbers of gradient evaluations (the neck), and a wave that hosts 64
// chains lasts as long as its busiest one (measured lane occupancy 4 %); one chain per wave lets every
// chain run at its own pace and the dispatcher balance the rest.
struct WTgtFunnel10 {
  static constexpr int EXT = EXT_NONE;
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 16;  // D <= 16: the chain's vectors live in LDS, 16 doubles apart (WavePolicy::VLDS)
  __device__ __forceinline__ void init(const Params&, int, double*) {}
  __device__ __forceinline__ static double lp_finish(double r) { return r; }
  __device__ __forceinline__ double grad(const double* q, double* g, int lane) const {
    const double logC = 0.91893853320467274178;  // log(sqrt(2 pi))
    const double log3 = 1.09861228866810969140;
    const double q0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(q[0])),
                                       __builtin_amdgcn_readfirstlane(__double2loint(q[0])));  // lane 0 holds q_0
    const double emq = wn_exp(-q0);
    const double ss = wn_rowsum0((lane >= 1 && lane < 11) ? q[0] * q[0] : 0.0);
    // (q0 / 9 and q0^2 / 18 as multiplications by the rounded reciprocals: two f64 divisions were a fifth of the
    // instructions of an evaluation; the result differs from the division's by at most one ulp, as the device exp()
    // differs from libm's)
    // (select form: both arms computed by every lane.  As `lane == 0 ? <expr> : <expr>` the compiler built a divergent if / else --
    //  two exec-mask regions and a branch per evaluation, ~100 cycles for a wave alone on its SIMD, tools/branch_cost.hip)
    double g_first = -5.0 - q0 * (1.0 / 9.0) + 0.5 * emq * ss, g_rest = -q[0] * emq;
    asm volatile("" : "+v"(g_first), "+v"(g_rest));  // (opaque: the compiler sank the two expressions back into an if / else)
    g[0] = lane == 0 ? g_first : (lane < 11 ? g_rest : 0.0);
    const double lp = (-(q0 * q0) * (1.0 / 18.0) - logC - log3) + (-0.5 * emq * ss - 10.0 * logC - 5.0 * q0);
    // s = np.exp(0.5*q0) overflowing makes the reference's lp -inf (targetDistr.py:75, WALNUTS.py:316)
    const double r = (0.5 * q0 > 709.782712893384) ? -INFINITY : lp;
    return lane == 0 ? r : 0.0;
  }
  // One whole leapfrog micro step (adaptiveIntegrators.py:72-84) with ONE row reduction (wn_k_orbit.hip): only lane 0's gradient needs
  // ss = sum q_i^2, so the kinetic energy of lanes 1..10 -- their second kick needs exp(-q0) alone -- rides in the same packed
  // reduction as ss, and lane 0's share is added afterwards.  Returns lp and KE complete (the same in every lane).  The sums are
  // associated differently from grad() + a separate (lp, KE) reduction: same values to the last bits, not the same bits.
  static constexpr bool FUSED_STEP = true;
  __device__ __forceinline__ void leap(double& q, double& v, double& g, double hh2, double hq, int lane, double& lp_out, double& ke_out) const {
    const double logC = 0.91893853320467274178, log3 = 1.09861228866810969140;
    v = v + hh2 * g; q = q + hq * v;
    const double q0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(q)), __builtin_amdgcn_readfirstlane(__double2loint(q)));
    const double emq = wn_exp(-q0);
    const bool rest = lane >= 1 && lane < 11;
    const double g_rest = rest ? -q * emq : 0.0;
    const double v_rest = v + hh2 * g_rest;
    double r = wn_pack32(rest ? q * q : 0.0, rest ? v_rest * v_rest : 0.0);
    r = wn_rowsum16(r);
    const double ss = wn_readlane(r, 0) + 0.0, ke_rest = wn_readlane(r, 32) + 0.0;
    const double g_first = -5.0 - q0 * (1.0 / 9.0) + 0.5 * emq * ss;
    double v_first = v + hh2 * g_first;  // (lane 0's value is the one that counts)
    asm volatile("" : "+v"(v_first));
    g = lane == 0 ? g_first : g_rest;
    v = lane == 0 ? v_first : v_rest;
    const double v0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    ke_out = ke_rest + v0 * v0;
    const double lp = (-(q0 * q0) * (1.0 / 18.0) - logC - log3) + (-0.5 * emq * ss - 10.0 * logC - 5.0 * q0);
    lp_out = (0.5 * q0 > 709.782712893384) ? -INFINITY : lp;
  }
};

// W waves (one workgroup) serve a chain: register e of wave w, lane l holds element (e*W + w)*64 + l.
// The 2-D toy targets of targetDistr.py (corrGauss, smileDistr, funnel1, modFunnel) for one chain per wave:
// lanes 0 and 1 hold q_0, q_1; every lane evaluates the lane-layout target on the broadcast pair (the same
// arithmetic, so the same bits) -- chosen when there are too few chains to fill the chip 64 per wave.
template <class LaneTgt>
struct WTgtPair {
  static constexpr int EXT = EXT_NONE;
  static constexpr int LDS_DOUBLES = 0;
  static constexpr int VEC_LDS = 16;  // D <= 16: the chain's vectors live in LDS, 16 doubles apart (WavePolicy::VLDS)
  __device__ __forceinline__ void init(const Params&, int, double*) {}
  __device__ __forceinline__ static double lp_finish(double r) { return r; }
  __device__ __forceinline__ double grad(const double* q, double* g, int lane) const {
    const double qq[2] = {wn_readlane(q[0], 0), wn_readlane(q[0], 1)};
    double gg[2];
    const double lp = LaneTgt::eval(qq, gg, 2, nullptr);
    g[0] = lane == 0 ? gg[0] : (lane == 1 ? gg[1] : 0.0);
    return lane == 0 ? lp : 0.0;
  }
};

// CL_: a CLUSTER of workgroups serves a chain -- workgroup `slice` of G holds elements [slice*DP, (slice+1)*DP)
// in its registers, every workgroup replicates the chain's scalars, and the only exchange is the sum / max
// of the per-workgroup partials through global memory (cl_exchange).  All G workgroups of a chain must be
// resident (the launcher checks the grid against the chip) and every spin is bounded.
template <int NE_, class Tgt, int W_ = 1, bool CL_ = false>
struct WavePolicy {
  static constexpr int NE = NE_;
  static constexpr int W = W_;
  static constexpr int ES = 64 * W_;    // element stride between registers
  static constexpr int DP = ES * NE_;
  static constexpr int NLC = 0;         // every sub-tree left endpoint lives in the slot store
  static constexpr int ALIGN_LOG2 = 0;  // chains do not share a wave: nothing to align
  static constexpr bool LANE_RNG = true; // uniform(): 64 uniforms (32 Philox blocks) at a time, one per lane
  static constexpr int NRED = 8;        // widest reduction (wn_chain.h NPB)
  static_assert(!CL_ || W_ > 1, "the cluster exchange hands results to the waves through the reduction LDS");
  static constexpr int LDS_DOUBLES = Tgt::LDS_DOUBLES + (W_ > 1 ? 2 * NRED * W_ : 0) + (CL_ ? NRED + 1 : 0);
  // Targets of at most 16 dimensions (one register per vector, lanes 0..15): every vector of the chain -- sub-tree
  // slots, orbit ends, proposals, the adaptive integrator's saved states -- lives in (dynamic) LDS, VS doubles apart,
  // instead of the workspace in memory: a tick then waits for no global load (they were a third of its time).
  static constexpr int VS = Tgt::VEC_LDS;
  static constexpr bool VLDS = VS > 0 && NE_ == 1 && W_ == 1 && !CL_;
  typedef __attribute__((address_space(3))) double lds_double;
  static size_t dyn_lds_bytes(int M) { return VLDS ? nvec(M) * VS * sizeof(double) : 0; }
  struct Vec {
    double x[NE_];
  };
  template <bool B>
  using Opt = typename std::conditional<B, Vec, EmptyVecW>::type;
  enum { NPARK = 9 };

  int D, cl, lane, wave, l0;
  int G, slice, goff;      // cluster: workgroups per chain, this one's index, its first element
  size_t vstride;          // doubles between consecutive vectors of a chain
  double* cred;            // cluster: partials [chain][2][G][NRED]
  unsigned* ccnt;          // cluster: arrival counter per chain
  int* cabort;             // cluster: set when a bounded spin gave up
  unsigned cdrop;          // test hook: exchange at which the last slice does not arrive (0 = never)
  unsigned epoch;
  bool aborted;
  unsigned long long t_start, t_real;  // diagnostic builds (-DWN_STAMPS)
  uint32_t cg;
  int nslot;
  int redpar;
  double* ldsr;      // cross-wave reduction buffers [2][NRED][W]
  // uniform(): lane l holds uniform 2 * rb_base + l of iteration rb_iter (wn_rng.h stream 0)
  uint32_t rb_iter, rb_base;
  double ru0;
  lds_double* lv;    // VLDS: this chain's vectors in LDS, [vector][VS]
  double* vbase;     // this chain's vectors: [2*nslot + COLD_N + 4 + WARM_N + NPARK][DP]
  ChainScalars* sc;  // this chain's parked scalars
  double *xq, *xg;   // split mode: position out / gradient in, [C][DP]
  const double* xlp; // split mode, EXT_LP: log density in, [C]
  Tgt tgt;

  __host__ __device__ static size_t nvec(int M) { return (size_t)(M + 1) * 2 + COLD_N + 4 + WARM_N + NPARK; }
  __host__ __device__ static size_t scalar_doubles() { return (sizeof(ChainScalars) + 63) / 64 * 8; }
  // workspace doubles for C chains: vectors, then parked scalars
  static size_t work_doubles(int C, int M, int G = 1) { return (size_t)C * (nvec(M) * DP * G + scalar_doubles()); }

  __device__ WavePolicy(const Params& prm, int block, double* lds = nullptr, double* dyn_lds = nullptr)
      : D(prm.D), lane(threadIdx.x & 63), wave(W_ > 1 ? (int)(threadIdx.x >> 6) : 0), nslot(prm.M + 1), redpar(0) {
    t_start = 0; t_real = 0;
#if defined(WN_STAMPS)
    t_real = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
#endif
    G = CL_ ? prm.cl_g : 1;
    cl = CL_ ? block / G : block;
    slice = CL_ ? block % G : 0;
    goff = slice * DP;
    cg = prm.chain0 + (uint32_t)cl;
    vstride = (size_t)DP * G;
    cred = prm.cl_red; ccnt = prm.cl_cnt; cabort = prm.cl_abort; cdrop = (unsigned)prm.cl_debug_drop;
    epoch = 0; aborted = false;
    l0 = wave * 64 + lane;
    ldsr = lds + Tgt::LDS_DOUBLES;
    const int c = cl < prm.C ? cl : 0;
    vbase = prm.work + (size_t)c * nvec(prm.M) * vstride + goff + l0;
    sc = reinterpret_cast<ChainScalars*>(prm.work + (size_t)prm.C * nvec(prm.M) * vstride + (size_t)c * scalar_doubles());
    xq = prm.xq ? prm.xq + (size_t)c * DP + l0 : nullptr;
    xg = prm.xg ? prm.xg + (size_t)c * DP + l0 : nullptr;
    xlp = prm.xlp ? prm.xlp + c : nullptr;
    tgt.init(prm, goff + l0, lds);
    rb_iter = 0xFFFFFFFFu; rb_base = 0u; ru0 = 0.0;
    lv = nullptr;
    if constexpr (VLDS) {
      lv = (lds_double*)dyn_lds;
      for (int k = lane; k < (int)nvec(prm.M) * VS; k += 64) lv[k] = 0.0;
    }
  }

  // (a MAPPED target says itself which element register e of a lane holds: time-major layout of the StockWatson series)
  template <class T_, class = void> struct is_mapped { static constexpr bool value = false; };
  template <class T_> struct is_mapped<T_, typename std::enable_if<T_::MAPPED>::type> { static constexpr bool value = true; };
  __device__ __forceinline__ bool valid(int e) const {
    if constexpr (is_mapped<Tgt>::value) return tgt.map_gidx(e, lane) >= 0;
    else return goff + e * ES + l0 < D;
  }
  __device__ __forceinline__ bool owner(int) const { return true; }
  __device__ __forceinline__ int gidx(int e) const {
    if constexpr (is_mapped<Tgt>::value) return tgt.map_gidx(e, lane);
    else return goff + e * ES + l0;
  }
  __device__ __forceinline__ bool writer() const { return l0 == 0 && slice == 0; }
  // per-workgroup writer and slot of per-chain scratch that every replica recomputes (warm-up quantile store)
  __device__ __forceinline__ bool wg_writer() const { return l0 == 0; }
  __device__ __forceinline__ int facs_slot() const { return cl * G + slice; }
  __device__ __forceinline__ int chain_local() const { return cl; }
  __device__ __forceinline__ uint32_t chain_global() const { return cg; }
  // the scalars are replicated: every lane of every wave votes the same, and the vote makes the
  // branch scalar
  __device__ __forceinline__ bool any(bool b) const { return (CL_ && aborted) ? false : __any(b); }
  __device__ __forceinline__ bool majority(bool a, bool) const { return any(a); }

  // Cluster step of a reduction: this workgroup's N partials go to global memory (agent-scope 8-byte
  // stores), one arrival is counted, everybody polls the counter (relaxed agent-scope loads, s_sleep,
  // bounded), then wave 0 sums / maxes the G partials lane-parallel -- the same order in every workgroup,
  // so all replicas of the chain get the same bits -- and hands them to the other waves through LDS.
  // Two parity buffers: a workgroup can overwrite parity p only after the next reduction's wait, which
  // everybody passes only after having read parity p.
  template <int N, bool MAXOP>
  __device__ __forceinline__ void cl_exchange(double* v) {
    epoch++;
    double* slot = cred + ((size_t)cl * 2 + (epoch & 1u)) * (size_t)G * NRED;
    if (l0 == 0) {
#pragma unroll
      for (int k = 0; k < N; ++k) __hip_atomic_store(slot + (size_t)slice * NRED + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // publication order: 8-byte agent-scope atomics on both sides, the stores drained before the arrival
      // (MI355X_MICROARCH "Valid forms": {8-B agent atomics both sides} + every storing wave's vmcnt(0) before
      // its counter add).  A release fence here would add an L2 write-back (~1.7 us) to every reduction.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(cdrop && epoch == cdrop && slice == G - 1))
        __hip_atomic_fetch_add(ccnt + cl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      const unsigned target = epoch * (unsigned)G;
      unsigned spins = 0;
      bool bad = false;
      // wrap-safe: the counter runs ahead of `target` by less than 2^31 (at most G arrivals of the next epoch)
      while ((int)(__hip_atomic_load(ccnt + cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(WN_CL_SLEEP);
        if (++spins > (1u << 22) || __hip_atomic_load(cabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = true; break; }
      }
      if (bad) __hip_atomic_store(cabort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      double* lbuf = ldsr + 2 * NRED * W_;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double x = MAXOP ? -INFINITY : 0.0;
        if (lane < G && !bad) x = __hip_atomic_load(slot + (size_t)lane * NRED + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = MAXOP ? wavemax(x) : wn_wavesum(x);
        if (lane == 0) lbuf[k] = x;
      }
      if (lane == 0) lbuf[NRED] = bad ? 1.0 : 0.0;
    }
    __syncthreads();
    const double* lbuf = ldsr + 2 * NRED * W_;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lbuf[k];
    if (lbuf[NRED] != 0.0) aborted = true;
    __syncthreads();
  }

  // sum over all lanes serving the chain; every replica gets the same bits.  Two buffers alternate:
  // a wave can reuse a buffer only after the barrier of the next reduction, behind which every wave
  // has finished reading it.
  template <int N>
  __device__ __forceinline__ void reduce(double* v) {
    static_assert(N <= NRED, "reduce width");
    if constexpr (VLDS) {
      // Only row 0 (lanes 0..15) holds data.  Four quantities are packed into the four rows of one register (three
      // permlane swaps), reduced together by ONE row reduction, and read back with v_readlane: 26 instructions for
      // four sums instead of 72, and the same bits as wn_wavesum gives for a quantity whose rows 1..3 are zero.
#pragma unroll
      for (int k0 = 0; k0 < N; k0 += 4) {
        const int m = N - k0 < 4 ? N - k0 : 4;
        double r;
        if (m == 1) r = v[k0];
        else if (m == 2) r = wn_pack32(v[k0], v[k0 + 1]);
        else r = wn_pack16(wn_pack32(v[k0], v[k0 + 2]), wn_pack32(v[k0 + 1], v[k0 + (m == 4 ? 3 : 2)]));
        r = wn_rowsum16(r);
#pragma unroll
        for (int k = 0; k < m; ++k) {
          const int src = m == 2 ? 32 * k : 16 * k;
          v[k0 + k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r), src), __builtin_amdgcn_readlane(__double2loint(r), src)) + 0.0;
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wn_wavesum(v[k]);
    if (W_ > 1) {
      double* buf = ldsr + (redpar & 1) * (NRED * W_);
      redpar++;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) buf[k * W_ + wave] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < W_; ++w) s += buf[k * W_ + w];
        v[k] = s;
      }
    }
    if (CL_) cl_exchange<N, false>(v);
  }

  // max over all lanes serving the chain (NaN-free inputs; max is exact, so every replica agrees)
  __device__ __forceinline__ static double wavemax(double x) {
    x = fmax(x, wn_dpp<0xB1>(x));
    x = fmax(x, wn_dpp<0x4E>(x));
    x = fmax(x, wn_dpp<0x141>(x));
    x = fmax(x, wn_dpp<0x140>(x));
#pragma unroll
    for (int d = 16; d < 64; d <<= 1) x = fmax(x, __shfl_xor(x, d));
    return x;
  }
  template <int N>
  __device__ __forceinline__ void reduce_max(double* v) {
    static_assert(N <= NRED, "reduce width");
    if constexpr (VLDS) {  // lanes >= VS hold copies of row-0 elements (vget): they count as the zeros they used to be
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = l0 < VS ? v[k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wavemax(v[k]);
    if (W_ > 1) {
      double* buf = ldsr + (redpar & 1) * (NRED * W_);
      redpar++;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) buf[k * W_ + wave] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double s = buf[k * W_];
#pragma unroll
        for (int w = 1; w < W_; ++w) s = fmax(s, buf[k * W_ + w]);
        v[k] = s;
      }
    }
    if (CL_) cl_exchange<N, true>(v);
  }

  // k-th uniform of (chain, iteration): the chain's scalars are replicated over the wave, so instead of every lane
  // running the same Philox block, lane l runs block base + l and the draw is a v_readlane (the generic window of
  // wn_chain.h costs one block, ~120 instructions, per two uniforms).  The window is moved at ONE code site per macro
  // step (rng_ensure, from Engine::rng_refill / start_iteration) -- a step draws at most five uniforms, the window is
  // kept four blocks ahead -- so the draws themselves carry no refill branch and no second copy of Philox.
  __device__ __forceinline__ void rng_ensure(const Params& prm, int iter, uint32_t k) {
    const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 1));
    const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane(iter);
    if (it != rb_iter || blk - rb_base > 28u) {  // blocks blk .. blk + 3 must be inside [rb_base, rb_base + 32)
      rb_iter = it; rb_base = blk;
      // lane l holds uniform 2 * rb_base + l: word pair l & 1 of block rb_base + (l >> 1) (the select is paid here, once
      // per 56 uniforms, not at every draw)
      Philox4 r = philox4x32_10(blk + (uint32_t)(lane >> 1), it, cg, 0u, (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32));
      ru0 = (lane & 1) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
    }
  }
  __device__ __forceinline__ double uniform(const Params&, int, uint32_t k) const {
    const int l = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)k) - 2u * rb_base);
    return wn_readlane(ru0, l);
  }
  __device__ __forceinline__ void grad(const Vec& q, Vec& g, double& lp_part) const { lp_part = tgt.grad(q.x, g.x, lane); }
  __device__ __forceinline__ double lp_finish(double r) const { return Tgt::lp_finish(r); }

  // ---- memory-resident vectors ------------------------------------------------------------------
  __device__ __forceinline__ double* vp(int v) const { return vbase + (size_t)v * vstride; }
  __device__ __forceinline__ void vput(int v, const Vec& x) const {
    if constexpr (VLDS) {
      if (l0 < VS) lv[v * VS + l0] = x.x[0];
      return;
    }
    double* d = vp(v);
#pragma unroll
    for (int e = 0; e < NE; ++e) d[e * ES] = x.x[e];
  }
  __device__ __forceinline__ void vget(int v, Vec& x) const {
    if constexpr (VLDS) {
      // (every lane loads: lanes >= VS get a copy of a row-0 element -- a masked load is an exec save / restore around
      //  it.  Nothing reads those lanes: stores, global I/O and the reductions of this layout take row 0 only.)
      x.x[0] = lv[v * VS + (l0 & (VS - 1))];
      return;
    }
    const double* s = vp(v);
#pragma unroll
    for (int e = 0; e < NE; ++e) x.x[e] = s[e * ES];
  }
  // two vectors under ONE lane mask (VLDS: a masked store is an exec save / restore around it)
  __device__ __forceinline__ void vput2(int v1, const Vec& x1, int v2, const Vec& x2) const {
    if constexpr (VLDS) {
      if (l0 < VS) { lv[v1 * VS + l0] = x1.x[0]; lv[v2 * VS + l0] = x2.x[0]; }
      return;
    }
    vput(v1, x1); vput(v2, x2);
  }
  __device__ __forceinline__ void slot_store(int slot, const Vec& q, const Vec& v) const { vput2(2 * slot, q, 2 * slot + 1, v); }
  __device__ __forceinline__ void slot_load(int slot, Vec& q, Vec& v) const { vget(2 * slot, q); vget(2 * slot + 1, v); }
  __device__ __forceinline__ void lslot_store(int, const Vec&, const Vec&) const {}
  __device__ __forceinline__ void lslot_load(int, Vec&, Vec&) const {}
  template <int K> __device__ __forceinline__ void cold_put(const Vec& x) const { vput(2 * nslot + K, x); }
  template <int K> __device__ __forceinline__ void cold_get(Vec& x) const { vget(2 * nslot + K, x); }
  __device__ __forceinline__ void prop_put(int which, const Vec& q, const Vec& g) const {
    vput2(2 * nslot + COLD_N + 2 * which, q, 2 * nslot + COLD_N + 2 * which + 1, g);
  }
  __device__ __forceinline__ void prop_get(int which, Vec& q, Vec& g) const {
    vget(2 * nslot + COLD_N + 2 * which, q); vget(2 * nslot + COLD_N + 2 * which + 1, g);
  }
  template <int K> __device__ __forceinline__ void warm_put(const Vec& x) const { vput(2 * nslot + COLD_N + 4 + K, x); }
  template <int K> __device__ __forceinline__ void warm_get(Vec& x) const { vget(2 * nslot + COLD_N + 4 + K, x); }
  // one 192-byte row: lane k < 24 stores d[k]
  __device__ __forceinline__ void diag_store(double* row, const double* d) const {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < NDIAG; ++k) v = l0 == k ? d[k] : v;
    if (l0 < NDIAG && slice == 0) row[l0] = v;
  }

  // ---- split mode ---------------------------------------------------------------------------------
  // StockWatson's param_constrain of an orbit state for the orbit statistics (Engine::orbit_stats): the target's own forward sums
  template <class T_, class = void> struct has_constrain { static constexpr bool value = false; };
  template <class T_> struct has_constrain<T_, typename std::enable_if<T_::HAS_CONSTRAIN>::type> { static constexpr bool value = true; };
  template <class F>
  __device__ __forceinline__ void sw_constrained(const Vec& q, F put) const {
    if constexpr (has_constrain<Tgt>::value && W_ == 1) tgt.constrain(q.x, lane, put);
  }
  template <int K> __device__ __forceinline__ void park_vec(const Vec& x) const { vput(2 * nslot + COLD_N + 4 + WARM_N + K, x); }
  template <int K> __device__ __forceinline__ void unpark_vec(Vec& x) const { vget(2 * nslot + COLD_N + 4 + WARM_N + K, x); }
  __device__ __forceinline__ double park_partial(double x) { double v[1] = {x}; reduce<1>(v); return v[0]; }
  __device__ __forceinline__ double unpark_partial(double x) const { return (l0 == 0 && slice == 0) ? x : 0.0; }
  __device__ __forceinline__ void park_scalars(const ChainScalars& s) const { if (l0 == 0) ChainScalars::copy(*sc, s); }
  __device__ __forceinline__ void unpark_scalars(ChainScalars& s) const { ChainScalars::copy(s, *sc); }
  __device__ __forceinline__ void ext_put_q(const Vec& q) const {
#pragma unroll
    for (int e = 0; e < NE; ++e) xq[e * ES] = q.x[e];  // the padding beyond D stays zero (vectors are zero there)
  }
  __device__ __forceinline__ void ext_get_q(Vec& q) const {
#pragma unroll
    for (int e = 0; e < NE; ++e) q.x[e] = xq[e * ES];
  }
  __device__ __forceinline__ void ext_zero_g() const {
#pragma unroll
    for (int e = 0; e < NE; ++e) xg[e * ES] = 0.0;
  }
  // gradient (and log density) of the position sent out by the previous launch
  __device__ __forceinline__ void ext_load_g(Vec& g) const {
#pragma unroll
    for (int e = 0; e < NE; ++e) g.x[e] = valid(e) ? xg[e * ES] : 0.0;
  }
  __device__ __forceinline__ void ext_lp(const Vec& q, const Vec& g, double& lp_part) const {
    if (Tgt::EXT == EXT_LP) {
      lp_part = l0 == 0 ? *xlp : 0.0;
    } else {
      double s = 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) s += q.x[e] * g.x[e];
      lp_part = s;
    }
  }
};

}  // namespace wn
