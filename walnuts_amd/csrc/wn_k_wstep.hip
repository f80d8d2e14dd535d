// wn_k_wstep.hip -- batched device form of the pseudocode-faithful variant walnuts/walnuts.py:
// walnuts_step (:279-359) for B independent chains of any dimension.
//
// What runs where.  The caller's numpy Generators must be consumed in the reference's order, and that order depends
// on the data (no uniform after a sub-U-turn :343, the categorical draw only after an accept :347-350), so one NUTS
// doubling is one exchange with the host:
//
//   host:   normal(size=D) per chain (:325)                                     -> wn_wstep_begin
//   per doubling:
//     host:   binomial(1, .5) (:330) and one choice(3) per macro step (:194)    -> wn_wstep_extend
//     device: extend_orbit (:211-276: stable_steps :144-182 both ways, leapfrog :74-95, weights :264-271),
//             sub_uturn of the extension (:62-70), logsumexp of its weights and of the old ones (:345)
//     host:   uniform (:346) for chains without a sub-U-turn, accept test      -> wn_wstep_commit
//     device: softmax / cumulative sum / search of the categorical pick (:348-350: numpy Generator.choice's
//             inversion with the host's uniform), orbit merge (:351), U-turn of the whole orbit (:352)
//
// One thread per chain; every D-vector lives in the handle's workspace, laid out [vector][element][chain] so that
// the threads of a wave touch consecutive addresses.  extend_orbit is written as a MACHINE that stops whenever it
// needs (logp, grad) at a new position: with a built-in target the kernel evaluates it in place and carries on
// (fused); with WN_T_EXTERNAL the kernel returns, the caller evaluates its Python (logp, grad) pair for the
// requested positions of all chains, and the next launch carries on (wn_wstep_ext_*) -- the reference's
// (logp, grad) callable contract (:379-380) at the reference's granularity.
#include <math.h>
#include <string.h>

#include <string>

#include "../../include/walnuts_amd.h"
#include "wn_launch.h"
#include "wn_target_eval.h"

using namespace wn;

namespace {

enum { V_TH0 = 0, V_RH0, V_G0, V_TH, V_RH, V_G, V_SEL, V_THL, V_RHL, V_GL, V_THR, V_RHR, V_GR, V_NVEC };
enum { ST_IDLE = 0, ST_INIT, ST_SF, ST_LF, ST_SB, ST_DONE };

struct WScal {  // per-chain scalars of the machine and of the orbit
  double lp0, lp, Hmin, Hmax, step, half, weight, p0;
  double lpL, lpR, wL, wR;
  int stage, n, k, ell, lf_ell, es, m, nmacro, backward, active, nw, sub_uturn;
  int sum_ell, sum_es;  // (statistics of wn_wstep_set_stats: chosen micro-step counts and forward ell_stable, summed over the macro steps)
};

struct WCtx {
  int kind, D, B, nmax;
  const double* tpar;
  const double* im;
  double macro_step, max_error;
  double* vec;   // [V_NVEC][D][B]
  double* eorb;  // [nmax][2][D][B] states of the extension in integration order
  double* ew;    // [nmax][B]       their log weights
  double* logw;  // [2*nmax][B]     log weights of the orbit so far (any order)
  WScal* sc;     // [B]
  const int* lottery;  // [B][nmax]
  double* xreq;  // external target: [B][D] positions to evaluate, chain-major
  const double* xlp;  // external: [B] log densities, [B][D] gradients
  const double* xg;
  int* npend;    // external: number of chains still waiting for an evaluation
  int* waiting;  // external: [B] 1 for the chains that asked for an evaluation in the last launch
};

struct Vecs {
  const WCtx& c;
  int b;
  double* lds;  // null: the chain's vectors live in the workspace, strided by B; else: [V_NVEC][D] in LDS (one chain per workgroup)
  __device__ double& operator()(int v, int j) const { return lds ? lds[v * c.D + j] : c.vec[((size_t)v * c.D + j) * c.B + b]; }
};
// the same for wstep_auto_kernel, whose vectors are ALWAYS in LDS: an LDS-address-space pointer (ds_read / ds_write; through `Vecs` the
// compiler has a generic pointer and every access of the kernel's hot loops was a FLAT instruction -- 373 of them)
typedef __attribute__((address_space(3))) double lds_f64;
struct VecsL {
  const WCtx& c;
  lds_f64* lds;
  static constexpr int b = 0;  // (the kernel hands every function a one-chain view of the handle)
  __device__ lds_f64& operator()(int v, int j) const { return lds[v * c.D + j]; }
};

template <class VT>
__device__ double ke_of(const WCtx& c, const VT& V, int vr) {  // 0.5*dot(inv_mass, rho**2)  (:124)
  double s = 0.0;
  for (int i = 0; i < c.D; ++i) s += c.im[i] * (V(vr, i) * V(vr, i));
  return 0.5 * s;
}
__device__ double mlogp(int ell, int es) {  // micro_steps_logp :197-208
  return (ell == es || ell == es / 2 || ell == es * 2) ? -1.0986122886681098 : -INFINITY;
}

// start one pass of stable_steps (:163-170) from (TH0, sgn*RH0) with (lp0, G0) known; TH = next evaluation point
template <class VT>
__device__ void stable_setup(const WCtx& c, const VT& V, WScal& s, double sgn) {
  s.ell = 1 << s.n;
  s.step = c.macro_step / (double)s.ell;
  s.half = 0.5 * s.step;
  for (int i = 0; i < c.D; ++i) { V(V_TH, i) = V(V_TH0, i); V(V_RH, i) = sgn * V(V_RH0, i); }
  const double Hc = -s.lp0 + ke_of(c, V, V_RH);
  s.Hmin = s.Hmax = Hc;
  for (int i = 0; i < c.D; ++i) V(V_RH, i) = V(V_RH, i) + s.half * V(V_G0, i);
  s.k = 0;
  for (int i = 0; i < c.D; ++i) V(V_TH, i) = V(V_TH, i) + (s.step * c.im[i]) * V(V_RH, i);
}
template <class VT>
__device__ void leap_setup(const WCtx& c, const VT& V, WScal& s) {  // leapfrog :88-91 (ell == 0 -> infinite step, Q9)
  s.step = c.macro_step / (double)s.lf_ell;
  s.half = 0.5 * s.step;
  for (int i = 0; i < c.D; ++i) { V(V_TH, i) = V(V_TH0, i); V(V_RH, i) = V(V_RH0, i) + s.half * V(V_G0, i); }
  s.k = 0;
  for (int i = 0; i < c.D; ++i) V(V_TH, i) = V(V_TH, i) + (s.step * c.im[i]) * V(V_RH, i);
}
template <class VT>
__device__ void macro_setup(const WCtx& c, const VT& V, WScal& s) {  // :252-253
  s.p0 = s.lp0 - ke_of(c, V, V_RH0);
  s.stage = ST_SF; s.n = 0;
  stable_setup(c, V, s, 1.0);
}

// (lp, G) = (logp, grad) at TH have arrived: advance to the next evaluation point (true) or to the end (false)
template <class VT>
__device__ bool advance(const WCtx& c, const VT& V, WScal& s) {
  const int D = c.D;
  if (s.stage == ST_INIT) {  // logp / grad of the initial state (:326): orbit = [(theta, rho)], weights = [-H]
    for (int i = 0; i < D; ++i) {
      const double g = V(V_G, i), th = V(V_TH, i), rh = V(V_RH, i);
      V(V_GL, i) = g; V(V_GR, i) = g;
      V(V_THL, i) = th; V(V_THR, i) = th; V(V_RHL, i) = rh; V(V_RHR, i) = rh; V(V_SEL, i) = th;
    }
    s.lpL = s.lpR = s.lp;
    const double w = s.lp - ke_of(c, V, V_RH);
    s.wL = s.wR = w;
    c.logw[V.b] = w; s.nw = 1;
    s.stage = ST_DONE;
    return false;
  }
  if (s.stage == ST_SF || s.stage == ST_SB) {
    if (s.k < s.ell - 1) {  // :172-176
      for (int i = 0; i < D; ++i) V(V_RH, i) = V(V_RH, i) + s.half * V(V_G, i);
      const double Hc = -s.lp + ke_of(c, V, V_RH);
      s.Hmin = (Hc < s.Hmin) ? Hc : s.Hmin; s.Hmax = (Hc > s.Hmax) ? Hc : s.Hmax;  // Python min() / max() order
      for (int i = 0; i < D; ++i) V(V_RH, i) = V(V_RH, i) + s.half * V(V_G, i);
      s.k++;
      for (int i = 0; i < D; ++i) V(V_TH, i) = V(V_TH, i) + (s.step * c.im[i]) * V(V_RH, i);
      return true;
    }
    for (int i = 0; i < D; ++i) V(V_RH, i) = V(V_RH, i) + s.half * V(V_G, i);  // :178-181
    const double Hc = -s.lp + ke_of(c, V, V_RH);
    s.Hmin = (Hc < s.Hmin) ? Hc : s.Hmin; s.Hmax = (Hc > s.Hmax) ? Hc : s.Hmax;
    if (!(s.Hmax - s.Hmin <= c.max_error) && s.n < 10) {
      s.n++;
      stable_setup(c, V, s, s.stage == ST_SF ? 1.0 : -1.0);
      return true;
    }
    if (s.stage == ST_SF) {  // :256-258
      s.es = s.ell;
      const int idx = c.lottery[(size_t)V.b * c.nmax + s.m];
      s.lf_ell = idx == 0 ? s.es / 2 : idx == 1 ? s.es : s.es * 2;
      s.sum_ell += s.lf_ell; s.sum_es += s.es;
      s.stage = ST_LF;
      leap_setup(c, V, s);
      return true;
    }
    // ST_SB done: the macro step is complete (:264-273); (TH0, RH0, lp0, G0) hold its end state
    const int en = s.ell;
    const double p1 = s.lp0 - ke_of(c, V, V_RH0);
    s.weight = p1 - s.p0 + mlogp(s.lf_ell, en) - mlogp(s.lf_ell, s.es) + s.weight;
    double* eo = c.eorb + (size_t)s.m * 2 * D * c.B;
    for (int i = 0; i < D; ++i) { eo[(size_t)i * c.B + V.b] = V(V_TH0, i); eo[((size_t)D + i) * c.B + V.b] = V(V_RH0, i); }
    c.ew[(size_t)s.m * c.B + V.b] = s.weight;
    s.m++;
    if (s.m == s.nmacro) { s.stage = ST_DONE; return false; }
    macro_setup(c, V, s);
    return true;
  }
  // ST_LF
  if (s.k < s.lf_ell - 1) {  // :92-93
    for (int i = 0; i < D; ++i) V(V_RH, i) = V(V_RH, i) + s.step * V(V_G, i);
    s.k++;
    for (int i = 0; i < D; ++i) V(V_TH, i) = V(V_TH, i) + (s.step * c.im[i]) * V(V_RH, i);
    return true;
  }
  // :94-95 end of the macro step's integration; then stable_steps from (theta, -rho) (:261)
  for (int i = 0; i < D; ++i) {
    V(V_RH0, i) = V(V_RH, i) + s.half * V(V_G, i);
    V(V_TH0, i) = V(V_TH, i);
    V(V_G0, i) = V(V_G, i);
  }
  s.lp0 = s.lp;
  s.stage = ST_SB; s.n = 0;
  stable_setup(c, V, s, -1.0);
  return true;
}

// first evaluation point of an extension (:242-248): from the orbit's first state with the momentum negated, or
// from its last state
template <class VT>
__device__ void extend_setup(const WCtx& c, const VT& V, WScal& s, int backward, int nmacro) {
  s.backward = backward; s.nmacro = nmacro; s.m = 0; s.sub_uturn = 0;
  for (int i = 0; i < c.D; ++i) {
    V(V_TH0, i) = backward ? V(V_THL, i) : V(V_THR, i);
    V(V_RH0, i) = backward ? -V(V_RHL, i) : V(V_RHR, i);
    V(V_G0, i) = backward ? V(V_GL, i) : V(V_GR, i);
  }
  s.lp0 = backward ? s.lpL : s.lpR;
  s.weight = backward ? s.wL : s.wR;
  macro_setup(c, V, s);
}

// uturn(state x, state y) (:16-33) between states of the extension given in LIST order (time order: the
// integration order reversed when going backward, :274-275)
__device__ bool ext_uturn(const WCtx& c, int b, int backward, int n, int x, int y) {
  const int ix = backward ? n - 1 - x : x, iy = backward ? n - 1 - y : y;
  const double* sx = c.eorb + (size_t)ix * 2 * c.D * c.B;
  const double* sy = c.eorb + (size_t)iy * 2 * c.D * c.B;
  double d1 = 0.0, d2 = 0.0;
  for (int i = 0; i < c.D; ++i) {
    const double diff = c.im[i] * (sy[(size_t)i * c.B + b] - sx[(size_t)i * c.B + b]);
    d1 += sx[((size_t)c.D + i) * c.B + b] * diff;
    d2 += sy[((size_t)c.D + i) * c.B + b] * diff;
  }
  return d1 < 0 || d2 < 0;
}
// scipy.special.logsumexp of n values with stride B
__device__ double lse_of(const double* a, int n, int B) {
  double mx = -INFINITY;
  bool nan = false;
  for (int i = 0; i < n; ++i) { const double x = a[(size_t)i * B]; if (isnan(x)) nan = true; mx = x > mx ? x : mx; }
  if (nan) mx = NAN;
  if (!isfinite(mx)) mx = 0.0;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += exp(a[(size_t)i * B] - mx);
  return log(s) + mx;
}
// the extension is complete: sub_uturn (:62-70) and the two logsumexp of :345
__device__ void extend_finish(const WCtx& c, int b, WScal& s, int* sub_out, double* lse_ext, double* lse_old) {
  const int n = s.nmacro;
  bool su = false;
  for (int span = n; span >= 2 && !su; span >>= 1)
    for (int i = 0; i < n / span && !su; ++i) su = ext_uturn(c, b, s.backward, n, span * i, span * (i + 1) - 1);
  s.sub_uturn = su ? 1 : 0;
  sub_out[b] = s.sub_uturn;
  lse_ext[b] = lse_of(c.ew + b, n, c.B);
  lse_old[b] = lse_of(c.logw + b, s.nw, c.B);
}

// evaluate the built-in target at TH until the machine is done (the loop of wstep_fused_kernel)
__device__ void run_machine(const WCtx& c, const Vecs& V, WScal& s) {
  for (;;) {
    double x[64], g[64];
    if (c.D <= 64) {
      for (int i = 0; i < c.D; ++i) x[i] = V(V_TH, i);
      s.lp = wn_target_eval(c.kind, c.D, c.tpar, x, g);
      for (int i = 0; i < c.D; ++i) V(V_G, i) = g[i];
    } else {
      double lp = 0.0;
      for (int i = 0; i < c.D; ++i) {
        const double xi = V(V_TH, i);
        double gi;
        lp += wn_target_eval(c.kind, 1, c.tpar ? c.tpar + i : nullptr, &xi, &gi);
        V(V_G, i) = gi;
      }
      s.lp = lp;
    }
    if (!advance(c, V, s)) break;
  }
}

// the same with the chain's vectors contiguous in LDS (wstep_auto_kernel): evaluated in place, no private arrays (a kernel with
// per-thread arrays in scratch memory runs only a fraction of its waves at a time)
template <class VT>
__device__ void run_machine_lds(const WCtx& c, const VT& V, WScal& s) {
  for (;;) {
    s.lp = wn_target_eval_p(c.kind, c.D, c.tpar, &V(V_TH, 0), &V(V_G, 0));
    if (!advance(c, V, s)) break;
  }
}
__global__ void wstep_fused_kernel(WCtx c, int mode /*0 begin, 1 extend*/, const double* theta, const double* rho,
                                   const int* backward, const int* active, int nmacro, int* sub_out, double* lse_ext,
                                   double* lse_old) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.B) return;
  WScal& s = c.sc[b];
  Vecs V{c, b, nullptr};
  if (mode == 0) {
    for (int i = 0; i < c.D; ++i) { V(V_TH, i) = theta[(size_t)b * c.D + i]; V(V_RH, i) = rho[(size_t)b * c.D + i]; }
    s.stage = ST_INIT; s.active = 1; s.nmacro = 0; s.m = 0;
  } else {
    s.active = active[b];
    if (!s.active) return;
    extend_setup(c, V, s, backward[b], nmacro);
  }
  run_machine(c, V, s);  // (positions are strided in the workspace: gather, evaluate with the scalar target code of wn_capi.hip, scatter)
  if (mode == 1) extend_finish(c, b, s, sub_out, lse_ext, lse_old);
}

// external target, one machine step per launch: consume (lp, grad) of the requested point, advance, request the next
__global__ void wstep_ext_kernel(WCtx c, int mode /*0 begin, 1 extend setup, 2 feed*/, const double* theta, const double* rho,
                                 const int* backward, const int* active, int nmacro, int* sub_out, double* lse_ext,
                                 double* lse_old) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.B) return;
  c.waiting[b] = 0;
  WScal& s = c.sc[b];
  Vecs V{c, b, nullptr};
  bool need = false;
  if (mode == 0) {
    for (int i = 0; i < c.D; ++i) { V(V_TH, i) = theta[(size_t)b * c.D + i]; V(V_RH, i) = rho[(size_t)b * c.D + i]; }
    s.stage = ST_INIT; s.active = 1; s.nmacro = 0; s.m = 0;
    need = true;
  } else if (mode == 1) {
    s.active = active[b];
    if (!s.active) return;
    extend_setup(c, V, s, backward[b], nmacro);
    need = true;
  } else {
    if (!s.active || s.stage == ST_DONE || s.stage == ST_IDLE) return;
    s.lp = c.xlp[b];
    for (int i = 0; i < c.D; ++i) V(V_G, i) = c.xg[(size_t)b * c.D + i];
    need = advance(c, V, s);
    if (!need && s.nmacro > 0 && s.m == s.nmacro) extend_finish(c, b, s, sub_out, lse_ext, lse_old);
  }
  if (need) {
    for (int i = 0; i < c.D; ++i) c.xreq[(size_t)b * c.D + i] = V(V_TH, i);
    atomicAdd(c.npend, 1);
    c.waiting[b] = 1;
  }
}

// accept / categorical pick (:347-350), orbit merge (:351), U-turn of the whole orbit (:352)
__device__ int commit_chain(const WCtx& c, const int b, WScal& s, const bool accept_b, const double u_choice_b, double* vlds = nullptr) {
  if (!s.active || s.sub_uturn) return 0;  // a sub-U-turn breaks out before the merge (:343)
  Vecs V{c, b, vlds};
  const int n = s.nmacro, D = c.D;
  if (accept_b) {
    // p = softmax(log_weights_ext) in LIST order; Generator.choice: cdf = p.cumsum(); cdf /= cdf[-1];
    // idx = cdf.searchsorted(u, side='right')
    double mx = -INFINITY;
    for (int i = 0; i < n; ++i) { const double x = c.ew[(size_t)i * c.B + b]; mx = x > mx ? x : mx; }
    double tot = 0.0;
    for (int i = 0; i < n; ++i) tot += exp(c.ew[(size_t)i * c.B + b] - mx);
    double cs = 0.0;
    for (int p = 0; p < n; ++p) { const int i = s.backward ? n - 1 - p : p; cs += exp(c.ew[(size_t)i * c.B + b] - mx) / tot; }
    const double last = cs;
    cs = 0.0;
    int idx = n - 1;
    for (int p = 0; p < n; ++p) {
      const int i = s.backward ? n - 1 - p : p;
      cs += exp(c.ew[(size_t)i * c.B + b] - mx) / tot;
      if (u_choice_b < cs / last) { idx = p; break; }
    }
    const int ii = s.backward ? n - 1 - idx : idx;
    const double* st = c.eorb + (size_t)ii * 2 * D * c.B;
    for (int i = 0; i < D; ++i) V(V_SEL, i) = st[(size_t)i * c.B + b];
  }
  // orbit = orbit_ext + orbit / orbit + orbit_ext: the new end is the LAST computed state
  const double* st = c.eorb + (size_t)(n - 1) * 2 * D * c.B;
  for (int i = 0; i < D; ++i) {
    const double th = st[(size_t)i * c.B + b], rh = st[((size_t)D + i) * c.B + b], g = V(V_G0, i);
    if (s.backward) { V(V_THL, i) = th; V(V_RHL, i) = rh; V(V_GL, i) = g; }
    else { V(V_THR, i) = th; V(V_RHR, i) = rh; V(V_GR, i) = g; }
  }
  if (s.backward) { s.lpL = s.lp0; s.wL = c.ew[(size_t)(n - 1) * c.B + b]; }
  else { s.lpR = s.lp0; s.wR = c.ew[(size_t)(n - 1) * c.B + b]; }
  for (int i = 0; i < n; ++i) c.logw[(size_t)(s.nw + i) * c.B + b] = c.ew[(size_t)i * c.B + b];
  s.nw += n;
  double d1 = 0.0, d2 = 0.0;  // uturn(orbit[0], orbit[-1], inv_mass)
  for (int i = 0; i < D; ++i) {
    const double diff = c.im[i] * (V(V_THR, i) - V(V_THL, i));
    d1 += V(V_RHL, i) * diff;
    d2 += V(V_RHR, i) * diff;
  }
  return (d1 < 0 || d2 < 0) ? 1 : 0;
}
__global__ void wstep_commit_kernel(WCtx c, const int* accept, const double* u_choice, int* uturn_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.B) return;
  uturn_out[b] = commit_chain(c, b, c.sc[b], accept[b] != 0, u_choice[b]);
}


// Whole transitions on the device with the ENGINE's random stream (wn_rng.h: Philox keyed by seed, global chain id and transition
// number) in place of the caller's numpy Generators: walnuts_step (walnuts.py:322-359) n_trans times per chain and launch, no host
// round trip.  Same transition kernel as the host-driven calls above, a different (documented) source of randomness: draw k of
// transition `it` of a chain is rng_uniform(seed, chain, it, k) in the reference's order of use -- direction bit (:330), one lottery
// index per macro step (:194), the acceptance uniform (:346) and, after an accept, the categorical uniform (:350); the momentum is
// rng_normal(seed, chain, it, j) / sqrt(inv_mass[j]) (:325).
__global__ void wstep_auto_kernel(WCtx c0, int* lottery, uint64_t seed, uint32_t chain0, int it0, int n_trans, int max_depth,
                                  double* theta_io, double* draws, int* stats) {
  // ONE CHAIN PER WAVEFRONT (lane 0 works): the machine's stages, the search depths and the orbit lengths differ from chain to chain,
  // and 64 chains in one wave each waited for every branch of every other (measured: 0.53 s per transition of a wave, 7.7e3
  // transitions/s for 4096 chains); alone in its wave a chain pays only its own path
  const int b = blockIdx.x;
  __shared__ double vl[V_NVEC * 64];
  __shared__ WScal sl;
  if (b >= c0.B || threadIdx.x != 0) return;
  // This chain's private view of the handle: batch size 1, every array offset to the chain's own contiguous part.  The
  // thread-per-chain kernels stride every array by the batch size (coalescing across 64 chains); a lone chain walking arrays
  // strided by 32 KB touches a new page with every element (measured: 64 chains 2.3 ms per transition, 256 chains 38 ms).
  WCtx c = c0;
  c.B = 1;
  c.eorb = c0.eorb + (size_t)b * c0.nmax * 2 * c0.D;
  c.ew = c0.ew + (size_t)b * c0.nmax;
  c.logw = c0.logw + (size_t)b * 2 * c0.nmax;
  c.vec = c0.vec + (size_t)b * V_NVEC * c0.D;
  c.lottery = lottery + (size_t)b * c0.nmax;
  int* const lot = lottery + (size_t)b * c0.nmax;
  // the chain's vectors and scalars in LDS (D <= 64: checked on the host)
  WScal& s = sl;
  VecsL V{c, (lds_f64*)vl};
  const uint32_t chain = chain0 + (uint32_t)b;
  int sub1 = 0;
  double lse2[2] = {0.0, 0.0};
  for (int tr = 0; tr < n_trans; ++tr) {
    const uint32_t it = (uint32_t)(it0 + tr);
    uint32_t k = 0;
    for (int i = 0; i < c.D; ++i) {
      V(V_TH, i) = theta_io[(size_t)b * c.D + i];
      V(V_RH, i) = rng_normal(seed, chain, it, (uint32_t)i) / sqrt(c.im[i]);
    }
    s.stage = ST_INIT; s.active = 1; s.nmacro = 0; s.m = 0; s.sub_uturn = 0;
    s.sum_ell = 0; s.sum_es = 0;
    int ndbl = 0, nacc = 0;  // (stats: doublings started, accepts)
    run_machine_lds(c, V, s);
    for (int depth = 0; depth < max_depth; ++depth) {
      const int n = 1 << depth;
      ndbl += 1;
      const int backward = rng_uniform(seed, chain, it, k++) < 0.5 ? 1 : 0;
      for (int m = 0; m < n; ++m) {
        const int idx = (int)(3.0 * rng_uniform(seed, chain, it, k++));
        lot[m] = idx > 2 ? 2 : idx;
      }
      extend_setup(c, V, s, backward, n);
      run_machine_lds(c, V, s);
      extend_finish(c, 0, s, &sub1, &lse2[0], &lse2[1]);
      if (s.sub_uturn) break;
      const double logu = log(rng_uniform(seed, chain, it, k++));
      const bool acc = logu < lse2[0] - lse2[1];
      const double uc = acc ? rng_uniform(seed, chain, it, k++) : 0.0;
      nacc += acc ? 1 : 0;
      if (commit_chain(c, 0, s, acc, uc, vl)) break;
    }
    if (stats) {
      int* o = stats + ((size_t)b * n_trans + tr) * 4;
      o[0] = ndbl; o[1] = nacc; o[2] = s.sum_ell; o[3] = s.sum_es;
    }
    for (int i = 0; i < c.D; ++i) {
      const double th = V(V_SEL, i);
      theta_io[(size_t)b * c.D + i] = th;
      if (draws) draws[((size_t)b * n_trans + tr) * c.D + i] = th;
    }
  }
}

}  // namespace (reopened below)

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same whole transitions with the WAVE doing the work -- lane i holds element i of every vector (D <= 64), vectors
// in registers, one wave reduction per energy, branch-lean (wstep_auto_kernel above lets lane 0 walk scalar loops over D with the
// vectors in LDS through a generic state machine: ~100x the instructions per gradient evaluation).  Gaussian targets (standard /
// diagonal: BASELINE configs[0]'s target), whose log density and gradient are element-wise; every other built-in target keeps the
// kernel above.  Same transition kernel (walnuts.py:279-359), same random stream and the same order of draws as the kernel above
// (direction bit, the extension's lottery indices, the acceptance uniform, the categorical uniform; momentum normals), so the two
// produce the same chains up to the association of their sums (tests/test_gpu_parity.py).
// ------------------------------------------------------------------------------------------------------------------------
#include "wn_policy_wave.h"  // wave sums, readlane helpers

namespace {

template <bool SMALL>  // SMALL: D <= 16 -- sums over one row of 16 lanes
__device__ __forceinline__ double wt_sum(double x) {
  if constexpr (SMALL) return wn_readlane(wn_rowsum16(x), 0);
  else return wn_wavesum(x);
}
// two sums at once
template <bool SMALL>
__device__ __forceinline__ void wt_sum2(double& a, double& b) {
  if constexpr (SMALL) {
    double r = wn_pack32(a, b);
    r = wn_rowsum16(r);
    a = wn_readlane(r, 0); b = wn_readlane(r, 32);
  } else { a = wn_wavesum(a); b = wn_wavesum(b); }
}

// One chain of the wave layout: what every kernel below does with it.  Lane i holds element i of every vector.
template <bool SMALL>
struct WvChain {
  int lane, D, li;
  bool in;
  double im, aa, macro, maxerr;
  const double* imv;          // inverse masses, all elements (the sub-U-turn checks walk whole states)
  double *eorb, *ew, *logw;   // this chain's: [m][2][D] states of the extension in integration order, their log weights, the orbit's so far
  // the orbit's two ends, the selected state, the number of log weights so far
  double thL, rhL, gL, thR, rhR, gR, sel, lpL, lpR, wL, wR;
  int nw;
  // the last state of the running extension
  double t0, r0, g0, lp0;
  int sum_ell, sum_es;  // (wn_wstep_set_stats)

  __device__ __forceinline__ void setup(const WCtx& c0, int b, int lane_) {
    lane = lane_; D = c0.D; in = lane < D; li = in ? lane : 0;
    eorb = c0.eorb + (size_t)b * c0.nmax * 2 * D;
    ew = c0.ew + (size_t)b * c0.nmax;
    logw = c0.logw + (size_t)b * 2 * c0.nmax;
    imv = c0.im;
    im = in ? c0.im[li] : 0.0;                                                    // inverse mass (0 beyond D: those lanes add nothing)
    aa = in ? (c0.kind == T_DIAGGAUSS ? c0.tpar[li] : 1.0) : 0.0;                 // precision of this lane's coordinate
    macro = c0.macro_step; maxerr = c0.max_error;
    sum_ell = sum_es = 0;
  }
  __device__ __forceinline__ double ke_of(double rho) const { return 0.5 * wt_sum<SMALL>(im * (rho * rho)); }  // 0.5*dot(inv_mass, rho**2)  (:124)
  // log density and gradient at theta (element-wise targets): lp = 1/2 sum theta_i g_i
  __device__ __forceinline__ double eval(double th, double& g) const { g = -(aa * th); return 0.5 * wt_sum<SMALL>(th * g); }
  __device__ __forceinline__ static double mlogp(int ell, int es) { return (ell == es || ell == es / 2 || ell == es * 2) ? -1.0986122886681098 : -INFINITY; }

  // stable_steps (:144-182) from (th0, sgn * rh0) with (lp0, g0) known: the smallest 2**n <= 1024 micro steps whose Hamiltonians
  // stay within max_error of each other (2**10 if none does).  One wave sum per micro step: H = sum_i (1/2 im rho_i^2 - 1/2 theta_i g_i).
  __device__ __forceinline__ int stable(double th0, double rh0, double g0_, double lp0_, double sgn) const {
    const double r0_ = sgn * rh0;
    const double H0 = -lp0_ + ke_of(r0_);
    int ell = 1;
#pragma unroll 1
    for (int n = 0;; ++n) {
      ell = 1 << n;
      const double step = macro / (double)ell, half = 0.5 * step;
      double th = th0, rh = r0_ + half * g0_;
      th = th + (step * im) * rh;
      double Hmin = H0, Hmax = H0;
      int k = ell;
#pragma unroll 1
      do {
        const double g = -(aa * th);
        rh = rh + half * g;
        const double Hc = wt_sum<SMALL>(0.5 * (im * (rh * rh)) - 0.5 * (th * g));
        Hmin = (Hc < Hmin) ? Hc : Hmin; Hmax = (Hc > Hmax) ? Hc : Hmax;  // Python min() / max() order
        rh = rh + half * g;
        th = th + (step * im) * rh;   // (one drift too many after the last step: nothing reads it)
      } while (--k);
      if (__builtin_amdgcn_readfirstlane((int)(Hmax - Hmin <= maxerr)) || n >= 10) break;
    }
    return ell;
  }

  // the transition's first state (:325-327): orbit = [(theta, rho)], log_weights = [-H(theta, rho)]
  __device__ __forceinline__ void begin(double th, double rh) {
    double g;
    const double lp = eval(th, g);
    thL = th; rhL = rh; gL = g; thR = th; rhR = rh; gR = g; sel = th;
    lpL = lp; lpR = lp;
    wL = lp - ke_of(rh); wR = wL;
    if (lane == 0) logw[0] = wL;
    nw = 1;
    t0 = th; r0 = rh; g0 = g; lp0 = lp;
  }

  // extend_orbit (:211-276): n macro steps from the orbit's first state with the momentum negated, or from its last state;
  // lot(m) in {0, 1, 2}: the m-th macro step's choose_micro_steps draw (:194)
  template <class Lot>
  __device__ __forceinline__ void extend(const int backward, const int n, Lot lot) {
    t0 = backward ? thL : thR; r0 = backward ? -rhL : rhR; g0 = backward ? gL : gR;
    lp0 = backward ? lpL : lpR;
    double weight = backward ? wL : wR;
#pragma unroll 1
    for (int m = 0; m < n; ++m) {
      const double p0 = lp0 - ke_of(r0);
      const int es = stable(t0, r0, g0, lp0, 1.0);
      const int idx = lot(m);
      const int ell = idx == 0 ? es / 2 : idx == 1 ? es : es * 2;  // choose_micro_steps (:194); es == 1 can give 0 (Q9): an infinite step
      sum_ell += ell; sum_es += es;
      {  // leapfrog (:74-95)
        const double step = macro / (double)ell, half = 0.5 * step;
        double t = t0, r = r0 + half * g0;
        t = t + (step * im) * r;
        double gg = -(aa * t);
#pragma unroll 1
        for (int j = 1; j < ell; ++j) {
          r = r + step * gg;
          t = t + (step * im) * r;
          gg = -(aa * t);
        }
        r0 = r + half * gg; t0 = t; g0 = gg;
      }
      double gchk;
      lp0 = eval(t0, gchk);
      const int en = stable(t0, r0, g0, lp0, -1.0);
      const double p1 = lp0 - ke_of(r0);
      weight = p1 - p0 + mlogp(ell, en) - mlogp(ell, es) + weight;  // :264-271
      if (in) { eorb[(size_t)m * 2 * D + li] = t0; eorb[(size_t)m * 2 * D + D + li] = r0; }
      if (lane == 0) ew[m] = weight;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }

  // sub_uturn (:62-70) over the extension in LIST order (time order: integration order reversed when going backward): check c of
  // the n-1 (span n >> l, i-th block) is lane c's; any order gives the same answer
  __device__ __forceinline__ bool sub_uturn(const int backward, const int n) const {
    bool su = false;
    for (int c1 = 0; c1 < n - 1; c1 += 64) {
      const int cc = c1 + lane;
      bool f = false;
      if (cc < n - 1) {
        const int l = 31 - __builtin_clz((unsigned)(cc + 1)), i = cc + 1 - (1 << l), span = n >> l;
        const int x = span * i, y = span * (i + 1) - 1;
        const int ix = backward ? n - 1 - x : x, iy = backward ? n - 1 - y : y;
        const double *sx = eorb + (size_t)ix * 2 * D, *sy = eorb + (size_t)iy * 2 * D;
        double d1 = 0.0, d2 = 0.0;
        for (int j = 0; j < D; ++j) {
          const double diff = imv[j] * (sy[j] - sx[j]);
          d1 += sx[D + j] * diff;
          d2 += sy[D + j] * diff;
        }
        f = d1 < 0 || d2 < 0;
      }
      su = su || __any(f);
    }
    return su;
  }
  // logsumexp of the extension's weights and of the orbit's so far (:345, scipy.special.logsumexp)
  __device__ __forceinline__ double lse(const double* a, int cnt) const {
    double mx = -INFINITY; bool nan = false;
    for (int j = lane; j < cnt; j += 64) { const double x = a[j]; nan = nan || isnan(x); mx = x > mx ? x : mx; }
    mx = WavePolicy<1, WTgtDiag<1>>::wavemax(mx);
    if (__any(nan)) mx = NAN;
    if (!isfinite(mx)) mx = 0.0;
    double sm = 0.0;
    for (int j = lane; j < cnt; j += 64) sm += exp(a[j] - mx);
    return log(wn_wavesum(sm)) + mx;
  }
  // after the acceptance draw: the categorical pick (:347-350) and the orbit merge (:351)
  __device__ __forceinline__ void commit(const bool acc, const double uc, const int backward, const int n) {
    if (acc) {  // p = softmax(log_weights_ext) in LIST order; Generator.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, 'right')
      double mx = -INFINITY;
      for (int j = 0; j < n; ++j) { const double x = ew[j]; mx = x > mx ? x : mx; }
      double tot = 0.0;
      for (int j = 0; j < n; ++j) tot += exp(ew[j] - mx);
      double cs = 0.0;
      for (int pp = 0; pp < n; ++pp) { const int j = backward ? n - 1 - pp : pp; cs += exp(ew[j] - mx) / tot; }
      const double last = cs;
      cs = 0.0;
      int idx = n - 1;
      for (int pp = 0; pp < n; ++pp) {
        const int j = backward ? n - 1 - pp : pp;
        cs += exp(ew[j] - mx) / tot;
        if (uc < cs / last) { idx = pp; break; }
      }
      const int ii = __builtin_amdgcn_readfirstlane(backward ? n - 1 - idx : idx);
      sel = in ? eorb[(size_t)ii * 2 * D + li] : 0.0;
    }
    // orbit = orbit_ext + orbit / orbit + orbit_ext (:351): the new end is the LAST computed state
    const double we = ew[n - 1];
    if (backward) { thL = t0; rhL = r0; gL = g0; lpL = lp0; wL = we; }
    else { thR = t0; rhR = r0; gR = g0; lpR = lp0; wR = we; }
    for (int j = lane; j < n; j += 64) logw[nw + j] = ew[j];
    nw += n;
  }
  __device__ __forceinline__ bool uturn() const {  // uturn(orbit[0], orbit[-1], inv_mass) (:352)
    double d1 = im * (thR - thL), d2 = rhR * d1;
    d1 = rhL * d1;
    wt_sum2<SMALL>(d1, d2);
    return __builtin_amdgcn_readfirstlane((int)(d1 < 0 || d2 < 0)) != 0;
  }

  // ---- the chain between two launches of the host-driven protocol (wn_wstep_begin / extend / commit on the wave kernels):
  // ten vectors of 64 doubles, then the scalars
  enum { ST_VECS = 10, ST_SCAL = ST_VECS * 64, ST_DOUBLES = ST_SCAL + 16 };
  __device__ __forceinline__ void store(double* st, int backward, int n, int active, int su) const {
    const double v[ST_VECS] = {thL, rhL, gL, thR, rhR, gR, sel, t0, r0, g0};
#pragma unroll
    for (int k = 0; k < ST_VECS; ++k) st[k * 64 + lane] = v[k];
    if (lane == 0) {
      double* q = st + ST_SCAL;
      q[0] = lpL; q[1] = lpR; q[2] = wL; q[3] = wR; q[4] = lp0;
      int* qi = reinterpret_cast<int*>(q + 5);
      qi[0] = nw; qi[1] = backward; qi[2] = n; qi[3] = active; qi[4] = su;
    }
  }
  __device__ __forceinline__ void load(const double* st, int& backward, int& n, int& active, int& su) {
    thL = st[0 * 64 + lane]; rhL = st[1 * 64 + lane]; gL = st[2 * 64 + lane]; thR = st[3 * 64 + lane]; rhR = st[4 * 64 + lane];
    gR = st[5 * 64 + lane]; sel = st[6 * 64 + lane]; t0 = st[7 * 64 + lane]; r0 = st[8 * 64 + lane]; g0 = st[9 * 64 + lane];
    const double* q = st + ST_SCAL;
    lpL = q[0]; lpR = q[1]; wL = q[2]; wR = q[3]; lp0 = q[4];
    const int* qi = reinterpret_cast<const int*>(q + 5);
    nw = qi[0]; backward = qi[1]; n = qi[2]; active = qi[3]; su = qi[4];
  }
};

template <bool SMALL>
__global__ __launch_bounds__(64) void wstep_wave_kernel(WCtx c0, int* lottery, uint64_t seed, uint32_t chain0, int it0, int n_trans, int max_depth,
                                                        double* theta_io, double* draws, int* stats) {
  const int b = blockIdx.x, lane = threadIdx.x, D = c0.D;
  if (b >= c0.B) return;
  WvChain<SMALL> ch;
  ch.setup(c0, b, lane);
  extern __shared__ int wt_lot_lds[];  // the extension's lottery indices, [nmax] (LDS operations of one wave execute in order)
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* const lot = (lds_int*)wt_lot_lds;
  (void)lottery;
  const uint32_t chain = chain0 + (uint32_t)b;
  const uint32_t seedlo = (uint32_t)seed, seedhi = (uint32_t)(seed >> 32);
  // window of uniforms: lane l holds uniform 2 * ub + l of transition uit
  uint32_t uit = 0xFFFFFFFFu, ub = 0;
  double uw = 0.0;
  auto uni = [&](uint32_t it, uint32_t k) -> double {
    const uint32_t blk = k >> 1;
    if (it != uit || blk - ub >= 32u) {
      uit = it; ub = blk;
      Philox4 r = philox4x32_10(blk + (uint32_t)(lane >> 1), it, chain, 0u, seedlo, seedhi);
      uw = (lane & 1) ? u53(r.w[2], r.w[3]) : u53(r.w[0], r.w[1]);
    }
    return wn_readlane(uw, (int)(k - 2u * ub));
  };
  for (int tr = 0; tr < n_trans; ++tr) {
    const uint32_t it = (uint32_t)(it0 + tr);
    uint32_t k = 0;
    const double th = ch.in ? theta_io[(size_t)b * D + ch.li] : 0.0;
    const double rh = ch.in ? rng_normal(seed, chain, it, (uint32_t)ch.li) / sqrt(ch.im) : 0.0;  // rho ~ mvnormal(0, inverse(inv_mass))  (:325)
    ch.begin(th, rh);
    ch.sum_ell = ch.sum_es = 0;
    int ndbl = 0, nacc = 0;  // (stats: what a reference run on the same stream must reproduce exactly)
#pragma unroll 1
    for (int depth = 0; depth < max_depth; ++depth) {
      const int n = 1 << depth;
      ndbl += 1;
      const int backward = uni(it, k++) < 0.5 ? 1 : 0;  // :330
      for (int m0 = 0; m0 < n; m0 += 32) {               // one lottery index per macro step (:194), drawn before the extension as the one-lane kernel does
        uit = 0xFFFFFFFFu;                               // a fresh window that STARTS at the block of uniform k + m0: its first 62 uniforms are inside
        (void)uni(it, k + (uint32_t)m0);
        const int off = (int)(k + (uint32_t)m0 - 2u * ub);
        const double u = __shfl(uw, (off + lane) & 63);
        const int idx = (int)(3.0 * u);
        if (lane < 32 && m0 + lane < n) lot[m0 + lane] = idx > 2 ? 2 : idx;
      }
      k += (uint32_t)n;
      ch.extend(backward, n, [&](int m) { return __builtin_amdgcn_readfirstlane((int)lot[m]); });
      if (ch.sub_uturn(backward, n)) break;
      const double lse_ext = ch.lse(ch.ew, n), lse_old = ch.lse(ch.logw, ch.nw);
      const double logu = log(uni(it, k++));  // :346
      const bool acc = logu < lse_ext - lse_old;
      double uc = 0.0;
      if (acc) { nacc += 1; uc = uni(it, k++); }  // :347-350
      ch.commit(acc, uc, backward, n);
      if (ch.uturn()) break;
    }
    if (ch.in) {
      theta_io[(size_t)b * D + ch.li] = ch.sel;
      if (draws) draws[((size_t)b * n_trans + tr) * D + ch.li] = ch.sel;
    }
    if (stats && lane < 4) stats[((size_t)b * n_trans + tr) * 4 + lane] = lane == 0 ? ndbl : lane == 1 ? nacc : lane == 2 ? ch.sum_ell : ch.sum_es;
  }
}

// ---- the host-driven protocol (wn_wstep_begin / extend / commit / result: the caller's numpy Generators decide, one exchange per
// doubling) on the same wave code: the chain's state lives in `wst` between launches.  Round 4's thread-per-chain kernels made 2.4e3
// transitions/s of walnuts_batch at 4096 chains (a lone thread walks every vector); the wave does an extension in microseconds.
template <bool SMALL>
__global__ __launch_bounds__(64) void wstep_wv_begin_kernel(WCtx c0, double* wst, const double* theta, const double* rho) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= c0.B) return;
  WvChain<SMALL> ch;
  ch.setup(c0, b, lane);
  ch.begin(ch.in ? theta[(size_t)b * c0.D + ch.li] : 0.0, ch.in ? rho[(size_t)b * c0.D + ch.li] : 0.0);
  ch.store(wst + (size_t)b * WvChain<SMALL>::ST_DOUBLES, 0, 0, 1, 0);
}
template <bool SMALL>
__global__ __launch_bounds__(64) void wstep_wv_extend_kernel(WCtx c0, double* wst, const int* lottery, const int* backward_in, const int* active_in,
                                                             int n, int* sub_out, double* lse_ext, double* lse_old) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= c0.B) return;
  double* const st = wst + (size_t)b * WvChain<SMALL>::ST_DOUBLES;
  int* const sti = reinterpret_cast<int*>(st + WvChain<SMALL>::ST_SCAL + 5);
  if (!active_in[b]) { if (lane == 0) sti[3] = 0; return; }  // (the commit launch leaves an inactive chain alone)
  extern __shared__ int wt_lot_lds[];
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* const lot = (lds_int*)wt_lot_lds;
  for (int m = lane; m < n; m += 64) lot[m] = lottery[(size_t)b * c0.nmax + m];
  WvChain<SMALL> ch;
  ch.setup(c0, b, lane);
  int bw_, n_, ac_, su_;
  ch.load(st, bw_, n_, ac_, su_);
  const int backward = backward_in[b];
  ch.extend(backward, n, [&](int m) { return __builtin_amdgcn_readfirstlane((int)lot[m]); });
  const bool su = ch.sub_uturn(backward, n);
  double le = 0.0, lo = 0.0;
  if (!su) { le = ch.lse(ch.ew, n); lo = ch.lse(ch.logw, ch.nw); }
  if (lane == 0) { sub_out[b] = su ? 1 : 0; lse_ext[b] = le; lse_old[b] = lo; }
  ch.store(st, backward, n, 1, su ? 1 : 0);
}
template <bool SMALL>
__global__ __launch_bounds__(64) void wstep_wv_commit_kernel(WCtx c0, double* wst, const int* accept, const double* u_choice, int* uturn_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= c0.B) return;
  double* const st = wst + (size_t)b * WvChain<SMALL>::ST_DOUBLES;
  WvChain<SMALL> ch;
  ch.setup(c0, b, lane);
  int backward, n, active, su;
  ch.load(st, backward, n, active, su);
  if (!active || su) { if (lane == 0) uturn_out[b] = 0; return; }  // a sub-U-turn breaks out before the merge (:343)
  ch.commit(accept[b] != 0, u_choice[b], backward, n);
  const bool ut = ch.uturn();
  if (lane == 0) uturn_out[b] = ut ? 1 : 0;
  ch.store(st, backward, n, active, su);
}
__global__ void wstep_wv_result_kernel(int B, int D, const double* wst, int st_doubles, double* theta_sel) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b < B && lane < D) theta_sel[(size_t)b * D + lane] = wst[(size_t)b * st_doubles + 6 * 64 + lane];
}

}  // namespace

namespace {

__global__ void wstep_result_kernel(WCtx c, double* theta_sel) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.B) return;
  Vecs V{c, b, nullptr};
  for (int i = 0; i < c.D; ++i) theta_sel[(size_t)b * c.D + i] = V(V_SEL, i);
}

}  // namespace

struct wn_wstep {
  WCtx c{};
  int device = 0, max_depth = 0;
  bool external = false;
  double *tpar = nullptr, *im = nullptr;
  int* lot = nullptr;
  double* wst = nullptr;   // wave kernels of the host-driven protocol: the chains' states between launches, [B][WvChain::ST_DOUBLES]
  uint32_t route = 0;      // WN_ROUTE_WSTEP_LEGACY or 0 (wn_wstep_set_route)
  int* stats = nullptr;    // caller-owned [B][n_transitions][4] of the next wn_wstep_run_device_rng calls (wn_wstep_set_stats), or null
  int* asub = nullptr;     // device-RNG transitions: per-chain scratch of extend_finish
  double* alse = nullptr;  // [2][B]
  std::string err;
};
static std::string g_wstep_err;

#define WCHK(call)                                                                \
  do {                                                                            \
    hipError_t _r = (call);                                                       \
    if (_r != hipSuccess) {                                                       \
      h->err = std::string(#call) + ": " + hipGetErrorString(_r);                 \
      return _r == hipErrorOutOfMemory ? WN_E_ALLOC : WN_E_HIP;                   \
    }                                                                             \
  } while (0)

extern "C" {

const char* wn_wstep_last_error(wn_wstep* h) { return h ? h->err.c_str() : g_wstep_err.c_str(); }

int wn_wstep_set_route(wn_wstep* h, uint32_t route) {
  if (!h) return WN_E_ARG;
  if (route & ~(uint32_t)WN_ROUTE_WSTEP_LEGACY) { h->err = "wn_wstep_set_route: only WN_ROUTE_WSTEP_LEGACY (or 0) applies to this handle"; return WN_E_ARG; }
  h->route = route;
  return WN_OK;
}

int wn_wstep_destroy(wn_wstep* h) {
  if (!h) return WN_OK;
  hipSetDevice(h->device);
  for (void* p : {(void*)h->tpar, (void*)h->im, (void*)h->lot, (void*)h->c.vec, (void*)h->c.eorb, (void*)h->c.ew, (void*)h->c.logw,
                  (void*)h->c.sc, (void*)h->c.npend, (void*)h->c.waiting, (void*)h->asub, (void*)h->alse, (void*)h->wst})
    if (p) hipFree(p);
  delete h;
  return WN_OK;
}

int wn_wstep_create(int32_t device, int32_t target_kind, const double* tpar_dev, size_t n_tpar, int32_t dim, int32_t B,
                    int32_t max_depth, const double* inv_mass_dev, double macro_step, double max_error, wn_wstep** out) {
  if (!out) { g_wstep_err = "null argument"; return WN_E_ARG; }
  *out = nullptr;
  if (dim < 1 || B < 1 || max_depth < 1 || !inv_mass_dev || !(macro_step > 0) || !(max_error > 0)) {
    g_wstep_err = "wn_wstep_create: bad argument"; return WN_E_ARG;
  }
  if (max_depth > 20) {  // the reference accepts any depth; the device stepper's orbit store is sized 2^(max_depth-1) states per chain
    g_wstep_err = "wn_wstep_create: max_nuts_depth beyond 20 is not supported by the device stepper (orbit store of 2^(depth-1) states per chain)";
    return WN_E_UNSUPPORTED;
  }
  const bool ext = target_kind == WN_T_EXTERNAL;
  if (!ext) {
    if (target_kind < 0 || target_kind > WN_T_TT_FUNNEL) { g_wstep_err = "unknown target"; return WN_E_ARG; }
    if ((target_kind == WN_T_DIAGGAUSS || target_kind == WN_T_DENSEGAUSS) && !tpar_dev) { g_wstep_err = "target needs parameters"; return WN_E_ARG; }
    if ((target_kind == WN_T_FUNNEL10 && dim != 11) ||
        ((target_kind == WN_T_CORRGAUSS || target_kind == WN_T_SMILE || target_kind == WN_T_FUNNEL1 || target_kind == WN_T_MODFUNNEL ||
          target_kind == WN_T_TT_CORRNORMAL) && dim != 2) || (target_kind == WN_T_TT_FUNNEL && dim < 2)) {
      g_wstep_err = "dimension does not fit the target"; return WN_E_ARG;
    }
    if (dim > 64 && target_kind != WN_T_STDGAUSS && target_kind != WN_T_DIAGGAUSS) {
      g_wstep_err = "built-in targets beyond dim 64: stdGauss / diagGauss only (others: pass (logp, grad) callables)"; return WN_E_UNSUPPORTED;
    }
  }
  wn_wstep* h = new wn_wstep();
  h->device = device; h->max_depth = max_depth; h->external = ext;
  if (hipSetDevice(device) != hipSuccess) { g_wstep_err = "hipSetDevice failed"; delete h; return WN_E_HIP; }
  WCtx& c = h->c;
  c.kind = target_kind; c.D = dim; c.B = B; c.nmax = 1 << (max_depth - 1);
  c.macro_step = macro_step; c.max_error = max_error;
  auto fail = [&](int rc) { g_wstep_err = h->err; wn_wstep_destroy(h); return rc; };
#define WCOPY(call)                                                                        \
  do {                                                                                     \
    hipError_t _r = (call);                                                                \
    if (_r != hipSuccess) { (void)hipGetLastError(); h->err = std::string(#call) + ": " + hipGetErrorString(_r); return fail(WN_E_HIP); } \
  } while (0)
#define WALLOC(ptr, bytes)                                                                 \
  do {                                                                                     \
    hipError_t _r = hipMalloc((void**)&(ptr), (bytes));                                    \
    if (_r != hipSuccess) { (ptr) = nullptr; (void)hipGetLastError(); h->err = std::string("hipMalloc: ") + hipGetErrorString(_r); \
                            return fail(_r == hipErrorOutOfMemory ? WN_E_ALLOC : WN_E_HIP); }   \
  } while (0)
  if (tpar_dev && n_tpar) {
    WALLOC(h->tpar, n_tpar * sizeof(double));
    WCOPY(hipMemcpy(h->tpar, tpar_dev, n_tpar * sizeof(double), hipMemcpyDeviceToDevice));
  }
  WALLOC(h->im, (size_t)dim * sizeof(double));
  WCOPY(hipMemcpy(h->im, inv_mass_dev, (size_t)dim * sizeof(double), hipMemcpyDeviceToDevice));
  WALLOC(h->lot, (size_t)B * c.nmax * sizeof(int));
  WALLOC(c.vec, (size_t)V_NVEC * dim * B * sizeof(double));
  WALLOC(c.eorb, (size_t)c.nmax * 2 * dim * B * sizeof(double));
  WALLOC(c.ew, (size_t)c.nmax * B * sizeof(double));
  WALLOC(c.logw, (size_t)2 * c.nmax * B * sizeof(double));
  WALLOC(c.sc, (size_t)B * sizeof(WScal));
  WALLOC(c.npend, 64);
  WALLOC(c.waiting, (size_t)B * sizeof(int));
  WALLOC(h->asub, (size_t)B * sizeof(int));
  WALLOC(h->alse, (size_t)2 * B * sizeof(double));
  if (!ext && (target_kind == WN_T_STDGAUSS || target_kind == WN_T_DIAGGAUSS) && dim <= 64 && c.nmax <= 8192)  // (wave_path below)
    WALLOC(h->wst, (size_t)B * WvChain<true>::ST_DOUBLES * sizeof(double));
#undef WALLOC
  WCOPY(hipMemset(c.sc, 0, (size_t)B * sizeof(WScal)));
  WCOPY(hipMemset(c.vec, 0, (size_t)V_NVEC * dim * B * sizeof(double)));
#undef WCOPY
  c.tpar = h->tpar; c.im = h->im; c.lottery = h->lot;
  *out = h;
  return WN_OK;
}

// element-wise Gaussian targets run the host-driven protocol on the wave kernels too (WN_WSTEP_LEGACY=1: the thread-per-chain kernels)
static bool wave_path(const wn_wstep* h) { return h->wst != nullptr && !(h->route & WN_ROUTE_WSTEP_LEGACY); }

static int wstep_launch(wn_wstep* h, int mode, const double* theta, const double* rho, const int* backward, const int* active,
                        int nmacro, int* sub_out, double* lse_ext, double* lse_old, hipStream_t st) {
  dim3 grid((h->c.B + 63) / 64), block(64);
  if (h->external) {
    WCHK(hipMemsetAsync(h->c.npend, 0, sizeof(int), st));
    hipLaunchKernelGGL(wstep_ext_kernel, grid, block, 0, st, h->c, mode, theta, rho, backward, active, nmacro, sub_out, lse_ext, lse_old);
  } else {
    hipLaunchKernelGGL(wstep_fused_kernel, grid, block, 0, st, h->c, mode, theta, rho, backward, active, nmacro, sub_out, lse_ext, lse_old);
  }
  WCHK(hipGetLastError());
  return WN_OK;
}

int wn_wstep_begin(wn_wstep* h, const double* theta_dev, const double* rho_dev, void* stream) {
  if (!h || !theta_dev || !rho_dev) { if (h) h->err = "null buffer"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  if (wave_path(h)) {
    if (h->c.D <= 16) hipLaunchKernelGGL(wstep_wv_begin_kernel<true>, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c, h->wst, theta_dev, rho_dev);
    else hipLaunchKernelGGL(wstep_wv_begin_kernel<false>, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c, h->wst, theta_dev, rho_dev);
    WCHK(hipGetLastError());
    return WN_OK;
  }
  return wstep_launch(h, 0, theta_dev, rho_dev, nullptr, nullptr, 0, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int wn_wstep_extend(wn_wstep* h, int32_t depth, const int32_t* backward_dev, const int32_t* lottery_dev, const int32_t* active_dev,
                    int32_t* sub_uturn_dev, double* lse_ext_dev, double* lse_old_dev, void* stream) {
  if (!h || !backward_dev || !lottery_dev || !active_dev || !sub_uturn_dev || !lse_ext_dev || !lse_old_dev) { if (h) h->err = "null buffer"; return WN_E_ARG; }
  if (depth < 0 || depth >= h->max_depth) { h->err = "depth out of range"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  const int n = 1 << depth;
  // lottery [B][n] -> the handle's [B][nmax]
  WCHK(hipMemcpy2DAsync(h->lot, (size_t)h->c.nmax * sizeof(int), lottery_dev, (size_t)n * sizeof(int), (size_t)n * sizeof(int), (size_t)h->c.B,
                        hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (wave_path(h)) {
    const size_t dyn = (size_t)h->c.nmax * sizeof(int);
    if (h->c.D <= 16) hipLaunchKernelGGL(wstep_wv_extend_kernel<true>, dim3(h->c.B), dim3(64), dyn, (hipStream_t)stream, h->c, h->wst, h->lot, backward_dev,
                                         active_dev, n, sub_uturn_dev, lse_ext_dev, lse_old_dev);
    else hipLaunchKernelGGL(wstep_wv_extend_kernel<false>, dim3(h->c.B), dim3(64), dyn, (hipStream_t)stream, h->c, h->wst, h->lot, backward_dev,
                            active_dev, n, sub_uturn_dev, lse_ext_dev, lse_old_dev);
    WCHK(hipGetLastError());
    return WN_OK;
  }
  return wstep_launch(h, 1, nullptr, nullptr, backward_dev, active_dev, n, sub_uturn_dev, lse_ext_dev, lse_old_dev, (hipStream_t)stream);
}

int wn_wstep_ext_pending(wn_wstep* h, int32_t* n_pending, void* stream) {
  if (!h || !h->external || !n_pending) { if (h) h->err = "wn_wstep_ext_pending needs a WN_T_EXTERNAL handle"; return WN_E_ARG; }
  int np = 0;
  WCHK(hipMemcpyAsync(&np, h->c.npend, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  WCHK(hipStreamSynchronize((hipStream_t)stream));
  *n_pending = np;
  return WN_OK;
}

int wn_wstep_ext_waiting(wn_wstep* h, int32_t* waiting_dev, void* stream) {
  if (!h || !h->external || !waiting_dev) { if (h) h->err = "wn_wstep_ext_waiting needs a WN_T_EXTERNAL handle and a [B] int32 device array"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  WCHK(hipMemcpyAsync(waiting_dev, h->c.waiting, (size_t)h->c.B * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return WN_OK;
}

int wn_wstep_ext_bind(wn_wstep* h, double* x_req_dev, const double* lp_dev, const double* grad_dev) {
  if (!h || !h->external || !x_req_dev || !lp_dev || !grad_dev) { if (h) h->err = "wn_wstep_ext_bind: bad argument"; return WN_E_ARG; }
  h->c.xreq = x_req_dev; h->c.xlp = lp_dev; h->c.xg = grad_dev;
  return WN_OK;
}

int wn_wstep_ext_feed(wn_wstep* h, int32_t* sub_uturn_dev, double* lse_ext_dev, double* lse_old_dev, void* stream) {
  if (!h || !h->external || !h->c.xreq) { if (h) h->err = "wn_wstep_ext_feed: bind the exchange buffers first"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  return wstep_launch(h, 2, nullptr, nullptr, nullptr, nullptr, 0, sub_uturn_dev, lse_ext_dev, lse_old_dev, (hipStream_t)stream);
}

int wn_wstep_commit(wn_wstep* h, const int32_t* accept_dev, const double* u_choice_dev, int32_t* uturn_dev, void* stream) {
  if (!h || !accept_dev || !u_choice_dev || !uturn_dev) { if (h) h->err = "null buffer"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  if (wave_path(h)) {
    if (h->c.D <= 16) hipLaunchKernelGGL(wstep_wv_commit_kernel<true>, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c, h->wst, accept_dev, u_choice_dev, uturn_dev);
    else hipLaunchKernelGGL(wstep_wv_commit_kernel<false>, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c, h->wst, accept_dev, u_choice_dev, uturn_dev);
    WCHK(hipGetLastError());
    return WN_OK;
  }
  hipLaunchKernelGGL(wstep_commit_kernel, dim3((h->c.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->c, accept_dev, u_choice_dev, uturn_dev);
  WCHK(hipGetLastError());
  return WN_OK;
}

int wn_wstep_run_device_rng(wn_wstep* h, uint64_t seed, int64_t chain_offset, int32_t iter0, int32_t n_transitions, double* theta_dev,
                            double* draws_dev, void* stream) {
  if (!h || !theta_dev || n_transitions < 1 || iter0 < 0) { if (h) h->err = "wn_wstep_run_device_rng: bad argument"; return WN_E_ARG; }
  if (h->external) { h->err = "wn_wstep_run_device_rng needs a built-in target (a host callable cannot run inside the kernel)"; return WN_E_UNSUPPORTED; }
  if (h->c.D > 64) { h->err = "wn_wstep_run_device_rng: dim <= 64 (the chain's vectors live in LDS)"; return WN_E_UNSUPPORTED; }
  if (chain_offset < 0 || (uint64_t)chain_offset + (uint64_t)h->c.B > (1ull << 32)) { h->err = "chain_offset + B must stay within [0, 2^32)"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  // element-wise targets: the wave-parallel kernel (lane = element); WN_WSTEP_LEGACY=1 keeps the one-lane kernel (A/B tests)
  if ((h->c.kind == T_STDGAUSS || h->c.kind == T_DIAGGAUSS) && h->c.nmax <= 8192 && !(h->route & WN_ROUTE_WSTEP_LEGACY)) {
    const size_t dyn = (size_t)h->c.nmax * sizeof(int);
    if (h->c.D <= 16) hipLaunchKernelGGL(wstep_wave_kernel<true>, dim3(h->c.B), dim3(64), dyn, (hipStream_t)stream, h->c, h->lot, seed, (uint32_t)chain_offset,
                                         (int)iter0, (int)n_transitions, h->max_depth, theta_dev, draws_dev, h->stats);
    else hipLaunchKernelGGL(wstep_wave_kernel<false>, dim3(h->c.B), dim3(64), dyn, (hipStream_t)stream, h->c, h->lot, seed, (uint32_t)chain_offset,
                            (int)iter0, (int)n_transitions, h->max_depth, theta_dev, draws_dev, h->stats);
    WCHK(hipGetLastError());
    return WN_OK;
  }
  hipLaunchKernelGGL(wstep_auto_kernel, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c, h->lot, seed, (uint32_t)chain_offset,
                     (int)iter0, (int)n_transitions, h->max_depth, theta_dev, draws_dev, h->stats);
  WCHK(hipGetLastError());
  return WN_OK;
}

int wn_wstep_set_stats(wn_wstep* h, int32_t* stats_dev) {
  if (!h) return WN_E_ARG;
  h->stats = stats_dev;
  return WN_OK;
}

int wn_wstep_result(wn_wstep* h, double* theta_sel_dev, void* stream) {
  if (!h || !theta_sel_dev) { if (h) h->err = "null buffer"; return WN_E_ARG; }
  WCHK(hipSetDevice(h->device));
  if (wave_path(h)) {
    hipLaunchKernelGGL(wstep_wv_result_kernel, dim3(h->c.B), dim3(64), 0, (hipStream_t)stream, h->c.B, h->c.D, (const double*)h->wst,
                       (int)WvChain<true>::ST_DOUBLES, theta_sel_dev);
    WCHK(hipGetLastError());
    return WN_OK;
  }
  hipLaunchKernelGGL(wstep_result_kernel, dim3((h->c.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->c, theta_sel_dev);
  WCHK(hipGetLastError());
  return WN_OK;
}

}  // extern "C"
