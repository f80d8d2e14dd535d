// wn_capi.hip -- the C ABI of include/walnuts_amd.h: engine object, kernel selection, launches.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/walnuts_amd.h"
#include "wn_launch.h"
#include "wn_policy_coop.h"  // CK_*
#include "wn_policy_wave.h"  // EXT_*
#include "wn_target_eval.h"

using namespace wn;

namespace wn {
size_t lane_work_doubles(int dmax, int C, int M);
size_t coop_work_doubles(int NT, int W, int C, int M);

static int lane_dmax(int target, int D) {
  switch (target) {
    case T_STDGAUSS:
    case T_DIAGGAUSS:
      return D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 24 ? 24 : 0;
    case T_DENSEGAUSS:
      return D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : 0;
    case T_FUNNEL10: return D == 11 ? 11 : 0;
    case T_CORRGAUSS:
    case T_SMILE:
    case T_FUNNEL1:
    case T_MODFUNNEL: return D == 2 ? 2 : 0;
  }
  return 0;
}

int choose_kernel(int target, int D, int waves_per_group, int integrator, int n_chains, unsigned route, KernelChoice* kc) {
  kc->G = 1;
  kc->route = route;
  const bool flow = integrator == I_ADAPT_FLOWD;  // (round 5: it runs in the cooperative MFMA layout too, one or two tiles per wave)
  const bool flow_nocoop = (route & WN_ROUTE_FLOW_NOCOOP) != 0;  // A-B: FlowD as before round 5 (lane / wave / split)
#ifdef WN_DIAG
  const int flow_ksmax = getenv("WN_FLOW_KSMAX") ? atoi(getenv("WN_FLOW_KSMAX")) : 32;  // diagnostic: largest k-step count FlowD takes to the coop layout
#else
  const int flow_ksmax = 32;
#endif
  char* nm = kc->name;  // per-choice storage: the engine owns its kernel name
  nm[0] = 0;
  const bool force_wave = (route & WN_ROUTE_FORCE_WAVE) != 0;  // std / diagonal Gaussians on the wave layout
  int dm = lane_dmax(target, D);
  if (force_wave && (target == T_STDGAUSS || target == T_DIAGGAUSS)) dm = 0;
  // Gaussians small enough for one chain per lane use it only when there are chains to fill the chip's
  // waves (64 per wave): at 4096 chains it occupies 64 of 1024 SIMDs and the 16-chain MFMA-group kernel is
  // 2-5x faster (std 16-D: fixedLeapFrog 5.4e8 vs 2.4e8, R2P 5.1e8 vs 9.5e7 evals/s; dense 16-D 7.7e8 vs 1.9e8),
  // while the lane kernel scales with the chain count (std 10-D fixedLeapFrog: 5.4e8 / 1.1e9 / 3.4e9 at 8k / 16k /
  // 64k chains against a flat 5.2e8; R2P crosses at 16k).  adaptLeapFrogFlowD has no MFMA-group kernel.
  const bool lane_gauss = (route & WN_ROUTE_LANE_GAUSS) != 0;  // always per lane
  if (!lane_gauss && waves_per_group != 64 && (target == T_STDGAUSS || target == T_DIAGGAUSS || target == T_DENSEGAUSS) &&
      !(D <= 4 && integrator == I_FIXED) && !(target == T_DENSEGAUSS && flow && flow_nocoop) &&
      n_chains < (integrator == I_FIXED ? 8192 : 16384))
    dm = 0;
  if ((target == T_CORRGAUSS || target == T_SMILE || target == T_FUNNEL1 || target == T_MODFUNNEL) && D == 2 &&
      waves_per_group != 64 && !lane_gauss && integrator != I_FIXED && n_chains < 16384) {
    // adaptive integrators with too few chains to fill the chip 64 per wave (4096 chains, R2P: 4.8e8 vs 3.8e8
    // evals/s on corrGauss; with fixedLeapFrog the lane kernel wins, 9.2e8 vs 5.0e8)
    kc->layout = 2; kc->p0 = 1; kc->p1 = target; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=1,target=%d>", target);
    return 0;
  }
  if (target == T_FUNNEL10 && waves_per_group != 64) {
    // per-chain work on the funnel is wildly uneven: one chain per wave (see WTgtFunnel10; measured at 4096 chains:
    // R2P 7.1e7 vs 4.5e7, fixedLeapFrog 1.6e8 vs 1.2e8 evaluations/s); waves_per_group = 64 asks for the
    // one-chain-per-lane kernel
    kc->layout = 2; kc->p0 = 1; kc->p1 = T_FUNNEL10; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=1,funnel10>");
    return 0;
  }
  if (dm) {
    kc->layout = 0; kc->p0 = dm; kc->p1 = target; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_lane_kernel<DMAX=%d,target=%d>", dm, target);
    return 0;
  }
  if (target == WN_T_STOCKWATSON) {
    if (D > 768 || D % 3) return -1;
    kc->layout = 2; kc->p0 = 12; kc->p1 = T_STOCKWATSON; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=12,stockwatson>");
    return 0;
  }
  if (target == WN_T_EXTERNAL) {  // caller-evaluated (logp, grad): split mode
    if (D > 1024) return -1;
    const int NE = D <= 512 ? 8 : 16;
    kc->layout = 3; kc->p0 = NE; kc->p1 = EXT_LP; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_split_tick_kernel<NE=%d,external>", NE);
    return 0;
  }
  if (target == T_DENSEGAUSS && D >= 2 && D <= 128 && integrator != I_FIXED && integrator != I_ADAPT_FLOWD && !dm && !(route & WN_ROUTE_DENSE_COOP)) {
    // (adaptLeapFrogFlowD stays on the cooperative layout: 3.1e8 against 2.7e8 evaluations/s here)
    // the adaptive integrators on the headline geometry (round 5): one chain per wavefront, the matrix in LDS shared by a workgroup's
    // chains (wn_k_wavelds.hip) -- a macro step is per-chain control flow, not a wide tile; WN_ROUTE_DENSE_COOP: the cooperative layout below
    kc->layout = 2; kc->p0 = 2; kc->p1 = T_DENSEGAUSS; kc->W = 1;
    snprintf(nm, sizeof(kc->name), "wn_wavelds_kernel<NE=2,dense,K=8>");
    return 0;
  }
  if (target == T_DENSEGAUSS || target == T_DIAGGAUSS || target == T_STDGAUSS) {
    const int W = (waves_per_group && waves_per_group != 64) ? waves_per_group : 4;  // 64 = "one chain per lane if possible"
    if (W != 4) return -1;
    const int ks = (D + 3) / 4;
    int KS = 0, NT = 0;
    if ((flow && (flow_nocoop || ks > flow_ksmax)) || (force_wave && target != T_DENSEGAUSS)) {
    } else if (target == T_DENSEGAUSS) {
      if (ks <= 16) { KS = 16; NT = 1; }
      else if (ks <= 25) { KS = 25; NT = 2; }
      else if (ks <= 32) { KS = 32; NT = 2; }
    } else {
      if (ks <= 16) { KS = 16; NT = 1; }
      else if (ks <= 32) { KS = 32; NT = 2; }
      // 129 <= D <= 256: four tiles per wave fit the register file only for fixedLeapFrog; the adaptive
      // kernels spill and run 5.5x slower than one chain per wavefront (std 256-D R2P: 6.7e7 vs 3.6e8 evals/s)
      else if (ks <= 64 && integrator == I_FIXED) { KS = 64; NT = 4; }
    }
    if (!KS && D > 4096 && D <= 64 * 4096 && target != T_DENSEGAUSS) {
      // std / diagonal Gaussian up to D = 2^18 (the range of the reference's Gaussian scaling scripts,
      // mainGaussESS.py:20): a cluster of ceil(D/4096) workgroups per chain keeps the chain's vectors in
      // registers; only the partial sums of the reductions cross workgroups
      kc->layout = 2; kc->p0 = 64; kc->p1 = T_DIAGGAUSS; kc->W = 4; kc->G = (D + 4095) / 4096;
      snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=16,W=4,diag,cluster=%d>", kc->G);
        return 0;
    }
    if (!KS && D > 1024 && D <= 4096 && target != T_DENSEGAUSS) {
      // std / diagonal Gaussian, 1024 < D <= 4096: four waves per chain
      kc->layout = 2; kc->p0 = 64; kc->p1 = T_DIAGGAUSS; kc->W = 4;
      snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=16,W=4,diag>");
        return 0;
    }
    if (!KS && D <= 1024) {
      // one chain per wavefront (wn_policy_wave.h): fused std / diagonal Gaussian, or the dense
      // Gaussian with its gradient from the batched MFMA kernel between chain-kernel launches
      const int NE = D <= 512 ? 8 : 16;
      kc->W = 1; kc->p0 = NE;
      if (target == T_DENSEGAUSS) {
        kc->layout = 3; kc->p1 = EXT_HALF_QG;
        snprintf(nm, sizeof(kc->name), "wn_split_tick_kernel<NE=%d,dense>+wn_dense_grad_kernel", NE);
      } else {
        kc->layout = 2; kc->p1 = T_DIAGGAUSS;
        snprintf(nm, sizeof(kc->name), "wn_wave_kernel<NE=%d,diag>", NE);
      }
        return 0;
    }
    if (!KS) return -1;
    kc->layout = 1; kc->p0 = KS; kc->p1 = (target == T_DENSEGAUSS) ? CK_DENSE : CK_DIAG; kc->W = W;
    snprintf(nm, sizeof(kc->name), "wn_coop_kernel<NT=%d,W=%d,KS=%d,%s>", NT, W, KS,
             kc->p1 == CK_DENSE ? "dense" : "diag");
    return 0;
  }
  return -1;
}

size_t kernel_work_doubles(const KernelChoice& kc, int C, int M) {
  if (kc.layout == 0) return lane_work_doubles(kc.p0, C, M);
  if (kc.layout == 2) return std::max(wave_work_doubles(kc.p0, C, M, kc.G), orbit_wide_geom(kc) ? std::max(orbit_wide_work_doubles(kc.p0, C, M), orbit2_work_doubles(kc.p0, C, M)) : (size_t)0);
  if (kc.layout == 3) return split_work_doubles(kc.p0, C, M);
  const int NT = kc.p0 <= 16 ? 1 : kc.p0 <= 32 ? 2 : 4;
  // the role-split kernel of the same layout (wn_k_fast.hip) pads every vector to 8 tiles x 4 registers x 64 lanes
  // whatever the dimension: (M+1) slots x (q, v) + 2 proposals + the momentum normals per 16-chain group
  const size_t fast = (((size_t)C + 15) / 16) * (size_t)((M + 1) * 2 + 3) * 2048;
  const size_t wide = orbit_wide_geom(kc) ? orbit_wide_work_doubles(orbit_wide_ne(kc), C, M) : (size_t)0;  // (std / diagonal Gaussian: wn_k_orbit.hip's wide mode)
  return std::max(std::max(std::max(coop_work_doubles(NT, kc.W, C, M), fast), fast4_work_doubles(C, M)), wide);
}

// sw_innov.stan:7-51, scalar form (wn_lpfun only; the sampler evaluates WTgtStockWatson in the wave kernel)
__device__ double wn_stockwatson_eval(const double* tpar, const double* x, double* g) {
  double r = 0.0;
  {
      const int T = (int)tpar[0];
      const double* y = tpar + 1;
      const double tS = x[0], sigma = exp(-0.5 * tS), gpri = exp(tS);
      const double *zin = x + 1, *xin = x + T, *tin = x + 2 * T;
      double z[257], xs[257], ez[257], a[258], b[258];
      z[1] = zin[0];
      for (int t = 2; t <= T - 1; ++t) z[t] = z[t - 1] + sigma * zin[t - 1];
      xs[1] = xin[0];
      for (int t = 2; t <= T; ++t) xs[t] = xs[t - 1] + sigma * xin[t - 1];
      double tau = tin[0];
      r = 5.0 * tS - 0.5 * gpri;
      for (int j = 1; j <= T - 2; ++j) r += -0.5 * zin[j] * zin[j];
      for (int j = 1; j <= T - 1; ++j) r += -0.5 * xin[j] * xin[j] - 0.5 * tin[j] * tin[j];
      for (int t = 1; t <= T; ++t) {
        if (t >= 2) { ez[t - 1] = exp(0.5 * z[t - 1]); tau += ez[t - 1] * tin[t - 1]; }
        const double w = exp(-xs[t]), res = y[t - 1] - tau;
        r += -0.5 * xs[t] - 0.5 * (res * res) * w;
        a[t] = res * w; b[t] = -0.5 + 0.5 * (res * res) * w;
      }
      double acc = 0.0, cacc = 0.0, dsig = 0.0;
      a[T + 1] = 0.0;
      for (int t = T; t >= 1; --t) a[t] += a[t + 1];  // A[t] = sum_{t' >= t} a[t']
      g[2 * T] = a[1];
      // c[k] = 1/2 ez[k] tauinn[k] A[k+1], k = T-1 .. 1
      for (int k = T - 1; k >= 1; --k) {
        g[2 * T + k] = ez[k] * a[k + 1] - tin[k];
        const double c = 0.5 * ez[k] * tin[k] * a[k + 1];
        if (k <= T - 2) g[1 + k] = sigma * cacc - zin[k];
        cacc += c;
        dsig += c * (z[k] - z[1]);
      }
      g[1] = cacc;
      for (int t = T; t >= 1; --t) {
        if (t <= T - 1) g[T + t] = sigma * acc - xin[t];
        acc += b[t];
        dsig += b[t] * (xs[t] - xs[1]);
      }
      g[T] = acc;
      g[0] = 5.0 - 0.5 * gpri - 0.5 * dsig;
  }
  return r;
}

__global__ void wn_lpfun_kernel(int kind, int D, int B, const double* tpar, const double* q,
                                double* lp, double* grad) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  lp[c] = kind == T_STOCKWATSON ? wn_stockwatson_eval(tpar, q + (size_t)c * D, grad + (size_t)c * D)
                                : wn_target_eval(kind, D, tpar, q + (size_t)c * D, grad + (size_t)c * D);
}

__global__ void wn_tape_kernel(uint64_t seed, uint32_t chain0, int C, int iter, int nu, int D,
                               double* u, double* z) {
  int c = blockIdx.x;
  if (c >= C) return;
  if (u)
    for (int k = threadIdx.x; k < nu; k += blockDim.x)
      u[(size_t)c * nu + k] = rng_uniform(seed, chain0 + c, (uint32_t)iter, (uint32_t)k);
  if (z)
    for (int j = threadIdx.x; j < D; j += blockDim.x)
      z[(size_t)c * D + j] = rng_normal(seed, chain0 + c, (uint32_t)iter, (uint32_t)j);
}

// generated = param_constrain(q, include_tp=True) for sw_innov.stan (mainSW.py:28-29; sw_innov.stan:17-35): each row of
// `samples` holds the 3T unconstrained parameters (tSigma, z1, zinn[T-2], x1, xinn[T-1], tau1, tauinn[T-1]) at its head;
// append sigma, gpri, z[T-1], x[T], tau[T].  One thread per draw, the recursions in the model's own order.
__global__ void wn_sw_constrain_kernel(double* samples, size_t nrows, int T) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int dg = 6 * T + 1;
  double* row = samples + r * dg;
  const double tS = row[0], sigma = exp(-0.5 * tS), gpri = exp(tS);
  const double *zinn = row + 2, *xinn = row + T + 1, *tin = row + 2 * T + 1;
  double* o = row + 3 * T;
  o[0] = sigma; o[1] = gpri;
  double *z = o + 2, *x = z + (T - 1), *tau = x + T;
  z[0] = row[1];
  for (int t = 1; t < T - 1; ++t) z[t] = z[t - 1] + sigma * zinn[t - 1];
  x[0] = row[T];
  for (int t = 1; t < T; ++t) x[t] = x[t - 1] + sigma * xinn[t - 1];
  tau[0] = row[2 * T];
  for (int t = 1; t < T; ++t) tau[t] = tau[t - 1] + exp(0.5 * z[t - 1]) * tin[t - 1];
}

// test hook (wn_debug_poison_lds): every workgroup owns a whole CU's LDS (160 KiB) and fills it with a bit pattern
__global__ __launch_bounds__(256) void wn_poison_lds_kernel(unsigned long long pattern, unsigned long long* sink) {
  extern __shared__ unsigned long long plds[];
  const int n = 160 * 1024 / 8;
  for (int i = threadIdx.x; i < n; i += blockDim.x) plds[i] = pattern;
  __syncthreads();
  if (threadIdx.x == 0 && plds[blockIdx.x % n] == 1ull) sink[0] = plds[1];  // (keeps the stores alive)
  // a little work so that the 4 x CUs workgroups spread over the whole chip instead of draining through a few CUs
  for (int k = 0; k < 200; ++k) __builtin_amdgcn_s_sleep(32);
}

__global__ void wn_fill_kernel(double* p, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace wn

struct wn_engine {
  wn_config cfg;
  int target = -1;
  double* tpar = nullptr;
  size_t ntpar = 0;
  double* work = nullptr;
  size_t work_doubles = 0;
  double* facs = nullptr;  // warm-up: sorted energyErrorInfFacs, [C][2][warmupIter]
  // split mode (layout 3): zero-padded precision matrix [DP][DP], position / gradient / log-density
  // exchange buffers [Cpad][DP], [Cpad], finished-chain counter, parameters of the run in flight
  double *apad = nullptr, *xq = nullptr, *xg = nullptr;
  int* sp_done = nullptr;
  // cluster layout: partial sums [C][2][G][8], arrival counters [C] followed by the abort flag
  double* cl_red = nullptr;
  unsigned* cl_cnt = nullptr;
  int DP = 0, Cpad = 0;
  Params sp_prm{};
  bool sp_active = false;
  long long sp_ticks = 0;
  KernelChoice kc{};
  bool have_kc = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int* h_poll = nullptr;          // pinned host words the split-mode loop polls the finished-chain counter through
  hipEvent_t ev_poll[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [part][batch parity]
  hipStream_t sp_side[3] = {nullptr, nullptr, nullptr};  // split mode: the streams of parts 1..3
  int* sp_done_side[3] = {nullptr, nullptr, nullptr};     // ... their finished-chain counters and pinned poll words: allocations of their own
  int* h_poll_side[3] = {nullptr, nullptr, nullptr};      // (copies on two streams that touch one allocation are ordered by the runtime: the parts ran one after the other)
  hipEvent_t sp_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t sp_fork1 = nullptr;
  hipStream_t sp_main = nullptr;  // split mode: stands in for the (uncapturable) legacy default stream
  hipEvent_t sp_fork0 = nullptr;
  int* kabort = nullptr;  // device flag: a kernel's bounded loop gave up (internal error)
  // The flag is latched into a run counter on the device behind every launch and copied to a pinned word in stream order
  // (asynchronously): nothing of the caller's stream is kept, and a later launch does not wait for an earlier one.
  int* h_abort = nullptr;         // pinned: number of runs that gave up so far, as of the last completed copy
  hipEvent_t ev_abort = nullptr;  // recorded behind the last copy
  bool kabort_pending = false;    // a launch since the counter was last read
  int aborts_reported = 0;
  bool timed = false;
  int debug_cluster_drop = 0;  // test hook (wn_debug_cluster_drop)
  std::string err;
};

static std::string g_create_err;

#define HIPCHK(e, call)                                                        \
  do {                                                                         \
    hipError_t _r = (call);                                                    \
    if (_r != hipSuccess) {                                                    \
      (e)->err = std::string(#call) + ": " + hipGetErrorString(_r);            \
      return WN_E_HIP;                                                         \
    }                                                                          \
  } while (0)

// device allocations: out-of-memory is the caller's problem to size (WN_E_ALLOC), anything else a HIP error
#define HIPALLOC(e, ptr, bytes)                                                \
  do {                                                                         \
    hipError_t _r = hipMalloc((void**)&(ptr), (bytes));                        \
    if (_r != hipSuccess) {                                                    \
      (ptr) = nullptr;                                                         \
      (void)hipGetLastError();                                                 \
      (e)->err = std::string("hipMalloc(" #ptr ", ") + std::to_string((size_t)(bytes)) + " B): " + hipGetErrorString(_r); \
      return _r == hipErrorOutOfMemory ? WN_E_ALLOC : WN_E_HIP;                \
    }                                                                          \
  } while (0)

#ifdef WN_DIAG
// diagnostic builds: the routing bits from the environment variables ABI <= 4 read inside the library (profiling scripts under tools/)
static unsigned wn_diag_route_env() {
  static const struct { const char* name; unsigned bit; } tab[] = {
      {"WN_ORBIT_LEGACY", WN_ROUTE_ORBIT_LEGACY}, {"WN_ORBIT_NOWIDE", WN_ROUTE_ORBIT_NOWIDE}, {"WN_ORBIT_WIDE_ADAPT", WN_ROUTE_ORBIT_WIDE_ADAPT},
      {"WN_ORBIT_ONEWAVE", WN_ROUTE_ORBIT_ONEWAVE}, {"WN_ORBIT_EXACT", WN_ROUTE_ORBIT_EXACT}, {"WN_ORBIT2_NOADAPT", WN_ROUTE_ORBIT2_NOADAPT},
      {"WN_ORBIT2_ANYC", WN_ROUTE_ORBIT2_ANYC}, {"WN_DIAG_COOP", WN_ROUTE_DIAG_COOP}, {"WN_DENSE_COOP", WN_ROUTE_DENSE_COOP},
      {"WN_FAST_LEGACY", WN_ROUTE_FAST_LEGACY}, {"WN_COOP_LEGACY", WN_ROUTE_COOP_LEGACY}, {"WN_FAST4_WHOLE", WN_ROUTE_FAST4_WHOLE},
      {"WN_FLOW_NOCOOP", WN_ROUTE_FLOW_NOCOOP}, {"WN_FORCE_WAVE", WN_ROUTE_FORCE_WAVE}, {"WN_LANE_GAUSS", WN_ROUTE_LANE_GAUSS},
      {"WN_SPLIT_EAGER", WN_ROUTE_SPLIT_EAGER}};
  unsigned r = 0;
  for (const auto& t : tab) if (getenv(t.name)) r |= t.bit;
  return r;
}
#endif

extern "C" {

int wn_abi_version(void) { return WN_ABI_VERSION; }

int wn_debug_cluster_drop(wn_engine* e, int32_t exchange) {
  if (!e || exchange < 0) return WN_E_ARG;
  e->debug_cluster_drop = exchange;
  return WN_OK;
}

const char* wn_last_error(wn_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int wn_create(const wn_config* cfg, wn_engine** out) {
  if (!cfg || !out) { g_create_err = "null argument"; return WN_E_ARG; }
  *out = nullptr;
  if (cfg->abi_version != WN_ABI_VERSION) { g_create_err = "abi_version mismatch"; return WN_E_ARG; }
  // WALNUTS.py:140,146 sys.exit on bad tuning targets; the analogous argument checks here
  if (cfg->dim < 1 || cfg->n_chains < 1) { g_create_err = "dim and n_chains must be >= 1"; return WN_E_ARG; }
  // the random stream is keyed by a 32-bit global chain id (wn_rng.h: one Philox counter word)
  if (cfg->chain_offset < 0 || (uint64_t)cfg->chain_offset + (uint64_t)cfg->n_chains > (1ull << 32)) {
    g_create_err = "chain_offset + n_chains must stay within [0, 2^32) (global chain ids are 32-bit)"; return WN_E_ARG;
  }
  if (cfg->M < 1 || cfg->M > MAX_M) { g_create_err = "M out of range [1,24]"; return WN_E_ARG; }
  if (cfg->integrator < WN_I_FIXED || cfg->integrator > WN_I_ADAPT_FLOWD) { g_create_err = "unknown integrator"; return WN_E_ARG; }
  if (cfg->min_c < 0 || cfg->max_c < cfg->min_c || cfg->max_c > 20) { g_create_err = "need 0 <= minC <= maxC <= 20"; return WN_E_ARG; }
  if (!(cfg->H0 > 0.0) || !(cfg->delta0 > 0.0) || cfg->step_size_rand_scale < 0.0 || cfg->step_size_rand_scale >= 1.0) {
    g_create_err = "need H0 > 0, delta0 > 0, 0 <= stepSizeRandScale < 1"; return WN_E_ARG;
  }
  if (!(cfg->r2p_prob0 > 0.0 && cfg->r2p_prob0 < 1.0)) { g_create_err = "R2Pprob0 must be in (0,1)"; return WN_E_ARG; }
  if (cfg->warmup_iter < 0) { g_create_err = "warmupIter must be >= 0"; return WN_E_ARG; }
  if (cfg->adapt_quant_bits != 0 && (cfg->adapt_quant_bits < 8 || cfg->adapt_quant_bits > 52)) { g_create_err = "adapt_quant_bits must be 0 (off) or in [8, 52]"; return WN_E_ARG; }
  if (cfg->split_parts < 0 || cfg->split_parts > 4 || cfg->split_ks < 0 || cfg->split_ks > 2) { g_create_err = "split_parts must be in [0, 4], split_ks in [0, 2]"; return WN_E_ARG; }
  if (cfg->route >> 17) { g_create_err = "unknown WN_ROUTE_* bit"; return WN_E_ARG; }
  if (cfg->warmup_iter > 0) {  // WALNUTS.py:140,146 sys.exit("bad adaptHtarget") / ("bad adaptDeltaTarget")
    if (cfg->adapt_h && (cfg->adapt_h_target < 0.0 || cfg->adapt_h_target > 1.0)) { g_create_err = "bad adaptHtarget"; return WN_E_ARG; }
    if (cfg->adapt_delta && cfg->adapt_delta_target < 0.0) { g_create_err = "bad adaptDeltaTarget"; return WN_E_ARG; }
    if (cfg->adapt_delta && !(cfg->adapt_delta_quantile >= 0.0 && cfg->adapt_delta_quantile <= 1.0)) { g_create_err = "adaptDeltaQuantile must be in [0,1]"; return WN_E_ARG; }
  }
  if (cfg->generated < WN_G_IDENTITY || cfg->generated > WN_G_SW_CONSTRAIN) { g_create_err = "unknown generated kind"; return WN_E_ARG; }
  if (cfg->generated == WN_G_SW_CONSTRAIN && (cfg->dim % 3 || cfg->dim < 9)) { g_create_err = "WN_G_SW_CONSTRAIN needs dim = 3T"; return WN_E_ARG; }
  if (cfg->generated == WN_G_FIRSTK && (cfg->gen_k < 1 || cfg->gen_k > cfg->dim)) { g_create_err = "gen_k out of range"; return WN_E_ARG; }
  wn_engine* e = new wn_engine();
  e->cfg = *cfg;
#ifdef WN_DIAG
  e->cfg.route |= wn_diag_route_env();  // diagnostic builds only (tools/build_variant.sh): the switches of ABI <= 4's environment
  if (getenv("WN_SPLIT_PARTS")) e->cfg.split_parts = atoi(getenv("WN_SPLIT_PARTS"));
  if (getenv("WN_SPLIT_KS")) e->cfg.split_ks = atoi(getenv("WN_SPLIT_KS"));
#endif
  hipError_t r = hipSetDevice(cfg->device);
  if (r != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(r); delete e; return WN_E_HIP; }
  r = hipEventCreate(&e->ev0);
  if (r == hipSuccess) r = hipEventCreate(&e->ev1);
  if (r != hipSuccess) { g_create_err = std::string("hipEventCreate: ") + hipGetErrorString(r); delete e; return WN_E_HIP; }
  r = hipMalloc((void**)&e->kabort, 64);
  if (r == hipSuccess) r = hipMemset(e->kabort, 0, 64);
  if (r == hipSuccess) r = hipHostMalloc((void**)&e->h_abort, 64);
  if (r == hipSuccess) { e->h_abort[0] = 0; r = hipEventCreateWithFlags(&e->ev_abort, hipEventDisableTiming); }
  if (r != hipSuccess) { g_create_err = std::string("hipMalloc: ") + hipGetErrorString(r); (void)hipGetLastError(); delete e; return r == hipErrorOutOfMemory ? WN_E_ALLOC : WN_E_HIP; }
  *out = e;
  return WN_OK;
}

static int check_abort(wn_engine* e, bool wait);

int wn_destroy(wn_engine* e) {
  if (!e) return WN_OK;
  hipSetDevice(e->cfg.device);
  // a run that gave up and was never asked about (wn_check_run / wn_last_run_ms) is reported here: the handle is released all the same
  const int rc_abort = e->kabort_pending ? check_abort(e, true) : WN_OK;
  if (rc_abort) { g_create_err = e->err; fprintf(stderr, "walnuts_amd: %s\n", e->err.c_str()); }
  if (e->h_abort) hipHostFree(e->h_abort);
  if (e->ev_abort) hipEventDestroy(e->ev_abort);
  if (e->tpar) hipFree(e->tpar);
  if (e->work) hipFree(e->work);
  if (e->facs) hipFree(e->facs);
  if (e->apad) hipFree(e->apad);
  if (e->xq) hipFree(e->xq);
  if (e->xg) hipFree(e->xg);
  if (e->sp_done) hipFree(e->sp_done);
  if (e->cl_red) hipFree(e->cl_red);
  if (e->cl_cnt) hipFree(e->cl_cnt);
  if (e->kabort) hipFree(e->kabort);
  if (e->h_poll) hipHostFree(e->h_poll);
  for (hipEvent_t ev : e->ev_poll) if (ev) hipEventDestroy(ev);
  if (e->sp_main) hipStreamDestroy(e->sp_main);
  for (hipStream_t ss : e->sp_side) if (ss) hipStreamDestroy(ss);
  for (int* d : e->sp_done_side) if (d) hipFree(d);
  for (int* h : e->h_poll_side) if (h) hipHostFree(h);
  for (hipEvent_t ev : e->sp_join) if (ev) hipEventDestroy(ev);
  if (e->sp_fork1) hipEventDestroy(e->sp_fork1);
  if (e->sp_fork0) hipEventDestroy(e->sp_fork0);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  delete e;
  return rc_abort;
}

int wn_set_target(wn_engine* e, int32_t kind, const double* params_dev, size_t n_params) {
  if (!e) return WN_E_ARG;
  const int D = e->cfg.dim;
  size_t need = 0;
  switch (kind) {
    case WN_T_STDGAUSS: need = 0; break;
    case WN_T_EXTERNAL: need = 0; break;
    case WN_T_STOCKWATSON:
      if (D % 3 || D < 9 || D > 768) { e->err = "StockWatson needs dim = 3T, 3 <= T <= 256"; return WN_E_ARG; }
      need = (size_t)D / 3 + 1;
      break;
    case WN_T_DIAGGAUSS: need = (size_t)D; break;
    case WN_T_DENSEGAUSS: need = (size_t)D * D; break;
    case WN_T_FUNNEL10: if (D != 11) { e->err = "funnel10 needs dim 11"; return WN_E_ARG; } break;
    case WN_T_CORRGAUSS: case WN_T_SMILE: case WN_T_FUNNEL1: case WN_T_MODFUNNEL:
      if (D != 2) { e->err = "target needs dim 2"; return WN_E_ARG; } break;
    default: e->err = "unknown target kind"; return WN_E_ARG;
  }
  if (need && (n_params != need || !params_dev)) { e->err = "wrong parameter count for target"; return WN_E_ARG; }
  if (e->cfg.generated == WN_G_SW_CONSTRAIN && kind != WN_T_STOCKWATSON) { e->err = "generated = WN_G_SW_CONSTRAIN needs the StockWatson target"; return WN_E_ARG; }
  KernelChoice kc;
  if (choose_kernel(kind, D, e->cfg.waves_per_group, e->cfg.integrator, e->cfg.n_chains, e->cfg.route, &kc)) {
    e->err = "no device kernel for this (target, dim) yet";
    return WN_E_UNSUPPORTED;
  }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  if (kc.G > 1) {
    // every workgroup of a chain spins on its siblings: the whole grid must be resident (one 4-wave
    // workgroup of 512-register waves per CU).  Checked before anything of the previous target is freed.
    hipDeviceProp_t prop;
    HIPCHK(e, hipGetDeviceProperties(&prop, e->cfg.device));
    if ((long long)e->cfg.n_chains * kc.G > prop.multiProcessorCount) {
      e->err = "cluster layout: n_chains * ceil(dim / 4096) must not exceed the number of CUs; run fewer chains per call";
      return WN_E_UNSUPPORTED;
    }
  }
  // from here on the previous target is gone: a failure below leaves the engine without a target
  // (wn_run then returns WN_E_NOTARGET) instead of an old kernel on buffers sized for the new one
  e->have_kc = false;
  e->target = -1;
  e->kc = KernelChoice{};
  if (e->tpar) { hipFree(e->tpar); e->tpar = nullptr; }
  if (e->work) { hipFree(e->work); e->work = nullptr; }
  size_t nt = need ? need : (size_t)D;  // stdGauss runs as the diagonal target with a == 1
  HIPALLOC(e, e->tpar, nt * sizeof(double));
  if (need) HIPCHK(e, hipMemcpy(e->tpar, params_dev, need * sizeof(double), hipMemcpyDeviceToDevice));
  else {
    hipLaunchKernelGGL(wn_fill_kernel, dim3((nt + 255) / 256), dim3(256), 0, 0, e->tpar, nt, 1.0);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipDeviceSynchronize());
  }
  e->ntpar = nt;
  e->work_doubles = kernel_work_doubles(kc, e->cfg.n_chains, e->cfg.M);
  HIPALLOC(e, e->work, e->work_doubles * sizeof(double));
  HIPCHK(e, hipMemset(e->work, 0, e->work_doubles * sizeof(double)));
  if (e->facs) { hipFree(e->facs); e->facs = nullptr; }
  if (e->cfg.warmup_iter > 0)  // one sorted copy per chain (per workgroup of a cluster: every replica keeps its own)
    HIPALLOC(e, e->facs, (size_t)e->cfg.n_chains * kc.G * 2 * e->cfg.warmup_iter * sizeof(double));
  if (e->apad) { hipFree(e->apad); e->apad = nullptr; }
  if (e->xq) { hipFree(e->xq); e->xq = nullptr; }
  if (e->xg) { hipFree(e->xg); e->xg = nullptr; }
  e->sp_active = false;
  if (e->cl_red) { hipFree(e->cl_red); e->cl_red = nullptr; }
  if (e->cl_cnt) { hipFree(e->cl_cnt); e->cl_cnt = nullptr; }
  if (kc.G > 1) {
    HIPALLOC(e, e->cl_red, (size_t)e->cfg.n_chains * 2 * kc.G * 8 * sizeof(double));
    HIPALLOC(e, e->cl_cnt, ((size_t)e->cfg.n_chains + 16) * sizeof(unsigned));
  }
  if (kc.layout == 3) {
    e->DP = 64 * kc.p0;
    e->Cpad = (e->cfg.n_chains + 63) / 64 * 64;
    const size_t nx = (size_t)e->Cpad * e->DP;
    if (!e->sp_done) HIPALLOC(e, e->sp_done, 64);
    if (kind == WN_T_DENSEGAUSS) {
      HIPALLOC(e, e->xq, nx * sizeof(double));
      HIPALLOC(e, e->xg, nx * sizeof(double));
      HIPCHK(e, hipMemset(e->xq, 0, nx * sizeof(double)));
      HIPCHK(e, hipMemset(e->xg, 0, nx * sizeof(double)));
      HIPALLOC(e, e->apad, (size_t)e->DP * e->DP * sizeof(double));
      HIPCHK(e, hipMemset(e->apad, 0, (size_t)e->DP * e->DP * sizeof(double)));
      HIPCHK(e, hipMemcpy2D(e->apad, (size_t)e->DP * sizeof(double), params_dev, (size_t)D * sizeof(double),
                            (size_t)D * sizeof(double), (size_t)D, hipMemcpyDeviceToDevice));
    }
  }
  const bool warm_run = e->cfg.warmup_iter > 0 && (e->cfg.adapt_h || e->cfg.adapt_delta);  // (otherwise nothing adapts: the plain kernels are exact)
  const bool fast4_geom = (kc.p1 != CK_DENSE || D <= 100) && !(kc.route & WN_ROUTE_FAST_LEGACY);     // wn_k_fast4.hip also runs the warm-up iterations
  if (kc.layout == 1 && e->cfg.integrator == WN_I_FIXED && (!warm_run || fast4_geom) && e->cfg.M <= 11 && D <= 112 && !(kc.p1 == CK_DENSE && kc.p0 == 32) &&
      !(kc.route & WN_ROUTE_COOP_LEGACY)) {
    // plain sampling runs of this engine go to the role-split kernel (wn_k_fast.hip); traced / orbit-statistics
    // runs and wn_integrator_step use the one-stream kernel of the same layout
    const int NT = kc.p0 <= 16 ? 1 : 2;
    snprintf(kc.name, sizeof(kc.name), "wn_fast_kernel<KS=%d,%s>", (kc.p1 != CK_DENSE && kc.p0 == 32) ? 28 : kc.p0, kc.p1 == CK_DENSE ? "dense" : "diag");
    (void)NT;
    // dense, D <= 100: wave pairs of four chains (wn_k_fast4.hip), traced runs too.  (4096 chains: 1.24e9 steps/s against the
    // seven-wave kernel's 1.12e9; 1024 chains are 256 pairs, one per CU, where that kernel fills 64 CUs: 3.9e8 against 2.8e8.)
    if (fast4_geom)
    {
      const int pr_ = fast4_pairs(e->cfg.n_chains);
      const bool halves = kc.p1 == CK_DENSE && D > 64 && pr_ == 1 && !(kc.route & WN_ROUTE_FAST4_WHOLE);  // (fast4_go: a group on two T waves)
      snprintf(kc.name, sizeof(kc.name), "wn_fast4_kernel<NS=%d%s,%s,pairs=%d%s>", D <= 32 ? 2 : D <= 64 ? 4 : D <= 96 ? 6 : 7, (kc.p1 == CK_DENSE && D > 96) ? ",tail" : "",
               kc.p1 == CK_DENSE ? "dense" : "diag", pr_, halves ? ",halves" : "");
    }
  }
  if (kc.layout == 2 && kc.p0 == 1 && kc.W == 1 && kc.G == 1 && e->cfg.M >= 1 &&
      (e->cfg.integrator == WN_I_FIXED || e->cfg.integrator == WN_I_ADAPT_D || e->cfg.integrator == WN_I_ADAPT_R2P) && !(kc.route & WN_ROUTE_ORBIT_LEGACY)) {
    // plain sampling runs go to the trajectory-first kernel (wn_k_orbit.hip); traced / orbit-statistics runs and
    // wn_integrator_step use the one-stream engine of the same layout
    const char* tn = kc.p1 == T_FUNNEL10 ? "funnel10" : kc.p1 == T_CORRGAUSS ? "corrGauss" : kc.p1 == T_SMILE ? "smile" : kc.p1 == T_FUNNEL1 ? "funnel1" : "modFunnel";
    snprintf(kc.name, sizeof(kc.name), "wn_orbit_kernel<%s,%s>", tn, e->cfg.integrator == WN_I_FIXED ? "fixedLeapFrog" : e->cfg.integrator == WN_I_ADAPT_D ? "adaptLeapFrogD" : "adaptLeapFrogR2P");
  }
  if ((e->cfg.warmup_iter == 0 || kc.p1 != T_STOCKWATSON || kc.layout != 2) && e->cfg.M >= 1 && orbit_wide_run(kc, e->cfg.integrator) && (kc.layout != 1 || e->cfg.dim <= 128 || (kc.p0 == 64 && e->cfg.dim <= 256)) &&
      !(kc.layout == 2 && kc.p1 == T_DIAGGAUSS && !(kc.p0 == 8 && e->cfg.dim > 128 && e->cfg.dim <= 512))) {
    // (round 5) StockWatson's plain sampling runs -- and the adaptive ones of dense / std / diagonal Gaussians of up to 128 dimensions: the
    // same kernel with vectors of 12 (2) registers per lane
    snprintf(kc.name, sizeof(kc.name), "wn_orbit_kernel<%s,%s>", kc.layout == 1 ? (e->cfg.dim <= 16 ? "diag,NE=1" : e->cfg.dim <= 128 ? "diag,NE=2,K=4" : "diag,NE=4,K=4") : kc.p1 == T_DIAGGAUSS ? (e->cfg.dim <= 256 ? "diag,NE=4,K=4" : "diag,NE=8,K=4") : kc.p1 == T_DENSEGAUSS ? "dense,NE=2,K=8" : "stockwatson,NE=12",
             e->cfg.integrator == WN_I_FIXED ? "fixedLeapFrog" : e->cfg.integrator == WN_I_ADAPT_D ? "adaptLeapFrogD" : "adaptLeapFrogR2P");
  }
  if (e->cfg.warmup_iter == 0 && e->cfg.M >= 1 && orbit2_run(kc, e->cfg.integrator, e->cfg.n_chains)) {
    // trajectory and bookkeeping on a wave each (wn_k_orbit2.hip)
    snprintf(kc.name, sizeof(kc.name), "wn_orbit2_kernel<stockwatson,NE=12,%s>", e->cfg.integrator == WN_I_FIXED ? "fixedLeapFrog" : e->cfg.integrator == WN_I_ADAPT_D ? "adaptLeapFrogD" : "adaptLeapFrogR2P");
  }
  e->kc = kc;
  e->have_kc = true;
  e->target = kind;
  return WN_OK;
}

// the fast kernels' bounded loops set kabort[0]; this one-thread kernel, enqueued behind every launch, counts such runs in kabort[1]
__global__ void wn_abort_latch_kernel(int* f) {
  if (f[0]) { f[1] += 1; f[0] = 0; }
}

// Has a run given up?  wait = false: only if the copy behind the last launch has completed (no synchronisation: used in front of a
// new launch, which must stay asynchronous); wait = true: waits for it (wn_check_run, wn_last_run_ms, wn_destroy).
static int check_abort(wn_engine* e, bool wait) {
  if (wait) { HIPCHK(e, hipEventSynchronize(e->ev_abort)); }
  else if (hipEventQuery(e->ev_abort) != hipSuccess) return WN_OK;  // still running: stays pending
  e->kabort_pending = false;
  const int n = e->h_abort[0];
  if (n != e->aborts_reported) {
    e->aborts_reported = n;
    e->err = "a kernel's tick loop exceeded its bound and gave up (internal error); the results of that run are incomplete";
    return WN_E_HIP;
  }
  return WN_OK;
}

static void fill_params(const wn_engine* e, Params* p) {
  memset(p, 0, sizeof(*p));
  const wn_config& c = e->cfg;
  p->D = c.dim; p->C = c.n_chains;
  p->integrator = c.integrator; p->M = c.M; p->minC = c.min_c; p->maxC = c.max_c;
  p->generated = c.generated; p->gen_k = c.gen_k;
  p->dg = c.generated == WN_G_IDENTITY ? c.dim : c.generated == WN_G_Q0_SUMSQ ? 2 : c.gen_k;
  if (c.generated == WN_G_SW_CONSTRAIN) {  // the sampler stores identity draws at the head of each (6T+1)-wide row; wn_sw_constrain_kernel fills the rest
    p->generated = G_IDENTITY;
    p->dg = 2 * c.dim + 1;
  }
  p->H = c.H0; p->s = c.step_size_rand_scale; p->delta = c.delta0; p->R2Pprob0 = c.r2p_prob0;
  p->logP = log(c.r2p_prob0); p->log1mP = log(1.0 - c.r2p_prob0);
  p->wtSumThresh = exp(WN_LOGZERO + 1.0);  // constants.py:14
  p->chain0 = (uint32_t)c.chain_offset;
  p->work = e->work;
  p->tpar = e->tpar;
  p->adaptH = c.adapt_h; p->adaptDelta = c.adapt_delta;
  p->adaptHtarget = c.adapt_h_target; p->adaptDeltaTarget = c.adapt_delta_target;
  p->adaptDeltaQuantile = c.adapt_delta_quantile;
  p->facs = e->facs;
  p->adaptQuantBits = c.adapt_quant_bits;
}

// split mode with the built-in dense-Gaussian gradient: begin, then {batched -A q, one tick of every
// chain} until every chain has finished.  The host polls the finished-chain counter once per batch
// of ticks; a launch whose chains have all finished is a few microseconds of early exits.
// the finished-chain counter goes to its pinned poll word by a one-thread kernel: a hipMemcpyAsync of four bytes held the HOST until
// the stream had drained (the kernel trace of a two-part run showed the parts taking turns, batch by batch)
__global__ void wn_poll_word_kernel(const int* src, int* dst) { *dst = *src; }

static int run_split_dense(wn_engine* e, Params p, hipStream_t st, int adapt, bool warm) {
  if (p.mode != MODE_SAMPLE) { e->err = "wn_integrator_step is not available in split mode"; return WN_E_UNSUPPORTED; }
  if (st == nullptr) {
    // The legacy default stream cannot be captured into a hipGraph -- and every caller that does not say otherwise is on it (until
    // round 3 the "graph" path of this function therefore always fell back to eager launches).  The run goes to a stream of the
    // engine's own, ordered behind the work already queued on the default stream; it is complete when this function returns.
    if (!e->sp_main) {
      HIPCHK(e, hipStreamCreateWithFlags(&e->sp_main, hipStreamNonBlocking));
      HIPCHK(e, hipEventCreateWithFlags(&e->sp_fork0, hipEventDisableTiming));
    }
    HIPCHK(e, hipEventRecord(e->sp_fork0, nullptr));
    HIPCHK(e, hipStreamWaitEvent(e->sp_main, e->sp_fork0, 0));
    st = e->sp_main;
  }
  // PARTS (opt-in: wn_config.split_parts = 2 [split_ks = 1|2]; measured in round 4).  The chains are independent, so the run
  // can be cut into sub-batches, each with its own stream, exchange rows, workspace, finished-chain counter and host thread, and
  // the chain kernel of one (HBM-bound) can run while the other is inside its gradient GEMM (matrix-pipe-bound).  Two parts with
  // 64 x 64 tiles (128 workgroups per GEMM, the two GEMMs side by side): +2.4 %, bit-identical draws -- each part's own
  // GEMM -> chain kernel -> GEMM sequence is as long as before and half the CUs idle during its chain kernel.  Two parts whose GEMMs
  // each fill the chip (k of a tile split over two workgroups that ADD into rows of xg the chain kernel leaves zeroed: KS = 2,
  // Params::xzero): -24 % -- the two GEMMs interleave instead of alternating (30 + 55 us) and a chain kernel that shares SIMDs with
  // FP64 MFMA waves takes 40 us instead of 13, at any s_setprio.  The default stays one part.
  int NP = 1;
  {
    const int parts_env = e->cfg.split_parts;
    int want = parts_env > 0 ? parts_env : 1;
#if defined(WN_STAMPS) || defined(WN_SPAN)
    want = 1;  // (the diagnostic rows are indexed by the local chain)
#endif
    const bool plain = !warm && !p.trace && !p.trace_n && !p.omin && !p.omax;
    while (want > 1 && !(plain && p.C % (64 * want) == 0 && want <= 4)) want >>= 1;
    NP = want < 1 ? 1 : want;
  }
  const int Cp = p.C / NP;
  const int ks_env = e->cfg.split_ks;
  const int KS = NP > 1 && ks_env == 2 ? 2 : 1;
  struct Part { Params p; hipStream_t st; hipGraph_t graph; hipGraphExec_t gexec; int* done_dev; int* h_poll; const double* xq; double* xg; };
  Part part[4];
  for (int k = 0; k < NP; ++k) {
    Part& q = part[k];
    q.graph = nullptr; q.gexec = nullptr;
    q.st = st;
    if (k > 0) {
      if (!e->sp_side[k - 1]) {
        HIPCHK(e, hipStreamCreateWithFlags(&e->sp_side[k - 1], hipStreamNonBlocking));
        HIPCHK(e, hipEventCreateWithFlags(&e->sp_join[k - 1], hipEventDisableTiming));
        HIPALLOC(e, e->sp_done_side[k - 1], 64);
        HIPCHK(e, hipHostMalloc((void**)&e->h_poll_side[k - 1], 2 * sizeof(int), hipHostMallocDefault));
      }
      q.st = e->sp_side[k - 1];
    }
    q.done_dev = k == 0 ? e->sp_done : e->sp_done_side[k - 1];
    q.h_poll = nullptr;
    q.p = p;
    const size_t c0 = (size_t)k * Cp;
    q.p.C = Cp;
    q.p.chain0 = p.chain0 + (uint32_t)c0;
    q.p.q0 = p.q0 + c0 * p.D;
    q.p.samples = p.samples + c0 * (size_t)(p.numIter + 1) * p.dg;
    q.p.diag = p.diag + c0 * (size_t)p.numIter * NDIAG;
    q.p.work = p.work + (size_t)k * kernel_work_doubles(e->kc, Cp, p.M);  // (the workspace is linear in the number of chains)
    q.xq = e->xq + c0 * e->DP; q.xg = e->xg + c0 * e->DP;
    q.p.xq = const_cast<double*>(q.xq); q.p.xg = q.xg; q.p.xlp = nullptr; q.p.sp_done = q.done_dev; q.p.xzero = KS > 1;
  }
  const int Cpad_p = NP == 1 ? e->Cpad : Cp;
  HIPCHK(e, hipMemsetAsync(e->sp_done, 0, 64, st));
  if (KS > 1) HIPCHK(e, hipMemsetAsync(e->xg, 0, (size_t)e->Cpad * e->DP * sizeof(double), st));
  if (NP > 1) {  // the side streams start behind everything queued so far
    if (!e->sp_fork1) HIPCHK(e, hipEventCreateWithFlags(&e->sp_fork1, hipEventDisableTiming));
    HIPCHK(e, hipEventRecord(e->sp_fork1, st));
    for (int k = 1; k < NP; ++k) {
      HIPCHK(e, hipStreamWaitEvent(part[k].st, e->sp_fork1, 0));
      HIPCHK(e, hipMemsetAsync(part[k].done_dev, 0, 64, part[k].st));
    }
  }
  hipError_t r = hipSuccess;
  for (int k = 0; k < NP && r == hipSuccess; ++k) r = launch_split_begin(e->kc, adapt, warm, part[k].p, part[k].st);
  if (r != hipSuccess) { e->err = std::string("kernel launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  e->sp_ticks = 0;
#if defined(WN_STAMPS) || defined(WN_SPAN)
  static unsigned long long* sbuf = nullptr;
  if (!sbuf) hipMalloc((void**)&sbuf, (size_t)p.C * 16 * sizeof(unsigned long long));
  hipMemset(sbuf, 0, (size_t)p.C * 16 * sizeof(unsigned long long));
  p.stamps = sbuf; part[0].p.stamps = sbuf;
#endif
  // every chain needs at most numIter * (2^M - 1) macro steps of at most 3 * 2^(maxC+2) gradient
  // evaluations (forward search, R2P re-integration, backward search; x3 Yoshida, x2 FlowD)
  const double cap = 64.0 + 1.05 * (double)p.numIter * (double)(1u << p.M) *
                                (adapt ? 9.0 * (double)(1u << (p.maxC + 2)) : 1.0);
  // The finished-chain counter is polled one batch BEHIND the launches: while the host waits for the counter of batch
  // k-1 (a copy into pinned memory + an event, nothing that blocks the stream), batch k is already queued, so the GPU
  // never waits for the host (a synchronising copy after every batch left it idle 19 % of the time).  The batch queued
  // when the last chain finishes is a few hundred early exits.
  if (!e->h_poll) {
    HIPCHK(e, hipHostMalloc((void**)&e->h_poll, 2 * sizeof(int), hipHostMallocDefault));
    for (hipEvent_t& ev : e->ev_poll) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  // One batch = 64 ticks = 128 launches.  By default the batch is captured ONCE per run (and per part) into a hipGraph and replayed:
  // the GPU then paces the ticks itself, whatever the host thread is doing (round 2's driver-run number for this path, 1.29e7 steps/s
  // against 1.9e7 in the builder's runs, was eager launches falling behind on a busy host).  WN_ROUTE_SPLIT_EAGER: eager launches.
  const bool eager = (e->cfg.route & WN_ROUTE_SPLIT_EAGER) != 0;
  auto grad = [&](const Part& q) { return launch_dense_grad(e->apad, q.xq, q.xg, e->DP, p.D, Cpad_p, q.done_dev, q.p.C, q.st, NP > 1 ? (KS > 1 ? 42 : 4) : 0); };
  auto enqueue_batch = [&](const Part& q) -> hipError_t {
    hipError_t rr = hipSuccess;
    for (int k = 0; k < 64 && rr == hipSuccess; ++k) {
      rr = grad(q);
      if (rr == hipSuccess) rr = launch_split_tick(e->kc, adapt, warm, q.p, q.st);
    }
    return rr;
  };
  // Each part is driven by a host thread of its own (part 0 by the caller's): one thread queues about 80 000 kernels per second,
  // graph replay or not, which is what ONE part consumes -- with a single thread the parts took turns batch by batch (kernel trace).
  const int device = e->cfg.device;
  auto drive = [&, device](int k, std::string* err) -> int {
    Part& q = part[k];
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice"; return WN_E_HIP; }
    struct Guard { Part& q; ~Guard() { if (q.gexec) hipGraphExecDestroy(q.gexec); if (q.graph) hipGraphDestroy(q.graph); q.gexec = nullptr; q.graph = nullptr; } } guard{q};
    if (!eager) {
      hipError_t rg = hipStreamBeginCapture(q.st, hipStreamCaptureModeThreadLocal);
      if (rg == hipSuccess) {
        const hipError_t rl = enqueue_batch(q);
        rg = hipStreamEndCapture(q.st, &q.graph);
        if (rg == hipSuccess && rl != hipSuccess) rg = rl;
      }
      if (rg == hipSuccess) rg = hipGraphInstantiate(&q.gexec, q.graph, nullptr, nullptr, 0);
#ifdef WN_DIAG
      if (getenv("WN_SPLIT_DEBUG")) fprintf(stderr, "[wn split] part %d of %d: hipGraph of 64 ticks: %s\n", k, NP, rg == hipSuccess ? "captured" : hipGetErrorString(rg));
#endif
      if (rg != hipSuccess) {  // (fall back to eager launches)
        (void)hipGetLastError();
        if (q.gexec) { hipGraphExecDestroy(q.gexec); q.gexec = nullptr; }
        if (q.graph) { hipGraphDestroy(q.graph); q.graph = nullptr; }
      }
    }
    int* const hp = k == 0 ? e->h_poll : e->h_poll_side[k - 1];
    double ticks = 0.0;
    for (int batch = 0;; ++batch) {
      if (ticks > cap) { *err = "split run exceeded its tick bound (internal error)"; return WN_E_HIP; }
      hipError_t rr = q.gexec ? hipGraphLaunch(q.gexec, q.st) : enqueue_batch(q);
      if (rr == hipSuccess) { hipLaunchKernelGGL(wn_poll_word_kernel, dim3(1), dim3(1), 0, q.st, q.done_dev, hp + (batch & 1)); rr = hipGetLastError(); }
      if (rr == hipSuccess) rr = hipEventRecord(e->ev_poll[2 * k + (batch & 1)], q.st);
      if (rr != hipSuccess) { *err = std::string("kernel launch: ") + hipGetErrorString(rr); return WN_E_HIP; }
      ticks += 64;
      if (k == 0) e->sp_ticks += 64;
      if (batch >= 1) {
        rr = hipEventSynchronize(e->ev_poll[2 * k + ((batch - 1) & 1)]);
        if (rr != hipSuccess) { *err = std::string("hipEventSynchronize: ") + hipGetErrorString(rr); return WN_E_HIP; }
        if (hp[(batch - 1) & 1] >= q.p.C) break;
      }
    }
    return WN_OK;
  };
  {
    std::string errs[4];
    int rcs[4] = {0, 0, 0, 0};
    std::vector<std::thread> th;
    for (int k = 1; k < NP; ++k) th.emplace_back([&, k]() { rcs[k] = drive(k, &errs[k]); });
    rcs[0] = drive(0, &errs[0]);
    for (std::thread& t : th) t.join();
    for (int k = 0; k < NP; ++k) if (rcs[k]) { e->err = errs[k]; hipDeviceSynchronize(); return rcs[k]; }
  }
  for (int k = 1; k < NP; ++k) {  // results are complete when wn_run returns, and the caller's stream is behind every part
    HIPCHK(e, hipEventRecord(e->sp_join[k - 1], part[k].st));
    HIPCHK(e, hipStreamWaitEvent(st, e->sp_join[k - 1], 0));
  }
  HIPCHK(e, hipStreamSynchronize(st));
#if defined(WN_STAMPS) || defined(WN_SPAN)
  {
    std::vector<unsigned long long> h((size_t)p.C * 16);
    hipMemcpy(h.data(), sbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double tot[16] = {0};
#ifdef WN_STAMPS
    for (int c = 0; c < p.C; ++c) for (int k = 0; k < 16; ++k) tot[k] += (double)h[(size_t)c * 16 + k];
    const char* names[16] = {"unpark", "grad_in+kick", "reduce", "C", "D_deep", "E_rest", "ticks", "A", "E_to_exp", "E_pick+stores",
                             "E_doubling_end", "B_pre+park", "fast_dots", "", "", ""};
    fprintf(stderr, "[WN_STAMPS split] 100 MHz ticks per chain-tick:");
    for (int k = 0; k < 13; ++k) if (k != 6) fprintf(stderr, " %s=%.1f", names[k], tot[k] / tot[6]);
    fprintf(stderr, " ticks/chain=%.0f\n", tot[6] / p.C);
#endif
    unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0; double dur = 0;
    for (int c = 0; c < p.C; ++c) {
      const unsigned long long a = h[(size_t)c * 16 + 13], b = h[(size_t)c * 16 + 14];
      if (!a) continue;
      if (a < s0) s0 = a; if (a > s1) s1 = a; if (b < e0) e0 = b; if (b > e1) e1 = b; dur += (double)(b - a);
    }
    {
      std::vector<long long> st, en;
      for (int c = 0; c < p.C; ++c) if (h[(size_t)c * 16 + 13]) { st.push_back((long long)(h[(size_t)c * 16 + 13] - s0)); en.push_back((long long)(h[(size_t)c * 16 + 14] - s0)); }
      std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
      fprintf(stderr, "[WN_SPAN] start percentiles (10 ns):");
      for (int q = 0; q <= 10; ++q) fprintf(stderr, " %lld", st[(st.size() - 1) * q / 10]);
      fprintf(stderr, "\n[WN_SPAN] end percentiles (10 ns):");
      for (int q = 0; q <= 10; ++q) fprintf(stderr, " %lld", en[(en.size() - 1) * q / 10]);
      fprintf(stderr, "\n[WN_SPAN] XCD-0 blocks (b%%8==0) start-end: ");
      for (int c = 0; c < p.C; c += 8) fprintf(stderr, " b%d:%lld-%lld", c, (long long)(h[(size_t)c * 16 + 13] - s0), (long long)(h[(size_t)c * 16 + 14] - s0));
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[WN_STAMPS split] launch 100, wall clock in 10 ns units: first start..last start %llu, first end..last end %llu, "
            "first start..last end %llu, mean wave duration %.0f\n", s1 - s0, e1 - e0, e1 - s0, dur / p.C);
  }
#endif
  return WN_OK;
}

static int launch(wn_engine* e, const Params& p, hipStream_t st, bool timed) {
  const int adapt = e->cfg.integrator == WN_I_FIXED ? 0 : e->cfg.integrator == WN_I_ADAPT_FLOWD ? 2 : 1;
  if (timed) HIPCHK(e, hipEventRecord(e->ev0, st));
  const bool warm = p.warmupIter > 0;
  if (e->kc.layout == 3) {
    if (e->kc.p1 != EXT_HALF_QG) { e->err = "external target: drive the run with wn_split_begin / wn_split_tick"; return WN_E_UNSUPPORTED; }
    int rc = run_split_dense(e, p, st, adapt, warm);
    if (rc) return rc;
    if (timed) { HIPCHK(e, hipEventRecord(e->ev1, st)); e->timed = true; }
    return WN_OK;
  }
  e->sp_ticks = 0;
  Params pc = p;
  if (e->kc.G > 1) {
    pc.cl_g = e->kc.G; pc.cl_red = e->cl_red; pc.cl_cnt = e->cl_cnt; pc.cl_abort = (int*)(e->cl_cnt + e->cfg.n_chains);
    pc.cl_debug_drop = e->debug_cluster_drop;  // test hook: wn_debug_cluster_drop (tests/test_gpu_sweep.py)
    HIPCHK(e, hipMemsetAsync(e->cl_cnt, 0, ((size_t)e->cfg.n_chains + 16) * sizeof(unsigned), st));
  }
  // chain queue of the persistent launches (wn_k_orbit.hip, staged targets): kabort[8], zeroed in stream order in front of every launch
  pc.dynq = e->kabort + 8;
  HIPCHK(e, hipMemsetAsync(e->kabort + 8, 0, sizeof(int), st));
  const Params& p_ = pc;
  const bool coop_legacy = (e->cfg.route & WN_ROUTE_COOP_LEGACY) != 0;  // A-B tests: one-stream kernel
  const bool fast_legacy = (e->cfg.route & WN_ROUTE_FAST_LEGACY) != 0;  // A-B tests: the seven-T-wave kernel for dense targets too
  // the tick-bound abort flag of the fast kernels belongs to THIS run: cleared in stream order before the launch; wn_last_run_ms,
  // the next wn_run and wn_destroy report a run that gave up
  if (e->kabort_pending) { int rc_ = check_abort(e, false); if (rc_) return rc_; }
  hipError_t r = (!coop_legacy && !fast_legacy && fast4_eligible(e->kc, adapt, warm, p_)) ? launch_fast4(e->kc, p_, e->kabort, st)
               : (!coop_legacy && fast_eligible(e->kc, adapt, warm, p_)) ? launch_fast(e->kc, p_, e->kabort, st)
               : orbit2_eligible(e->kc, adapt, warm, p_) ? launch_orbit2(e->kc, p_, e->kabort, st)
               : orbit_eligible(e->kc, adapt, warm, p_) ? launch_orbit(e->kc, p_, st)
               : e->kc.layout == 0 ? launch_lane(e->kc, adapt, warm, p_, st)
               : e->kc.layout == 1 ? launch_coop(e->kc, adapt, warm, p_, st) : launch_wave(e->kc, adapt, warm, p_, st);
  if (r != hipSuccess) { e->err = std::string("kernel launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  hipLaunchKernelGGL(wn_abort_latch_kernel, dim3(1), dim3(1), 0, st, e->kabort);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(e->h_abort, e->kabort + 1, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(e, hipEventRecord(e->ev_abort, st));
  e->kabort_pending = true;
  if (e->cfg.generated == WN_G_SW_CONSTRAIN && p.mode == MODE_SAMPLE) {
    const size_t nrows = (size_t)p.C * (p.numIter + 1);
    hipLaunchKernelGGL(wn_sw_constrain_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, st, p.samples, nrows, p.D / 3);
    HIPCHK(e, hipGetLastError());
  }
  if (timed) { HIPCHK(e, hipEventRecord(e->ev1, st)); e->timed = true; }
  if (e->kc.G > 1) {  // a cluster run reports a spin that gave up instead of returning unreduced sums
    int ab = 0;
    HIPCHK(e, hipMemcpyAsync(&ab, pc.cl_abort, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (ab) { e->err = "cluster layout: a cross-workgroup reduction timed out (workgroups not co-resident?)"; return WN_E_HIP; }
  }
  return WN_OK;
}

int wn_run(wn_engine* e, const double* q0_dev, uint64_t seed, int32_t n_iter, double* samples_dev,
           double* diag_dev, double* trace_dev, int32_t trace_cap, int32_t* trace_n_dev, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->have_kc) { e->err = "wn_set_target has not been called"; return WN_E_NOTARGET; }
  if (!q0_dev || !samples_dev || !diag_dev || n_iter < 0) { e->err = "null buffer or negative n_iter"; return WN_E_ARG; }
  if (trace_dev && (!trace_n_dev || trace_cap < 1)) { e->err = "trace needs trace_n_dev and trace_cap >= 1"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  Params p;
  fill_params(e, &p);
  p.mode = MODE_SAMPLE;
  p.numIter = n_iter; p.seed = seed;
  p.warmupIter = e->cfg.warmup_iter;
  if (p.warmupIter > 0 && (e->cfg.adapt_h || e->cfg.adapt_delta)) {
    p.facs = e->facs;  // allocated in wn_set_target (no allocation on the launch path)
  } else {
    p.warmupIter = 0;  // nothing adapts: the plain kernels are exact
  }
  p.q0 = q0_dev; p.samples = samples_dev; p.diag = diag_dev;
  p.trace = trace_dev; p.trace_cap = trace_cap; p.trace_n = trace_n_dev;
#ifdef WN_STAMPS
  {  // diagnostic build: per-phase cycle stamps, dumped to $WN_STAMPS_OUT after the run
    static unsigned long long* sbuf = nullptr;
    const size_t nw = (size_t)(p.C + 15) * 16 * 4;  // one 16-entry row per wave in every layout (the pair kernel: eight rows per workgroup of >= 4 chains)
    static size_t scap = 0;
    if (nw > scap) { if (sbuf) hipFree(sbuf); hipMalloc((void**)&sbuf, nw * sizeof(unsigned long long)); scap = nw; }
    hipMemset(sbuf, 0, nw * sizeof(unsigned long long));
    p.stamps = sbuf;
    int rc = launch(e, p, (hipStream_t)stream, true);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), sbuf, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double tot[16] = {0};
    for (size_t w = 0; w < nw / 16; ++w) for (int k = 0; k < 16; ++k) tot[k] += (double)h[w * 16 + k];
    const char* names[16] = {"A_setup", "B_leapfrog+grad", "B_ke+reduce", "C_after_micro", "D_uturn", "E_rest", "ticks", "loop_head",
                             "E_to_exp", "E_pick+stores", "E_doubling_end", "", "", "", "", ""};
    if (strstr(e->kc.name, "wn_fast4_kernel")) {  // rows: up to 8 waves per workgroup (T waves 0..pairs-1, then the B waves)
      const int pairs = fast4_pairs(p.C);
      const size_t groups = ((size_t)(p.C + 3) / 4 + pairs - 1) / pairs;
      for (int w = 0; w < pairs + 2 * ((pairs + 1) / 2) + (pairs == 1 ? 1 : 0); ++w) {  // (+1: the second T wave of the halves mode)
        double a[8] = {0};
        for (size_t gI = 0; gI < groups; ++gI) for (int k = 0; k < 8; ++k) a[k] += (double)h[(gI * 8 + w) * 16 + k];
        if (a[6] <= 0) continue;
        // waves 0 .. 2 quads - 1: the bookkeeper (R, I alternating), then the T waves
        const int quads = (pairs + 1) / 2;
        if (w >= 2 * quads) fprintf(stderr, "[WN_STAMPS fast4] T wave %d cycles/tick: fragment requests=%.0f wait for the command=%.0f commands+halfstep=%.0f mfma=%.0f halfstep+checks+sums=%.0f stores=%.0f (ticks %.0f)\n",
                               w, a[7] / a[6], a[5] / a[6], a[0] / a[6], a[1] / a[6], a[2] / a[6], (a[3] + a[4]) / a[6], a[6] / groups);
        else if (w & 1) fprintf(stderr, "[WN_STAMPS fast4] I wave %d cycles/tick: normals+window=%.0f wait for R=%.0f message=%.0f issue=%.0f command=%.0f\n",
                     w, a[0] / a[6], a[5] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6]);
        else fprintf(stderr, "[WN_STAMPS fast4] R wave %d cycles/tick: wait for T=%.0f sums+verdicts=%.0f retire=%.0f message=%.0f\n",
                     w, a[5] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6]);
      }
      return rc;
    }
    if (strstr(e->kc.name, "wn_orbit2_kernel")) {  // two rows of 8 per chain: T wave, R wave
      double t[8] = {0}, r[8] = {0};
      for (int c = 0; c < p.C; ++c) for (int k = 0; k < 8; ++k) { t[k] += (double)h[((size_t)c * 2) * 8 + k]; r[k] += (double)h[((size_t)c * 2 + 1) * 8 + k]; }
      fprintf(stderr, "[WN_STAMPS orbit2] T wave cycles/leaf: step+gradient+sums=%.0f publish+stores+abort poll=%.0f draws=%.0f wait for a command=%.0f wait for ring space=%.0f "
              "command work + rest=%.0f (leaves %.0f)\n", t[0] / t[6], t[1] / t[6], t[5] / t[6], t[2] / t[6], t[3] / t[6], t[4] / t[6], t[6]);
      fprintf(stderr, "[WN_STAMPS orbit2] R wave cycles/retired leaf: wait for leaves=%.0f draws+scans+ballots=%.0f checks=%.0f events+accumulators=%.0f rest=%.0f "
              "(leaves %.0f, %.2f per sub-batch)\n", r[2] / r[6], r[0] / r[6], r[1] / r[6], r[3] / r[6], r[4] / r[6], r[6], r[6] / r[7]);
      return rc;
    }
    if (strstr(e->kc.name, "wn_orbit_kernel")) {  // one row per chain
      const char* on[16] = {"T_draws", "T_forward", "T_r2p+backward", "T_finish+ring", "R_scans+exp", "R_checks", "macro_steps", "R_events+accumulators",
                            "iteration_start", "doubling/iteration_end", "blocks", "evals_run", "redos", "", "", ""};
      auto row = [&](const char* what, const double* a) {
        fprintf(stderr, "[WN_STAMPS orbit] %s: cycles per macro step:", what);
        for (int k = 0; k < 10; ++k) if (k != 6) fprintf(stderr, " %s=%.0f", on[k], a[k] / a[6]);
        double c = 0; for (int k = 0; k < 10; ++k) if (k != 6) c += a[k];
        fprintf(stderr, " | total=%.0f macro steps=%.0f leaves/block=%.2f evals run/macro step=%.2f redos/block=%.4f cycles/eval run=%.0f\n", c / a[6], a[6], a[6] / a[10], a[11] / a[6],
                a[12] / a[10], c / a[11]);
      };
      double tot2[16] = {0};
      size_t wmax = 0; double cmax = -1;
      for (int c = 0; c < p.C; ++c) {
        double cc = 0;
        for (int k = 0; k < 16; ++k) { tot2[k] += (double)h[(size_t)c * 16 + k]; if (k < 10 && k != 6) cc += (double)h[(size_t)c * 16 + k]; }
        if (cc > cmax) { cmax = cc; wmax = c; }
      }
      row("all chains", tot2);
      double am[16]; for (int k = 0; k < 16; ++k) am[k] = (double)h[wmax * 16 + k];
      char nm[64]; snprintf(nm, sizeof nm, "longest chain %zu", wmax);
      row(nm, am);
      return rc;
    }
    if (strstr(e->kc.name, "wn_fast_kernel")) {  // rows: 8 waves per 16-chain group (T waves 0..6, B wave 7)
      const size_t groups = (size_t)(p.C + 15) / 16;
      for (int w = 0; w < 8; ++w) {
        double a[8] = {0};
        for (size_t gI = 0; gI < groups; ++gI) for (int k = 0; k < 8; ++k) a[k] += (double)h[(gI * 8 + w) * 16 + k];
        if (a[6] <= 0) continue;
        if (w < 7) fprintf(stderr, "[WN_STAMPS fast] T wave %d cycles/tick: cmd+halfstep=%.0f waitA=%.0f mfma+halfstep=%.0f lp/ke partials=%.0f checks(on-chip)+stores=%.0f "
                                    "deeper checks=%.0f waitB=%.0f\n", w, a[0] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6], a[4] / a[6], a[7] / a[6], a[5] / a[6]);
        else fprintf(stderr, "[WN_STAMPS fast] B wave cycles/tick: read partials=%.0f waitA=%.0f bookkeeper=%.0f cmd+normals=%.0f waitB=%.0f (ticks %.0f)\n",
                     a[0] / a[6], a[1] / a[6], a[2] / a[6], a[4] / a[6], a[5] / a[6], a[6] / groups);
      }
      return rc;
    }
    fprintf(stderr, "[WN_STAMPS] per wave-tick cycles:");
    for (int k = 0; k < 11; ++k) if (k != 6) fprintf(stderr, " %s=%.0f", names[k], tot[k] / tot[6]);
    fprintf(stderr, " ticks/wave=%.0f\n", tot[6] / (nw / 16));
    {  // the wave that ran longest
      size_t wmax = 0; double cmax = -1;
      for (size_t w = 0; w < nw / 16; ++w) { double c = 0; for (int k = 0; k < 16; ++k) if (k != 6) c += (double)h[w * 16 + k]; if (c > cmax) { cmax = c; wmax = w; } }
      fprintf(stderr, "[WN_STAMPS] longest wave %zu (cycles %.3g):", wmax, cmax);
      for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.0f", k == 6 ? "ticks" : names[k], k == 6 ? (double)h[wmax * 16 + 6] : (double)h[wmax * 16 + k] / (double)h[wmax * 16 + 6]);
      fprintf(stderr, "\n");
    }
    return rc;
  }
#endif
  return launch(e, p, (hipStream_t)stream, true);
}

int wn_run_orbit_stats(wn_engine* e, const double* q0_dev, uint64_t seed, int32_t n_iter, double* samples_dev,
                       double* diag_dev, double* orbit_min_dev, double* orbit_max_dev, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->have_kc) { e->err = "wn_set_target has not been called"; return WN_E_NOTARGET; }
  if (!q0_dev || !samples_dev || !diag_dev || !orbit_min_dev || !orbit_max_dev || n_iter < 0) { e->err = "null buffer or negative n_iter"; return WN_E_ARG; }
  // (generated = WN_G_SW_CONSTRAIN, round 5: the wave kernel applies param_constrain to every orbit state -- Engine::orbit_stats ->
  //  WTgtStockWatson::constrain --, as the reference's recordOrbitStats does with any generated(), WALNUTS.py:182-184,331-333)
  HIPCHK(e, hipSetDevice(e->cfg.device));
  Params p;
  fill_params(e, &p);
  p.mode = MODE_SAMPLE;
  p.numIter = n_iter; p.seed = seed;
  p.warmupIter = (e->cfg.warmup_iter > 0 && (e->cfg.adapt_h || e->cfg.adapt_delta)) ? e->cfg.warmup_iter : 0;
  p.q0 = q0_dev; p.samples = samples_dev; p.diag = diag_dev;
  p.omin = orbit_min_dev; p.omax = orbit_max_dev;
  if (e->cfg.generated == WN_G_SW_CONSTRAIN) p.generated = G_SW_CONSTRAIN;  // (identity draws as in wn_run; the orbit statistics see the transform)
  return launch(e, p, (hipStream_t)stream, true);
}

// ---- caller-evaluated target (WN_T_EXTERNAL): the reference's lpFun plugin as a batched callable ----
int wn_split_begin(wn_engine* e, const double* q0_dev, uint64_t seed, int32_t n_iter, double* samples_dev,
                   double* diag_dev, double* q_eval_dev, const double* grad_dev, const double* lp_dev, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->have_kc) { e->err = "wn_set_target has not been called"; return WN_E_NOTARGET; }
  if (e->kc.layout != 3 || e->kc.p1 != EXT_LP) { e->err = "wn_split_begin needs the WN_T_EXTERNAL target"; return WN_E_ARG; }
  if (!q0_dev || !samples_dev || !diag_dev || !q_eval_dev || !grad_dev || !lp_dev || n_iter < 0) { e->err = "null buffer or negative n_iter"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  Params p;
  fill_params(e, &p);
  p.mode = MODE_SAMPLE;
  p.numIter = n_iter; p.seed = seed;
  p.warmupIter = (e->cfg.warmup_iter > 0 && (e->cfg.adapt_h || e->cfg.adapt_delta)) ? e->cfg.warmup_iter : 0;
  p.q0 = q0_dev; p.samples = samples_dev; p.diag = diag_dev;
  p.xq = q_eval_dev; p.xg = const_cast<double*>(grad_dev); p.xlp = lp_dev; p.sp_done = e->sp_done;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(e, hipMemsetAsync(e->sp_done, 0, 64, st));
  hipError_t r = launch_split_begin(e->kc, e->cfg.integrator == WN_I_FIXED ? 0 : e->cfg.integrator == WN_I_ADAPT_FLOWD ? 2 : 1, p.warmupIter > 0, p, st);
  if (r != hipSuccess) { e->err = std::string("kernel launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  e->sp_prm = p;
  e->sp_active = true;
  e->sp_ticks = 0;
  return WN_OK;
}

int wn_split_ld(wn_engine* e, int32_t* ld) {
  if (!e || !ld) return WN_E_ARG;
  if (!e->have_kc || e->kc.layout != 3) { e->err = "wn_split_ld needs a split-mode target"; return WN_E_ARG; }
  *ld = e->DP;
  return WN_OK;
}

int wn_split_tick(wn_engine* e, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->sp_active) { e->err = "wn_split_begin has not been called"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipError_t r = launch_split_tick(e->kc, e->cfg.integrator == WN_I_FIXED ? 0 : e->cfg.integrator == WN_I_ADAPT_FLOWD ? 2 : 1, e->sp_prm.warmupIter > 0, e->sp_prm, (hipStream_t)stream);
  if (r != hipSuccess) { e->err = std::string("kernel launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  e->sp_ticks++;
  return WN_OK;
}

int wn_split_done(wn_engine* e, int32_t* n_done, void* stream) {
  if (!e || !n_done) return WN_E_ARG;
  if (!e->sp_active) { e->err = "wn_split_begin has not been called"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  int done = 0;
  HIPCHK(e, hipMemcpyAsync(&done, e->sp_done, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
  *n_done = done;
  return WN_OK;
}

int wn_integrator_step(wn_engine* e, const double* q_dev, const double* v_dev, const double* g_dev,
                       const double* ham0_dev, const double* h_dev, const int32_t* xi_dev,
                       const double* u_r2p_dev, double* q1_dev, double* v1_dev, double* g1_dev,
                       int32_t* iout_dev, double* dout_dev, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->have_kc) { e->err = "wn_set_target has not been called"; return WN_E_NOTARGET; }
  if (!q_dev || !v_dev || !g_dev || !ham0_dev || !h_dev || !xi_dev || !q1_dev || !v1_dev || !g1_dev || !iout_dev || !dout_dev ||
      (e->cfg.integrator == WN_I_ADAPT_R2P && !u_r2p_dev)) { e->err = "null buffer"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  Params p;
  fill_params(e, &p);
  p.mode = MODE_KAT;
  p.numIter = 0;
  p.warmupIter = 1;  // the instantiation that tracks igrConst; no adaptation runs in this mode
  if (!p.facs) p.adaptDelta = 0;
  p.kq = q_dev; p.kv = v_dev; p.kg = g_dev; p.kham0 = ham0_dev; p.kh = h_dev; p.kxi = xi_dev;
  p.ku_r2p = u_r2p_dev; p.kq1 = q1_dev; p.kv1 = v1_dev; p.kg1 = g1_dev; p.kiout = iout_dev; p.kdout = dout_dev;
  return launch(e, p, (hipStream_t)stream, false);
}

int wn_lpfun(wn_engine* e, const double* q_dev, double* lp_dev, double* grad_dev, void* stream) {
  if (!e) return WN_E_ARG;
  if (!e->have_kc) { e->err = "wn_set_target has not been called"; return WN_E_NOTARGET; }
  if (!q_dev || !lp_dev || !grad_dev) { e->err = "null buffer"; return WN_E_ARG; }
  if (e->target == WN_T_EXTERNAL) { e->err = "the external target is evaluated by the caller"; return WN_E_UNSUPPORTED; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  const int B = e->cfg.n_chains;
  hipLaunchKernelGGL(wn_lpfun_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->target,
                     e->cfg.dim, B, e->tpar, q_dev, lp_dev, grad_dev);
  HIPCHK(e, hipGetLastError());
  return WN_OK;
}

static int check_target_dim(int32_t target_kind, const double* tpar_dev, int32_t dim) {
  if (dim < 1) { g_create_err = "bad dimension"; return WN_E_ARG; }
  if (target_kind < 0 || target_kind > T_TT_FUNNEL) { g_create_err = "unknown target"; return WN_E_ARG; }
  if ((target_kind == T_DIAGGAUSS || target_kind == T_DENSEGAUSS) && !tpar_dev) { g_create_err = "target needs parameters"; return WN_E_ARG; }
  if ((target_kind == T_FUNNEL10 && dim != 11) ||
      ((target_kind == T_CORRGAUSS || target_kind == T_SMILE || target_kind == T_FUNNEL1 || target_kind == T_MODFUNNEL ||
        target_kind == T_TT_CORRNORMAL) && dim != 2) || (target_kind == T_TT_FUNNEL && dim < 2)) {
    g_create_err = "dimension does not fit the target"; return WN_E_ARG;
  }
  return WN_OK;
}

int wn_lpfun_kind(int32_t device, int32_t target_kind, const double* tpar_dev, int32_t dim, int32_t B,
                  const double* q_dev, double* lp_dev, double* grad_dev, void* stream) {
  if (B < 1 || !q_dev || !lp_dev || !grad_dev) { g_create_err = "wn_lpfun_kind: bad argument"; return WN_E_ARG; }
  int rc = check_target_dim(target_kind, tpar_dev, dim);
  if (rc) return rc;
  hipError_t r = hipSetDevice(device);
  if (r != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(r); return WN_E_HIP; }
  hipLaunchKernelGGL(wn_lpfun_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, target_kind, dim, B,
                     tpar_dev, q_dev, lp_dev, grad_dev);
  r = hipGetLastError();
  if (r != hipSuccess) { g_create_err = std::string("wn_lpfun_kind launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  return WN_OK;
}

int wn_rng_tape(wn_engine* e, uint64_t seed, int32_t iter, int32_t n_u, double* u_dev, double* z_dev,
                void* stream) {
  if (!e) return WN_E_ARG;
  if (iter < 0 || n_u < 0) { e->err = "negative iter or n_u"; return WN_E_ARG; }
  HIPCHK(e, hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(wn_tape_kernel, dim3(e->cfg.n_chains), dim3(64), 0, (hipStream_t)stream, seed,
                     (uint32_t)e->cfg.chain_offset, e->cfg.n_chains, iter, n_u, e->cfg.dim, u_dev, z_dev);
  HIPCHK(e, hipGetLastError());
  return WN_OK;
}

int wn_last_run_ms(wn_engine* e, float* ms) {
  if (!e || !ms) return WN_E_ARG;
  if (!e->timed) { e->err = "no timed run yet"; return WN_E_ARG; }
  HIPCHK(e, hipEventSynchronize(e->ev1));
  HIPCHK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
  if (e->kabort_pending) return check_abort(e, true);
  return WN_OK;
}

int wn_check_run(wn_engine* e) {
  if (!e) return WN_E_ARG;
  HIPCHK(e, hipSetDevice(e->cfg.device));
  return e->kabort_pending ? check_abort(e, true) : WN_OK;
}

int wn_debug_poison_lds(int32_t device, uint64_t pattern, void* stream) {
  hipError_t r = hipSetDevice(device);
  if (r != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(r); return WN_E_HIP; }
  unsigned long long* sink = nullptr;  // (allocated on THIS device, for this call)
  if (hipMalloc((void**)&sink, 64) != hipSuccess) { g_create_err = "hipMalloc failed"; return WN_E_ALLOC; }
  hipDeviceProp_t prop;
  r = hipGetDeviceProperties(&prop, device);
  if (r == hipSuccess) r = hipFuncSetAttribute((const void*)wn_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (r != hipSuccess) { g_create_err = std::string("wn_debug_poison_lds: ") + hipGetErrorString(r); return WN_E_HIP; }
  hipLaunchKernelGGL(wn_poison_lds_kernel, dim3(4 * prop.multiProcessorCount), dim3(256), 160 * 1024, (hipStream_t)stream, (unsigned long long)pattern, sink);
  r = hipGetLastError();
  if (r == hipSuccess) r = hipStreamSynchronize((hipStream_t)stream);
  hipFree(sink);
  if (r != hipSuccess) { g_create_err = std::string("wn_debug_poison_lds launch: ") + hipGetErrorString(r); return WN_E_HIP; }
  return WN_OK;
}

int wn_last_split_ticks(wn_engine* e, int64_t* ticks) {
  if (!e || !ticks) return WN_E_ARG;
  *ticks = (int64_t)e->sp_ticks;
  return WN_OK;
}

const char* wn_kernel_name(wn_engine* e) { return (e && e->have_kc) ? e->kc.name : ""; }  // engine-owned storage

}  // extern "C"
