/*
 * walnuts_amd.h -- C ABI of the MI355X-native batched WALNUTS/NUTS engine.
 *
 * The reference (bob-carpenter/walnuts) has no FFI: its boundary is Python call signatures
 * (SURVEY.md 8b).  This ABI is what sits *behind* Python functions of those shapes
 * (walnuts_amd/WALNUTS.py mirrors WALNUTSpy/WALNUTS.py:111-129, walnuts_amd/walnuts.py mirrors
 * walnuts/walnuts.py:362-373); each entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions: plain C types only; every pointer named *_dev is device memory on the
 * engine's GPU, owned by the caller; all functions return 0 on success or a negative
 * WN_E_* code, never throw and never exit(); one engine per GPU, not thread-safe per engine,
 * safe across engines.  All arithmetic is IEEE binary64 ("f64").
 */
#ifndef WALNUTS_AMD_H
#define WALNUTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_ABI_VERSION 5
#define WN_NDIAG 24      /* diagnostics columns, WALNUTSpy/WALNUTS.py:670-693 */
#define WN_TRACE_NSCAL 12 /* trace row = 12 scalars + q[D] + v[D] */

/* error codes */
enum {
  WN_OK = 0,
  WN_E_ARG = -1,         /* bad argument (the reference: ValueError / sys.exit on bad targets) */
  WN_E_UNSUPPORTED = -2, /* configuration has no device kernel (yet) */
  WN_E_HIP = -3,         /* HIP runtime error, see wn_last_error */
  WN_E_NOTARGET = -4,    /* wn_set_target not called */
  WN_E_ALLOC = -5        /* device memory exhausted (hipErrorOutOfMemory); the engine keeps no target */
};

/* target plugin lpFun(q) -> [lp, grad]  (WALNUTSpy/targetDistr.py; adaptiveIntegrators.py passes
 * it as `lpFun`).  On the device the plugin is an enum + parameter buffer. */
enum {
  WN_T_STDGAUSS = 0,   /* targetDistr.py:18-21 (D <= 262144; beyond 4096: n_chains * ceil(D/4096) <= 256) */
  WN_T_FUNNEL10 = 1,   /* targetDistr.py:74-78 (D = 11) */
  WN_T_DENSEGAUSS = 2, /* lp = -1/2 q'Aq, grad = -Aq; params = A row-major D*D, D <= 1024 (SURVEY.md 8d cfg 2/4) */
  WN_T_DIAGGAUSS = 3,  /* params = diag(A), D <= 262144 as for WN_T_STDGAUSS (SURVEY.md 8d cfg 2 diagonal variant) */
  WN_T_CORRGAUSS = 4,  /* targetDistr.py:25-31 (D = 2) */
  WN_T_SMILE = 5,      /* targetDistr.py:34-38 (D = 2) */
  WN_T_FUNNEL1 = 6,    /* targetDistr.py:88-93 (D = 2) */
  WN_T_MODFUNNEL = 7,  /* targetDistr.py:41-51 (D = 2) */
  /* (lpdf, grad) pairs of test/targets.py, used with walnuts() (walnuts/walnuts.py:362);
   * wn_wstep_* / wn_lpfun only */
  WN_T_TT_CORRNORMAL = 8, /* test/targets.py:9-15 (D = 2; note its grad[0] = -q0 + rho*q1) */
  WN_T_TT_FUNNEL = 9,     /* test/targets.py:23-29 (any D >= 2; no log-scale term) */
  /* lpFun evaluated by the CALLER (targetDistr.py:18 `lpFun(q) -> [lp, grad]` as a batched callable
   * on device buffers): wn_split_begin / wn_split_tick below; D <= 1024 */
  WN_T_EXTERNAL = 10,
  /* StockWatson stochastic volatility, WALNUTSpy_examples/StockWatson/sw_innov.stan:7-51 as
   * bridgestan's log_density_gradient evaluates it (mainSW.py:22): params = [T, y[1..T]], D = 3T,
   * T <= 256 (BASELINE.json config 5: T = 252, D = 756) */
  WN_T_STOCKWATSON = 11
};

/* macro-step integrator plugin  integrator(q,v,g,Ham0,h,xi,lpFun,delta,auxPar)
 * (WALNUTSpy/adaptiveIntegrators.py) */
enum {
  WN_I_FIXED = 0,     /* fixedLeapFrog    adaptiveIntegrators.py:49-59  */
  WN_I_ADAPT_D = 1,   /* adaptLeapFrogD   adaptiveIntegrators.py:65-137 */
  WN_I_ADAPT_R2P = 2, /* adaptLeapFrogR2P adaptiveIntegrators.py:361-475 */
  WN_I_ADAPT_YOSHIDA = 3, /* adaptYoshidaD  adaptiveIntegrators.py:142-240 */
  WN_I_ADAPT_FLOWD = 4    /* adaptLeapFrogFlowD adaptiveIntegrators.py:246-356 (lane / wave / split layouts) */
};

/* generated(q): what is stored per draw (WALNUTS.py:113,164,695) */
enum {
  WN_G_IDENTITY = 0, /* lambda q: q */
  WN_G_Q0_SUMSQ = 1, /* [q[0], sum(q*q)]  WALNUTSpy_examples/gaussian/mainGaussESS.py:29-30 */
  WN_G_FIRSTK = 2,   /* q[0:k]            WALNUTSpy_examples/funnel/mainFunnel.py:17-18 */
  /* StockWatson: param_constrain(q, include_tp=True) (mainSW.py:28-29) for sw_innov.stan -- the 3T unconstrained
   * parameters followed by the transformed parameters in declaration order: sigma, gpri, z[T-1], x[T], tau[T]
   * (sw_innov.stan:17-35); dg = 6T + 1.  Needs the WN_T_STOCKWATSON target. */
  WN_G_SW_CONSTRAIN = 3
};

/* Kernel routing (wn_config.route, ABI 5).  The reference's boundary is keyword arguments (WALNUTS.py:111-129), not ambient
 * state: which of several equivalent kernels runs a configuration is part of the engine's configuration too.  0 = WN_ROUTE_AUTO:
 * the library chooses (what every caller should pass).  The bits below select the OTHER implementation of the same reference
 * function where two exist -- the test suite compares them with each other and with the oracle -- and never change what is
 * computed, only by which kernel (results agree to the bits or to rtol 1e-6, as tests/ states case by case).  Until ABI 4 these
 * were environment variables of the same names (WN_ORBIT_LEGACY=1 ...), read inside the library on every wn_set_target /
 * launch; the default build reads no environment variable any more (a -DWN_DIAG build ORs the old variables in at wn_create). */
enum {
  WN_ROUTE_AUTO = 0,
  WN_ROUTE_ORBIT_LEGACY = 1 << 0,     /* plain runs of the one-chain-per-wave layouts on the one-stream engine (wn_wave_kernel /
                                         wn_wavelds_kernel / wn_coop_kernel) instead of the trajectory-first kernels (wn_k_orbit*.hip) */
  WN_ROUTE_ORBIT_NOWIDE = 1 << 1,     /* ... only the wide mode (vectors of NE registers per lane) off */
  WN_ROUTE_ORBIT_WIDE_ADAPT = 1 << 2, /* StockWatson: adaptive integrators' plain runs on the one-wave wide mode too */
  WN_ROUTE_ORBIT_ONEWAVE = 1 << 3,    /* StockWatson: no two-wave kernel (wn_k_orbit2.hip) */
  WN_ROUTE_ORBIT_EXACT = 1 << 4,      /* trajectory-first kernels with the one-stream engine's micro-step arithmetic (two reductions
                                         per micro step) instead of the fused step: bit-identical to that engine */
  WN_ROUTE_ORBIT2_NOADAPT = 1 << 5,   /* wn_k_orbit2.hip for fixedLeapFrog only */
  WN_ROUTE_ORBIT2_ANYC = 1 << 6,      /* wn_k_orbit2.hip whatever the chain count (default: only while chains leave SIMDs idle) */
  WN_ROUTE_DIAG_COOP = 1 << 7,        /* std / diagonal Gaussians: the cooperative 16-chain layout instead of wn_orbit_kernel's wide mode */
  WN_ROUTE_DENSE_COOP = 1 << 8,       /* dense Gaussian D <= 128, adaptive integrators: the cooperative MFMA layout instead of one chain per wave */
  WN_ROUTE_FAST_LEGACY = 1 << 9,      /* fixedLeapFrog Gaussians: the seven-T-wave kernel (wn_k_fast.hip) instead of wn_k_fast4.hip */
  WN_ROUTE_COOP_LEGACY = 1 << 10,     /* ... the one-stream cooperative kernel instead of either */
  WN_ROUTE_FAST4_WHOLE = 1 << 11,     /* wn_k_fast4.hip: a group on ONE T wave also when the launch has < 2 groups per CU */
  WN_ROUTE_FLOW_NOCOOP = 1 << 12,     /* adaptLeapFrogFlowD off the cooperative layout (lane / wave / split) */
  WN_ROUTE_FORCE_WAVE = 1 << 13,      /* std / diagonal Gaussians on the one-chain-per-wave layout */
  WN_ROUTE_LANE_GAUSS = 1 << 14,      /* small Gaussians one chain per lane whatever the chain count */
  WN_ROUTE_SPLIT_EAGER = 1 << 15,     /* split mode: eager launches instead of the hipGraph of 64 ticks */
  WN_ROUTE_WSTEP_LEGACY = 1 << 16     /* wn_wstep_*: the one-lane transition kernels instead of the wave kernels (wn_wstep_set_route) */
};

/* keyword arguments of WALNUTS(...) (WALNUTS.py:111-129) + integratorAuxPar
 * (adaptiveIntegrators.py:36-44) as a flat struct. */
typedef struct wn_config {
  int32_t abi_version; /* WN_ABI_VERSION */
  int32_t device;      /* HIP device ordinal */
  int32_t dim;         /* D = q0.size */
  int32_t n_chains;    /* chains resident on this engine */
  int64_t chain_offset; /* global id of local chain 0; RNG is keyed by global id;
                           0 <= chain_offset, chain_offset + n_chains <= 2^32 (ids are 32-bit) */
  int32_t integrator;  /* WN_I_* */
  int32_t M;           /* max doublings, WALNUTS.py:121 */
  int32_t min_c;       /* integratorAuxPar.minC */
  int32_t max_c;       /* integratorAuxPar.maxC */
  double H0;           /* WALNUTS.py:115 */
  double step_size_rand_scale; /* WALNUTS.py:116 */
  double delta0;       /* WALNUTS.py:117 */
  double r2p_prob0;    /* integratorAuxPar.R2Pprob0 */
  int32_t generated;   /* WN_G_* */
  int32_t gen_k;
  int32_t warmup_iter; /* WALNUTS.py:120: iterations 1..warmup_iter adapt H and delta (:701-712) */
  int32_t waves_per_group; /* layout hint: 0 = auto (chosen from target, dim, integrator, n_chains); 4 = waves sharing
                              one 16-chain MFMA group; 64 = one chain per lane where a lane kernel exists */
  int32_t adapt_h;         /* WALNUTS.py:123 adaptH */
  int32_t adapt_delta;     /* WALNUTS.py:125 adaptDelta */
  double adapt_h_target;       /* WALNUTS.py:124 adaptHtarget: P2 quantile 1-target of log igrConst */
  double adapt_delta_target;   /* WALNUTS.py:126 */
  double adapt_delta_quantile; /* WALNUTS.py:127 */
  /* ---- ABI 5 ---- */
  uint32_t route;          /* WN_ROUTE_* bits; 0 = automatic (see above) */
  int32_t split_parts;     /* split mode (dense Gaussian 128 < D <= 1024): 0 / 1 = one part (default), 2 / 4 = sub-batches on streams of
                              their own (was WN_SPLIT_PARTS) */
  int32_t split_ks;        /* ... 2 = k of a gradient tile split over two workgroups (was WN_SPLIT_KS); 0 / 1 = no */
  int32_t adapt_quant_bits; /* 0 = the reference's arithmetic.  b in [8, 52] (opt-in, NOT the reference): the warm-up's adapted H and
                               delta (WALNUTS.py:708-712) are rounded to b significant bits at the end of every warm-up iteration.
                               With warm-up a run is a feedback loop through H / delta that amplifies last-bit differences of the
                               energy sums by ~1e3 per iteration on ill-conditioned targets -- the reference's own result moves by
                               more than 1e-6 within ~15 iterations when ONE input coordinate moves by one ulp
                               (tests/studies/warmup_sensitivity.py) -- and rounding the loop variable cuts the loop: with b = 20 two
                               implementations that differ in summation order stay within 1e-6 of each other through the warm-up
                               ("warm-up tolerance contract").  The oracle has the same switch (adaptQuantBits). */
} wn_config;

typedef struct wn_engine wn_engine;

/* Construct an engine on cfg->device; allocates the engine's own workspace. */
int wn_create(const wn_config *cfg, wn_engine **out);

/* Select the target plugin (replaces passing `lpFun`, WALNUTS.py:111).  params_dev: device
 * buffer of n_params doubles (see WN_T_*), copied into engine-owned storage. */
int wn_set_target(wn_engine *e, int32_t kind, const double *params_dev, size_t n_params);

/* WALNUTS(lpFun, q0, ..., numIter) for all resident chains (WALNUTS.py:189-695).
 *   q0_dev      [C][D]                chain-major initial positions
 *   samples_dev [C][n_iter+1][dg]     row 0 = generated(q0)   (WALNUTS.py:164,695)
 *   diag_dev    [C][n_iter][24]       (WALNUTS.py:670-693)
 *   trace_dev   [C][trace_cap][12+2D] or NULL: one row per integrator call
 *               {iter, doubling, xi, h, Ham0, lp, If, Ib, c, lwt, nEvalF, nEvalB, q', v'}
 *   trace_n_dev [C] int32 or NULL: rows wanted per chain (may exceed trace_cap)
 *   stream      hipStream_t (0 = default stream).  Asynchronous: returns after the launch. */
int wn_run(wn_engine *e, const double *q0_dev, uint64_t seed, int32_t n_iter, double *samples_dev,
           double *diag_dev, double *trace_dev, int32_t trace_cap, int32_t *trace_n_dev,
           void *stream);

/* ---- caller-evaluated target (WN_T_EXTERNAL) ----------------------------------------------------
 * The reference calls lpFun(q) once per leapfrog step of one chain (adaptiveIntegrators.py:53,78);
 * here every chain advances by one gradient evaluation per wn_split_tick and the caller evaluates
 * its target for ALL chains in between:
 *
 *   wn_split_ld(e, &ld);                                  // row stride of the exchange buffers
 *   wn_split_begin(e, q0, seed, n_iter, samples, diag, q_eval, grad, lp, stream);
 *   do {
 *     lp[c], grad[c][0:D] = lpFun(q_eval[c][0:D]) for every chain c   (caller, on the same stream)
 *     wn_split_tick(e, stream);
 *     every so often: wn_split_done(e, &n_done, stream);  // synchronises
 *   } while (n_done < n_chains);
 *
 * q_eval_dev / grad_dev are caller-owned [n_chains][ld] doubles, lp_dev [n_chains]; they must stay
 * valid until the run has finished.  A NaN / -inf lp is a per-chain event as in the reference
 * (mainSW.py:24-26, WALNUTS.py:316-319).  Rows of finished chains keep their last position; their
 * lp / grad are ignored. */
int wn_split_ld(wn_engine *e, int32_t *ld);
int wn_split_begin(wn_engine *e, const double *q0_dev, uint64_t seed, int32_t n_iter,
                   double *samples_dev, double *diag_dev, double *q_eval_dev,
                   const double *grad_dev, const double *lp_dev, void *stream);
int wn_split_tick(wn_engine *e, void *stream);
int wn_split_done(wn_engine *e, int32_t *n_done, void *stream);

/* wn_run with recordOrbitStats=True (WALNUTS.py:128,182-184,724-725): additionally fills
 * orbit_min_dev / orbit_max_dev [C][n_iter][dg] with the per-iteration minimum / maximum of
 * generated(q) over the states of the orbit: identity, first-k, [q0, sum q^2] and (round 5) WN_G_SW_CONSTRAIN -- StockWatson's
 * param_constrain (mainSW.py:28-29) is applied to every orbit state inside the kernel, as the reference's recordOrbitStats does with
 * any generated(). */
int wn_run_orbit_stats(wn_engine *e, const double *q0_dev, uint64_t seed, int32_t n_iter,
                       double *samples_dev, double *diag_dev, double *orbit_min_dev,
                       double *orbit_max_dev, void *stream);

/* One macro step for a batch of independent states -- the integrator plugin call
 * integrator(q,v,g,Ham0,h,xi,lpFun,delta,auxPar) -> integratorReturn
 * (adaptiveIntegrators.py:17-29,49,65,361).  Arrays are [B][D] / [B]; u_r2p_dev holds the
 * uniform R2P consumes (adaptiveIntegrators.py:392).  iout_dev [B][5] = nEvalF,nEvalB,If,Ib,c;
 * dout_dev [B][4] = lp, lwt, H1 (Hamiltonian of the returned state), igrConst.  Uses cfg->integrator,
 * delta0, min_c, max_c, r2p_prob0; B must equal cfg->n_chains. */
int wn_integrator_step(wn_engine *e, const double *q_dev, const double *v_dev, const double *g_dev,
                       const double *ham0_dev, const double *h_dev, const int32_t *xi_dev,
                       const double *u_r2p_dev, double *q1_dev, double *v1_dev, double *g1_dev,
                       int32_t *iout_dev, double *dout_dev, void *stream);

/* lpFun(q) -> (lp, grad) for a batch: q_dev [B][D] -> lp_dev [B], grad_dev [B][D]. */
int wn_lpfun(wn_engine *e, const double *q_dev, double *lp_dev, double *grad_dev, void *stream);

/* Stand-alone lpFun for any WN_T_* kind (no engine): logp(theta) / grad(theta) of the
 * (logp, grad) callable pair that walnuts() takes (walnuts/walnuts.py:379-380). */
int wn_lpfun_kind(int32_t device, int32_t target_kind, const double *tpar_dev, int32_t dim, int32_t B,
                  const double *q_dev, double *lp_dev, double *grad_dev, void *stream);

/* ---- walnuts_step(...) of the pseudocode-faithful variant (walnuts/walnuts.py:279-359), batched -------------
 * B independent chains of dimension D advance by one transition.  The caller's numpy Generators are consumed on
 * the host in the reference's order, which depends on the data (no uniform after a sub-U-turn :343, the categorical
 * draw only after an accept :347-350): one NUTS doubling is one exchange.
 *
 *   wn_wstep_create(device, kind, tpar, n_tpar, D, B, max_nuts_depth, inv_mass, macro_step, max_error, &h);
 *   per transition:
 *     rho[b] = inv_mass**-0.5 * rng_b.normal(size=D)                       (:322-325)
 *     wn_wstep_begin(h, theta, rho)                                        orbit = [(theta, rho)], weights = [-H]  (:326)
 *     for depth in range(max_nuts_depth), while a chain is active:         (:328)
 *       backward[b] = rng_b.binomial(1, .5); lottery[b][m] = rng_b.choice(3) for the 2**depth macro steps (:330,:194)
 *       wn_wstep_extend(h, depth, backward, lottery, active, sub_uturn, lse_ext, lse_old)
 *           device: extend_orbit (:211-276), sub_uturn of the extension (:343), both logsumexp of :345
 *       active[b] &= !sub_uturn[b];  accept[b] = log(rng_b.uniform(0,1)) < lse_ext[b] - lse_old[b]   (:346-347)
 *       u_choice[b] = rng_b.random() if accept[b]                          (Generator.choice(orbit_ext, p=softmax) :350)
 *       wn_wstep_commit(h, accept, u_choice, uturn)
 *           device: categorical pick by inversion, orbit merge (:351), uturn(orbit[0], orbit[-1]) (:352)
 *       active[b] &= !uturn[b]
 *     wn_wstep_result(h, theta_next)                                       theta_selected (:359)
 *
 * kind = a WN_T_* built-in (dim <= 64, or any dim for WN_T_STDGAUSS / WN_T_DIAGGAUSS), or WN_T_EXTERNAL: the target
 * is the caller's (logp, grad) pair (:379-380).  Then every call above leaves chains waiting for an evaluation:
 *     wn_wstep_ext_bind(h, x_req, lp, grad)                                caller-owned [B][D], [B], [B][D]
 *     after begin / extend:  do { wn_wstep_ext_pending(h, &n); if (!n) break;
 *                                 wn_wstep_ext_waiting(h, waiting);    [B] int32: 1 = this chain asked for an evaluation
 *                                 lp[b], grad[b] = logp(x_req[b]), grad(x_req[b]) for the chains with waiting[b];
 *                                 wn_wstep_ext_feed(h, sub_uturn, lse_ext, lse_old); } while (1);
 * (only the waiting chains' x_req rows are positions the reference would evaluate: a chain whose extension has finished
 * keeps a stale row, and a feed ignores whatever is passed for it).
 * All arrays are device memory; int arrays are int32.  Not thread-safe per handle. */
typedef struct wn_wstep wn_wstep;
int wn_wstep_create(int32_t device, int32_t target_kind, const double *tpar_dev, size_t n_tpar, int32_t dim,
                    int32_t B, int32_t max_depth, const double *inv_mass_dev, double macro_step,
                    double max_error, wn_wstep **out);
int wn_wstep_begin(wn_wstep *h, const double *theta_dev, const double *rho_dev, void *stream);
int wn_wstep_extend(wn_wstep *h, int32_t depth, const int32_t *backward_dev, const int32_t *lottery_dev,
                    const int32_t *active_dev, int32_t *sub_uturn_dev, double *lse_ext_dev,
                    double *lse_old_dev, void *stream);
int wn_wstep_commit(wn_wstep *h, const int32_t *accept_dev, const double *u_choice_dev, int32_t *uturn_dev,
                    void *stream);
int wn_wstep_result(wn_wstep *h, double *theta_sel_dev, void *stream);
/* n_transitions whole transitions of every chain in ONE launch, with the engine's own random stream (wn_rng.h: Philox keyed by seed,
 * global chain id = chain_offset + b, transition number iter0 + i) in place of the caller's numpy Generators: the same transition
 * kernel (walnuts.py:322-359), no host round trip per doubling.  theta_dev [B][D] is read and overwritten with the last states;
 * draws_dev [B][n_transitions][D] (or NULL) receives every state.  Built-in targets only. */
int wn_wstep_run_device_rng(wn_wstep *h, uint64_t seed, int64_t chain_offset, int32_t iter0, int32_t n_transitions,
                            double *theta_dev, double *draws_dev, void *stream);
/* (new; test / diagnostics) integer statistics of the following wn_wstep_run_device_rng calls: stats_dev [B][n_transitions][4] int32 =
 * {doublings started (binomial draws, walnuts.py:330), accepts (:347), sum of the chosen micro-step counts (choose_micro_steps, :194),
 * sum of the forward ell_stable (:250)} per chain and transition -- what tests/golden/walnuts_device.npz holds for the unmodified
 * reference run on the same random stream.  NULL detaches.  The buffer is caller-owned. */
int wn_wstep_set_stats(wn_wstep *h, int32_t *stats_dev);
/* (ABI 5) kernel routing of this handle: WN_ROUTE_WSTEP_LEGACY or 0 (was the environment variable WN_WSTEP_LEGACY). */
int wn_wstep_set_route(wn_wstep *h, uint32_t route);
int wn_wstep_ext_bind(wn_wstep *h, double *x_req_dev, const double *lp_dev, const double *grad_dev);
int wn_wstep_ext_pending(wn_wstep *h, int32_t *n_pending, void *stream);
int wn_wstep_ext_waiting(wn_wstep *h, int32_t *waiting_dev, void *stream);
int wn_wstep_ext_feed(wn_wstep *h, int32_t *sub_uturn_dev, double *lse_ext_dev, double *lse_old_dev,
                      void *stream);
const char *wn_wstep_last_error(wn_wstep *h); /* h may be NULL: last creation error */
int wn_wstep_destroy(wn_wstep *h);

/* Export the engine's random stream (wn_rng.h) so the oracle / the Python reference can be
 * driven by the same randomness: u_dev [C][n_u] uniforms, z_dev [C][D] normals of iteration
 * `iter` (1-based) for every resident chain.  Either pointer may be NULL. */
int wn_rng_tape(wn_engine *e, uint64_t seed, int32_t iter, int32_t n_u, double *u_dev,
                double *z_dev, void *stream);

/* Kernel time of the last wn_run in milliseconds, measured with HIP events recorded on the
 * launch stream (synchronises on the stop event). */
int wn_last_run_ms(wn_engine *e, float *ms);
/* (new) Waits for the last wn_run / wn_run_orbit_stats of this engine and reports whether it completed: WN_E_HIP when a kernel's
 * bounded loop gave up (internal error; the samples of that run are incomplete).  wn_run itself stays asynchronous -- it only
 * reports an earlier run that has already finished -- so a caller that wants the verdict of a run asks here (or through
 * wn_last_run_ms, which waits anyway); wn_destroy reports a run nobody asked about.  Replaces nothing in the reference: its
 * Python loop cannot fail this way. */
int wn_check_run(wn_engine *e);

/* Split mode (dense Gaussian with 128 < D <= 1024, SURVEY 8d cfg 4): chain-kernel ticks the last wn_run launched -- each one
 * {batched gradient kernel, one tick of every chain}, in batches of 64 captured into a hipGraph.  0 for fused kernels.  (New:
 * measurement only -- bench.py reports time per tick beside the kernels' own durations.) */
int wn_last_split_ticks(wn_engine *e, int64_t *ticks);

/* Test hook: fills the LDS (160 KiB) of every CU of `device` with the 64-bit `pattern` (4 x CUs workgroups that own a whole CU's
 * LDS each).  LDS keeps what the previous kernel left there; the sampler kernels must not depend on it
 * (tests/test_gpu_lds.py runs every kernel family behind NaN and garbage patterns and compares bits). */
int wn_debug_poison_lds(int32_t device, uint64_t pattern, void *stream);
/* Test hook (ABI 5; was the environment variable WN_DEBUG_CLUSTER_DROP): cluster layout (D > 4096), the last slice of every chain skips
 * its arrival at cross-workgroup exchange number `exchange` of the following runs (0 = never, the default): the siblings' bounded spins
 * must give up and wn_run must report it (tests/test_gpu_sweep.py). */
int wn_debug_cluster_drop(wn_engine *e, int32_t exchange);

/* Which kernel the engine dispatches to: "wn_coop_kernel<NT=2,W=4,KS=25,dense>", "wn_lane_kernel<DMAX=..>" ...
 * The string is owned by the engine: valid until the next wn_set_target or wn_destroy of THIS engine;
 * "" before the first successful wn_set_target. */
const char *wn_kernel_name(wn_engine *e);

const char *wn_last_error(wn_engine *e); /* e may be NULL: last creation error */
int wn_destroy(wn_engine *e);
int wn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
