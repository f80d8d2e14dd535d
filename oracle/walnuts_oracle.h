/*
 * walnuts_oracle.h -- CPU restatement of the WALNUTS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the parity oracle for the MI355X engine in walnuts_amd/.  It is a
 * plain-C, single-chain, strictly sequential restatement of the reference sampler
 * (bob-carpenter/walnuts, WALNUTSpy/WALNUTS.py + adaptiveIntegrators.py + targetDistr.py),
 * following the reference's order of arithmetic and order of random draws.  Every
 * function cites the reference file:line it restates.
 *
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under walnuts_amd/ imports, links or calls it.
 *
 * Pinning: the reference holds no golden vectors, so the oracle is
 * pinned against outputs of the reference itself, generated in the build container by
 * tests/golden/make_golden.py (tape-replay of the unmodified Python reference) and
 * committed as tests/golden/ *.npz.  tests/test_oracle_golden.py checks every fixture.
 */
#ifndef WALNUTS_ORACLE_H
#define WALNUTS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* target kinds (values shared with include/walnuts_amd.h) */
enum {
  WNO_T_STDGAUSS = 0,    /* targetDistr.py:18-21 */
  WNO_T_FUNNEL10 = 1,    /* targetDistr.py:74-78 */
  WNO_T_DENSEGAUSS = 2,  /* SURVEY.md 8d cfg 2/4: lp=-1/2 q'Aq, grad=-Aq, tpar = A row-major d*d */
  WNO_T_DIAGGAUSS = 3,   /* SURVEY.md 8d cfg 2 diagonal variant: tpar = diag(A), d */
  WNO_T_CORRGAUSS = 4,   /* targetDistr.py:25-31 */
  WNO_T_SMILE = 5,       /* targetDistr.py:34-38 */
  WNO_T_FUNNEL1 = 6,     /* targetDistr.py:88-93 */
  WNO_T_MODFUNNEL = 7,   /* targetDistr.py:41-51 */
  /* 8, 9: test/targets.py pairs (walnuts() variant); 10: caller-evaluated target of the product ABI */
  WNO_T_STOCKWATSON = 11 /* WALNUTSpy_examples/StockWatson/sw_innov.stan:7-51; tpar = [T, y[T]], d = 3T */
};

/* integrator kinds */
enum {
  WNO_I_FIXED = 0, /* adaptiveIntegrators.py:49-59  fixedLeapFrog   */
  WNO_I_ADAPT_D = 1, /* adaptiveIntegrators.py:65-137 adaptLeapFrogD  */
  WNO_I_ADAPT_R2P = 2, /* adaptiveIntegrators.py:361-475 adaptLeapFrogR2P */
  WNO_I_ADAPT_YOSHIDA = 3, /* adaptiveIntegrators.py:142-240 adaptYoshidaD (4th order, 3 gradients per micro step) */
  WNO_I_ADAPT_FLOWD = 4    /* adaptiveIntegrators.py:246-356 adaptLeapFrogFlowD (flow-error criterion, 2 gradients per micro step) */
};

/* generated() kinds: what is stored per draw */
enum {
  WNO_G_IDENTITY = 0,  /* WALNUTS.py:113 default lambda q: q */
  WNO_G_Q0_SUMSQ = 1,  /* mainGaussESS.py:29-30  [q[0], sum(q*q)] */
  WNO_G_FIRSTK = 2,    /* mainFunnel.py:17-18    q[0:k] */
  WNO_G_SW_CONSTRAIN = 3 /* mainSW.py:28-29  param_constrain(q, include_tp=True) of sw_innov.stan (:17-35): [q, sigma, gpri, z, x, tau] */
};

/* rng modes */
enum {
  WNO_RNG_PHILOX = 0, /* engine stream: Philox4x32-10 keyed (seed; chain, iter, stream) */
  WNO_RNG_TAPE = 1    /* explicit per-iteration tapes (used for the Python-reference goldens) */
};

#define WNO_NDIAG 24
/* one trace row = 12 scalars + q[d] + v[d] */
#define WNO_TRACE_NSCAL 12

typedef struct {
  int d;
  int target;
  const double *tpar;
  int integrator;
  double H0;                /* WALNUTS.py:115 */
  double stepSizeRandScale; /* WALNUTS.py:116 */
  double delta0;            /* WALNUTS.py:117 */
  int numIter;              /* WALNUTS.py:119 */
  int warmupIter;           /* WALNUTS.py:120 */
  int M;                    /* WALNUTS.py:121 */
  int minC, maxC;           /* adaptiveIntegrators.py:37 */
  double R2Pprob0;          /* adaptiveIntegrators.py:37 */
  int generated;
  int gen_k;
  /* warm-up adaptation, WALNUTS.py:123-127, 139-147, 313, 701-712 (+ P2quantile.py) */
  int adaptH;
  double adaptHtarget;
  int adaptDelta;
  double adaptDeltaTarget;
  double adaptDeltaQuantile;
  /* rng */
  int rng_mode;
  uint64_t seed;
  uint32_t chain;
  const double *tape_u; /* [numIter][tape_u_stride] uniforms in call order, per iteration */
  int tape_u_stride;
  const double *tape_z; /* [numIter][d] standard normals, per iteration */
  /* recordOrbitStats (WALNUTS.py:128,182-184,274-276,331-333,...): per-iteration min / max of
   * generated(q) over the orbit states; [numIter][dg] each, or NULL */
  double *orbitMin;
  double *orbitMax;
  /* optional per-integrator-call trace */
  double *trace; /* [trace_cap][WNO_TRACE_NSCAL + 2*d] */
  int trace_cap;
  /* NOT the reference (opt-in, 0 = off = WALNUTS.py:708-712 as written): the adapted delta and H rounded to this many explicit
   * mantissa bits (8..52) whenever the warm-up recomputes them.  Cuts the feedback loop through H / delta that amplifies
   * last-bit differences of the energy sums (tests/studies/warmup_sensitivity.py); the engine has the same switch
   * (wn_config.adapt_quant_bits) and rounds with the same integer arithmetic. */
  int adaptQuantBits;
} wno_config;

/* sizes */
int wno_gen_dim(const wno_config *cfg);

/* lpFun(q) -> (lp, grad)   targetDistr.py  */
int wno_lpfun(int target, int d, const double *tpar, const double *q, double *lp, double *grad);

/* subTreePlan(nleaf) -> (nleaf-1) x 2 ints   WALNUTS.py:22-41 */
int wno_subtreeplan(int nleaf, int *checks);

/* stopCondition(qm,vm,qp,vp)   WALNUTS.py:95-97 */
int wno_stopcondition(int d, const double *qm, const double *vm, const double *qp, const double *vp);

/* one macro-step integrator call; `u_r2p` is the uniform consumed by R2P
 * (adaptiveIntegrators.py:392); returns fields of integratorReturn
 * (adaptiveIntegrators.py:17-29).  iout = {nEvalF, nEvalB, If, Ib, c}; dout = {lp, lwt, igrConst}. */
int wno_integrator(int integrator, int target, int d, const double *tpar, const double *q,
                   const double *v, const double *g, double Ham0, double h, int xi, double delta,
                   int minC, int maxC, double R2Pprob0, double u_r2p, double *q1, double *v1,
                   double *g1, int *iout, double *dout);

/* WALNUTS(...) for one chain.  samples: [(numIter+1)][dg] (row r = column r of the
 * reference's (dg, numIter+1) array); diag: [numIter][24]; returns <0 on error, else the
 * number of trace rows written (0 when cfg->trace is NULL).  n_u_used (optional, may be
 * NULL): [numIter] number of uniforms consumed per iteration. */
int wno_walnuts(const wno_config *cfg, const double *q0, double *samples, double *diag,
                int *n_u_used);

/* P2quantile (WALNUTSpy/P2quantile.py:16-86): push xs[0..n-1], return quantile() after each push */
void wno_p2quantile(double prob, int n, const double *xs, double *q_after);

/* np.quantile(a[0:n], q), method "linear" (numpy/lib/_function_base_impl.py), used by
 * WALNUTS.py:706 */
double wno_np_quantile(const double *a, int n, double q);

/* engine RNG, restated independently of walnuts_amd/csrc/wn_rng.h (DESIGN.md "RNG"). */
void wno_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double wno_rng_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t k);
double wno_rng_normal(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t j);
void wno_rng_tape(uint64_t seed, uint32_t chain, uint32_t iter, int nu, double *u, int nz,
                  double *z);

const char *wno_version(void);

#ifdef __cplusplus
}
#endif
#endif
