/*
 * walnuts_oracle.c -- CPU restatement of the WALNUTS hot path.  TEST INFRASTRUCTURE ONLY
 * (see walnuts_oracle.h).  Plain C99, no FMA contraction (build with -ffp-contract=off),
 * arithmetic in the reference's order.  Citations are relative to /root/reference/.
 *
 * Parity status: PINNED against outputs of the unmodified Python reference run in the build
 * container (tests/golden/make_golden.py -> tests/golden/ *.npz; the reference has no golden
 * vectors of its own).  The dense/diagonal Gaussian targets are not in the reference
 * (SURVEY.md 8d defines them); they are pinned through a Python callable of the same
 * arithmetic passed to the reference sampler as lpFun.
 */
#define _GNU_SOURCE
#include "walnuts_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* WALNUTSpy/constants.py:13-14 */
#define LOGZERO (-700.0)
static double wt_sum_thresh(void) { return exp(LOGZERO + 1.0); }

const char *wno_version(void) { return "walnuts-oracle r1"; }

/* ------------------------------------------------------------------------------------------
 * engine RNG (the build's own stream, DESIGN.md "RNG"); not a reference restatement.
 * Philox4x32-10 (Salmon et al. 2011), counter = (block, iter, chain, stream).
 * ---------------------------------------------------------------------------------------- */
void wno_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void rng_block(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t stream, uint32_t blk,
                      uint64_t *x0, uint64_t *x1) {
  uint32_t ctr[4] = {blk, iter, chain, stream};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  wno_philox4x32_10(ctr, key, w);
  *x0 = ((uint64_t)w[1] << 32) | w[0];
  *x1 = ((uint64_t)w[3] << 32) | w[2];
}

double wno_rng_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t k) {
  uint64_t x0, x1;
  rng_block(seed, chain, iter, 0u, k >> 1, &x0, &x1);
  uint64_t x = (k & 1u) ? x1 : x0;
  return (double)(x >> 11) * 0x1.0p-53;
}

double wno_rng_normal(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t j) {
  uint64_t x0, x1;
  rng_block(seed, chain, iter, 1u, j >> 1, &x0, &x1);
  double u1 = ((double)(x0 >> 12) + 0.5) * 0x1.0p-52;
  double u2 = (double)(x1 >> 11) * 0x1.0p-53;
  double r = sqrt(-2.0 * log(u1));
  double th = 6.283185307179586476925286766559 * u2;
  return (j & 1u) ? r * sin(th) : r * cos(th);
}

void wno_rng_tape(uint64_t seed, uint32_t chain, uint32_t iter, int nu, double *u, int nz,
                  double *z) {
  for (int k = 0; k < nu; ++k) u[k] = wno_rng_uniform(seed, chain, iter, (uint32_t)k);
  for (int j = 0; j < nz; ++j) z[j] = wno_rng_normal(seed, chain, iter, (uint32_t)j);
}

/* ------------------------------------------------------------------------------------------
 * summation helpers
 * ---------------------------------------------------------------------------------------- */

/* cfg->adaptQuantBits (opt-in; NOT in the reference): x rounded to b explicit mantissa bits, ties away from zero, by integer
 * arithmetic on the bit pattern -- the same operation as the engine's wn_quant_bits (walnuts_amd/csrc/wn_chain.h) */
static double quant_bits(double x, int b) {
  unsigned long long u;
  if (b <= 0 || b >= 52) return x;
  memcpy(&u, &x, 8);
  if (((u >> 52) & 0x7ffull) == 0x7ffull) return x; /* inf / nan: unchanged (an integer test: immune to floating-point contraction) */
  u += 1ull << (51 - b);
  u &= ~((1ull << (52 - b)) - 1ull);
  memcpy(&x, &u, 8);
  return x;
}

/* Python builtin sum(x*y): strict left to right (WALNUTS.py:97,256,308; adaptiveIntegrators.py:55) */
static double seqdot(int n, const double *x, const double *y) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

/* numpy np.sum of a contiguous float64 array: pairwise summation with 8 accumulators and
 * 128-element blocks (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum);
 * used by targetDistr.py:19 (np.sum(q*q)).  Checked against np.sum in tests. */
static double np_pairwise_sum_sq(const double *q, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += q[i] * q[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    int i;
    for (int j = 0; j < 8; ++j) r[j] = q[j] * q[j];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += q[i + j] * q[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += q[i] * q[i];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_sq(q, n2) + np_pairwise_sum_sq(q + n2, n - n2);
  }
}

/* ------------------------------------------------------------------------------------------
 * targets: lpFun(q, hessian=False) -> [lp, grad]      WALNUTSpy/targetDistr.py
 * ---------------------------------------------------------------------------------------- */
int wno_lpfun(int target, int d, const double *tpar, const double *q, double *lp, double *grad) {
  switch (target) {
    case WNO_T_STDGAUSS: { /* targetDistr.py:18-21 */
      *lp = -0.5 * np_pairwise_sum_sq(q, d);
      for (int i = 0; i < d; ++i) grad[i] = -q[i];
      return 0;
    }
    case WNO_T_FUNNEL10: { /* targetDistr.py:74-78; scipy.stats.norm.logpdf(x,loc,scale) =
                              -((x-loc)/scale)**2/2 - log(sqrt(2*pi)) - log(scale) */
      if (d != 11) return -2;
      const double logC = log(sqrt(2.0 * M_PI));
      double x0 = (q[0] - 0.0) / 3.0;
      double lp0 = -(x0 * x0) / 2.0 - logC - log(3.0);
      double scale = exp(0.5 * q[0]);
      double lscale = log(scale);
      double s = 0.0; /* builtin sum: 0 + t1 + t2 + ... */
      for (int i = 1; i < 11; ++i) {
        double x = (q[i] - 0.0) / scale;
        double t = -(x * x) / 2.0 - logC - lscale;
        s += t;
      }
      *lp = lp0 + s;
      double emq = exp(-q[0]);
      double ss = 0.0;
      for (int i = 1; i < 11; ++i) ss += q[i] * q[i];
      grad[0] = -5.0 - q[0] / 9.0 + 0.5 * emq * ss;
      for (int i = 1; i < 11; ++i) grad[i] = -q[i] * emq;
      return 0;
    }
    case WNO_T_DENSEGAUSS: { /* lp = -1/2 q'Aq, grad = -Aq (row dots, left to right) */
      double s = 0.0;
      for (int i = 0; i < d; ++i) {
        double a = 0.0;
        const double *row = tpar + (size_t)i * d;
        for (int j = 0; j < d; ++j) a += row[j] * q[j];
        grad[i] = -a;
      }
      for (int i = 0; i < d; ++i) s += q[i] * grad[i];
      *lp = 0.5 * s;
      return 0;
    }
    case WNO_T_DIAGGAUSS: { /* SURVEY.md 8d diagonal variant */
      double s = 0.0;
      for (int i = 0; i < d; ++i) grad[i] = -(tpar[i] * q[i]);
      for (int i = 0; i < d; ++i) s += q[i] * grad[i];
      *lp = 0.5 * s;
      return 0;
    }
    case WNO_T_CORRGAUSS: { /* targetDistr.py:25-31 */
      if (d != 2) return -2;
      double rho = 0.5;
      double tmp = 1.0 - rho * rho;
      double r = q[1] - rho * q[0];
      *lp = -0.5 * pow(q[0], 2.0) - (0.5 / tmp) * pow(r, 2.0); /* numpy scalar ** -> pow */
      grad[0] = -(q[0] - rho * q[1]) / tmp;
      grad[1] = -(q[1] - rho * q[0]) / tmp;
      return 0;
    }
    case WNO_T_SMILE: { /* targetDistr.py:34-38 */
      if (d != 2) return -2;
      double q0s = pow(q[0], 2.0);
      double r = q[1] - q0s;
      *lp = -0.5 * q0s - 0.5 * pow(r, 2.0);
      grad[0] = -q[0] + 2.0 * q[0] * q[1] - 2.0 * pow(q[0], 3.0);
      grad[1] = q0s - q[1];
      return 0;
    }
    case WNO_T_FUNNEL1: { /* targetDistr.py:88-93 */
      if (d != 2) return -2;
      const double logC = log(sqrt(2.0 * M_PI));
      double x0 = q[0] / 3.0;
      double scale = exp(0.5 * q[0]);
      double x1 = q[1] / scale;
      *lp = (-(x0 * x0) / 2.0 - logC - log(3.0)) + (-(x1 * x1) / 2.0 - logC - log(scale));
      double emq = exp(-q[0]);
      grad[0] = -0.5 - q[0] / 9.0 + 0.5 * pow(q[1], 2.0) * emq;
      grad[1] = -q[1] * emq;
      return 0;
    }
    case WNO_T_MODFUNNEL: { /* targetDistr.py:41-51 */
      if (d != 2) return -2;
      double x = q[0], y = q[1];
      double t1 = exp(-3.0 * x);
      double t2 = 1.0 + t1;
      double t3 = 0.1e1 / t2;
      double t4 = pow(y, 2.0);
      double t5 = -0.1e1 / 0.2e1;
      *lp = t5 * (t2 * t4 + log(t3) + pow(x, 2.0));
      grad[0] = 0.3e1 / 0.2e1 * t1 * (t4 - t3) - x;
      grad[1] = -y * t2;
      return 0;
    }
    case WNO_T_STOCKWATSON: {
      /* WALNUTSpy_examples/StockWatson/sw_innov.stan:7-51 through bridgestan's
       * log_density_gradient (mainSW.py:22): all parameters unconstrained (no Jacobian), constants
       * dropped (propto).  bridgestan / Stan are absent from the reference tree and from this image
       * (no version pinned): PARITY UNPINNED for this target -- the restatement follows
       * the Stan program line by line and is checked against finite differences and an independent
       * NumPy restatement in tests/.  tpar = [T, y[1..T]];  d = 3T.
       * Parameter order (sw_innov.stan:7-14): tSigma, z1, zinn[T-2], x1, xinn[T-1], tau1, tauinn[T-1]. */
      const int T = (int)tpar[0];
      const double *y = tpar + 1;
      if (T < 3 || d != 3 * T) return -2;
      const double tS = q[0];
      const double *zin = q + 1;      /* zin[0] = z1, zin[j] = zinn[j], j = 1..T-2 */
      const double *xin = q + T;      /* xin[0] = x1, xin[j] = xinn[j], j = 1..T-1 */
      const double *tin = q + 2 * T;  /* tin[0] = tau1, tin[j] = tauinn[j], j = 1..T-1 */
      const double sigma = exp(-0.5 * tS); /* :24 */
      const double gpri = exp(tS);         /* :25 */
      double *z = (double *)malloc(sizeof(double) * (size_t)(8 * T));
      if (!z) return -3;
      double *x = z + T, *tau = x + T, *ez = tau + T, *a = ez + T, *b = a + T, *c = b + T, *A = c + T;
      z[1] = zin[0]; /* :27-28 (1-based arrays as in the Stan program) */
      for (int t = 2; t <= T - 1; ++t) z[t] = z[t - 1] + sigma * zin[t - 1];
      double *x1b = x - 1; /* x1b[t], t = 1..T */
      x1b[1] = xin[0];     /* :30-31 */
      for (int t = 2; t <= T; ++t) x1b[t] = x1b[t - 1] + sigma * xin[t - 1];
      double *tau1b = tau - 1;
      tau1b[1] = tin[0]; /* :33-34 */
      for (int t = 2; t <= T; ++t) {
        ez[t - 1] = exp(0.5 * z[t - 1]);
        tau1b[t] = tau1b[t - 1] + ez[t - 1] * tin[t - 1];
      }
      double r = 5.0 * tS - 0.5 * gpri; /* :40 */
      double sz = 0.0, sx = 0.0, st = 0.0;
      for (int j = 1; j <= T - 2; ++j) sz += zin[j] * zin[j]; /* :44-46 normal(0,1), constants dropped */
      for (int j = 1; j <= T - 1; ++j) sx += xin[j] * xin[j];
      for (int j = 1; j <= T - 1; ++j) st += tin[j] * tin[j];
      r += -0.5 * sz - 0.5 * sx - 0.5 * st;
      double *a1 = a - 1, *b1 = b - 1;
      double so = 0.0;
      for (int t = 1; t <= T; ++t) { /* :47-49  y[t] ~ normal(tau[t], exp(0.5*x[t])) */
        const double w = exp(-x1b[t]), res = y[t - 1] - tau1b[t];
        so += -0.5 * x1b[t] - 0.5 * (res * res) * w;
        a1[t] = res * w;                      /* d/dtau[t] */
        b1[t] = -0.5 + 0.5 * (res * res) * w; /* d/dx[t] */
      }
      r += so;
      *lp = r;
      /* reverse sweep */
      double *A1 = A - 1; /* A1[s] = sum_{t >= s} a[t] */
      double acc = 0.0;
      for (int t = T; t >= 1; --t) { acc += a1[t]; A1[t] = acc; }
      grad[2 * T] = A1[1]; /* tau1 */
      for (int s = 2; s <= T; ++s) {
        grad[2 * T + s - 1] = ez[s - 1] * A1[s] - tin[s - 1];  /* tauinn[s-1] */
        c[s - 1] = 0.5 * ez[s - 1] * tin[s - 1] * A1[s];      /* d/dz[s-1] */
      }
      double dsig = 0.0; /* sum over states of (dL/dstate) * (state - first state) / sigma-chain */
      acc = 0.0;
      for (int k = T - 1; k >= 1; --k) { /* z[k] = z1 + sigma * sum_{j<k} zinn[j] */
        if (k <= T - 2) grad[1 + k] = sigma * acc - zin[k]; /* zinn[k] feeds z[k+1..T-1] */
        acc += c[k];
        dsig += c[k] * (z[k] - z[1]);
      }
      grad[1] = acc; /* z1 */
      acc = 0.0;
      for (int t = T; t >= 1; --t) { /* x[t] = x1 + sigma * sum_{j<t} xinn[j] */
        if (t <= T - 1) grad[T + t] = sigma * acc - xin[t]; /* xinn[t] feeds x[t+1..T] */
        acc += b1[t];
        dsig += b1[t] * (x1b[t] - x1b[1]);
      }
      grad[T] = acc; /* x1 */
      grad[0] = 5.0 - 0.5 * gpri - 0.5 * dsig; /* dsigma/dtSigma = -sigma/2 */
      free(z);
      return 0;
    }
    default:
      return -1;
  }
}

/* ------------------------------------------------------------------------------------------
 * subTreePlan(nleaf)    WALNUTS.py:22-41
 * ---------------------------------------------------------------------------------------- */
static void sub_uturn_rec(int a, int b, int *checks, int *k) {
  if (a != b) {
    int m = (int)floor((a + b) / 2.0); /* WALNUTS.py:36 */
    sub_uturn_rec(a, m, checks, k);
    sub_uturn_rec(m + 1, b, checks, k);
    checks[2 * (*k)] = a; /* Uturn(a,b)  WALNUTS.py:27-32 */
    checks[2 * (*k) + 1] = b;
    (*k)++;
  }
}
int wno_subtreeplan(int nleaf, int *checks) {
  int k = 0;
  sub_uturn_rec(1, nleaf, checks, &k);
  return k; /* nleaf-1 rows */
}

/* stopCondition   WALNUTS.py:95-97 */
int wno_stopcondition(int d, const double *qm, const double *vm, const double *qp,
                      const double *vp) {
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < d; ++i) {
    double tmp = qp[i] - qm[i];
    s1 += vp[i] * tmp;
  }
  if (s1 < 0.0) return 1;
  for (int i = 0; i < d; ++i) {
    double tmp = qp[i] - qm[i];
    s2 += vm[i] * tmp;
  }
  return s2 < 0.0;
}

/* ------------------------------------------------------------------------------------------
 * integrators          WALNUTSpy/adaptiveIntegrators.py
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int target, d;
  const double *tpar;
} lpctx;

typedef struct {
  double lp, lwt, igrConst;
  int nEvalF, nEvalB, If, Ib, c;
} igr_ret;

/* One pass of 2**c micro steps from (q, vv0, g) (adaptiveIntegrators.py:70-84 / 112-126).
 * On return qq,vv,gg hold the end state, *fnew the last lp; returns all(isfinite(Hams));
 * *maxErr = abs(Hams[0]-Hams[-1]); *maxAbsDiff = max(abs(diff(Hams))); *Hlast = Hams[-1]. */
static int g_yoshida = 0; /* set by integrator_impl for adaptYoshidaD (single-threaded test tool) */
static int g_flow = 0;    /* set by integrator_impl for adaptLeapFrogFlowD */
/* np.max(np.abs(a - b)) over d elements (NaN-propagating) */
static double np_max_absdiff(int d, const double *a, const double *b) {
  double m = -INFINITY;
  int nan = 0;
  for (int k = 0; k < d; ++k) {
    double x = fabs(a[k] - b[k]);
    if (isnan(x)) nan = 1;
    else if (x > m) m = x;
  }
  return nan ? NAN : m;
}
/* Python's builtin max(err, x): x if x > err else err */
static double py_max(double err, double x) { return x > err ? x : err; }
static int micro_pass(const lpctx *t, int c, double h, const double *q, const double *vv0,
                      const double *g, double Ham0, double *qq, double *vv, double *gg,
                      double *vh, double *fnew, double *hh_out, double *maxErr,
                      double *maxAbsDiff, double *Hlast, int *nEval) {
  int d = t->d;
  long nstep = 1L << c;
  double hh = h / (double)nstep;
  memcpy(qq, q, sizeof(double) * d);
  memcpy(vv, vv0, sizeof(double) * d);
  memcpy(gg, g, sizeof(double) * d);
  int allfin = isfinite(Ham0) ? 1 : 0;
  double prev = Ham0, mad = -INFINITY;
  int madnan = 0;
  double flow_max = -INFINITY;
  int flow_nan = 0;
  for (long i = 1; i <= nstep; ++i) {
    if (g_yoshida) { /* adaptiveIntegrators.py:143-144,157-173: three leapfrog sub-steps */
      const double firstLast = 1.351207191959658, middle = -1.702414383919315;
      for (int sub = 0; sub < 3; ++sub) {
        const double w = (sub == 1) ? middle : firstLast;
        const double a = 0.5 * w * hh, bq = w * hh; /* 0.5*firstLast*hh ; firstLast*hh */
        for (int k = 0; k < d; ++k) vh[k] = vv[k] + a * gg[k];
        for (int k = 0; k < d; ++k) qq[k] = qq[k] + bq * vh[k];
        wno_lpfun(t->target, d, t->tpar, qq, fnew, gg);
        (*nEval)++;
        for (int k = 0; k < d; ++k) vv[k] = vh[k] + a * gg[k];
      }
    } else if (g_flow) { /* adaptLeapFrogFlowD, adaptiveIntegrators.py:262-290 (forward) / 319-343 (backward) */
      double *qold = (double *)malloc(sizeof(double) * (size_t)(6 * d));
      double *gold = qold + d, *vold = gold + d, *qMid = vold + d, *gMid = qMid + d, *tmp = gMid + d;
      double hh2 = 0.5 * hh, fMid;
      for (int k = 0; k < d; ++k) vh[k] = vv[k] + hh2 * gg[k];                     /* :263 */
      memcpy(qold, qq, sizeof(double) * d); memcpy(gold, gg, sizeof(double) * d);  /* :264-266 */
      memcpy(vold, vv, sizeof(double) * d);
      for (int k = 0; k < d; ++k) qq[k] = qq[k] + hh * vh[k];                      /* :267 */
      wno_lpfun(t->target, d, t->tpar, qq, fnew, gg);                              /* :268 */
      (*nEval)++;
      for (int k = 0; k < d; ++k) vv[k] = vh[k] + hh2 * gg[k];                     /* :270 */
      for (int k = 0; k < d; ++k) qMid[k] = 0.5 * (qq[k] + qold[k]) + (hh / 8.0) * (vold[k] - vv[k]); /* :272 */
      wno_lpfun(t->target, d, t->tpar, qMid, &fMid, gMid);                         /* :273 */
      (*nEval)++;
      const double hh_sq = hh * hh, h6 = hh / 6.0;
      for (int k = 0; k < d; ++k)                                                  /* :276 qf */
        tmp[k] = qold[k] + hh * vold[k] + hh_sq * ((1.0 / 6.0) * gold[k] + (1.0 / 3.0) * gMid[k]);
      double err = np_max_absdiff(d, tmp, qq);                                     /* :277 */
      for (int k = 0; k < d; ++k) tmp[k] = vold[k] + h6 * (gold[k] + gg[k] + 4.0 * gMid[k]); /* :279 vf */
      err = py_max(err, np_max_absdiff(d, tmp, vv));                               /* :280 */
      for (int k = 0; k < d; ++k)                                                  /* :282 qb */
        tmp[k] = qq[k] - hh * vv[k] + hh_sq * ((1.0 / 6.0) * gg[k] + (1.0 / 3.0) * gMid[k]);
      err = py_max(err, np_max_absdiff(d, tmp, qold));                             /* :283 */
      for (int k = 0; k < d; ++k) tmp[k] = -(-vv[k] + h6 * (gold[k] + gg[k] + 4.0 * gMid[k])); /* :285 vb */
      err = py_max(err, np_max_absdiff(d, tmp, vold));                             /* :287 */
      if (isnan(err)) flow_nan = 1; /* maxErr = np.max(Errs)  :294 */
      else if (err > flow_max) flow_max = err;
      free(qold);
    } else {
    double hh2 = 0.5 * hh;
    for (int k = 0; k < d; ++k) vh[k] = vv[k] + hh2 * gg[k];
    for (int k = 0; k < d; ++k) qq[k] = qq[k] + hh * vh[k];
    wno_lpfun(t->target, d, t->tpar, qq, fnew, gg);
    (*nEval)++;
    for (int k = 0; k < d; ++k) vv[k] = vh[k] + hh2 * gg[k];
    }
    double Hi = -(*fnew) + 0.5 * seqdot(d, vv, vv);
    if (!isfinite(Hi)) allfin = 0;
    double df = fabs(Hi - prev);
    if (isnan(df)) madnan = 1; /* np.max propagates nan */
    else if (df > mad) mad = df;
    prev = Hi;
  }
  *hh_out = hh;
  *maxErr = g_flow ? (flow_nan ? NAN : flow_max) : fabs(Ham0 - prev);
  *maxAbsDiff = madnan ? NAN : mad;
  *Hlast = prev;
  return allfin;
}

static void integrator_impl(int kind, const lpctx *t, const double *q, const double *v,
                            const double *g, double Ham0, double h, int xi, double delta, int minC,
                            int maxC, double R2Pprob0, double (*unif)(void *), void *uctx,
                            double *q1, double *v1, double *g1, igr_ret *ret, double *work) {
  int d = t->d;
  double *qq = work, *vv = work + d, *gg = work + 2 * d, *vh = work + 3 * d;
  double *vv0 = work + 4 * d;
  double *qO = work + 5 * d, *vO = work + 6 * d, *gO = work + 7 * d;

  if (kind == WNO_I_FIXED) { /* adaptiveIntegrators.py:49-59 */
    double hh2 = 0.5 * h, fnew;
    for (int k = 0; k < d; ++k) vh[k] = (double)xi * v[k] + hh2 * g[k];
    for (int k = 0; k < d; ++k) q1[k] = q[k] + h * vh[k];
    wno_lpfun(t->target, d, t->tpar, q1, &fnew, g1);
    for (int k = 0; k < d; ++k) vv[k] = vh[k] + hh2 * g1[k];
    double H1 = -fnew + 0.5 * seqdot(d, vv, vv);
    for (int k = 0; k < d; ++k) v1[k] = (double)xi * vv[k];
    ret->lp = fnew;
    ret->nEvalF = 1; ret->nEvalB = 0; ret->If = 0; ret->Ib = 0; ret->c = 0;
    ret->lwt = 0.0;
    ret->igrConst = h * pow(fmax(1.0e-10, fabs(Ham0 - H1)), -1.0 / 3.0);
    return;
  }

  g_yoshida = (kind == WNO_I_ADAPT_YOSHIDA);
  if (kind == WNO_I_ADAPT_YOSHIDA) kind = WNO_I_ADAPT_D; /* same search/weight logic, :146-240 */
  g_flow = (kind == WNO_I_ADAPT_FLOWD);
  if (kind == WNO_I_ADAPT_FLOWD) { /* :246-356: the search of adaptLeapFrogD with c = 0..maxC (minC is ignored) */
    kind = WNO_I_ADAPT_D;
    minC = 0;
  }
  /* forward search  adaptiveIntegrators.py:67-94 (D) / 363-389 (R2P) */
  int nEvalF = 0, nEvalB = 0;
  int If = maxC;
  double fnew = NAN, hh = h, maxErr, mad = NAN, Hlast = Ham0;
  for (int k = 0; k < d; ++k) vv0[k] = (double)xi * v[k];
  for (int c = minC; c <= maxC; ++c) {
    int fin = micro_pass(t, c, h, q, vv0, g, Ham0, qq, vv, gg, vh, &fnew, &hh, &maxErr, &mad,
                         &Hlast, &nEvalF);
    if (fin && maxErr < delta) {
      If = c;
      break;
    }
  }
  int cSim = If;
  double lwtf = 0.0;
  int maxTry, Ib;
  if (kind == WNO_I_ADAPT_R2P) {
    double u = unif(uctx); /* adaptiveIntegrators.py:392 */
    if (u < R2Pprob0) {    /* :393-399 */
      cSim = If;
    } else { /* :400-424 re-integrate at c = If+1 */
      micro_pass(t, If + 1, h, q, vv0, g, Ham0, qq, vv, gg, vh, &fnew, &hh, &maxErr, &mad, &Hlast,
                 &nEvalF);
      cSim = If + 1;
    }
    if (cSim == If) { /* :430-437 */
      maxTry = If - 1;
      Ib = If;
      lwtf = log(R2Pprob0);
    } else {
      maxTry = maxC;
      Ib = maxC;
      lwtf = log(1.0 - R2Pprob0);
    }
  } else { /* D: :104-111 backward tries c = minC..If-1 */
    maxTry = If - 1;
    Ib = If;
  }
  memcpy(qO, qq, sizeof(double) * d);
  memcpy(vO, vv, sizeof(double) * d);
  memcpy(gO, gg, sizeof(double) * d);
  double fOut = fnew;
  double igrConst = hh * pow(mad, -1.0 / 3.0); /* :101 / :399 / :424 */

  /* backward search  :107-132 (D) / :440-464 (R2P) */
  int do_back = (kind == WNO_I_ADAPT_D) ? (If > minC) : (maxTry >= minC);
  if (do_back) {
    double H0b = Hlast;
    double *nv = vv0; /* reuse: -vOut */
    for (int k = 0; k < d; ++k) nv[k] = -vO[k];
    for (int c = minC; c <= maxTry; ++c) {
      double f2, hh2_, me, md, hl;
      int fin = micro_pass(t, c, h, qO, nv, gO, H0b, qq, vv, gg, vh, &f2, &hh2_, &me, &md, &hl,
                           &nEvalB);
      if (fin && me < delta) {
        Ib = c;
        break;
      }
    }
  }
  for (int k = 0; k < d; ++k) {
    q1[k] = qO[k];
    v1[k] = (double)xi * vO[k];
    g1[k] = gO[k];
  }
  ret->lp = fOut;
  ret->nEvalF = nEvalF;
  ret->nEvalB = nEvalB;
  ret->If = If;
  ret->Ib = Ib;
  ret->igrConst = igrConst;
  if (kind == WNO_I_ADAPT_D) { /* :135-137 */
    ret->c = If;
    ret->lwt = (If != Ib) ? LOGZERO : 0.0;
  } else { /* :466-475 */
    double lwtb = LOGZERO;
    if (cSim == Ib) lwtb = log(R2Pprob0);
    else if (cSim == Ib + 1) lwtb = log(1.0 - R2Pprob0);
    ret->c = cSim;
    ret->lwt = lwtb - lwtf;
  }
}

static double const_unif(void *ctx) { return *(double *)ctx; }

int wno_integrator(int integrator, int target, int d, const double *tpar, const double *q,
                   const double *v, const double *g, double Ham0, double h, int xi, double delta,
                   int minC, int maxC, double R2Pprob0, double u_r2p, double *q1, double *v1,
                   double *g1, int *iout, double *dout) {
  lpctx t = {target, d, tpar};
  double *work = (double *)malloc(sizeof(double) * 8 * (size_t)d);
  if (!work) return -3;
  igr_ret r;
  integrator_impl(integrator, &t, q, v, g, Ham0, h, xi, delta, minC, maxC, R2Pprob0, const_unif,
                  &u_r2p, q1, v1, g1, &r, work);
  iout[0] = r.nEvalF; iout[1] = r.nEvalB; iout[2] = r.If; iout[3] = r.Ib; iout[4] = r.c;
  dout[0] = r.lp; dout[1] = r.lwt; dout[2] = r.igrConst;
  free(work);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * stateStore      WALNUTS.py:48-88
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int d3, n;
  double *stack; /* [n][d3] */
  int *id;
  int *used;
} state_store;

static int ss_push(state_store *s, int id, const double *q, const double *v, const double *g) {
  int d = s->d3 / 3;
  for (int i = 0; i < s->n; ++i) { /* :57-63 first free slot */
    if (!s->used[i]) {
      double *p = s->stack + (size_t)i * s->d3;
      memcpy(p, q, sizeof(double) * d);
      memcpy(p + d, v, sizeof(double) * d);
      memcpy(p + 2 * d, g, sizeof(double) * d);
      s->id[i] = id;
      s->used[i] = 1;
      return 0;
    }
  }
  return -1; /* :64-65 "stack full" */
}
static const double *ss_read(const state_store *s, int id) { /* :67-73 */
  for (int i = 0; i < s->n; ++i)
    if (s->id[i] == id && s->used[i]) return s->stack + (size_t)i * s->d3;
  return NULL;
}
static void ss_delete_range(state_store *s, int from, int to) { /* :75-76 */
  for (int i = 0; i < s->n; ++i)
    if (s->id[i] >= from && s->id[i] <= to) s->used[i] = 0;
}
static void ss_reset(state_store *s) { /* :79-80 */
  for (int i = 0; i < s->n; ++i) s->used[i] = 0;
}

/* ------------------------------------------------------------------------------------------
 * P2quantile      WALNUTSpy/P2quantile.py:16-86   (Jain & Chlamtac 1985)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  long npush;
  double p;
  double x[5]; /* first five values (:43-44) */
  double q[5]; /* zeros until the 5th push, then the sorted x (self.q = self.x aliases, :47-48) */
  long n[5];
  double npp[5];
} p2q;

static void p2_init(p2q *s, double prob) { /* :17-27 */
  s->npush = 0;
  s->p = prob;
  for (int i = 0; i < 5; ++i) { s->x[i] = 0.0; s->q[i] = 0.0; s->n[i] = i + 1; }
  s->npp[0] = 1.0; s->npp[1] = 1.0 + 2.0 * prob; s->npp[2] = 1.0 + 4.0 * prob;
  s->npp[3] = 3.0 + 2.0 * prob; s->npp[4] = 5.0;
}
static double p2_quantile(const p2q *s) { return s->q[2]; } /* :28-29 */

static double dsign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : (x == 0 ? 0.0 : NAN)); }

static void p2_push(p2q *s, double xi) { /* pushCore :41-86 */
  s->npush += 1;
  if (s->npush <= 5) s->x[s->npush - 1] = xi;
  if (s->npush == 5) { /* :46-48 np.sort, q = x */
    for (int a = 0; a < 5; ++a) s->q[a] = s->x[a];
    for (int a = 1; a < 5; ++a) {
      double v = s->q[a];
      int b = a - 1;
      /* np.sort puts NaN last */
      while (b >= 0 && (s->q[b] > v || (isnan(s->q[b]) && !isnan(v)))) { s->q[b + 1] = s->q[b]; --b; }
      s->q[b + 1] = v;
    }
  } else if (s->npush > 5) {
    int k; /* findInterval :31-38; returns None when no branch matches (xi == q[4] or NaN) */
    if (xi < s->q[0]) k = 0;
    else if (xi > s->q[4]) k = 5;
    else {
      k = -1;
      for (int i = 0; i < 4; ++i)
        if (xi < s->q[i + 1]) { k = i + 1; break; }
    }
    if (k == 0) { s->q[0] = xi; k = 1; }
    else if (k == 5) { s->q[4] = xi; k = 4; }
    /* self.n[k:5] += 1 ; k None -> n[None:5] = every marker (:61) */
    for (int i = (k < 0 ? 0 : k); i < 5; ++i) s->n[i] += 1;
    double nn = (double)s->npush, pp = s->p;
    s->npp[0] = 1.0;
    s->npp[1] = 0.5 * (nn - 1) * pp + 1.0;
    s->npp[2] = (nn - 1) * pp + 1.0;
    s->npp[3] = (nn - 1) * (1 + pp) / 2.0 + 1;
    s->npp[4] = nn;
    for (int i = 2; i < 5; ++i) { /* :71-86 */
      long ni = s->n[i - 1], nip = s->n[i], nim = s->n[i - 2];
      double di = s->npp[i - 1] - (double)ni;
      if ((di >= 1.0 && nip - ni > 1) || (di <= -1.0 && nim - ni < -1)) {
        long d = (long)dsign(di);
        double qi = s->q[i - 1];
        double qip = qi + ((double)d / (double)(nip - nim)) *
                              ((double)(ni - nim + d) * (s->q[i] - qi) / (double)(nip - ni) +
                               (double)(nip - ni - d) * (qi - s->q[i - 2]) / (double)(ni - nim));
        if (s->q[i - 2] < qip && qip < s->q[i]) s->q[i - 1] = qip;
        else s->q[i - 1] = qi + (double)d * (s->q[i + d - 1] - qi) / (double)(s->n[i + d - 1] - s->n[i - 1]);
        s->n[i - 1] += d;
      }
    }
  }
}

void wno_p2quantile(double prob, int n, const double *xs, double *q_after) {
  p2q s;
  p2_init(&s, prob);
  for (int i = 0; i < n; ++i) { p2_push(&s, xs[i]); q_after[i] = p2_quantile(&s); }
}

/* np.quantile(a[0:n], q), method='linear': virtual index (n-1)*q, neighbours floor/floor+1,
 * _lerp(a, b, t) = a + (b-a)*t, replaced by b - (b-a)*(1-t) where t >= 0.5; NaN if any NaN. */
static int cmp_double(const void *x, const void *y) {
  double a = *(const double *)x, b = *(const double *)y;
  if (isnan(a)) return isnan(b) ? 0 : 1;
  if (isnan(b)) return -1;
  return a < b ? -1 : (a > b ? 1 : 0);
}
double wno_np_quantile(const double *a, int n, double q) {
  double *t = (double *)malloc(sizeof(double) * (size_t)n);
  memcpy(t, a, sizeof(double) * (size_t)n);
  qsort(t, (size_t)n, sizeof(double), cmp_double);
  double res;
  if (isnan(t[n - 1])) {
    res = NAN;
  } else {
    double vi = (double)(n - 1) * q;
    long prev = (long)floor(vi), next = prev + 1;
    double gamma = vi - (double)prev;
    if (vi >= (double)(n - 1)) { prev = n - 1; next = n - 1; gamma = vi - (-1.0); }
    if (vi < 0) { prev = 0; next = 0; gamma = vi - 0.0; }
    double A = t[prev], B = t[next], d = B - A;
    res = (gamma >= 0.5) ? (B - d * (1 - gamma)) : (A + d * gamma);
  }
  free(t);
  return res;
}

/* ------------------------------------------------------------------------------------------
 * random stream as the sampler sees it (Q1: one global stream, data-dependent order)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const wno_config *cfg;
  uint32_t iter; /* 1-based iterN */
  uint32_t k;    /* uniforms consumed in this iteration */
  int overflow;
} rng_state;

static double next_unif(void *ctx) {
  rng_state *r = (rng_state *)ctx;
  const wno_config *c = r->cfg;
  double u;
  if (c->rng_mode == WNO_RNG_TAPE) {
    if ((int)r->k >= c->tape_u_stride) {
      r->overflow = 1;
      u = 0.5;
    } else {
      u = c->tape_u[(size_t)(r->iter - 1) * c->tape_u_stride + r->k];
    }
  } else {
    u = wno_rng_uniform(c->seed, c->chain, r->iter, r->k);
  }
  r->k++;
  return u;
}
static double next_normal(rng_state *r, int j) {
  const wno_config *c = r->cfg;
  if (c->rng_mode == WNO_RNG_TAPE) return c->tape_z[(size_t)(r->iter - 1) * c->d + j];
  return wno_rng_normal(c->seed, c->chain, r->iter, (uint32_t)j);
}
/* numpy legacy uniform(low, high) = low + (high-low)*U  */
static double unif_lh(rng_state *r, double low, double high) {
  return low + (high - low) * next_unif(r);
}

/* ------------------------------------------------------------------------------------------
 * generated(q)
 * ---------------------------------------------------------------------------------------- */
int wno_gen_dim(const wno_config *cfg) {
  switch (cfg->generated) {
    case WNO_G_IDENTITY: return cfg->d;
    case WNO_G_Q0_SUMSQ: return 2;
    case WNO_G_FIRSTK: return cfg->gen_k;
    case WNO_G_SW_CONSTRAIN: return (cfg->d % 3 == 0 && cfg->d >= 9) ? 2 * cfg->d + 1 : -1;
    default: return -1;
  }
}
static void generated(const wno_config *cfg, const double *q, double *out) {
  switch (cfg->generated) {
    case WNO_G_IDENTITY: memcpy(out, q, sizeof(double) * cfg->d); break;
    case WNO_G_Q0_SUMSQ: out[0] = q[0]; out[1] = seqdot(cfg->d, q, q); break; /* mainGaussESS.py:29-30 */
    case WNO_G_FIRSTK: memcpy(out, q, sizeof(double) * cfg->gen_k); break;
    case WNO_G_SW_CONSTRAIN: { /* mainSW.py:28-29 generatedInnov(q) = param_constrain(q, include_tp=True) for sw_innov.stan: the 3T
                                * unconstrained parameters, then the transformed parameters in declaration order -- sigma, gpri,
                                * z[T-1], x[T], tau[T] (sw_innov.stan:17-35), the recursions in the model's own order */
      const int T = cfg->d / 3;
      memcpy(out, q, sizeof(double) * cfg->d);
      const double tS = q[0], sigma = exp(-0.5 * tS), gpri = exp(tS);
      const double *zinn = q + 2, *xinn = q + T + 1, *tin = q + 2 * T + 1;
      double *o = out + 3 * T;
      o[0] = sigma; o[1] = gpri;
      double *z = o + 2, *x = z + (T - 1), *tau = x + T;
      z[0] = q[1];
      for (int t = 1; t < T - 1; ++t) z[t] = z[t - 1] + sigma * zinn[t - 1];
      x[0] = q[T];
      for (int t = 1; t < T; ++t) x[t] = x[t - 1] + sigma * xinn[t - 1];
      tau[0] = q[2 * T];
      for (int t = 1; t < T; ++t) tau[t] = tau[t - 1] + exp(0.5 * z[t - 1]) * tin[t - 1];
      break;
    }
  }
}

/* orbitMin/orbitMax update: np.minimum / np.maximum (nan-propagating), WALNUTS.py:332-333 etc. */
static void orbit_stats(const wno_config *cfg, int iterN, const double *q, int init) {
  if (!cfg->orbitMin) return;
  const int dg = wno_gen_dim(cfg);
  double tmp[1024];
  double *g = dg <= 1024 ? tmp : (double *)malloc(sizeof(double) * dg);
  generated(cfg, q, g);
  double *mn = cfg->orbitMin + (size_t)(iterN - 1) * dg, *mx = cfg->orbitMax + (size_t)(iterN - 1) * dg;
  for (int k = 0; k < dg; ++k) {
    if (init) { mn[k] = g[k]; mx[k] = g[k]; }
    else {
      mn[k] = (isnan(g[k]) || isnan(mn[k])) ? NAN : (g[k] < mn[k] ? g[k] : mn[k]);
      mx[k] = (isnan(g[k]) || isnan(mx[k])) ? NAN : (g[k] > mx[k] ? g[k] : mx[k]);
    }
  }
  if (g != tmp) free(g);
}

/* np.min / np.max over a slice (nan-propagating) */
static double np_max(const double *x, int lo, int hi) {
  double m = x[lo];
  for (int i = lo; i <= hi; ++i) {
    if (isnan(x[i])) return NAN;
    if (x[i] > m) m = x[i];
  }
  return m;
}
static double np_min(const double *x, int lo, int hi) {
  double m = x[lo];
  for (int i = lo; i <= hi; ++i) {
    if (isnan(x[i])) return NAN;
    if (x[i] < m) m = x[i];
  }
  return m;
}

/* ------------------------------------------------------------------------------------------
 * WALNUTS(...)          WALNUTSpy/WALNUTS.py:111-727, one chain
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const wno_config *cfg;
  lpctx t;
  rng_state rng;
  double *work;
  int ntrace;
  int iterN, idoubl;
  void *igrQ; /* p2q* */
  int warm;
} sampler_ctx;

static void call_integrator(sampler_ctx *S, const double *q, const double *v, const double *g,
                            double Ham0, double h, int xi, double delta, double *q1, double *v1,
                            double *g1, igr_ret *r) {
  const wno_config *c = S->cfg;
  integrator_impl(c->integrator, &S->t, q, v, g, Ham0, h, xi, delta, c->minC, c->maxC,
                  c->R2Pprob0, next_unif, &S->rng, q1, v1, g1, r, S->work);
  /* if(warmup and adaptH): igrConstQ.push(np.log(intOut.igrConst))  WALNUTS.py:313,347,411,454,498,541
     (right after Hs/Ifs/... are recorded and before the non-finite check; nothing in between reads it) */
  if (S->warm) p2_push((p2q *)S->igrQ, log(r->igrConst));
  if (c->trace && S->ntrace < c->trace_cap) {
    int d = c->d;
    double *row = c->trace + (size_t)S->ntrace * (WNO_TRACE_NSCAL + 2 * d);
    row[0] = S->iterN; row[1] = S->idoubl; row[2] = xi; row[3] = h; row[4] = Ham0;
    row[5] = r->lp; row[6] = r->If; row[7] = r->Ib; row[8] = r->c; row[9] = r->lwt;
    row[10] = r->nEvalF; row[11] = r->nEvalB;
    memcpy(row + WNO_TRACE_NSCAL, q1, sizeof(double) * d);
    memcpy(row + WNO_TRACE_NSCAL + d, v1, sizeof(double) * d);
  }
  S->ntrace++;
}

int wno_walnuts(const wno_config *cfg, const double *q0, double *samples, double *diag,
                int *n_u_used) {
  const int d = cfg->d, M = cfg->M, numIter = cfg->numIter;
  if (cfg->warmupIter < 0) return -10;
  /* :139-147 (the reference sys.exit()s on bad targets) */
  if (cfg->adaptH && (cfg->adaptHtarget < 0.0 || cfg->adaptHtarget > 1.0)) return -13;
  if (cfg->adaptDelta && cfg->adaptDeltaTarget < 0.0) return -14;
  p2q igrConstQ;
  p2_init(&igrConstQ, 1.0 - cfg->adaptHtarget); /* :141 */
  double *energyErrorInfFacs = (double *)calloc((size_t)(cfg->warmupIter > 0 ? cfg->warmupIter : 1), sizeof(double)); /* :147 */
  if (M < 1 || M > 24 || d < 1) return -11;
  const int dg = wno_gen_dim(cfg);
  if (dg < 1) return -12;
  const double wtSumThresh = wt_sum_thresh();
  double H = cfg->H0;        /* WALNUTS.py:137 */
  double delta = cfg->delta0; /* :144 */
  const double s = cfg->stepSizeRandScale;
  int rc = 0;

  /* plans[i] = subTreePlan(2**i)  :152-154 */
  int **plans = (int **)calloc(M, sizeof(int *));
  for (int i = 0; i < M; ++i) {
    int nleaf = 1 << i;
    plans[i] = (int *)malloc(sizeof(int) * 2 * (nleaf > 1 ? nleaf - 1 : 1));
    wno_subtreeplan(nleaf, plans[i]);
  }
  /* :167 stateStore(3*d, 2*(M+1)+1) */
  state_store st;
  st.d3 = 3 * d;
  st.n = 2 * (M + 1) + 1;
  st.stack = (double *)calloc((size_t)st.n * st.d3, sizeof(double));
  st.id = (int *)calloc(st.n, sizeof(int));
  st.used = (int *)calloc(st.n, sizeof(int));
  /* :170-174 */
  const int NH = 1 << M;
  double *Hs = (double *)calloc(NH, sizeof(double));
  int *Ifs = (int *)calloc(NH, sizeof(int));
  int *Ibs = (int *)calloc(NH, sizeof(int));
  int *cs = (int *)calloc(NH, sizeof(int));
  double *lwts = (double *)calloc(NH, sizeof(double));

  double *buf = (double *)calloc((size_t)16 * d, sizeof(double));
  double *qc = buf, *v = buf + d, *qp = buf + 2 * d, *vp = buf + 3 * d, *gp = buf + 4 * d;
  double *qm = buf + 5 * d, *vm = buf + 6 * d, *gm = buf + 7 * d, *qProp = buf + 8 * d;
  double *qPropLast = buf + 9 * d, *qtmp = buf + 10 * d, *vtmp = buf + 11 * d;
  double *q1 = buf + 12 * d, *v1 = buf + 13 * d, *g1 = buf + 14 * d;

  sampler_ctx S;
  S.cfg = cfg;
  S.t.target = cfg->target; S.t.d = d; S.t.tpar = cfg->tpar;
  S.rng.cfg = cfg; S.rng.iter = 0; S.rng.k = 0; S.rng.overflow = 0;
  S.work = (double *)calloc((size_t)8 * d, sizeof(double));
  S.ntrace = 0;
  S.igrQ = &igrConstQ;
  S.warm = 0;

  memcpy(qc, q0, sizeof(double) * d); /* :177 */
  generated(cfg, q0, samples);        /* :164 samples[:,0] = generated(q0) */

  for (int iterN = 1; iterN <= numIter; ++iterN) { /* :189 */
    S.rng.iter = (uint32_t)iterN;
    S.rng.k = 0;
    S.iterN = iterN;
    const int warmup = iterN <= cfg->warmupIter; /* :209 */
    S.warm = warmup && cfg->adaptH;          /* igrConstQ.push(...) at every integrator call :313 */
    int nevalF = 0, nevalB = 0, Lold_ = 0, L_ = 0;
    double orbitLen_ = 0.0, orbitLenSam_ = 0.0;
    int NdoublSampled_ = 0, NdoublComputed_ = 0;
    double indexStat_ = 0.0, indexStatOld_ = 0.0, timeLenF_ = 0.0, timeLenB_ = 0.0;

    /* :216 B = floor(uniform(0,2,size=M)) */
    int B[32];
    for (int i = 0; i < M; ++i) B[i] = (int)floor(unif_lh(&S.rng, 0.0, 2.0));
    int nleft = 0; /* :220 */
    for (int i = 0; i < M; ++i) nleft += B[i] * (1 << i);
    const int I0 = nleft; /* :223 */
    Ifs[I0] = 0; Ibs[I0] = 0; cs[I0] = 0; lwts[I0] = 0.0;
    int a = 0, b = 0, maxFint = 0, maxBint = 0; /* :230-233 */

    for (int j = 0; j < d; ++j) v[j] = next_normal(&S.rng, j); /* :236 */
    memcpy(qp, qc, sizeof(double) * d); memcpy(qm, qc, sizeof(double) * d); /* :239-242 */
    memcpy(vp, v, sizeof(double) * d);  memcpy(vm, v, sizeof(double) * d);
    memcpy(qProp, qc, sizeof(double) * d); memcpy(qPropLast, qc, sizeof(double) * d); /* :245-246 */
    double f0;
    wno_lpfun(cfg->target, d, cfg->tpar, qc, &f0, gp); /* :249 */
    memcpy(gm, gp, sizeof(double) * d);                /* :252-253 */
    Hs[I0] = -f0 + 0.5 * seqdot(d, v, v);              /* :256 */
    const double multinomialLscale = Hs[I0];           /* :259 */
    double WoldSum = 1.0, WnewSum = 0.0;
    double lwtSumb = 0.0, lwtSumf = 0.0; /* :263-264 */
    int forcedReject = 0, bothEndsPassive = 0, stopCode = 0;
    orbit_stats(cfg, iterN, qc, 1); /* :274-276 */

    for (int i = 0; i < M; ++i) { /* :281 */
      S.idoubl = i;
      int xi = B[i] ? -1 : 1;      /* :283 (-1)**B[i] */
      int at = a + xi * (1 << i);  /* :285-286 */
      int bt = b + xi * (1 << i);
      int expandFurther = 1;
      memcpy(qPropLast, qProp, sizeof(double) * d); /* :292 */
      Lold_ = L_;
      indexStatOld_ = indexStat_;
      igr_ret r;

      if (i == 0) { /* :296 */
        double HLoc = unif_lh(&S.rng, H * (1 - s), H * (1 + s)); /* :298 */
        orbitLen_ += HLoc;                                      /* :300 */
        if (xi == 1) { /* :301-328 */
          call_integrator(&S, qp, vp, gp, Hs[I0], HLoc, xi, delta, q1, v1, g1, &r);
          memcpy(qp, q1, sizeof(double) * d); memcpy(vp, v1, sizeof(double) * d);
          memcpy(gp, g1, sizeof(double) * d);
          nevalF += r.nEvalF; nevalB += r.nEvalB;
          Hs[I0 + 1] = -r.lp + 0.5 * seqdot(d, vp, vp);
          Ifs[I0 + 1] = r.If; Ibs[I0 + 1] = r.Ib; cs[I0 + 1] = r.c; lwts[I0 + 1] = r.lwt;
          maxFint = 1;
          timeLenF_ = HLoc;
          if (!isfinite(Hs[I0 + 1])) { forcedReject = 1; stopCode = 999; break; } /* :316-319 */
          lwtSumf = lwts[I0 + 1];
          double Wnew = exp(-Hs[I0 + 1] + multinomialLscale + lwtSumf);
          memcpy(qProp, qp, sizeof(double) * d); /* :326 */
          L_ = 1;
          indexStat_ = timeLenF_;
          orbit_stats(cfg, iterN, qp, 0); /* :331-333 */
          WoldSum = 1.0; WnewSum = Wnew; /* :368-369 */
        } else { /* :335-361 */
          call_integrator(&S, qm, vm, gm, Hs[I0], HLoc, xi, delta, q1, v1, g1, &r);
          memcpy(qm, q1, sizeof(double) * d); memcpy(vm, v1, sizeof(double) * d);
          memcpy(gm, g1, sizeof(double) * d);
          nevalF += r.nEvalF; nevalB += r.nEvalB;
          Hs[I0 - 1] = -r.lp + 0.5 * seqdot(d, vm, vm);
          Ifs[I0 - 1] = r.If; Ibs[I0 - 1] = r.Ib; cs[I0 - 1] = r.c; lwts[I0 - 1] = r.lwt;
          maxBint = -1;
          timeLenB_ = HLoc;
          if (!isfinite(Hs[I0 - 1])) { forcedReject = 1; stopCode = 999; break; }
          lwtSumb = lwts[I0 - 1];
          double Wnew = exp(-Hs[I0 - 1] + multinomialLscale + lwtSumb);
          memcpy(qProp, qm, sizeof(double) * d);
          L_ = -1;
          indexStat_ = -timeLenB_;
          orbit_stats(cfg, iterN, qm, 0); /* :364-366 */
          WoldSum = 1.0; WnewSum = Wnew;
        }
      } else { /* :381 */
        const int nrow = (1 << i) - 1;
        const int *pl = plans[i];
        WnewSum = 0.0; /* :390 */
        for (int j = 0; j < nrow; ++j) { /* :392 */
          int p0, p1;
          if (xi == 1) { p0 = b + pl[2 * j]; p1 = b + pl[2 * j + 1]; } /* :385 */
          else { p0 = a - pl[2 * j]; p1 = a - pl[2 * j + 1]; }         /* :387 */
          if (abs(p0 - p1) == 1) { /* :394 */
            double HLoc1[2];
            HLoc1[0] = unif_lh(&S.rng, H * (1 - s), H * (1 + s)); /* :395 size=2 */
            HLoc1[1] = unif_lh(&S.rng, H * (1 - s), H * (1 + s));
            if (xi == -1) { /* :398-481 backward */
              int i1 = p0;
              call_integrator(&S, qm, vm, gm, Hs[I0 + i1 + 1], HLoc1[0], xi, delta, q1, v1, g1, &r);
              memcpy(qm, q1, sizeof(double) * d); memcpy(vm, v1, sizeof(double) * d);
              memcpy(gm, g1, sizeof(double) * d);
              nevalF += r.nEvalF; nevalB += r.nEvalB;
              Hs[I0 + i1] = -r.lp + 0.5 * seqdot(d, vm, vm);
              Ifs[I0 + i1] = r.If; Ibs[I0 + i1] = r.Ib; cs[I0 + i1] = r.c; lwts[I0 + i1] = r.lwt;
              maxBint = i1;
              timeLenB_ += HLoc1[0];
              if (!isfinite(Hs[I0 + i1])) { forcedReject = 1; stopCode = 999; break; }
              lwtSumb += lwts[I0 + i1]; /* :420 */
              double Wnew = exp(-Hs[I0 + i1] + multinomialLscale + lwtSumb);
              WnewSum += Wnew;
              if (WnewSum > wtSumThresh && next_unif(&S.rng) < Wnew / WnewSum) { /* :426 */
                memcpy(qProp, qm, sizeof(double) * d);
                L_ = i1;
                indexStat_ = -timeLenB_;
              }
              if (ss_push(&st, i1, qm, vm, gm)) { rc = -20; goto done; } /* :431 */
              orbitLen_ += HLoc1[0];
              orbit_stats(cfg, iterN, qm, 0); /* :434-436 */
              memcpy(qtmp, qm, sizeof(double) * d); /* :438-439 */
              memcpy(vtmp, vm, sizeof(double) * d);
              int i2 = p1; /* :442 */
              call_integrator(&S, qm, vm, gm, Hs[I0 + i2 + 1], HLoc1[1], xi, delta, q1, v1, g1, &r);
              memcpy(qm, q1, sizeof(double) * d); memcpy(vm, v1, sizeof(double) * d);
              memcpy(gm, g1, sizeof(double) * d);
              nevalF += r.nEvalF; nevalB += r.nEvalB;
              Hs[I0 + i2] = -r.lp + 0.5 * seqdot(d, vm, vm);
              Ifs[I0 + i2] = r.If; Ibs[I0 + i2] = r.Ib; cs[I0 + i2] = r.c; lwts[I0 + i2] = r.lwt;
              maxBint = i2;
              timeLenB_ += HLoc1[1];
              if (!isfinite(Hs[I0 + i2])) { forcedReject = 1; break; } /* :457-459 (Q4) */
              /* Q2: lwtSumb is NOT updated with lwts[I0+i2] here (:461-463) */
              Wnew = exp(-Hs[I0 + i2] + multinomialLscale + lwtSumb);
              WnewSum += Wnew;
              if (WnewSum > wtSumThresh && next_unif(&S.rng) < Wnew / WnewSum) {
                memcpy(qProp, qm, sizeof(double) * d);
                L_ = i2;
                indexStat_ = -timeLenB_;
              }
              if (ss_push(&st, i2, qm, vm, gm)) { rc = -20; goto done; } /* :470 */
              orbitLen_ += HLoc1[1];
              orbit_stats(cfg, iterN, qm, 0); /* :474-476 */
              if (wno_stopcondition(d, qm, vm, qtmp, vtmp)) { expandFurther = 0; break; } /* :479 */
            } else { /* :485-570 forward */
              int i1 = p0;
              call_integrator(&S, qp, vp, gp, Hs[I0 + i1 - 1], HLoc1[0], xi, delta, q1, v1, g1, &r);
              memcpy(qp, q1, sizeof(double) * d); memcpy(vp, v1, sizeof(double) * d);
              memcpy(gp, g1, sizeof(double) * d);
              nevalF += r.nEvalF; nevalB += r.nEvalB;
              Hs[I0 + i1] = -r.lp + 0.5 * seqdot(d, vp, vp);
              Ifs[I0 + i1] = r.If; Ibs[I0 + i1] = r.Ib; cs[I0 + i1] = r.c; lwts[I0 + i1] = r.lwt;
              maxFint = i1;
              timeLenF_ += HLoc1[0];
              if (!isfinite(Hs[I0 + i1])) { forcedReject = 1; stopCode = 999; break; }
              lwtSumf += lwts[I0 + i1]; /* :507 */
              double Wnew = exp(-Hs[I0 + i1] + multinomialLscale + lwtSumf);
              WnewSum += Wnew;
              if (WnewSum > wtSumThresh && next_unif(&S.rng) < Wnew / WnewSum) { /* :512 */
                memcpy(qProp, qp, sizeof(double) * d);
                L_ = i1;
                indexStat_ = timeLenF_;
              }
              if (ss_push(&st, i1, qp, vp, gp)) { rc = -20; goto done; } /* :518 */
              orbitLen_ += HLoc1[0];
              orbit_stats(cfg, iterN, qp, 0); /* :521-523 */
              memcpy(qtmp, qp, sizeof(double) * d); /* :525-526 */
              memcpy(vtmp, vp, sizeof(double) * d);
              int i2 = p1; /* :529 */
              call_integrator(&S, qp, vp, gp, Hs[I0 + i2 - 1], HLoc1[1], xi, delta, q1, v1, g1, &r);
              memcpy(qp, q1, sizeof(double) * d); memcpy(vp, v1, sizeof(double) * d);
              memcpy(gp, g1, sizeof(double) * d);
              nevalF += r.nEvalF; nevalB += r.nEvalB;
              Hs[I0 + i2] = -r.lp + 0.5 * seqdot(d, vp, vp);
              Ifs[I0 + i2] = r.If; Ibs[I0 + i2] = r.Ib; cs[I0 + i2] = r.c; lwts[I0 + i2] = r.lwt;
              maxFint = i2;
              timeLenF_ += HLoc1[1];
              if (!isfinite(Hs[I0 + i2])) { forcedReject = 1; break; } /* :544-546 */
              lwtSumf += lwts[I0 + i2];                                 /* :550 */
              Wnew = exp(-Hs[I0 + i2] + multinomialLscale + lwtSumf);
              WnewSum += Wnew;
              if (WnewSum > wtSumThresh && next_unif(&S.rng) < Wnew / WnewSum) {
                memcpy(qProp, qp, sizeof(double) * d);
                L_ = i2;
                indexStat_ = timeLenF_;
              }
              if (ss_push(&st, i2, qp, vp, gp)) { rc = -20; goto done; } /* :560 */
              orbitLen_ += HLoc1[1];
              orbit_stats(cfg, iterN, qp, 0); /* :564-566 */
              if (wno_stopcondition(d, qtmp, vtmp, qp, vp)) { expandFurther = 0; break; } /* :568 */
            }
          } else { /* :572-587 inner row */
            int im = p0 < p1 ? p0 : p1;
            int ip = p0 < p1 ? p1 : p0;
            ss_delete_range(&st, im + 1, ip - 1);
            const double *statep = ss_read(&st, ip);
            const double *statem = ss_read(&st, im);
            if (!statep || !statem) { rc = -21; goto done; }
            if (wno_stopcondition(d, statem, statem + d, statep, statep + d)) {
              expandFurther = 0;
              break;
            }
          }
        } /* rows */
      }

      if (forcedReject) break; /* :590-592 */

      indexStat_ = indexStat_ / (timeLenF_ + timeLenB_); /* :595 (Q6) */

      if (!expandFurther) { /* :597-605 */
        memcpy(qProp, qPropLast, sizeof(double) * d);
        L_ = Lold_;
        indexStat_ = indexStatOld_;
        NdoublSampled_ = i;
        NdoublComputed_ = i + 1;
        stopCode = 5;
        break;
      } else {
        if (!(next_unif(&S.rng) < WnewSum / WoldSum)) { /* :613-616 */
          L_ = Lold_;
          indexStat_ = indexStatOld_;
          memcpy(qProp, qPropLast, sizeof(double) * d);
        }
        int joinedCrit = wno_stopcondition(d, qm, vm, qp, vp);                    /* :622 */
        bothEndsPassive = (lwtSumb < LOGZERO + 1.0) && (lwtSumf < LOGZERO + 1.0); /* :624 */
        if (joinedCrit || bothEndsPassive) {
          stopCode = joinedCrit ? 4 : -4;
          NdoublSampled_ = i + 1;
          NdoublComputed_ = i + 1;
          orbitLenSam_ = orbitLen_;
          break;
        }
      }
      WoldSum += WnewSum; /* :641-648 */
      orbitLenSam_ = orbitLen_;
      NdoublSampled_ = i + 1;
      NdoublComputed_ = i + 1;
      a = a < at ? a : at;
      b = b > bt ? b : bt;
      ss_reset(&st);
    } /* doublings */
    /* NOTE: the reference never calls stateReset() after a break, so stale ids can survive
       into the next iteration's store; ids are re-pushed before being read, and stateRead
       returns the first *used* match, so restate that literally: reset only where it does. */

    memcpy(qc, qProp, sizeof(double) * d); /* :653 */

    /* diagnostics :660-693.  usedSteps = maxBint..-1, 1..maxFint */
    double *drow = diag + (size_t)(iterN - 1) * WNO_NDIAG;
    int lo = I0 + maxBint, hi = I0 + maxFint;
    double orbitEnergyError = np_max(Hs, lo, hi) - np_min(Hs, lo, hi);
    int minIf = 1 << 30, maxIf = -(1 << 30), minc = 1 << 30, maxc = -(1 << 30);
    double minlw = INFINITY, maxlw = -INFINITY;
    int nused = 0, nneq = 0, nzero = 0, lwnan = 0;
    for (int idx = lo; idx <= hi; ++idx) {
      if (idx == I0) continue;
      nused++;
      if (Ifs[idx] < minIf) minIf = Ifs[idx];
      if (Ifs[idx] > maxIf) maxIf = Ifs[idx];
      if (cs[idx] < minc) minc = cs[idx];
      if (cs[idx] > maxc) maxc = cs[idx];
      if (isnan(lwts[idx])) lwnan = 1;
      if (lwts[idx] < minlw) minlw = lwts[idx];
      if (lwts[idx] > maxlw) maxlw = lwts[idx];
      nneq += (Ifs[idx] != Ibs[idx]);
      nzero += (Ifs[idx] == 0);
    }
    if (lwnan) minlw = maxlw = NAN;
    drow[0] = L_;
    drow[1] = NdoublSampled_;
    drow[2] = orbitLen_;
    drow[3] = orbitLenSam_;
    drow[4] = maxFint;
    drow[5] = maxBint;
    drow[6] = nevalF;
    drow[7] = nevalB;
    drow[8] = minIf;
    drow[9] = maxIf;
    drow[10] = minlw;
    drow[11] = maxlw;
    drow[12] = 1.0 * bothEndsPassive;
    drow[13] = 1.0 * ((lwtSumb < LOGZERO + 1.0) || (lwtSumf < LOGZERO + 1.0));
    drow[14] = (double)nneq / (double)nused;
    drow[15] = H;
    drow[16] = (double)nzero / (double)nused;
    drow[17] = orbitEnergyError;
    drow[18] = delta;
    drow[19] = 1.0 * stopCode;
    drow[20] = NdoublComputed_;
    drow[21] = minc;
    drow[22] = maxc;
    drow[23] = indexStat_;

    generated(cfg, qc, samples + (size_t)iterN * dg); /* :695 */
    if (n_u_used) n_u_used[iterN - 1] = (int)S.rng.k;

    if (warmup) { /* :701-712 */
      if (cfg->adaptDelta) energyErrorInfFacs[iterN - 1] = orbitEnergyError / delta;
      if (cfg->adaptDelta && iterN > 10)
        delta = quant_bits(cfg->adaptDeltaTarget / wno_np_quantile(energyErrorInfFacs, iterN, cfg->adaptDeltaQuantile), cfg->adaptQuantBits);
      if (cfg->adaptH && igrConstQ.npush > 10)
        H = quant_bits(pow(delta, 1.0 / 3.0) * exp(p2_quantile(&igrConstQ)), cfg->adaptQuantBits);
    }
  }
  if (S.rng.overflow) rc = -30;

done:
  for (int i = 0; i < M; ++i) free(plans[i]);
  free(plans);
  free(st.stack); free(st.id); free(st.used);
  free(Hs); free(Ifs); free(Ibs); free(cs); free(lwts);
  free(buf);
  free(S.work);
  free(energyErrorInfFacs);
  return rc < 0 ? rc : S.ntrace;
}
