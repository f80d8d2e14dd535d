// wn_target_eval.h -- scalar lpFun(q) -> lp, grad of every built-in target (one thread, plain loops): the plumbing
// entry points wn_lpfun / wn_lpfun_kind and the walnuts.py-variant step kernels (wn_k_wstep.hip).  The samplers of
// WALNUTS.py evaluate their targets inside the fused kernels (wn_targets.h, wn_policy_*.h).
#pragma once
#include <math.h>

#include "wn_chain.h"

namespace wn {

// lpFun(q) -> lp, grad for any target kind, scalar loops (test / plumbing entry points; the
// samplers evaluate their targets inside the fused kernels)
enum { T_TT_CORRNORMAL = 8, T_TT_FUNNEL = 9 };
// (XP / GP: pointer types of the position and the gradient -- plain `double*`, or LDS-address-space pointers: through a generic pointer
// to LDS every access is a FLAT instruction)
template <class XP, class GP>
__device__ inline double wn_target_eval_p(int kind, int D, const double* tpar, XP x, GP g) {
  const double logC = 0.91893853320467274178, log3 = 1.09861228866810969140;
  double r = 0.0;
  switch (kind) {
    case T_STDGAUSS:
      for (int i = 0; i < D; ++i) { g[i] = -x[i]; r += x[i] * x[i]; }
      r = -0.5 * r;
      break;
    case T_DIAGGAUSS:
      for (int i = 0; i < D; ++i) { g[i] = -(tpar[i] * x[i]); r += x[i] * g[i]; }
      r = 0.5 * r;
      break;
    case T_DENSEGAUSS:
      for (int i = 0; i < D; ++i) {
        double a = 0.0;
        for (int j = 0; j < D; ++j) a += tpar[(size_t)i * D + j] * x[j];
        g[i] = -a;
        r += x[i] * g[i];
      }
      r = 0.5 * r;
      break;
    case T_FUNNEL10: {
      double emq = exp(-x[0]), ss = 0.0;
      for (int i = 1; i < 11; ++i) ss += x[i] * x[i];
      g[0] = -5.0 - x[0] / 9.0 + 0.5 * emq * ss;
      for (int i = 1; i < 11; ++i) g[i] = -x[i] * emq;
      r = (-(x[0] * x[0]) / 18.0 - logC - log3) + (-0.5 * emq * ss - 10.0 * logC - 5.0 * x[0]);
      r = (0.5 * x[0] > 709.782712893384) ? -INFINITY : r;
    } break;
    case T_CORRGAUSS: {
      const double rho = 0.5, tmp = 1.0 - rho * rho;
      double rr = x[1] - rho * x[0];
      g[0] = -(x[0] - rho * x[1]) / tmp; g[1] = -(x[1] - rho * x[0]) / tmp;
      r = -0.5 * (x[0] * x[0]) - (0.5 / tmp) * (rr * rr);
    } break;
    case T_SMILE: {
      double q0s = x[0] * x[0], rr = x[1] - q0s;
      g[0] = -x[0] + 2.0 * x[0] * x[1] - 2.0 * (q0s * x[0]); g[1] = q0s - x[1];
      r = -0.5 * q0s - 0.5 * (rr * rr);
    } break;
    case T_FUNNEL1: {
      double emq = exp(-x[0]);
      g[0] = -0.5 - x[0] / 9.0 + 0.5 * (x[1] * x[1]) * emq; g[1] = -x[1] * emq;
      r = (-(x[0] * x[0]) / 18.0 - logC - log3) + (-0.5 * (x[1] * x[1]) * emq - logC - 0.5 * x[0]);
      r = (0.5 * x[0] > 709.782712893384) ? -INFINITY : r;
    } break;
    case T_MODFUNNEL: {
      double t1 = exp(-3.0 * x[0]), t2 = 1.0 + t1, t3 = 1.0 / t2, t4 = x[1] * x[1];
      g[0] = 1.5 * t1 * (t4 - t3) - x[0]; g[1] = -x[1] * t2;
      r = -0.5 * (t2 * t4 + log(t3) + x[0] * x[0]);
    } break;
    case T_TT_CORRNORMAL: {  // test/targets.py:9-15
      const double rho = 0.5;
      double rr = x[1] - rho * x[0];
      g[0] = -x[0] + rho * x[1];
      g[1] = (-x[1] + rho * x[0]) / (1.0 - rho * rho);
      r = -0.5 * (x[0] * x[0]) - 0.5 / (1.0 - rho * rho) * (rr * rr);
    } break;
    case T_TT_FUNNEL: {  // test/targets.py:23-29
      double ss = 0.0;
      for (int i = 1; i < D; ++i) ss += x[i] * x[i];
      const double e = exp(0.5 * x[0]);
      g[0] = -(x[0] / 9.0 - 0.25 * ss / e);
      for (int i = 1; i < D; ++i) g[i] = -1.0 / e * x[i];
      r = -0.5 * (x[0] * x[0]) / 9.0 - 0.5 * ss / e;
    } break;
  }
  return r;
}
__device__ inline double wn_target_eval(int kind, int D, const double* tpar, const double* x, double* g) {
  return wn_target_eval_p<const double*, double*>(kind, D, tpar, x, g);
}

}  // namespace wn
