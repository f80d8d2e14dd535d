// wn_targets.h -- device forms of the reference's target plugins lpFun(q) -> [lp, grad]
// (WALNUTSpy/targetDistr.py) for the one-chain-per-lane layout.  Each returns lp and fills g.
// Gaussian-family targets for the cooperative MFMA layout live in wn_policy_coop.h.
#pragma once
#include "wn_chain.h"

namespace wn {

// lp = -1/2 sum a_i q_i^2, grad = -a_i q_i.  stdGauss (targetDistr.py:18-21) is a == 1.
template <int DMAX>
struct TgtDiag {
  static constexpr int kind = T_DIAGGAUSS;
  static WN_HD double eval(const double* q, double* g, int D, const double* tpar) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < DMAX; ++e) {
      if (e < D) {
        g[e] = -(tpar[e] * q[e]);
        s += q[e] * g[e];
      } else {
        g[e] = 0.0;
      }
    }
    return 0.5 * s;
  }
};

// lp = -1/2 q'Aq, grad = -Aq, A row-major D*D, read through scalar loads
template <int DMAX>
struct TgtDense {
  static constexpr int kind = T_DENSEGAUSS;
  static WN_HD double eval(const double* q, double* g, int D, const double* tpar) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
      double a = 0.0;
      if (i < D) {
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
          if (j < D) a += tpar[i * D + j] * q[j];
      }
      g[i] = -a;
      s += q[i] * g[i];
    }
    return 0.5 * s;
  }
};

// funnel10 (targetDistr.py:74-78): q0 ~ N(0, 3^2), q_i | q0 ~ N(0, e^{q0}), i = 1..10, with the
// normalising constants scipy.stats.norm.logpdf includes.
struct TgtFunnel10 {
  static constexpr int kind = T_FUNNEL10;
  static WN_HD double eval(const double* q, double* g, int, const double*) {
    const double logC = 0.91893853320467274178;  // log(sqrt(2 pi))
    const double log3 = 1.09861228866810969140;
    const double emq = exp(-q[0]);
    double ss = 0.0;
#pragma unroll
    for (int i = 1; i < 11; ++i) ss += q[i] * q[i];
    g[0] = -5.0 - q[0] / 9.0 + 0.5 * emq * ss;
#pragma unroll
    for (int i = 1; i < 11; ++i) g[i] = -q[i] * emq;
    // sum_i [-(q_i/s)^2/2 - logC - log s], s = e^{q0/2}.  The reference evaluates log(s) with
    // s = np.exp(0.5*q0) (targetDistr.py:75): when that overflows its lp is -inf and the sampler
    // force-rejects (WALNUTS.py:316); keep that behaviour.
    const double lp = (-(q[0] * q[0]) / 18.0 - logC - log3) + (-0.5 * emq * ss - 10.0 * logC - 5.0 * q[0]);
    return (0.5 * q[0] > 709.782712893384) ? -INFINITY : lp;
  }
};

struct TgtCorrGauss {  // targetDistr.py:25-31
  static constexpr int kind = T_CORRGAUSS;
  static WN_HD double eval(const double* q, double* g, int, const double*) {
    const double rho = 0.5, tmp = 1.0 - rho * rho;
    double r = q[1] - rho * q[0];
    g[0] = -(q[0] - rho * q[1]) / tmp;
    g[1] = -(q[1] - rho * q[0]) / tmp;
    return -0.5 * (q[0] * q[0]) - (0.5 / tmp) * (r * r);
  }
};

struct TgtSmile {  // targetDistr.py:34-38
  static constexpr int kind = T_SMILE;
  static WN_HD double eval(const double* q, double* g, int, const double*) {
    double q0s = q[0] * q[0], r = q[1] - q0s;
    g[0] = -q[0] + 2.0 * q[0] * q[1] - 2.0 * (q0s * q[0]);
    g[1] = q0s - q[1];
    return -0.5 * q0s - 0.5 * (r * r);
  }
};

struct TgtFunnel1 {  // targetDistr.py:88-93
  static constexpr int kind = T_FUNNEL1;
  static WN_HD double eval(const double* q, double* g, int, const double*) {
    const double logC = 0.91893853320467274178, log3 = 1.09861228866810969140;
    const double emq = exp(-q[0]);
    g[0] = -0.5 - q[0] / 9.0 + 0.5 * (q[1] * q[1]) * emq;
    g[1] = -q[1] * emq;
    const double lp = (-(q[0] * q[0]) / 18.0 - logC - log3) + (-0.5 * (q[1] * q[1]) * emq - logC - 0.5 * q[0]);
    return (0.5 * q[0] > 709.782712893384) ? -INFINITY : lp;  // log(np.exp(0.5*q0)) overflow, as above
  }
};

struct TgtModFunnel {  // targetDistr.py:41-51
  static constexpr int kind = T_MODFUNNEL;
  static WN_HD double eval(const double* q, double* g, int, const double*) {
    double x = q[0], y = q[1];
    double t1 = exp(-3.0 * x), t2 = 1.0 + t1, t3 = 1.0 / t2, t4 = y * y;
    g[0] = 1.5 * t1 * (t4 - t3) - x;
    g[1] = -y * t2;
    return -0.5 * (t2 * t4 + log(t3) + x * x);
  }
};

}  // namespace wn
