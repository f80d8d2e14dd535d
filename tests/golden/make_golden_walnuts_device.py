#!/usr/bin/env python3
"""Golden transitions of walnuts_step() (walnuts/walnuts.py:279-359) ON THE ENGINE'S RANDOM STREAM: pins `walnuts_device`
(wn_wstep_run_device_rng: wstep_wave_kernel and wstep_auto_kernel, wn_k_wstep.hip) to the reference.

The UNMODIFIED reference walnuts() is run in this container with a Generator stand-in whose normal / binomial / choice /
uniform serve the engine's Philox stream (wn_rng.h; here through the oracle's independent restatement wno_rng_*) in the
order the device kernels consume it -- the tape trick of make_golden.py, for walnuts_step:
    rng.normal(size=D)             -> rng_normal(seed, chain, it, j), j < D; starts transition `it` (k = 0)       (:325)
    rng.binomial(1, 0.5)           -> 1 if U(k++) < 0.5 else 0                                                    (:330)
    rng.choice([e//2, e, 2e])      -> a[min(int(3 U(k++)), 2)]                                                    (:194)
    rng.uniform(0, 1)              -> U(k++)                                                                      (:346)
    rng.choice(orbit_ext, p=p)     -> numpy's own inversion: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(U(k++), 'right')  (:350)
with U(k) = rng_uniform(seed, chain, it, k).  The stand-in also LOGS what the reference did with each draw, which gives the integer
columns a device run must reproduce exactly: doublings started, accepts, sum of the chosen micro-step counts, sum of the forward
`ell_stable`, uniforms consumed.

Run:  PYTHONDONTWRITEBYTECODE=1 python3 -B tests/golden/make_golden_walnuts_device.py
Writes tests/golden/walnuts_device.npz (inputs + outputs only).
"""
import importlib.util
import os
import sys
import time

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402  (the engine's RNG restated on the CPU)

spec = importlib.util.spec_from_file_location("ref_walnuts", "/root/reference/walnuts/walnuts.py")
ref = importlib.util.module_from_spec(spec)
import tqdm  # noqa: F401,E402
spec.loader.exec_module(ref)
ref.trange = lambda n, **k: range(n)


class EngineStream:
    """numpy.random.Generator stand-in: the engine's Philox stream of one chain, transition by transition."""

    def __init__(self, seed, chain, it0):
        self.seed, self.chain, self.it, self.k = int(seed), int(chain), int(it0) - 1, 0
        self.log = []  # per transition: [doublings, accepts, sum ell, sum ell_stable (forward searches), uniforms]

    def _u(self):
        u = ol.lib().wno_rng_uniform(self.seed, self.chain, self.it, self.k)
        self.k += 1
        self.log[-1][4] += 1
        return u

    def normal(self, size=None):
        self.it += 1
        self.k = 0
        self.log.append([0, 0, 0, 0, 0])
        return np.array([ol.lib().wno_rng_normal(self.seed, self.chain, self.it, j) for j in range(int(size))])

    def binomial(self, n, p):
        assert n == 1 and p == 0.5
        self.log[-1][0] += 1
        return 1 if self._u() < 0.5 else 0

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * self._u()

    def choice(self, a, p=None):
        u = self._u()
        if p is None:  # choose_micro_steps: uniform over the three candidates
            idx = min(int(3.0 * u), 2)
            self.log[-1][2] += int(a[idx])
            self.log[-1][3] += int(a[1])
            return np.asarray(a)[idx]  # (np.int64 as Generator.choice returns: ell = 0 -- Q9, ell_stable = 1 halved -- then divides to inf, not ZeroDivisionError)
        self.log[-1][1] += 1
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return a[int(cdf.searchsorted(u, side="right"))]


def run_case(name, out, seed, C, T, D, inv_mass, macro_step, depth, max_error, prec=None, it0=1, scale=1.0):
    rng0 = np.random.default_rng(seed)
    theta0 = scale * rng0.normal(size=(C, D))
    if prec is None:
        lpdf = lambda q: -0.5 * np.dot(q, q)
        grad = lambda q: -q
    else:
        lpdf = lambda q: 0.5 * np.dot(q, -(prec * q))  # the engine's diagonal Gaussian: g = -(a .* q), lp = 1/2 q.g
        grad = lambda q: -(prec * q)
    draws = np.empty((C, T, D))
    ints = np.empty((C, T, 5), dtype=np.int64)
    t0 = time.time()
    for c in range(C):
        es = EngineStream(seed, c, it0)
        draws[c] = ref.walnuts(es, theta0[c], lpdf, grad, inv_mass, macro_step, depth, max_error, 0, T)
        ints[c] = np.array(es.log)
    out[name + "_cfg"] = np.array([seed, C, T, D, depth, it0], dtype=np.int64)
    out[name + "_par"] = np.array([macro_step, max_error])
    out[name + "_inv_mass"] = np.asarray(inv_mass, dtype=np.float64)
    out[name + "_prec"] = np.ones(D) if prec is None else np.asarray(prec, dtype=np.float64)
    out[name + "_theta0"] = theta0
    out[name + "_draws"] = draws
    out[name + "_ints"] = ints
    print("%s: %d chains x %d transitions in %.1f s; doublings mean %.2f, micro steps per transition mean %.0f max %d" %
          (name, C, T, time.time() - t0, ints[:, :, 0].mean(), ints[:, :, 2].mean(), ints[:, :, 2].max()))


if __name__ == "__main__":
    out = {}
    # BASELINE configs[0]'s target and test/test.py's parameters (macro_step 2.0, max depth 10, max_error 0.1), 10-D standard normal
    run_case("std10", out, 20251004, 128, 30, 10, np.ones(10), 2.0, 10, 0.1)
    # a diagonal mass matrix (inverse masses 0.25 .. 4) on the same target
    run_case("std10_mass", out, 77, 32, 20, 10, np.geomspace(0.25, 4.0, 10), 2.0, 10, 0.1)
    # a diagonal-precision Gaussian (T_DIAGGAUSS), D = 24 > 16: the wave kernel's full-wave sums
    run_case("diag24", out, 5, 16, 12, 24, np.ones(24), 1.5, 8, 0.2, prec=np.geomspace(0.5, 8.0, 24), it0=7)
    np.savez_compressed(os.path.join(HERE, "walnuts_device.npz"), **out)
