#!/usr/bin/env python3
"""Golden fixture: the UNMODIFIED Python reference sampler at BASELINE config 5's size (D = 756, minC = 3, M = 14,
WALNUTSpy_examples/StockWatson/mainSW.py:41-80) driven by a NumPy restatement of sw_innov.stan as lpFun.

bridgestan / Stan are absent here, so the TARGET stays unpinned against Stan; what this pins is the
SAMPLER at D = 756 against the reference itself: the reference's WALNUTS() + adaptLeapFrogD / adaptLeapFrogR2P /
fixedLeapFrog run a few iterations on `sw_lpfun` below with the engine's random stream served through the tape of
make_golden.py, and inputs + outputs are stored.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 -B tests/golden/make_golden_sw_sampler.py
"""
import os
import sys
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference; run_reference, Tape)

SW = np.load(os.path.join(HERE, "stockwatson_data.npz"))
T = int(SW["T"][0]); Y = SW["y"]


def sw_lpfun(q, hessian=False):
    """sw_innov.stan:7-51 as bridgestan's log_density_gradient would evaluate it (unconstrained parameters tSigma, z1,
    zinn[T-2], x1, xinn[T-1], tau1, tauinn[T-1]; constants dropped), value and analytic gradient with NumPy cumulative sums."""
    q = np.asarray(q, dtype=np.float64)
    tS, z1, zinn = q[0], q[1], q[2:T]
    x1, xinn = q[T], q[T + 1:2 * T]
    tau1, tauinn = q[2 * T], q[2 * T + 1:3 * T]
    sigma = np.exp(-0.5 * tS)
    z = z1 + np.concatenate([[0.0], np.cumsum(sigma * zinn)])      # z[1..T-1]
    x = x1 + np.concatenate([[0.0], np.cumsum(sigma * xinn)])      # x[1..T]
    ez = np.exp(0.5 * z)
    tau = tau1 + np.concatenate([[0.0], np.cumsum(ez * tauinn)])   # tau[1..T]
    w = np.exp(-x)
    res = Y - tau
    lp = 5.0 * tS - 0.5 * np.exp(tS) - 0.5 * np.sum(zinn * zinn) - 0.5 * np.sum(xinn * xinn) - 0.5 * np.sum(tauinn * tauinn) \
        + np.sum(-0.5 * x - 0.5 * res * res * w)
    a = res * w                                   # d lp / d tau[t]
    b = -0.5 + 0.5 * res * res * w                # d lp / d x[t]
    A = np.cumsum(a[::-1])[::-1]                  # A[t] = sum_{t' >= t} a[t']
    B = np.cumsum(b[::-1])[::-1]
    cc = 0.5 * ez * tauinn * A[1:]                # d lp / d z[j], j = 1..T-1
    CS = np.cumsum(cc[::-1])[::-1]
    g = np.empty(3 * T)
    g[1] = CS[0]
    g[2:T] = sigma * CS[1:] - zinn
    g[T] = B[0]
    g[T + 1:2 * T] = sigma * B[1:] - xinn
    g[2 * T] = A[0]
    g[2 * T + 1:3 * T] = ez * A[1:] - tauinn
    g[0] = 5.0 - 0.5 * np.exp(tS) - 0.5 * sigma * (np.sum(zinn * CS[1:]) + np.sum(xinn * B[1:]))
    if not np.isfinite(lp):
        return [np.nan, 0.0 * q]  # mainSW.py:22-26: a numeric exception is a NaN log density
    return [lp, g]


def main():
    import oracle_lib as ol
    # the restatement against its own complex-step derivative
    rng = np.random.default_rng(3)
    q = 0.3 * rng.normal(size=3 * T)
    lp, g = sw_lpfun(q)
    for j in (0, 1, 5, T - 1, T, T + 7, 2 * T - 1, 2 * T, 2 * T + 9, 3 * T - 1):
        h = 1e-6
        qp = q.copy(); qp[j] += h
        qm = q.copy(); qm[j] -= h
        fd = (sw_lpfun(qp)[0] - sw_lpfun(qm)[0]) / (2 * h)
        assert abs(fd - g[j]) <= 2e-5 * max(1.0, abs(g[j])), (j, fd, g[j])
    lo, go = ol.lpfun(ol.T_STOCKWATSON, q, np.concatenate([[float(T)], Y]))
    assert abs(lo - lp) <= 1e-11 * abs(lp) and np.allclose(go, g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    out = {}
    runs = {"D": ("D", dict(H0=0.1, delta0=0.3, numIter=3), 3), "R2P": ("R2P", dict(H0=0.1, delta0=0.3, numIter=3), 3),
            "fixed": ("fixed", dict(H0=0.002, numIter=1), 0)}
    for name, (ig, kw, minC) in runs.items():
        integ, kind = mg.INTEGRATORS[ig]
        chain = {"D": 5, "R2P": 6, "fixed": 7}[name]
        # M = 14 as mainSW.py; the fixedLeapFrog run (H0 = 0.002: 2047 leaves per iteration) is cut at M = 7 to stay small
        M = 7 if name == "fixed" else 14
        r = mg.run_reference(sw_lpfun, np.zeros(3 * T), integ, seed=mg.SEED, chain=chain, M=M, warmupIter=0,
                             igrAux=mg.ai.integratorAuxPar(minC=minC), **kw)
        tr = r["trace"]
        pre = f"sw756_{name}_"
        out[pre + "cfg"] = np.array([kind, kw["H0"], kw.get("delta0", 0.05), kw["numIter"], M, minC, chain, mg.SEED], dtype=np.float64)
        out[pre + "samples"] = r["samples"]
        out[pre + "diagnostics"] = r["diagnostics"]
        out[pre + "trace_scalars"] = tr[:, :12]
        keep = np.unique(np.r_[0:min(6, len(tr)), max(0, len(tr) - 3):len(tr)])
        out[pre + "trace_rows"] = keep.astype(np.int32)
        out[pre + "trace_qv"] = tr[keep, 12:]
        out[pre + "tape_u"] = r["tape_u"]; out[pre + "tape_z"] = r["tape_z"]; out[pre + "n_u"] = r["n_u"]
        print(name, "calls", len(tr), "evals", r["diagnostics"][:, 6].sum() + r["diagnostics"][:, 7].sum(), "Ndoubl", r["diagnostics"][:, 20])
    path = os.path.join(HERE, "stockwatson_sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
