"""Rank-normalised split bulk ESS (Vehtari, Gelman, Simpson, Carpenter, Buerkner 2021), the
default method of ArviZ's `az.ess`, which the reference's experiment scripts use
(WALNUTSpy_examples/gaussian/mainGaussESS.py:51).  ArviZ is absent here (ESS parity
unpinned), so this is an independent implementation validated on AR(1) series with known ESS.
NumPy only; runs on the host on the (small) gathered draws, not on the hot path.
"""
import numpy as np


def _autocov(x):
    """autocovariance along the last axis via FFT, biased (1/n) estimator"""
    n = x.shape[-1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=-1, keepdims=True)
    f = np.fft.rfft(xc, n=m, axis=-1)
    ac = np.fft.irfft(f * np.conj(f), n=m, axis=-1)[..., :n]
    return ac / n


def _z_scale(x):
    """rank-normalise all draws jointly (Blom offsets, as ArviZ/posterior do)"""
    from scipy.stats import norm, rankdata
    r = rankdata(x.reshape(-1), method="average").reshape(x.shape)
    return norm.ppf((r - 0.375) / (x.size + 0.25))


def ess(x, split=True, rank_normalise=True):
    """x: (chains, draws) -> bulk effective sample size of all draws together."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    if split:
        n = x.shape[1] // 2
        if n < 2:
            return float("nan")
        x = np.concatenate([x[:, :n], x[:, -n:]], axis=0)
    if rank_normalise:
        x = _z_scale(x)
    m, n = x.shape
    if n < 4:
        return float("nan")
    acov = _autocov(x)
    chain_mean = x.mean(axis=1)
    mean_var = acov[:, 0].mean() * n / (n - 1.0)
    var_plus = mean_var * (n - 1.0) / n
    if m > 1:
        var_plus += chain_mean.var(ddof=1)
    if not var_plus > 0:
        return float("nan")
    rho = 1.0 - (mean_var - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    # Geyer's initial positive / monotone sequence on pair sums
    t = 1
    rho_even, rho_odd = 1.0, rho[1]
    tau = 0.0
    pair_sums = []
    while t < n - 2:
        s = rho_even + rho_odd
        if not (s > 0):
            break
        pair_sums.append(s)
        rho_even, rho_odd = rho[t + 1], rho[t + 2]
        t += 2
    if not pair_sums:
        return float(m * n)
    ps = np.minimum.accumulate(np.array(pair_sums))
    tau = -1.0 + 2.0 * ps.sum()
    tau = max(tau, 1.0 / np.log10(m * n))
    return float(m * n / tau)
