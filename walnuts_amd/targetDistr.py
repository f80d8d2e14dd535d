"""Device forms of the reference's target plugins (WALNUTSpy/targetDistr.py).

The reference passes a Python callable lpFun(q, hessian=False) -> [lp, grad]
(targetDistr.py:18).  A Python callable cannot run inside a fused GPU kernel, so a target here
is a descriptor (kind enum + parameter buffer) that the kernels evaluate in place; calling the
descriptor evaluates it on the GPU through the same C ABI (wn_lpfun), keeping the reference's
call shape for scripts that probe a target by hand.
"""
import numpy as np

from . import _lib as L


class DeviceTarget:
    def __init__(self, name, kind, dim=None, params=None):
        self.name, self.kind, self.dim, self.params = name, kind, dim, params

    def check_dim(self, d):
        if self.dim is not None and d != self.dim:
            raise ValueError(f"target {self.name} needs dimension {self.dim}, got {d}")

    def __call__(self, q, hessian=False):
        """lpFun(q) -> [lp, grad], evaluated on the GPU."""
        from .engine import Engine
        if hessian:
            raise NotImplementedError("Hessians are only used by integrators outside the hot path")
        q = np.asarray(q, dtype=np.float64)
        self.check_dim(q.size)
        e = Engine(q.size, 1).set_target(self.kind, self.params)
        lp, g = e.lpfun(q.reshape(1, -1))
        out = [float(lp.cpu()[0]), g.cpu().numpy()[0]]
        e.close()
        return out

    def __repr__(self):
        return f"<walnuts_amd target {self.name}>"


class BatchedTarget:
    """The lpFun plugin evaluated by the caller for ALL chains at once, between launches of the
    chain kernel (include/walnuts_amd.h, WN_T_EXTERNAL): `fn(q)` gets the positions as a device
    tensor view [C, D] (float64, row stride >= D) and returns (lp [C], grad [C, D])."""

    def __init__(self, fn, dim=None, name="batched"):
        self.fn, self.dim, self.name = fn, dim, name

    def check_dim(self, d):
        if self.dim is not None and d != self.dim:
            raise ValueError(f"target {self.name} needs dimension {self.dim}, got {d}")


def batched(fn, dim=None):
    return BatchedTarget(fn, dim)


def host_callable(lpFun):
    """Wrap a reference-style per-chain host callable lpFun(q) -> [lp, grad] (targetDistr.py:18):
    positions go to the host, lpFun runs once per chain, results go back.  Keeps the reference's
    plugin contract exactly; the speed is the host callable's."""

    def fn(qv):
        import torch
        q = qv.cpu().numpy()
        lp = np.empty(q.shape[0]); g = np.empty_like(q)
        for c in range(q.shape[0]):
            r = lpFun(q[c].copy())
            lp[c] = r[0]; g[c] = np.asarray(r[1], dtype=np.float64)
        return torch.as_tensor(lp, device=qv.device), torch.as_tensor(g, device=qv.device)

    return BatchedTarget(fn, None, getattr(lpFun, "__name__", "host_callable"))


stdGauss = DeviceTarget("stdGauss", L.T_STDGAUSS)              # targetDistr.py:18-21
corrGauss = DeviceTarget("corrGauss", L.T_CORRGAUSS, 2)        # targetDistr.py:25-31
smileDistr = DeviceTarget("smileDistr", L.T_SMILE, 2)          # targetDistr.py:34-38
modFunnel = DeviceTarget("modFunnel", L.T_MODFUNNEL, 2)        # targetDistr.py:41-51
funnel10 = DeviceTarget("funnel10", L.T_FUNNEL10, 11)          # targetDistr.py:74-78
funnel1 = DeviceTarget("funnel1", L.T_FUNNEL1, 2)              # targetDistr.py:88-93


modFunnelH = modFunnel  # targetDistr.py:53-54 with hessian=False (Hessians are used by integrators outside the path)


def _funnel10rescaled(q):
    """targetDistr.py:81-86: funnel10 in the coordinates qb = S*q, S = (3, 1, ..., 1); batched torch form
    (split mode: evaluated for all chains between chain-kernel launches)."""
    import math
    import torch
    S = torch.ones(11, dtype=q.dtype, device=q.device); S[0] = 3.0
    qb = q * S
    q0 = qb[:, 0]
    emq = torch.exp(-q0)
    ss = (qb[:, 1:] * qb[:, 1:]).sum(dim=1)
    logC = 0.5 * math.log(2.0 * math.pi)
    lp = (-(q0 * q0) / 18.0 - logC - math.log(3.0)) + (-0.5 * emq * ss - 10.0 * logC - 5.0 * q0)
    g = torch.empty_like(qb)
    g[:, 0] = -5.0 - q0 / 9.0 + 0.5 * emq * ss
    g[:, 1:] = -qb[:, 1:] * emq[:, None]
    return lp, g * S


funnel10rescaled = BatchedTarget(_funnel10rescaled, 11, "funnel10rescaled")  # targetDistr.py:81-86


def denseGauss(A):
    """lp = -1/2 q'Aq, grad = -Aq for a symmetric precision matrix A (cfg 2/4)."""
    A = np.asarray(A, dtype=np.float64) if not hasattr(A, "data_ptr") else A
    d = A.shape[0]
    if A.shape != (d, d):
        raise ValueError("A must be square")
    return DeviceTarget(f"denseGauss{d}", L.T_DENSEGAUSS, d, A)


def diagGauss(a):
    """lp = -1/2 sum a_i q_i^2 (diagonal precision)."""
    a = np.asarray(a, dtype=np.float64) if not hasattr(a, "data_ptr") else a
    return DeviceTarget(f"diagGauss{a.shape[0]}", L.T_DIAGGAUSS, int(a.shape[0]), a)


def stockWatson(y):
    """StockWatson stochastic-volatility model of WALNUTSpy_examples/StockWatson/sw_innov.stan as
    mainSW.py:15-26 evaluates it through bridgestan (unconstrained parameters, D = 3T, T <= 256);
    `y` is the observation vector of swdata.json."""
    y = np.asarray(y, dtype=np.float64)
    T = int(y.shape[0])
    if not 3 <= T <= 256:
        raise ValueError("StockWatson needs 3 <= T <= 256")
    return DeviceTarget(f"stockWatson{T}", L.T_STOCKWATSON, 3 * T, np.concatenate([[float(T)], y]))


def stockWatsonConstrain(y):
    """Host form of mainSW.py:28-29 `generatedInnov(q) = param_constrain(q, include_tp=True)` for
    sw_innov.stan: the 3T unconstrained parameters followed by the transformed parameters in
    declaration order (sigma, gpri, z[T-1], x[T], tau[T]; sw_innov.stan:17-35).  Pass it as
    `generated=`: identity draws are stored on the device and the transform is applied to each stored
    draw on the host, once per iteration, as in the reference.  The device form of the same transform is
    `walnuts_amd.sw_constrain` (include/walnuts_amd.h WN_G_SW_CONSTRAIN); this host form is what it is tested against."""
    T = int(np.asarray(y).shape[0])

    def gen(q):
        q = np.asarray(q, dtype=np.float64)
        tS, z1, zinn = q[0], q[1], q[2:T]
        x1, xinn = q[T], q[T + 1:2 * T]
        tau1, tauinn = q[2 * T], q[2 * T + 1:3 * T]
        sigma, gpri = np.exp(-0.5 * tS), np.exp(tS)
        z = np.empty(T - 1); x = np.empty(T); tau = np.empty(T)
        z[0] = z1
        for t in range(1, T - 1):
            z[t] = z[t - 1] + sigma * zinn[t - 1]
        x[0] = x1
        for t in range(1, T):
            x[t] = x[t - 1] + sigma * xinn[t - 1]
        tau[0] = tau1
        for t in range(1, T):
            tau[t] = tau[t - 1] + np.exp(0.5 * z[t - 1]) * tauinn[t - 1]
        return np.concatenate([q, [sigma, gpri], z, x, tau])

    return gen
