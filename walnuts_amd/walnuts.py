"""walnuts() / walnuts_step() with the reference's signature (walnuts/walnuts.py:279-408).

The caller's numpy Generator is consumed on the host in exactly the reference's order --
normal(size=D) :325, binomial(1,0.5) :330, one choice([ell//2, ell, 2*ell]) per macro step :194,
uniform(0,1) :346, choice(orbit_ext, p=softmax) :350 -- so a seeded run reproduces the reference's
draws.  Everything that touches the target (energy, the stable_steps search, the leapfrog
integration: extend_orbit :211-276) runs on the GPU through wn_ext_orbit; the host keeps only the
orbit bookkeeping (U-turn tests and the categorical selection on the returned orbit).

`logp` / `grad` are device descriptors from walnuts_amd.targets (mirrors of test/targets.py) or
walnuts_amd.targetDistr; both arguments must describe the same target.
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib as L
from .targetDistr import DeviceTarget


def disable_runtime_warnings():  # walnuts.py:7-13
    warnings.filterwarnings("ignore", category=RuntimeWarning)


def uturn(theta_rho1, theta_rho2, inv_mass):  # walnuts.py:16-33
    theta1, rho1 = theta_rho1
    theta2, rho2 = theta_rho2
    diff = inv_mass * (theta2 - theta1)
    return np.dot(rho1, diff) < 0 or np.dot(rho2, diff) < 0


def sub_uturn(orbit, start_ignore, end_ignore, inv_mass):  # walnuts.py:62-70 (ignores start/end, Q9)
    length = np.shape(orbit)[0]
    span = length
    while span >= 2:
        for i in range(length // span):
            if uturn(orbit[span * i], orbit[span * (i + 1) - 1], inv_mass):
                return True
        span = span // 2
    return False


def _target_of(logp, grad):
    t = logp if isinstance(logp, DeviceTarget) else getattr(logp, "target", None)
    g = grad if isinstance(grad, DeviceTarget) else getattr(grad, "target", None)
    if t is None or g is None:
        raise TypeError("logp and grad must be device target descriptors (walnuts_amd.targets / "
                        "walnuts_amd.targetDistr): a Python callable cannot run on the GPU")
    if t is not g:
        raise ValueError("logp and grad describe different targets")
    return t


class _Device:
    """device buffers + the wn_ext_orbit call for one chain"""

    def __init__(self, target, D, inv_mass):
        import torch
        if not torch.cuda.is_available():
            from .engine import WalnutsError
            raise WalnutsError("walnuts_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
        self.torch, self.lib, self.D, self.target = torch, L.lib(), D, target
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.im = torch.as_tensor(np.ascontiguousarray(inv_mass, dtype=np.float64), device=self.dev)
        self.tpar = None
        if target.params is not None:
            self.tpar = torch.as_tensor(np.ascontiguousarray(np.asarray(target.params, dtype=np.float64)),
                                        device=self.dev).reshape(-1)


    def extend(self, theta, rho, n_macro, lottery, weight0, macro_step, max_error):
        t = self.torch
        P = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        th = t.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=self.dev).reshape(1, -1)
        rh = t.as_tensor(np.ascontiguousarray(rho, dtype=np.float64), device=self.dev).reshape(1, -1)
        lot = t.as_tensor(np.ascontiguousarray(lottery, dtype=np.int32), device=self.dev).reshape(1, -1)
        w0 = t.as_tensor(np.array([weight0], dtype=np.float64), device=self.dev)
        orbit = t.empty((1, n_macro, 2, self.D), dtype=t.float64, device=self.dev)
        logw = t.empty((1, n_macro), dtype=t.float64, device=self.dev)
        rc = self.lib.wn_ext_orbit(self.dev.index, self.target.kind, P(self.tpar), self.D, 1, P(th), P(rh),
                                   P(self.im), float(macro_step), float(max_error), int(n_macro), P(lot),
                                   P(w0), P(orbit), P(logw), None,
                                   C.c_void_p(t.cuda.current_stream(self.dev).cuda_stream))
        if rc:
            raise RuntimeError(self.lib.wn_last_error(None).decode())
        o = orbit.cpu().numpy()[0]
        return [(o[m, 0], o[m, 1]) for m in range(n_macro)], logw.cpu().numpy()[0].tolist()

    def minus_H(self, theta, rho, inv_mass):
        """-H(theta, rho) = logp(theta) - 0.5*dot(inv_mass, rho**2) (walnuts.py:127-141); logp on the GPU."""
        t = self.torch
        th = t.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=self.dev).reshape(1, -1)
        lp = t.empty(1, dtype=t.float64, device=self.dev)
        g = t.empty_like(th)
        rc = self.lib.wn_lpfun_kind(self.dev.index, self.target.kind,
                                    C.c_void_p(self.tpar.data_ptr()) if self.tpar is not None else None,
                                    self.D, 1, C.c_void_p(th.data_ptr()), C.c_void_p(lp.data_ptr()),
                                    C.c_void_p(g.data_ptr()),
                                    C.c_void_p(t.cuda.current_stream(self.dev).cuda_stream))
        if rc:
            raise RuntimeError(self.lib.wn_last_error(None).decode())
        return float(lp.cpu()[0]) - 0.5 * np.dot(inv_mass, rho ** 2)


def _walnuts_step(dev, rng, theta, inv_mass, macro_step, max_nuts_depth, max_error):
    from scipy.special import logsumexp, softmax
    L_mass = inv_mass ** -0.5
    D = theta.size
    rho = L_mass * rng.normal(size=D)                       # :325
    orbit, log_weights = [(theta, rho)], [dev.minus_H(theta, rho, inv_mass)]  # :326
    theta_selected = theta
    for depth in range(max_nuts_depth):                     # :328
        num_macro_steps = 2 ** depth
        going_backward = bool(rng.binomial(1, 0.5))         # :330
        # one choice([ell//2, ell, 2*ell]) per macro step (:194,:256): the index does not depend on ell
        lottery = [int(rng.choice(3)) for _ in range(num_macro_steps)]
        if going_backward:                                  # extend_orbit :242-248
            th0, rh0 = orbit[0]
            rh0 = -rh0
            w0 = log_weights[0]
        else:
            th0, rh0 = orbit[-1]
            w0 = log_weights[-1]
        orbit_ext, log_weights_ext = dev.extend(th0, rh0, num_macro_steps, lottery, w0, macro_step, max_error)
        if going_backward:                                  # :274-275
            orbit_ext, log_weights_ext = orbit_ext[::-1], log_weights_ext[::-1]
        if sub_uturn(orbit_ext, 0, num_macro_steps, inv_mass):  # :343
            break
        log_accept_prob_ub = logsumexp(log_weights_ext) - logsumexp(log_weights)  # :345
        log_uniform = np.log(rng.uniform(0, 1))             # :346
        if log_uniform < log_accept_prob_ub:                # :347-350
            p = softmax(log_weights_ext)
            theta_selected, _ = orbit_ext[int(rng.choice(len(orbit_ext), p=p))]
        orbit = orbit_ext + orbit if going_backward else orbit + orbit_ext  # :351
        if uturn(orbit[0], orbit[-1], inv_mass):            # :352
            break
        log_weights = log_weights_ext + log_weights if going_backward else log_weights + log_weights_ext
    return theta_selected


def _validate(theta, inv_mass, macro_step, max_nuts_depth, max_error):  # walnuts.py:307-320
    theta = np.array(theta, dtype=np.float64)
    inv_mass = np.array(inv_mass, dtype=np.float64)
    if theta.ndim != 1:
        raise ValueError("theta not a vector")
    if inv_mass.ndim != 1:
        raise ValueError("inv_mass not a vector")
    if theta.size != inv_mass.size:
        raise ValueError("size mismatch between theta and inv_mass")
    if not macro_step > 0:
        raise ValueError("non-positive macro_step")
    if not max_nuts_depth > 0:
        raise ValueError("non-positive max_nuts_depth")
    if not max_error > 0:
        raise ValueError("non-positive max_error")
    return theta, inv_mass


def walnuts_step(rng, theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error):
    """Next state of the WALNUTS chain from `theta` (walnuts.py:279-359)."""
    theta, inv_mass = _validate(theta, inv_mass, macro_step, max_nuts_depth, max_error)
    target = _target_of(logp, grad)
    target.check_dim(theta.size)
    dev = _Device(target, theta.size, inv_mass)
    return _walnuts_step(dev, rng, theta, inv_mass, macro_step, max_nuts_depth, max_error)


def walnuts(rng, theta_init, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup,
            iter_sample):
    """Markov chain of `iter_sample` draws, shape (iter_sample, D) (walnuts.py:362-408).  Warm-up
    draws are discarded; this variant has no adaptation (walnuts.py:406)."""
    disable_runtime_warnings()
    theta = np.array(theta_init, dtype=np.float64)
    D = theta.size
    draws = np.empty((iter_sample, D))
    dev = None
    for i in range(iter_warmup + iter_sample):
        theta, inv_mass_a = _validate(theta, inv_mass, macro_step, max_nuts_depth, max_error)
        if dev is None:
            target = _target_of(logp, grad)
            target.check_dim(D)
            dev = _Device(target, D, inv_mass_a)
        theta = _walnuts_step(dev, rng, theta, inv_mass_a, macro_step, max_nuts_depth, max_error)
        if i >= iter_warmup:
            draws[i - iter_warmup] = theta
    return draws
