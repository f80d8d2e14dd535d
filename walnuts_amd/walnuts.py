"""walnuts() / walnuts_step() / walnuts_stan() with the reference's signatures (walnuts/walnuts.py:279-408,
walnuts/walnuts_stan.py:4-15), plus the batched forms walnuts_batch() / walnuts_step_batch().

Everything a transition computes -- stable_steps searches, leapfrog integration, weights, sub-U-turn and U-turn
tests, logsumexp accept bound, categorical pick -- runs on the GPU for all chains at once (csrc/wn_k_wstep.hip
behind the wn_wstep_* entry points).  The host only consumes each caller-owned numpy Generator, in exactly the
reference's order: normal(size=D) :325, then per doubling binomial(1, .5) :330, one choice(3) per macro step
:194, uniform(0, 1) :346 unless the extension had a sub-U-turn :343, and the one random() of
Generator.choice(orbit_ext, p=...) :350 after an accept -- so a seeded run reproduces the reference's draws
and leaves the Generator at the reference's stream position.

`logp` / `grad` are either device descriptors (walnuts_amd.targets mirrors test/targets.py;
walnuts_amd.targetDistr) or plain Python callables logp(theta) -> float, grad(theta) -> (D,) as in the
reference (:379-380): the device then stops wherever the algorithm needs them, the pair is called once per
waiting chain on the host, and the device carries on (the reference's contract at the reference's granularity).
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _lib as L
from .targetDistr import DeviceTarget


def _quiet():
    warnings.filterwarnings("ignore", category=RuntimeWarning)  # as the reference does at the top of walnuts() (:391)


def _check_args(theta, inv_mass, macro_step, max_nuts_depth, max_error):
    """The reference's argument errors (walnuts.py:307-320), same messages; theta may be (D,) or (C, D)."""
    theta = np.array(theta, dtype=np.float64)
    inv_mass = np.array(inv_mass, dtype=np.float64)
    if theta.ndim not in (1, 2) or 0 in theta.shape:
        raise ValueError("theta not a vector")
    if inv_mass.ndim != 1:
        raise ValueError("inv_mass not a vector")
    if theta.shape[-1] != inv_mass.size:
        raise ValueError("size mismatch between theta and inv_mass")
    if not macro_step > 0:
        raise ValueError("non-positive macro_step")
    if not max_nuts_depth > 0:
        raise ValueError("non-positive max_nuts_depth")
    if not max_error > 0:
        raise ValueError("non-positive max_error")
    return theta, inv_mass


def _resolve(logp, grad):
    """-> (device descriptor, None) or (None, (logp, grad) host callables)."""
    t = logp if isinstance(logp, DeviceTarget) else getattr(logp, "target", None)
    g = grad if isinstance(grad, DeviceTarget) else getattr(grad, "target", None)
    if t is not None or g is not None:
        if t is not g:
            raise ValueError("logp and grad describe different targets")
        return t, None
    if not (callable(logp) and callable(grad)):
        raise TypeError("logp and grad must be callables (theta -> float, theta -> (D,)) or walnuts_amd target descriptors")
    return None, (logp, grad)


_HOSTRNG = None


def _hostrng_lib():
    """libwalnuts_hostrng.so (csrc/wn_hostrng.c): numpy's own C implementations of Generator.normal / binomial / integers / uniform /
    random (libnpyrandom) called on the callers' bit generators -- the same streams as per-chain Python calls, without 4096 x 5
    interpreter round trips per doubling.  None when it is not built (the per-chain Python calls below are then used: same draws)."""
    global _HOSTRNG
    if _HOSTRNG is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwalnuts_hostrng.so")
        try:
            # libnpyrandom's distributions are linked statically: a helper built against another numpy could drift from what the
            # caller's Generator methods draw, so it is only used with the numpy it was built for (csrc/Makefile records the version)
            with open(path + ".numpy_version") as f:
                if f.read().strip() != np.__version__:
                    raise OSError("libwalnuts_hostrng.so was built for another numpy")
            lib = C.CDLL(path)
            for name in ("wn_hr_momentum", "wn_hr_extend_draws", "wn_hr_accept_uniforms", "wn_hr_choice_uniforms"):
                getattr(lib, name).restype = None
            _HOSTRNG = lib
        except OSError:
            _HOSTRNG = False
    return _HOSTRNG or None


def _bitgen_pointers(rngs):
    """(void*)[B] of the Generators' bitgen_t, or None when some `rng` is not a numpy Generator with a ctypes interface (stand-ins)
    or when one Generator serves several chains (the C path draws all chains' accept uniforms and then all choice uniforms, the
    per-chain Python path interleaves them per chain: the two orders are the same only for distinct Generators).  The C path reads the
    bit generators' states directly, without Generator's lock: do not draw from these Generators on another thread during a call."""
    try:
        if len({id(g) for g in rngs}) != len(rngs):
            return None
        ptrs = []
        for g in rngs:
            if not isinstance(g, np.random.Generator):
                return None
            a = g.bit_generator.ctypes.bit_generator
            ptrs.append(a.value if hasattr(a, "value") else int(a))
        return (C.c_void_p * len(ptrs))(*ptrs)
    except Exception:
        return None


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


class _Stepper:
    """The wn_wstep handle of B chains and its exchange buffers."""

    def __init__(self, target, fns, B, D, inv_mass, macro_step, max_nuts_depth, max_error, route=0):
        import torch
        if not torch.cuda.is_available():
            from .engine import WalnutsError
            raise WalnutsError("walnuts_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
        self.t, self.lib, self.B, self.D, self.fns, self.depth = torch, L.lib(), B, D, fns, int(max_nuts_depth)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        dv = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.dev)
        self.dv = dv
        im = dv(inv_mass)
        tpar = dv(np.asarray(target.params, dtype=np.float64)).reshape(-1) if (target is not None and target.params is not None) else None
        kind = L.T_EXTERNAL if target is None else target.kind
        h = C.c_void_p()
        rc = self.lib.wn_wstep_create(self.dev.index, kind, self.p(tpar), tpar.numel() if tpar is not None else 0, D, B, self.depth,
                                      self.p(im), float(macro_step), float(max_error), C.byref(h))
        if rc:
            msg = self.lib.wn_wstep_last_error(None).decode()
            raise NotImplementedError(msg) if rc == L.E_UNSUPPORTED else (MemoryError(msg) if rc == L.E_ALLOC else ValueError(msg))
        self.h = h
        if route:  # WN_ROUTE_WSTEP_LEGACY: the one-lane kernels (A/B comparisons; include/walnuts_amd.h)
            self._ck(self.lib.wn_wstep_set_route(self.h, L.route_bits(route)))
        z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=self.dev)
        self.sub, self.uturn = z(B, dt=torch.int32), z(B, dt=torch.int32)
        self.lse_ext, self.lse_old = z(B), z(B)
        if fns is not None:
            self.xreq, self.xlp, self.xg = z(B, D), z(B), z(B, D)
            self.waiting = z(B, dt=torch.int32)
            self._ck(self.lib.wn_wstep_ext_bind(self.h, self.p(self.xreq), self.p(self.xlp), self.p(self.xg)))

    @staticmethod
    def p(x):
        return C.c_void_p(x.data_ptr()) if x is not None else None

    def _st(self):
        return C.c_void_p(self.t.cuda.current_stream(self.dev).cuda_stream)

    def _ck(self, rc):
        if rc:
            raise RuntimeError(self.lib.wn_wstep_last_error(self.h).decode())

    def _serve(self, waiting=None):
        """Host (logp, grad) pair: evaluate the positions the device asked for until no chain waits.  Only the chains
        that asked in this round are evaluated (wn_wstep_ext_waiting): a chain whose extension has finished keeps a stale
        row in xreq, which the reference never evaluates (side effects, exceptions, NaN positions of user callables)."""
        if self.fns is None:
            return
        logp, grad = self.fns
        n = C.c_int32(0)
        while True:
            self._ck(self.lib.wn_wstep_ext_pending(self.h, C.byref(n), self._st()))
            if n.value == 0:
                return
            self._ck(self.lib.wn_wstep_ext_waiting(self.h, self.p(self.waiting), self._st()))
            x = self.xreq.cpu().numpy()
            w = self.waiting.cpu().numpy()
            lp = np.zeros(self.B); g = np.zeros((self.B, self.D))
            for b in np.flatnonzero(w):
                lp[b] = logp(x[b])
                g[b] = grad(x[b])
            self.xlp.copy_(self.dv(lp)); self.xg.copy_(self.dv(g))
            self._ck(self.lib.wn_wstep_ext_feed(self.h, self.p(self.sub), self.p(self.lse_ext), self.p(self.lse_old), self._st()))

    def step(self, rngs, theta, inv_mass):
        """One transition of every chain (walnuts.py:322-359); theta (B, D) -> next states (B, D)."""
        t, B, D = self.t, self.B, self.D
        L_mass = inv_mass ** -0.5
        hr = _hostrng_lib()
        if hr is not None:  # the Generators' C interface: the same draws from the same streams, no per-chain interpreter calls
            self._bg = _bitgen_pointers(rngs)  # rebuilt every step (B attribute reads): a Generator replaced in the caller's list is seen
            if self._bg is None:
                hr = None
        bg = self._bg if hr is not None else None
        if hr is not None:
            rho = np.empty((B, D)); lm = np.ascontiguousarray(L_mass, dtype=np.float64)
            hr.wn_hr_momentum(bg, B, D, _np(lm), _np(rho))                              # :325
        else:
            rho = np.stack([L_mass * rngs[b].normal(size=D) for b in range(B)])      # :325
        th_d, rho_d = self.dv(theta), self.dv(rho)
        self._ck(self.lib.wn_wstep_begin(self.h, self.p(th_d), self.p(rho_d), self._st()))
        self._serve(range(B))
        active = np.ones(B, dtype=np.int32)
        for depth in range(self.depth):                                               # :328
            n = 1 << depth
            if not active.any():
                break
            backward = np.zeros(B, dtype=np.int32); lottery = np.zeros((B, n), dtype=np.int32)
            if hr is not None:
                act = None
                scratch = np.empty(n, dtype=np.uint64)
                hr.wn_hr_extend_draws(bg, _np(active), B, n, _np(backward), _np(lottery), _np(scratch))  # :330, :194
            else:
                act = [b for b in range(B) if active[b]]
                for b in act:
                    backward[b] = int(bool(rngs[b].binomial(1, 0.5)))                     # :330
                    lottery[b] = [int(rngs[b].choice(3)) for _ in range(n)]              # :194, one per macro step (:256)
            bw_d, lo_d, ac_d = self.dv(backward, t.int32), self.dv(lottery, t.int32), self.dv(active, t.int32)
            self._ck(self.lib.wn_wstep_extend(self.h, depth, self.p(bw_d), self.p(lo_d), self.p(ac_d), self.p(self.sub),
                                              self.p(self.lse_ext), self.p(self.lse_old), self._st()))
            self._serve(act)
            sub = self.sub.cpu().numpy(); le = self.lse_ext.cpu().numpy(); lo = self.lse_old.cpu().numpy()
            accept = np.zeros(B, dtype=np.int32); u_choice = np.zeros(B)
            if hr is not None:
                sub = np.ascontiguousarray(sub, dtype=np.int32)
                u = np.ones(B)
                hr.wn_hr_accept_uniforms(bg, _np(active), _np(sub), B, _np(u))         # :346 (not after a sub-U-turn, :343)
                live = (active != 0) & (sub == 0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    accept = (live & (np.log(u) < le - lo)).astype(np.int32)           # :345-347
                hr.wn_hr_choice_uniforms(bg, _np(accept), B, _np(u_choice))            # Generator.choice(orbit_ext, p=p) :350
                active[(active != 0) & (sub != 0)] = 0
            else:
                for b in act:
                    if sub[b]:                                                             # :343
                        active[b] = 0
                        continue
                    log_uniform = np.log(rngs[b].uniform(0, 1))                           # :346
                    if log_uniform < le[b] - lo[b]:                                       # :345-347
                        accept[b] = 1
                        u_choice[b] = rngs[b].random()                                    # Generator.choice(orbit_ext, p=p) :350
            ac_d2, uc_d = self.dv(accept, t.int32), self.dv(u_choice)
            self._ck(self.lib.wn_wstep_commit(self.h, self.p(ac_d2), self.p(uc_d), self.p(self.uturn), self._st()))
            ut = self.uturn.cpu().numpy()
            active[(active != 0) & (ut != 0)] = 0                                      # :352
        out = t.empty((B, D), dtype=t.float64, device=self.dev)
        self._ck(self.lib.wn_wstep_result(self.h, self.p(out), self._st()))
        return out.cpu().numpy()

    def close(self):
        if getattr(self, "h", None):
            self.t.cuda.synchronize(self.dev)
            self.lib.wn_wstep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stepper(theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, route=0):
    target, fns = _resolve(logp, grad)
    B, D = theta.shape
    if target is not None:
        target.check_dim(D)
    return _Stepper(target, fns, B, D, inv_mass, macro_step, max_nuts_depth, max_error, route)


def walnuts_step_batch(rngs, theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error):
    """Next states (C, D) of C independent WALNUTS chains from theta (C, D); rngs = one numpy Generator per chain."""
    theta, inv_mass = _check_args(theta, inv_mass, macro_step, max_nuts_depth, max_error)
    theta = np.atleast_2d(theta)
    if len(rngs) != theta.shape[0]:
        raise ValueError("one Generator per chain")
    s = _stepper(theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error)
    try:
        return s.step(rngs, theta, inv_mass)
    finally:
        s.close()


def walnuts_step(rng, theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error):
    """Next state of the WALNUTS chain from `theta` (walnuts.py:279-359)."""
    theta, inv_mass = _check_args(theta, inv_mass, macro_step, max_nuts_depth, max_error)
    if theta.ndim != 1:
        raise ValueError("theta not a vector")
    return walnuts_step_batch([rng], theta[None, :], logp, grad, inv_mass, macro_step, max_nuts_depth, max_error)[0]


def walnuts_batch(rngs, theta_init, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup, iter_sample, *, route=0):
    """C chains at once: theta_init (C, D), one Generator per chain -> draws (C, iter_sample, D).  Chain c consumes
    rngs[c] exactly as walnuts(rngs[c], theta_init[c], ...) would."""
    _quiet()
    theta, inv_mass = _check_args(theta_init, inv_mass, macro_step, max_nuts_depth, max_error)
    theta = np.atleast_2d(theta)
    Cn, D = theta.shape
    if len(rngs) != Cn:
        raise ValueError("one Generator per chain")
    draws = np.empty((Cn, iter_sample, D))
    s = _stepper(theta, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, route)
    try:
        for i in range(iter_warmup + iter_sample):
            theta = s.step(rngs, theta, inv_mass)
            if i >= iter_warmup:
                draws[:, i - iter_warmup] = theta
    finally:
        s.close()
    return draws


def walnuts_device(seed, theta_init, target, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup, iter_sample,
                   chain_offset=0, transitions_per_launch=64, iter0=1, return_stats=False, route=0):
    """C chains with the DEVICE's random stream instead of numpy Generators: theta_init (C, D) -> draws (C, iter_sample, D).
    The transition kernel is walnuts_step's (walnuts.py:279-359); the randomness is the engine's Philox stream keyed by
    (seed, chain_offset + chain, transition number) -- wn_rng.h -- so a chain's draws do not depend on how chains are batched
    or sharded, and whole transitions run on the GPU with no host round trip per doubling (`walnuts_batch` keeps the
    reference's Generator contract and pays ~4 host exchanges per doubling for it).  `target` is a walnuts_amd device target
    (e.g. walnuts_amd.targetDistr.stdGauss), not a host callable.  `iter0` is the transition number of the first transition (the
    stream's `iteration` key).  return_stats=True also returns int32 [C, iter_warmup + iter_sample, 4] = {doublings started, accepts, sum of
    the chosen micro-step counts, sum of the forward ell_stable} per transition (wn_wstep_set_stats): the integers the unmodified
    reference reproduces when it is driven with the same stream (tests/golden/make_golden_walnuts_device.py)."""
    _quiet()
    theta, inv_mass = _check_args(theta_init, inv_mass, macro_step, max_nuts_depth, max_error)
    theta = np.atleast_2d(theta)
    Cn, D = theta.shape
    tgt, fns = _resolve(target, target)
    if tgt is None:
        raise ValueError("walnuts_device needs a device target descriptor; host callables go through walnuts_batch")
    s = _stepper(theta, target, target, inv_mass, macro_step, max_nuts_depth, max_error, route)
    try:
        t = s.t
        th = s.dv(theta)
        n_tot = int(iter_warmup) + int(iter_sample)
        out = t.empty((Cn, int(iter_sample), D), dtype=t.float64, device=s.dev)
        stats = []
        it = int(iter0)
        done = 0
        while done < n_tot:
            n = min(int(transitions_per_launch), n_tot - done)
            buf = t.empty((Cn, n, D), dtype=t.float64, device=s.dev)
            if return_stats:
                stats.append(t.zeros((Cn, n, 4), dtype=t.int32, device=s.dev))
                s._ck(s.lib.wn_wstep_set_stats(s.h, s.p(stats[-1])))
            s._ck(s.lib.wn_wstep_run_device_rng(s.h, int(seed), int(chain_offset), it, n, s.p(th), s.p(buf), s._st()))
            lo = max(done, int(iter_warmup))
            if done + n > lo:
                out[:, lo - int(iter_warmup):done + n - int(iter_warmup)] = buf[:, lo - done:]
            done += n; it += n
        if return_stats:
            s._ck(s.lib.wn_wstep_set_stats(s.h, None))
            return out.cpu().numpy(), t.cat(stats, dim=1).cpu().numpy()
        return out.cpu().numpy()
    finally:
        s.close()


def walnuts(rng, theta_init, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup, iter_sample):
    """Markov chain of `iter_sample` draws, shape (iter_sample, D) (walnuts.py:362-408).  Warm-up draws are
    discarded; this variant has no adaptation (walnuts.py:406)."""
    theta = np.array(theta_init, dtype=np.float64)
    if theta.ndim != 1:
        raise ValueError("theta not a vector")
    return walnuts_batch([rng], theta[None, :], logp, grad, inv_mass, macro_step, max_nuts_depth, max_error,
                         iter_warmup, iter_sample)[0]


def walnuts_stan(rng, theta_init, stan_file, data_file, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup,
                 iter_sample):
    """walnuts() on a Stan model through bridgestan (walnuts/walnuts_stan.py:4-43): logp = StanModel.log_density,
    grad = the gradient half of StanModel.log_density_gradient, both passed to walnuts() as host callables.
    (The reference's own file stops with NameErrors before sampling -- SURVEY.md Q10; this is its documented intent.)
    bridgestan is not part of this package: without it the call fails with the ImportError below.  For the one
    Stan model of the reference's examples, StockWatson, `walnuts_amd.targetDistr.stockWatson(y)` is a fused device
    target that needs no Stan toolchain."""
    try:
        import bridgestan as bs
    except ImportError as ex:
        raise ImportError("walnuts_stan needs the bridgestan package (and a Stan toolchain) to compile "
                          f"{stan_file!r}; it is not installed here.  Pass logp / grad callables or a walnuts_amd device "
                          "target to walnuts() instead.") from ex
    model = bs.StanModel(stan_file, data_file)

    def logp(theta):
        return model.log_density(theta)

    def grad(theta):
        return model.log_density_gradient(theta)[1]

    return walnuts(rng, theta_init, logp, grad, inv_mass, macro_step, max_nuts_depth, max_error, iter_warmup, iter_sample)
