"""Engine: thin object over the C ABI.  PyTorch-ROCm tensors are used only as device buffers."""
import ctypes as C

import numpy as np

from . import _lib as L


class WalnutsError(RuntimeError):
    pass


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise WalnutsError("walnuts_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
    return torch


class Engine:
    """One engine per GPU: C chains of dimension D with a fixed sampler configuration.

    Mirrors the keyword arguments of the reference's WALNUTS(...) (WALNUTSpy/WALNUTS.py:111-129)
    and integratorAuxPar (adaptiveIntegrators.py:36-44)."""

    def __init__(self, dim, n_chains, integrator=L.I_FIXED, H0=0.2, stepSizeRandScale=0.2,
                 delta0=0.05, M=10, minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=L.G_IDENTITY,
                 gen_k=0, warmupIter=0, chain_offset=0, device=None, waves_per_group=0, adaptH=True,
                 adaptHtarget=0.8, adaptDelta=True, adaptDeltaTarget=0.6, adaptDeltaQuantile=0.9, route=0, split_parts=0,
                 split_ks=0, adaptQuantBits=0):
        """route: 0 (the library chooses the kernel) or WN_ROUTE_* names / bits (include/walnuts_amd.h: the other implementation of
        the same function, for A/B comparisons); adaptQuantBits: 0 = the reference's arithmetic, b = the adapted H / delta rounded
        to b mantissa bits (opt-in; DESIGN 9, warm-up tolerance contract)."""
        torch = _torch()
        self.torch = torch
        self.lib = L.lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        cfg = L.WnConfig()
        cfg.abi_version = L.WN_ABI_VERSION
        cfg.device = self.device.index
        cfg.dim = int(dim); cfg.n_chains = int(n_chains); cfg.chain_offset = int(chain_offset)
        cfg.integrator = int(integrator); cfg.M = int(M); cfg.min_c = int(minC); cfg.max_c = int(maxC)
        cfg.H0 = float(H0); cfg.step_size_rand_scale = float(stepSizeRandScale)
        cfg.delta0 = float(delta0); cfg.r2p_prob0 = float(R2Pprob0)
        cfg.generated = int(generated); cfg.gen_k = int(gen_k); cfg.warmup_iter = int(warmupIter)
        cfg.waves_per_group = int(waves_per_group)
        cfg.adapt_h = int(bool(adaptH)); cfg.adapt_delta = int(bool(adaptDelta))
        cfg.adapt_h_target = float(adaptHtarget); cfg.adapt_delta_target = float(adaptDeltaTarget)
        cfg.adapt_delta_quantile = float(adaptDeltaQuantile)
        cfg.route = L.route_bits(route); cfg.split_parts = int(split_parts); cfg.split_ks = int(split_ks)
        cfg.adapt_quant_bits = int(adaptQuantBits)
        self.cfg = cfg
        self.dim, self.n_chains = int(dim), int(n_chains)
        self.dg = {L.G_IDENTITY: self.dim, L.G_Q0_SUMSQ: 2, L.G_FIRSTK: int(gen_k), L.G_SW_CONSTRAIN: 2 * self.dim + 1}[int(generated)]
        h = C.c_void_p()
        rc = self.lib.wn_create(C.byref(cfg), C.byref(h))
        if rc:
            msg = self.lib.wn_last_error(None).decode()
            if rc == L.E_UNSUPPORTED:
                raise NotImplementedError(msg)
            raise ValueError(msg) if rc == L.E_ARG else WalnutsError(msg)
        self.h = h
        self._keep = []

    def _check(self, rc):
        if rc:
            msg = self.lib.wn_last_error(self.h).decode()
            if rc == L.E_UNSUPPORTED:
                raise NotImplementedError(msg)
            if rc == L.E_ALLOC:
                raise MemoryError(msg)
            raise ValueError(msg) if rc == L.E_ARG else WalnutsError(f"rc={rc}: {msg}")

    def _dev(self, a, dtype=None):
        torch = self.torch
        dtype = dtype or torch.float64
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device).contiguous()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_target(self, kind, params=None):
        p = self._dev(params).reshape(-1) if params is not None else None
        self.torch.cuda.synchronize(self.device)
        self._check(self.lib.wn_set_target(self.h, int(kind), C.c_void_p(p.data_ptr()) if p is not None else None,
                                           p.numel() if p is not None else 0))
        self.kernel_name = self.lib.wn_kernel_name(self.h).decode()
        return self

    def run(self, q0, seed, n_iter, trace_cap=0):
        """Returns device tensors samples [C, n_iter+1, dg], diag [C, n_iter, 24]
        (+ trace [C, cap, 12+2D], trace_n [C] when trace_cap > 0).  Asynchronous."""
        torch = self.torch
        q0 = self._dev(q0).reshape(self.n_chains, self.dim)
        samples = torch.zeros((self.n_chains, n_iter + 1, self.dg), dtype=torch.float64, device=self.device)
        diag = torch.zeros((self.n_chains, n_iter, L.NDIAG), dtype=torch.float64, device=self.device)
        trace = tn = None
        if trace_cap:
            trace = torch.zeros((self.n_chains, trace_cap, L.TRACE_NSCAL + 2 * self.dim),
                                dtype=torch.float64, device=self.device)
            tn = torch.zeros(self.n_chains, dtype=torch.int32, device=self.device)
        self._check(self.lib.wn_run(self.h, C.c_void_p(q0.data_ptr()), C.c_uint64(int(seed)), int(n_iter),
                                    C.c_void_p(samples.data_ptr()), C.c_void_p(diag.data_ptr()),
                                    C.c_void_p(trace.data_ptr()) if trace is not None else None,
                                    int(trace_cap), C.c_void_p(tn.data_ptr()) if tn is not None else None,
                                    self._stream()))
        self._keep = [q0]
        return (samples, diag, trace, tn) if trace_cap else (samples, diag)

    def run_orbit_stats(self, q0, seed, n_iter):
        """run() with recordOrbitStats: returns samples, diag, orbit_min, orbit_max ([C, n_iter, dg])."""
        torch = self.torch
        q0 = self._dev(q0).reshape(self.n_chains, self.dim)
        samples = torch.zeros((self.n_chains, n_iter + 1, self.dg), dtype=torch.float64, device=self.device)
        diag = torch.zeros((self.n_chains, n_iter, L.NDIAG), dtype=torch.float64, device=self.device)
        omin = torch.zeros((self.n_chains, n_iter, self.dg), dtype=torch.float64, device=self.device)
        omax = torch.zeros_like(omin)
        P = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.wn_run_orbit_stats(self.h, P(q0), C.c_uint64(int(seed)), int(n_iter), P(samples),
                                                P(diag), P(omin), P(omax), self._stream()))
        self._keep = [q0]
        return samples, diag, omin, omax

    def run_external(self, q0, seed, n_iter, fn, poll=32):
        """Run with a caller-evaluated target (WN_T_EXTERNAL): `fn(q)` takes the positions of all
        chains as a device tensor view [C, D] and returns (lp [C], grad [C, D]) device tensors --
        the reference's lpFun plugin (targetDistr.py:18) as a batched callable.  Every chain
        advances by one gradient evaluation per call of fn."""
        torch = self.torch
        q0 = self._dev(q0).reshape(self.n_chains, self.dim)
        samples = torch.zeros((self.n_chains, n_iter + 1, self.dg), dtype=torch.float64, device=self.device)
        diag = torch.zeros((self.n_chains, n_iter, L.NDIAG), dtype=torch.float64, device=self.device)
        ld = C.c_int32()
        self._check(self.lib.wn_split_ld(self.h, C.byref(ld)))
        xq = torch.zeros((self.n_chains, ld.value), dtype=torch.float64, device=self.device)
        xg = torch.zeros_like(xq)
        xlp = torch.zeros(self.n_chains, dtype=torch.float64, device=self.device)
        P = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.wn_split_begin(self.h, P(q0), C.c_uint64(int(seed)), int(n_iter), P(samples), P(diag),
                                            P(xq), P(xg), P(xlp), self._stream()))
        qview, gview = xq[:, :self.dim], xg[:, :self.dim]
        done = C.c_int32(0)
        ticks = 0
        adaptive = self.cfg.integrator != L.I_FIXED
        cap = 64 + 1.05 * n_iter * (1 << self.cfg.M) * (9.0 * (1 << (self.cfg.max_c + 2)) if adaptive else 1.0)
        while done.value < self.n_chains:
            if ticks > cap:
                raise WalnutsError("external-target run exceeded its tick bound (internal error)")
            for _ in range(poll):
                lp, g = fn(qview)
                xlp.copy_(torch.as_tensor(lp, dtype=torch.float64, device=self.device).reshape(self.n_chains))
                gview.copy_(torch.as_tensor(g, dtype=torch.float64, device=self.device).reshape(self.n_chains, self.dim))
                self._check(self.lib.wn_split_tick(self.h, self._stream()))
            ticks += poll
            self._check(self.lib.wn_split_done(self.h, C.byref(done), self._stream()))
        self.ext_ticks = ticks
        self._keep = [q0, xq, xg, xlp]
        return samples, diag

    def check(self):
        """Waits for the last run and raises WalnutsError if a kernel's bounded loop gave up (wn_check_run)."""
        self._check(self.lib.wn_check_run(self.h))

    def last_run_ms(self):
        ms = C.c_float()
        self._check(self.lib.wn_last_run_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_split_ticks(self):
        """split mode: {gradient kernel, chain tick} pairs the last run launched (0 for fused kernels)"""
        n = C.c_int64()
        self._check(self.lib.wn_last_split_ticks(self.h, C.byref(n)))
        return int(n.value)

    def integrator_step(self, q, v, g, Ham0, h, xi, u_r2p=None):
        torch = self.torch
        B, D = self.n_chains, self.dim
        q, v, g = (self._dev(x).reshape(B, D) for x in (q, v, g))
        Ham0 = self._dev(Ham0).reshape(B); h = self._dev(h).reshape(B)
        xi = self._dev(xi, torch.int32).reshape(B)
        u = self._dev(u_r2p if u_r2p is not None else np.zeros(B)).reshape(B)
        q1 = torch.empty_like(q); v1 = torch.empty_like(q); g1 = torch.empty_like(q)
        io = torch.zeros((B, 5), dtype=torch.int32, device=self.device)
        do = torch.zeros((B, 4), dtype=torch.float64, device=self.device)
        P = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.wn_integrator_step(self.h, P(q), P(v), P(g), P(Ham0), P(h), P(xi), P(u),
                                                P(q1), P(v1), P(g1), P(io), P(do), self._stream()))
        torch.cuda.synchronize(self.device)
        return q1, v1, g1, io, do

    def lpfun(self, q):
        torch = self.torch
        q = self._dev(q).reshape(self.n_chains, self.dim)
        lp = torch.empty(self.n_chains, dtype=torch.float64, device=self.device)
        g = torch.empty_like(q)
        P = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.wn_lpfun(self.h, P(q), P(lp), P(g), self._stream()))
        torch.cuda.synchronize(self.device)
        return lp, g

    def rng_tape(self, seed, it, n_u):
        torch = self.torch
        u = torch.empty((self.n_chains, max(n_u, 1)), dtype=torch.float64, device=self.device)
        z = torch.empty((self.n_chains, self.dim), dtype=torch.float64, device=self.device)
        P = lambda t: C.c_void_p(t.data_ptr())
        self._check(self.lib.wn_rng_tape(self.h, C.c_uint64(int(seed)), int(it), int(n_u), P(u), P(z), self._stream()))
        torch.cuda.synchronize(self.device)
        return u[:, :n_u], z

    def close(self):
        if getattr(self, "h", None):
            self.torch.cuda.synchronize(self.device)
            h, self.h = self.h, None
            rc = self.lib.wn_destroy(h)  # reports a run that gave up and was never asked about
            if rc:
                raise WalnutsError(f"rc={rc}: {self.lib.wn_last_error(None).decode()}")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
