"""WALNUTS(...) with the reference's signature (WALNUTSpy/WALNUTS.py:111-129), batched on the GPU.

Differences a caller sees:
  * lpFun is a device descriptor (walnuts_amd.targetDistr: fused in the kernel), a
    targetDistr.batched(fn) device callable evaluated for all chains between chain-kernel launches,
    or any reference-style host callable lpFun(q) -> [lp, grad] (slow: one host call per chain and
    gradient evaluation); integrator is a device descriptor (walnuts_amd.adaptiveIntegrators);
  * q0 may be (D,) -- one chain, returns exactly the reference's shapes -- or (C, D) for C
    independent chains, returning arrays with a leading chain axis;
  * randomness comes from the engine's counter-based stream keyed by (seed, chain, iteration)
    instead of NumPy's global state; pass seed= and chain_offset=.
Warm-up adaptation of H and delta (WALNUTS.py:701-712, P2quantile.py) runs per chain on the device.
"""
import sys

import numpy as np

from . import _lib as L
from . import adaptiveIntegrators as ai
from .engine import Engine
from .targetDistr import BatchedTarget, DeviceTarget, host_callable


class Generated:
    """Device forms of the `generated` argument (WALNUTS.py:113)."""

    def __init__(self, kind, k=0):
        self.kind, self.k = kind, k


identity = Generated(L.G_IDENTITY)
q0_sumsq = Generated(L.G_Q0_SUMSQ)  # mainGaussESS.py:29-30  [q[0], sum(q*q)]
# mainSW.py:28-29 generatedInnov(q) = param_constrain(q, include_tp=True) for sw_innov.stan, on the device (needs the
# targetDistr.stockWatson target): q followed by sigma, gpri, z[T-1], x[T], tau[T]
sw_constrain = Generated(L.G_SW_CONSTRAIN)


def firstk(k):
    return Generated(L.G_FIRSTK, k)  # mainFunnel.py:17-18


def WALNUTS(lpFun, q0, generated=identity, integrator=ai.fixedLeapFrog, H0=0.2,
            stepSizeRandScale=0.2, delta0=0.05, numIter=2000, warmupIter=1000, M=10,
            igrAux=None, adaptH=True, adaptHtarget=0.8, adaptDelta=True, adaptDeltaTarget=0.6,
            adaptDeltaQuantile=0.9, recordOrbitStats=False, *, seed=0, chain_offset=0, device=None,
            return_torch=False, trace_cap=0, route=0, adaptQuantBits=0, split_parts=0, split_ks=0):
    """Keyword-only extras: `route` (0 = the library chooses the kernel; WN_ROUTE_* names select the other implementation of the
    same function, include/walnuts_amd.h), `adaptQuantBits` (0 = the reference's arithmetic; b: adapted H / delta rounded to b
    mantissa bits, opt-in), `split_parts` / `split_ks` (split mode sub-batches)."""
    if igrAux is None:
        igrAux = ai.integratorAuxPar()
    if not isinstance(lpFun, (DeviceTarget, BatchedTarget)):
        if not callable(lpFun):
            raise TypeError("lpFun must be a walnuts_amd.targetDistr descriptor or a callable lpFun(q) -> [lp, grad]")
        lpFun = host_callable(lpFun)  # the reference's plugin contract, evaluated on the host
    if not isinstance(integrator, ai.DeviceIntegrator):
        raise TypeError("integrator must be one of walnuts_amd.adaptiveIntegrators.{fixedLeapFrog,"
                        "adaptLeapFrogD,adaptLeapFrogR2P,adaptYoshidaD,adaptLeapFrogFlowD}")
    # the reference sys.exit()s on bad adaptation targets whatever warmupIter is (WALNUTS.py:140,146)
    if adaptH and (adaptHtarget < 0.0 or adaptHtarget > 1.0):
        sys.exit("bad adaptHtarget")
    if adaptDelta and adaptDeltaTarget < 0.0:
        sys.exit("bad adaptDeltaTarget")
    host_gen = None
    if not isinstance(generated, Generated):
        if not callable(generated):
            raise TypeError("generated must be callable or a walnuts_amd.WALNUTS.Generated")
        host_gen, generated = generated, identity  # applied to the stored draws on the host
    q0a = q0 if hasattr(q0, "data_ptr") else np.asarray(q0, dtype=np.float64)
    single = q0a.ndim == 1
    C, D = (1, q0a.shape[0]) if single else q0a.shape
    lpFun.check_dim(D)
    # Beyond D = 4096 a chain occupies a cluster of ceil(D/4096) workgroups that must all be resident (256 CUs):
    # more chains than fit run in consecutive launches; the RNG is keyed by the global chain id, so the
    # result does not depend on the split.
    if isinstance(lpFun, DeviceTarget) and lpFun.kind in (L.T_STDGAUSS, L.T_DIAGGAUSS) and D > 4096 and not single:
        per = max(1, 256 // (-(-D // 4096)))
        if C > per:
            kw = dict(generated=host_gen if host_gen is not None else generated, integrator=integrator, H0=H0,
                      stepSizeRandScale=stepSizeRandScale, delta0=delta0, numIter=numIter, warmupIter=warmupIter, M=M,
                      igrAux=igrAux, adaptH=adaptH, adaptHtarget=adaptHtarget, adaptDelta=adaptDelta,
                      adaptDeltaTarget=adaptDeltaTarget, adaptDeltaQuantile=adaptDeltaQuantile,
                      recordOrbitStats=recordOrbitStats, seed=seed, device=device, return_torch=return_torch,
                      trace_cap=trace_cap, route=route, adaptQuantBits=adaptQuantBits)
            parts = [WALNUTS(lpFun, q0a[lo:lo + per], chain_offset=chain_offset + lo, **kw) for lo in range(0, C, per)]
            import torch as _t
            return tuple((_t.cat(list(x)) if return_torch else np.concatenate(list(x))) for x in zip(*parts))
    eng = Engine(D, C, integrator=integrator.kind, H0=H0, stepSizeRandScale=stepSizeRandScale,
                 delta0=delta0, M=M, minC=igrAux.minC, maxC=igrAux.maxC, R2Pprob0=igrAux.R2Pprob0,
                 generated=generated.kind, gen_k=generated.k, warmupIter=warmupIter,
                 chain_offset=chain_offset, device=device, adaptH=adaptH, adaptHtarget=adaptHtarget,
                 adaptDelta=adaptDelta, adaptDeltaTarget=adaptDeltaTarget,
                 adaptDeltaQuantile=adaptDeltaQuantile, route=route, adaptQuantBits=adaptQuantBits, split_parts=split_parts,
                 split_ks=split_ks)
    if isinstance(lpFun, BatchedTarget):
        if recordOrbitStats or trace_cap:
            raise NotImplementedError("recordOrbitStats / trace need a device target descriptor")
        eng.set_target(L.T_EXTERNAL)
        out = eng.run_external(q0a, seed, numIter, lpFun.fn)
    elif recordOrbitStats:
        eng.set_target(lpFun.kind, lpFun.params)
        if host_gen is not None:
            raise NotImplementedError("recordOrbitStats needs a device form of generated (identity, firstk(k), q0_sumsq, sw_constrain)")
        samples, diag, omin, omax = eng.run_orbit_stats(q0a, seed, numIter)
        out = (samples, diag, omin.transpose(1, 2), omax.transpose(1, 2))  # (dg, numIter) per chain (WALNUTS.py:183-184)
    else:
        eng.set_target(lpFun.kind, lpFun.params)
        out = eng.run(q0a, seed, numIter, trace_cap=trace_cap)
    samples, diag = out[0], out[1]
    samples = samples.transpose(1, 2)  # [C, dg, numIter+1] as the reference's (dg, numIter+1)
    if return_torch:
        eng.torch.cuda.synchronize(eng.device)
        if host_gen is not None:  # a host callable sees host arrays; its results go back to the device
            sh = samples.cpu().numpy()
            sh = np.stack([np.stack([np.asarray(host_gen(sh[c, :, i])) for i in range(numIter + 1)], axis=1)
                           for c in range(C)])
            samples = eng.torch.as_tensor(sh, device=eng.device)
        res = (samples, diag) + tuple(out[2:])
        eng.close()
        return res
    samples = samples.cpu().numpy(); diag = diag.cpu().numpy()
    extra = tuple(t.cpu().numpy() for t in out[2:])
    eng.close()
    if host_gen is not None:
        samples = np.stack([np.stack([np.asarray(host_gen(samples[c, :, i])) for i in range(numIter + 1)], axis=1)
                            for c in range(C)])
    if single:
        return (samples[0], diag[0]) + tuple(x[0] for x in extra)
    return (samples, diag) + extra
