"""Macro-step integrator plugins (mirror of WALNUTSpy/adaptiveIntegrators.py).

The reference passes a Python function integrator(q,v,g,Ham0,h,xi,lpFun,delta,auxPar) ->
integratorReturn (adaptiveIntegrators.py:49,65,361).  Here an integrator is a descriptor the
fused kernel switches on; calling it runs one macro step on the GPU through wn_integrator_step,
with the same argument order and return fields.
"""
import numpy as np

from . import _lib as L


class integratorReturn:  # adaptiveIntegrators.py:17-29
    def __init__(self, q1, v1, lp1, grad1, nEvalF, nEvalB, If, Ib, c, lwt, igrConst):
        self.q, self.v, self.lp, self.grad = q1, v1, lp1, grad1
        self.nEvalF, self.nEvalB, self.If, self.Ib, self.c = nEvalF, nEvalB, If, Ib, c
        self.lwt, self.igrConst = lwt, igrConst

    def __str__(self):
        return f"dim: {len(self.q)} If: {self.If} Ib: {self.Ib} c: {self.c}"


class integratorAuxPar:  # adaptiveIntegrators.py:36-44
    def __init__(self, minC=0, maxC=10, R2Pprob0=2.0 / 3.0, maxFPiter=30, FPtol=1.0e-8,
                 FPNewton=False, rescaledGradThresh=5.0):
        self.minC, self.maxC, self.R2Pprob0 = minC, maxC, R2Pprob0
        self.maxFPiter, self.FPtol, self.FPNewton = maxFPiter, FPtol, FPNewton
        self.rescaledGradThresh = rescaledGradThresh


class DeviceIntegrator:
    def __init__(self, name, kind):
        self.name, self.kind = name, kind

    def __call__(self, q, v, g, Ham0, h, xi, lpFun, delta, auxPar, u_r2p=None):
        """One macro step on the GPU.  `u_r2p` replaces the global-stream uniform R2P draws
        (adaptiveIntegrators.py:392); it defaults to a fresh NumPy uniform like the reference."""
        from .engine import Engine
        q = np.asarray(q, dtype=np.float64)
        lpFun.check_dim(q.size)
        if u_r2p is None:
            u_r2p = np.random.uniform()
        e = Engine(q.size, 1, integrator=self.kind, delta0=delta, minC=auxPar.minC, maxC=auxPar.maxC,
                   R2Pprob0=auxPar.R2Pprob0).set_target(lpFun.kind, lpFun.params)
        q1, v1, g1, io, do = e.integrator_step(q, v, g, [Ham0], [h], [xi], [u_r2p])
        io = io.cpu().numpy()[0]; do = do.cpu().numpy()[0]
        igr = float(do[3])  # adaptiveIntegrators.py:59 / 101 / 399 / 424 / 300
        out = integratorReturn(q1.cpu().numpy()[0], v1.cpu().numpy()[0], float(do[0]), g1.cpu().numpy()[0],
                               int(io[0]), int(io[1]), int(io[2]), int(io[3]), int(io[4]), float(do[1]), igr)
        e.close()
        return out

    def __repr__(self):
        return f"<walnuts_amd integrator {self.name}>"


fixedLeapFrog = DeviceIntegrator("fixedLeapFrog", L.I_FIXED)            # adaptiveIntegrators.py:49
adaptLeapFrogD = DeviceIntegrator("adaptLeapFrogD", L.I_ADAPT_D)        # adaptiveIntegrators.py:65
adaptLeapFrogR2P = DeviceIntegrator("adaptLeapFrogR2P", L.I_ADAPT_R2P)  # adaptiveIntegrators.py:361
adaptYoshidaD = DeviceIntegrator("adaptYoshidaD", L.I_ADAPT_YOSHIDA)    # adaptiveIntegrators.py:142
adaptLeapFrogFlowD = DeviceIntegrator("adaptLeapFrogFlowD", L.I_ADAPT_FLOWD)  # adaptiveIntegrators.py:246
