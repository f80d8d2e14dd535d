"""ctypes driver for tests/hostsim (CPU instantiation of the engine's chain state machine).

Test tool only -- see tests/hostsim/hostsim.cpp.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "hostsim", "hostsim.cpp")
_SO = os.path.join(_HERE, "hostsim", "libhostsim.so")
_CSRC = os.path.join(_ROOT, "walnuts_amd", "csrc")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class Params(C.Structure):
    _fields_ = [
        ("D", C.c_int), ("C", C.c_int), ("mode", C.c_int),
        ("integrator", C.c_int), ("M", C.c_int), ("numIter", C.c_int), ("minC", C.c_int),
        ("maxC", C.c_int), ("generated", C.c_int), ("gen_k", C.c_int), ("dg", C.c_int),
        ("H", C.c_double), ("s", C.c_double), ("delta", C.c_double), ("R2Pprob0", C.c_double),
        ("logP", C.c_double), ("log1mP", C.c_double), ("wtSumThresh", C.c_double),
        ("seed", C.c_uint64), ("chain0", C.c_uint32),
        ("q0", dp), ("samples", dp), ("diag", dp), ("trace", dp), ("trace_cap", C.c_int),
        ("trace_n", ip), ("work", dp), ("tpar", dp),
        ("kq", dp), ("kv", dp), ("kg", dp), ("kham0", dp), ("kh", dp), ("ku_r2p", dp),
        ("kxi", ip), ("kq1", dp), ("kv1", dp), ("kg1", dp), ("kdout", dp), ("kiout", ip),
        ("warmupIter", C.c_int), ("adaptH", C.c_int), ("adaptDelta", C.c_int),
        ("adaptHtarget", C.c_double), ("adaptDeltaTarget", C.c_double), ("adaptDeltaQuantile", C.c_double),
        ("facs", dp), ("adaptQuantBits", C.c_int), ("omin", dp), ("omax", dp), ("xq", dp), ("xg", dp), ("xlp", dp), ("sp_done", C.c_void_p), ("cl_g", C.c_int), ("cl_red", dp), ("cl_cnt", C.c_void_p), ("cl_abort", C.c_void_p), ("cl_debug_drop", C.c_int), ("dynq", C.c_void_p), ("stamps", C.c_void_p),
    ]


def build():
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                               "-I" + _CSRC, "-o", _SO, _SRC])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.hostsim_run.argtypes = [C.POINTER(Params), C.c_int]
        _lib.hostsim_run.restype = C.c_int
        assert _lib.hostsim_params_size() == C.sizeof(Params)
    return _lib


def _p(a, t=dp):
    return a.ctypes.data_as(t) if a is not None else None


def walnuts(target, q0, integrator=0, tpar=None, H0=0.2, stepSizeRandScale=0.2, delta0=0.05,
            numIter=10, M=10, minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, seed=0,
            chain0=0, trace_cap=0, warmupIter=0, adaptH=True, adaptHtarget=0.8, adaptDelta=True,
            adaptDeltaTarget=0.6, adaptDeltaQuantile=0.9, recordOrbitStats=False, adaptQuantBits=0):
    q0 = np.ascontiguousarray(np.atleast_2d(np.asarray(q0, float)))
    Cn, D = q0.shape
    prm = Params()
    prm.adaptQuantBits = adaptQuantBits
    prm.D = D; prm.C = Cn; prm.mode = 0
    prm.integrator = integrator; prm.M = M; prm.numIter = numIter; prm.minC = minC; prm.maxC = maxC
    prm.generated = generated; prm.gen_k = gen_k
    dg = {0: D, 1: 2, 2: gen_k}[generated]
    prm.dg = dg
    prm.H = H0; prm.s = stepSizeRandScale; prm.delta = delta0; prm.R2Pprob0 = R2Pprob0
    prm.logP = math.log(R2Pprob0); prm.log1mP = math.log(1.0 - R2Pprob0)
    prm.wtSumThresh = math.exp(-699.0)
    prm.seed = seed; prm.chain0 = chain0
    if target == 0:  # stdGauss == diag with a = 1
        tpar = np.ones(D)
    tp = np.ascontiguousarray(np.asarray(tpar, float)) if tpar is not None else None
    samples = np.zeros((Cn, numIter + 1, dg)); diag = np.zeros((Cn, numIter, 24))
    trace = np.zeros((Cn, max(trace_cap, 1), 12 + 2 * D)); tn = np.zeros(Cn, dtype=np.int32)
    prm.q0 = _p(q0); prm.samples = _p(samples); prm.diag = _p(diag)
    prm.trace = _p(trace); prm.trace_cap = trace_cap; prm.trace_n = _p(tn, ip)
    prm.tpar = _p(tp)
    prm.warmupIter = warmupIter; prm.adaptH = int(adaptH); prm.adaptDelta = int(adaptDelta)
    prm.adaptHtarget = adaptHtarget; prm.adaptDeltaTarget = adaptDeltaTarget
    prm.adaptDeltaQuantile = adaptDeltaQuantile
    facs = np.zeros((Cn, 2, max(warmupIter, 1)))
    prm.facs = _p(facs)
    omin = omax = None
    if recordOrbitStats:
        omin = np.zeros((Cn, numIter, dg)); omax = np.zeros((Cn, numIter, dg))
        prm.omin = _p(omin); prm.omax = _p(omax)
    rc = lib().hostsim_run(C.byref(prm), target)
    if rc:
        raise RuntimeError(f"hostsim rc={rc}")
    if recordOrbitStats:
        return dict(samples=samples.transpose(0, 2, 1).copy(), diagnostics=diag, trace=trace, trace_n=tn,
                    orbitMin=omin.transpose(0, 2, 1).copy(), orbitMax=omax.transpose(0, 2, 1).copy())
    return dict(samples=samples.transpose(0, 2, 1).copy(), diagnostics=diag, trace=trace, trace_n=tn)


# ---- role-split headline path: the bookkeeper of walnuts_amd/csrc/wn_fast.h on the CPU (tests/hostsim/fastsim.cpp)
_FSRC = os.path.join(_HERE, "hostsim", "fastsim.cpp")
_FSO = os.path.join(_HERE, "hostsim", "libfastsim.so")
_flib = None


def fast_lib():
    global _flib
    if _flib is None:
        deps = [_FSRC] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
        if not os.path.exists(_FSO) or os.path.getmtime(_FSO) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                                   "-I" + _CSRC, "-o", _FSO, _FSRC])
        _flib = C.CDLL(_FSO)
        _flib.fastsim_run2.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
        _flib.fastsim_run2.restype = C.c_long
        assert _flib.fastsim_params_size() == C.sizeof(Params)
    return _flib


def fast_walnuts(target, q0, tpar=None, H0=0.2, stepSizeRandScale=0.2, delta0=0.05, numIter=10, M=10, generated=0, gen_k=0,
                 seed=0, chain0=0, nlc=2, trace_cap=0, lag=2, warmupIter=0, adaptH=True, adaptHtarget=0.8, adaptDelta=True,
                 adaptDeltaTarget=0.6, adaptDeltaQuantile=0.9, adaptQuantBits=0):
    """fixedLeapFrog on a std / diagonal / dense Gaussian through FastBookT<lag> + the scalar T-wave stand-in
    (lag 2: wn_k_fast.hip, lag 3: wn_k_fast4.hip)."""
    q0 = np.ascontiguousarray(np.atleast_2d(np.asarray(q0, float)))
    Cn, D = q0.shape
    prm = Params()
    prm.adaptQuantBits = adaptQuantBits
    prm.D = D; prm.C = Cn; prm.mode = 0
    prm.integrator = 0; prm.M = M; prm.numIter = numIter; prm.maxC = 10
    prm.generated = generated; prm.gen_k = gen_k
    dg = {0: D, 1: 2, 2: gen_k}[generated]
    prm.dg = dg
    prm.H = H0; prm.s = stepSizeRandScale; prm.delta = delta0; prm.R2Pprob0 = 2.0 / 3.0
    prm.wtSumThresh = math.exp(-699.0)
    prm.seed = seed; prm.chain0 = chain0
    tp = np.ascontiguousarray(np.asarray(tpar, float)) if tpar is not None else None
    samples = np.zeros((Cn, numIter + 1, dg)); diag = np.zeros((Cn, numIter, 24))
    prm.q0 = _p(q0); prm.samples = _p(samples); prm.diag = _p(diag); prm.tpar = _p(tp)
    trace = tn = None
    if trace_cap:
        trace = np.zeros((Cn, trace_cap, 12 + 2 * D)); tn = np.zeros(Cn, dtype=np.int32)
        prm.trace = _p(trace); prm.trace_cap = trace_cap
        prm.trace_n = _p(tn, ip)
    # warm-up adaptation of H and delta (WALNUTS.py:701-712): the two-stream form only (lag = 4, the headline kernel's bookkeeper)
    assert warmupIter == 0 or lag == 4
    prm.warmupIter = warmupIter if (adaptH or adaptDelta) else 0
    prm.adaptH = int(adaptH); prm.adaptDelta = int(adaptDelta)
    prm.adaptHtarget = adaptHtarget; prm.adaptDeltaTarget = adaptDeltaTarget; prm.adaptDeltaQuantile = adaptDeltaQuantile
    facs = np.zeros((Cn, 2, max(warmupIter, 1)))
    prm.facs = _p(facs)
    idle = C.c_double(0.0)
    ticks = fast_lib().fastsim_run2(C.byref(prm), target, nlc, C.byref(idle), lag)
    if ticks < 0:
        raise RuntimeError("fastsim: a chain exceeded its tick bound")
    return dict(samples=samples.transpose(0, 2, 1).copy(), diagnostics=diag, ticks=ticks, idle_frac=idle.value,
                trace=trace, trace_n=tn)
