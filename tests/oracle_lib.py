"""ctypes binding of the CPU oracle (oracle/libwalnuts_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the walnuts_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

T_STDGAUSS, T_FUNNEL10, T_DENSEGAUSS, T_DIAGGAUSS, T_CORRGAUSS, T_SMILE, T_FUNNEL1, T_MODFUNNEL = range(8)
T_STOCKWATSON = 11
I_FIXED, I_ADAPT_D, I_ADAPT_R2P, I_ADAPT_YOSHIDA, I_ADAPT_FLOWD = range(5)
G_IDENTITY, G_Q0_SUMSQ, G_FIRSTK, G_SW_CONSTRAIN = range(4)
RNG_PHILOX, RNG_TAPE = 0, 1
NDIAG = 24
TRACE_NSCAL = 12

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class WnoConfig(C.Structure):
    _fields_ = [
        ("d", C.c_int), ("target", C.c_int), ("tpar", c_double_p), ("integrator", C.c_int),
        ("H0", C.c_double), ("stepSizeRandScale", C.c_double), ("delta0", C.c_double),
        ("numIter", C.c_int), ("warmupIter", C.c_int), ("M", C.c_int),
        ("minC", C.c_int), ("maxC", C.c_int), ("R2Pprob0", C.c_double),
        ("generated", C.c_int), ("gen_k", C.c_int),
        ("adaptH", C.c_int), ("adaptHtarget", C.c_double), ("adaptDelta", C.c_int),
        ("adaptDeltaTarget", C.c_double), ("adaptDeltaQuantile", C.c_double),
        ("rng_mode", C.c_int), ("seed", C.c_uint64), ("chain", C.c_uint32),
        ("tape_u", c_double_p), ("tape_u_stride", C.c_int), ("tape_z", c_double_p),
        ("orbitMin", c_double_p), ("orbitMax", c_double_p),
        ("trace", c_double_p), ("trace_cap", C.c_int), ("adaptQuantBits", C.c_int),
    ]


def build(force=False, asan=False):
    name = "libwalnuts_oracle_asan.so" if asan else "libwalnuts_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "walnuts_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return path


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.wno_walnuts.restype = C.c_int
        L.wno_walnuts.argtypes = [C.POINTER(WnoConfig), c_double_p, c_double_p, c_double_p, c_int_p]
        L.wno_gen_dim.restype = C.c_int
        L.wno_gen_dim.argtypes = [C.POINTER(WnoConfig)]
        L.wno_lpfun.restype = C.c_int
        L.wno_lpfun.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        L.wno_subtreeplan.restype = C.c_int
        L.wno_subtreeplan.argtypes = [C.c_int, c_int_p]
        L.wno_stopcondition.restype = C.c_int
        L.wno_stopcondition.argtypes = [C.c_int] + [c_double_p] * 4
        L.wno_integrator.restype = C.c_int
        L.wno_integrator.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                     c_double_p, C.c_double, C.c_double, C.c_int, C.c_double,
                                     C.c_int, C.c_int, C.c_double, C.c_double, c_double_p,
                                     c_double_p, c_double_p, c_int_p, c_double_p]
        L.wno_philox4x32_10.restype = None
        L.wno_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.wno_rng_uniform.restype = C.c_double
        L.wno_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.wno_rng_normal.restype = C.c_double
        L.wno_rng_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.wno_rng_tape.restype = None
        L.wno_rng_tape.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, c_double_p,
                                   C.c_int, c_double_p]
        L.wno_p2quantile.restype = None
        L.wno_p2quantile.argtypes = [C.c_double, C.c_int, c_double_p, c_double_p]
        L.wno_np_quantile.restype = C.c_double
        L.wno_np_quantile.argtypes = [c_double_p, C.c_int, C.c_double]
        L.wno_version.restype = C.c_char_p
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().wno_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def rng_tape(seed, chain, it, nu, nz):
    u = np.empty(max(nu, 1)); z = np.empty(max(nz, 1))
    lib().wno_rng_tape(seed, chain, it, nu, _dp(u), nz, _dp(z))
    return u[:nu], z[:nz]


def lpfun(target, q, tpar=None):
    q = _f64(q); d = q.size
    tp = _f64(tpar) if tpar is not None else None
    lp = C.c_double(); g = np.empty(d)
    rc = lib().wno_lpfun(target, d, _dp(tp), _dp(q), C.byref(lp), _dp(g))
    if rc:
        raise ValueError(f"wno_lpfun rc={rc}")
    return lp.value, g


def subtreeplan(nleaf):
    out = np.zeros((max(nleaf - 1, 1), 2), dtype=np.int32)
    k = lib().wno_subtreeplan(nleaf, out.ctypes.data_as(c_int_p))
    return out[:k]


def stopcondition(qm, vm, qp, vp):
    qm, vm, qp, vp = map(_f64, (qm, vm, qp, vp))
    return bool(lib().wno_stopcondition(qm.size, _dp(qm), _dp(vm), _dp(qp), _dp(vp)))


def integrator(kind, target, q, v, g, Ham0, h, xi, delta, tpar=None, minC=0, maxC=10,
               R2Pprob0=2.0 / 3.0, u_r2p=0.0):
    q, v, g = map(_f64, (q, v, g)); d = q.size
    tp = _f64(tpar) if tpar is not None else None
    q1 = np.empty(d); v1 = np.empty(d); g1 = np.empty(d)
    io = np.zeros(5, dtype=np.int32); do = np.zeros(3)
    rc = lib().wno_integrator(kind, target, d, _dp(tp), _dp(q), _dp(v), _dp(g), Ham0, h, xi, delta,
                              minC, maxC, R2Pprob0, u_r2p, _dp(q1), _dp(v1), _dp(g1),
                              io.ctypes.data_as(c_int_p), _dp(do))
    if rc:
        raise ValueError(f"wno_integrator rc={rc}")
    return dict(q=q1, v=v1, grad=g1, nEvalF=int(io[0]), nEvalB=int(io[1]), If=int(io[2]),
                Ib=int(io[3]), c=int(io[4]), lp=do[0], lwt=do[1], igrConst=do[2])


def p2quantile(prob, xs):
    xs = _f64(xs); out = np.empty_like(xs)
    lib().wno_p2quantile(prob, xs.size, _dp(xs), _dp(out))
    return out


def np_quantile(a, q):
    a = _f64(a)
    return lib().wno_np_quantile(_dp(a), a.size, q)


def walnuts(target, q0, integrator=I_FIXED, tpar=None, H0=0.2, stepSizeRandScale=0.2, delta0=0.05,
            numIter=10, warmupIter=0, M=10, minC=0, maxC=10, R2Pprob0=2.0 / 3.0,
            generated=G_IDENTITY, gen_k=0, seed=0, chain=0, tape_u=None, tape_z=None,
            trace_cap=0, adaptH=True, adaptHtarget=0.8, adaptDelta=True, adaptDeltaTarget=0.6,
            adaptDeltaQuantile=0.9, recordOrbitStats=False, adaptQuantBits=0):
    """One chain of the oracle's WALNUTS.  Returns dict(samples (dg,numIter+1), diagnostics
    (numIter,24), trace (ncalls, 12+2d), n_u (numIter,))."""
    q0 = _f64(q0); d = q0.size
    cfg = WnoConfig()
    cfg.d = d; cfg.target = target
    tp = _f64(tpar) if tpar is not None else None
    cfg.tpar = _dp(tp)
    cfg.integrator = integrator
    cfg.H0 = H0; cfg.stepSizeRandScale = stepSizeRandScale; cfg.delta0 = delta0
    cfg.numIter = numIter; cfg.warmupIter = warmupIter; cfg.M = M
    cfg.minC = minC; cfg.maxC = maxC; cfg.R2Pprob0 = R2Pprob0
    cfg.generated = generated; cfg.gen_k = gen_k
    cfg.adaptH = int(adaptH); cfg.adaptHtarget = adaptHtarget; cfg.adaptDelta = int(adaptDelta)
    cfg.adaptDeltaTarget = adaptDeltaTarget; cfg.adaptDeltaQuantile = adaptDeltaQuantile
    cfg.seed = seed; cfg.chain = chain
    cfg.adaptQuantBits = int(adaptQuantBits)
    if tape_u is not None:
        tu = _f64(tape_u); tz = _f64(tape_z)
        assert tu.shape[0] == numIter and tz.shape == (numIter, d)
        cfg.rng_mode = RNG_TAPE
        cfg.tape_u = _dp(tu); cfg.tape_u_stride = tu.shape[1]; cfg.tape_z = _dp(tz)
    else:
        cfg.rng_mode = RNG_PHILOX
    ncol = TRACE_NSCAL + 2 * d
    tr = np.zeros((max(trace_cap, 1), ncol)) if trace_cap else None
    cfg.trace = _dp(tr); cfg.trace_cap = trace_cap
    dg = lib().wno_gen_dim(C.byref(cfg))
    omin = omax = None
    if recordOrbitStats:
        omin = np.zeros((numIter, dg)); omax = np.zeros((numIter, dg))
        cfg.orbitMin = _dp(omin); cfg.orbitMax = _dp(omax)
    samples = np.zeros((numIter + 1, dg)); diag = np.zeros((numIter, NDIAG))
    nu = np.zeros(numIter, dtype=np.int32)
    rc = lib().wno_walnuts(C.byref(cfg), _dp(q0), _dp(samples), _dp(diag),
                           nu.ctypes.data_as(c_int_p))
    if rc < 0:
        raise RuntimeError(f"wno_walnuts rc={rc}")
    out = dict(samples=samples.T.copy(), diagnostics=diag, n_u=nu, ncalls=rc)
    if recordOrbitStats:
        out["orbitMin"] = omin.T.copy(); out["orbitMax"] = omax.T.copy()
    if tr is not None:
        out["trace"] = tr[:min(rc, trace_cap)]
    return out
