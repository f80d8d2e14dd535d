"""ctypes binding of libwalnuts_amd.so (the C ABI declared in include/walnuts_amd.h).

There is no CPU fallback: if the HIP library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WN_LIB=<name> selects a diagnostic build made by tools/build_variant.sh (profiling only)
_VARIANT = os.environ.get("WN_LIB", "")
LIB_PATH = os.path.join(_HERE, f"libwalnuts_amd_{_VARIANT}.so" if _VARIANT else "libwalnuts_amd.so")

WN_ABI_VERSION = 5
NDIAG = 24
TRACE_NSCAL = 12

# enums of include/walnuts_amd.h
T_STDGAUSS, T_FUNNEL10, T_DENSEGAUSS, T_DIAGGAUSS, T_CORRGAUSS, T_SMILE, T_FUNNEL1, T_MODFUNNEL = range(8)
T_EXTERNAL = 10
T_STOCKWATSON = 11
I_FIXED, I_ADAPT_D, I_ADAPT_R2P, I_ADAPT_YOSHIDA, I_ADAPT_FLOWD = range(5)
G_IDENTITY, G_Q0_SUMSQ, G_FIRSTK, G_SW_CONSTRAIN = range(4)
E_ARG, E_UNSUPPORTED, E_HIP, E_NOTARGET, E_ALLOC = -1, -2, -3, -4, -5
# WN_ROUTE_* (wn_config.route): which of two equivalent kernels runs a configuration; 0 = the library chooses
ROUTE = {name: 1 << k for k, name in enumerate(
    ["ORBIT_LEGACY", "ORBIT_NOWIDE", "ORBIT_WIDE_ADAPT", "ORBIT_ONEWAVE", "ORBIT_EXACT", "ORBIT2_NOADAPT", "ORBIT2_ANYC", "DIAG_COOP", "DENSE_COOP",
     "FAST_LEGACY", "COOP_LEGACY", "FAST4_WHOLE", "FLOW_NOCOOP", "FORCE_WAVE", "LANE_GAUSS", "SPLIT_EAGER", "WSTEP_LEGACY"])}


def route_bits(route):
    """int, a name ("ORBIT_LEGACY"), or an iterable of names -> the WN_ROUTE_* bit mask"""
    if route is None:
        return 0
    if isinstance(route, int):
        return route
    if isinstance(route, str):
        route = [route]
    r = 0
    for name in route:
        r |= ROUTE[name[9:] if name.startswith("WN_ROUTE_") else name]
    return r


class WnConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32),
        ("n_chains", C.c_int32), ("chain_offset", C.c_int64), ("integrator", C.c_int32),
        ("M", C.c_int32), ("min_c", C.c_int32), ("max_c", C.c_int32),
        ("H0", C.c_double), ("step_size_rand_scale", C.c_double), ("delta0", C.c_double),
        ("r2p_prob0", C.c_double), ("generated", C.c_int32), ("gen_k", C.c_int32),
        ("warmup_iter", C.c_int32), ("waves_per_group", C.c_int32),
        ("adapt_h", C.c_int32), ("adapt_delta", C.c_int32), ("adapt_h_target", C.c_double),
        ("adapt_delta_target", C.c_double), ("adapt_delta_quantile", C.c_double),
        ("route", C.c_uint32), ("split_parts", C.c_int32), ("split_ks", C.c_int32), ("adapt_quant_bits", C.c_int32),
    ]


# every symbol include/walnuts_amd.h declares
SYMBOLS = ["wn_abi_version", "wn_create", "wn_set_target", "wn_run", "wn_run_orbit_stats", "wn_split_ld", "wn_split_begin", "wn_split_tick", "wn_split_done", "wn_integrator_step",
           "wn_lpfun", "wn_lpfun_kind", "wn_wstep_create", "wn_wstep_begin", "wn_wstep_extend", "wn_wstep_commit", "wn_wstep_result", "wn_wstep_run_device_rng", "wn_wstep_set_stats", "wn_wstep_set_route",
           "wn_wstep_ext_bind", "wn_wstep_ext_pending", "wn_wstep_ext_waiting", "wn_wstep_ext_feed", "wn_wstep_last_error", "wn_wstep_destroy", "wn_rng_tape", "wn_last_run_ms", "wn_check_run", "wn_last_split_ticks", "wn_debug_poison_lds", "wn_debug_cluster_drop", "wn_kernel_name", "wn_last_error",
           "wn_destroy"]

_lib = None


def lib():
    """Load the HIP library; raises ImportError (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the walnuts_amd engine is HIP-only and has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C walnuts_amd/csrc`.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.wn_abi_version.restype = C.c_int
    L.wn_create.restype = C.c_int
    L.wn_create.argtypes = [C.POINTER(WnConfig), C.POINTER(vp)]
    L.wn_set_target.restype = C.c_int
    L.wn_set_target.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.wn_run.restype = C.c_int
    L.wn_run.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp, vp, vp, C.c_int32, vp, vp]
    L.wn_run_orbit_stats.restype = C.c_int
    L.wn_run_orbit_stats.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp, vp, vp, vp, vp]
    L.wn_split_begin.restype = C.c_int
    L.wn_split_begin.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.wn_split_ld.restype = C.c_int
    L.wn_split_ld.argtypes = [vp, C.POINTER(C.c_int32)]
    L.wn_split_tick.restype = C.c_int
    L.wn_split_tick.argtypes = [vp, vp]
    L.wn_split_done.restype = C.c_int
    L.wn_split_done.argtypes = [vp, C.POINTER(C.c_int32), vp]
    L.wn_integrator_step.restype = C.c_int
    L.wn_integrator_step.argtypes = [vp] * 14
    L.wn_lpfun.restype = C.c_int
    L.wn_lpfun.argtypes = [vp, vp, vp, vp, vp]
    L.wn_lpfun_kind.restype = C.c_int
    L.wn_lpfun_kind.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.wn_wstep_create.restype = C.c_int
    L.wn_wstep_create.argtypes = [C.c_int32, C.c_int32, vp, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, vp, C.c_double,
                                  C.c_double, C.POINTER(vp)]
    for name, nargs in (("wn_wstep_begin", 4), ("wn_wstep_commit", 5), ("wn_wstep_result", 3), ("wn_wstep_ext_bind", 4),
                        ("wn_wstep_ext_feed", 5)):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp] * nargs
    L.wn_wstep_extend.restype = C.c_int
    L.wn_wstep_extend.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.wn_wstep_ext_pending.restype = C.c_int
    L.wn_wstep_ext_pending.argtypes = [vp, C.POINTER(C.c_int32), vp]
    L.wn_wstep_run_device_rng.restype = C.c_int
    L.wn_wstep_run_device_rng.argtypes = [vp, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]
    L.wn_wstep_set_stats.restype = C.c_int
    L.wn_wstep_set_stats.argtypes = [vp, vp]
    L.wn_wstep_set_route.restype = C.c_int
    L.wn_wstep_set_route.argtypes = [vp, C.c_uint32]
    L.wn_wstep_ext_waiting.restype = C.c_int
    L.wn_wstep_ext_waiting.argtypes = [vp, vp, vp]
    L.wn_wstep_last_error.restype = C.c_char_p
    L.wn_wstep_last_error.argtypes = [vp]
    L.wn_wstep_destroy.restype = C.c_int
    L.wn_wstep_destroy.argtypes = [vp]
    L.wn_rng_tape.restype = C.c_int
    L.wn_rng_tape.argtypes = [vp, C.c_uint64, C.c_int32, C.c_int32, vp, vp, vp]
    L.wn_check_run.restype = C.c_int
    L.wn_check_run.argtypes = [vp]
    L.wn_last_run_ms.restype = C.c_int
    L.wn_last_run_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.wn_last_split_ticks.restype = C.c_int
    L.wn_last_split_ticks.argtypes = [vp, C.POINTER(C.c_int64)]
    L.wn_debug_poison_lds.restype = C.c_int
    L.wn_debug_poison_lds.argtypes = [C.c_int32, C.c_uint64, vp]
    L.wn_debug_cluster_drop.restype = C.c_int
    L.wn_debug_cluster_drop.argtypes = [vp, C.c_int32]
    L.wn_kernel_name.restype = C.c_char_p
    L.wn_kernel_name.argtypes = [vp]
    L.wn_last_error.restype = C.c_char_p
    L.wn_last_error.argtypes = [vp]
    L.wn_destroy.restype = C.c_int
    L.wn_destroy.argtypes = [vp]
    if L.wn_abi_version() != WN_ABI_VERSION:
        raise ImportError("libwalnuts_amd.so ABI version mismatch; rebuild it")
    _lib = L
    return L
