"""walnuts_amd -- MI355X-native batched WALNUTS/NUTS leapfrog-and-orbit engine.

Drop-in for the hot path of bob-carpenter/walnuts (WALNUTSpy.WALNUTS + adaptiveIntegrators +
targetDistr): the same call shapes, executed by hand-written gfx950 HIP kernels behind a C ABI
(include/walnuts_amd.h).  There is no CPU fallback.
"""
from . import _lib
from . import adaptiveIntegrators, targetDistr, targets
from .walnuts import walnuts, walnuts_step, walnuts_batch, walnuts_step_batch, walnuts_stan, walnuts_device
from .WALNUTS import WALNUTS, Generated, identity, q0_sumsq, firstk, sw_constrain
from .engine import Engine, WalnutsError

__all__ = ["WALNUTS", "walnuts", "walnuts_step", "walnuts_batch", "walnuts_step_batch", "walnuts_stan", "walnuts_device", "targets", "Engine", "WalnutsError", "adaptiveIntegrators", "targetDistr", "Generated",
           "identity", "q0_sumsq", "firstk", "sw_constrain"]
