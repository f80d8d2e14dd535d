"""Device mirrors of the (lpdf, grad) pairs in the reference's test/targets.py, for walnuts().
Each `*_lpdf` / `*_grad` name is bound to the same descriptor; walnuts() checks that the pair
belongs together."""
from . import _lib as L
from .targetDistr import DeviceTarget

T_TT_CORRNORMAL, T_TT_FUNNEL = 8, 9

_std = DeviceTarget("standard_normal", L.T_STDGAUSS)                    # test/targets.py:4-7
_corr = DeviceTarget("correlated_normal", T_TT_CORRNORMAL, 2)           # test/targets.py:9-15
_rosen = DeviceTarget("rosenbrock", L.T_SMILE, 2)                       # test/targets.py:17-21
_funnel = DeviceTarget("funnel", T_TT_FUNNEL)                           # test/targets.py:23-29

standard_normal_lpdf = standard_normal_grad = _std
correlated_normal_lpdf = correlated_normal_grad = _corr
rosenbrock_lpdf = rosenbrock_grad = _rosen
funnel_lpdf = funnel_grad = _funnel
