"""LDS keeps what the previous kernel on a CU left there.  Every kernel family that uses LDS -- the role-split and wave-pair
kernels of the headline path, the one-stream MFMA-group kernel, the one-chain-per-wave kernels (funnel10: dynamic LDS
vectors; StockWatson: LDS transpose), split mode's chain kernel -- is run three times: as is, behind a kernel that fills all
160 KiB of every CU's LDS with NaNs, and behind one that fills it with a garbage pattern of well-formed small doubles.
Draws and diagnostics must be bit-identical (round 2's b9f800c was a bookkeeper reading never-written LDS entries)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SEED = 20251121


@pytest.fixture(scope="module")
def wa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import walnuts_amd
    return walnuts_amd


def _poison(wa, pattern):
    import torch
    from walnuts_amd import _lib as L
    rc = L.lib().wn_debug_poison_lds(torch.cuda.current_device(), pattern, None)
    assert rc == 0
    torch.cuda.synchronize()


def _three_runs(wa, make, q0, n_iter, route=0):
    """make(route) -> engine; route = WN_ROUTE_* names (the engine's routing argument, include/walnuts_amd.h)"""
    outs = []
    for pattern in (None, 0x7FF8000000000000, 0x3FF0000000000001, 0x0000000100000001):
        if pattern is not None:
            _poison(wa, pattern)
        e = make(route)
        name = e.kernel_name
        s, d = e.run(q0, SEED, n_iter)
        outs.append((s.cpu().numpy().copy(), d.cpu().numpy().copy()))
        e.close()
    for s, d in outs[1:]:
        np.testing.assert_array_equal(s, outs[0][0])
        np.testing.assert_array_equal(d, outs[0][1])
    return name


def _dense(D, C):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    return make_dense_problem(D, 1e4, C, SEED)


def test_headline_kernels_do_not_read_stale_lds(wa):
    A, q0 = _dense(100, 4096)
    mk = lambda r=0: wa.Engine(100, 4096, integrator=ol.I_FIXED, H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, route=r).set_target(ol.T_DENSEGAUSS, A)
    assert "wn_fast4_kernel" in _three_runs(wa, mk, q0, 3)
    assert "wn_fast_kernel" in _three_runs(wa, mk, q0, 3, route="FAST_LEGACY")
    assert "wn_coop_kernel" in _three_runs(wa, mk, q0, 2, route="COOP_LEGACY")
    A, q1 = _dense(100, 1024)
    mk1 = lambda r=0: wa.Engine(100, 1024, integrator=ol.I_FIXED, H0=0.5, M=8, generated=ol.G_IDENTITY, route=r).set_target(ol.T_DENSEGAUSS, A)
    assert "pairs=1" in _three_runs(wa, mk1, q1, 3)


def test_one_stream_adaptive_kernel_does_not_read_stale_lds(wa):
    A, q0 = _dense(64, 512)
    mk = lambda r=0: wa.Engine(64, 512, integrator=ol.I_ADAPT_R2P, H0=0.5, delta0=0.3, M=7, route=r).set_target(ol.T_DENSEGAUSS, A)
    assert "wn_coop_kernel" in _three_runs(wa, mk, q0, 3, route="DENSE_COOP")
    # round 5: these runs are one chain per wavefront with the matrix staged in LDS (every word of the staging array is written)
    assert "wn_orbit_kernel<dense" in _three_runs(wa, mk, q0, 3)
    assert "wn_wavelds_kernel" in _three_runs(wa, mk, q0, 3, route="ORBIT_LEGACY")
    A1, q1 = _dense(100, 300)
    mk1 = lambda r=0: wa.Engine(100, 300, integrator=ol.I_ADAPT_D, H0=0.5, delta0=0.3, M=6, route=r).set_target(ol.T_DENSEGAUSS, A1)
    assert "wn_orbit_kernel<dense" in _three_runs(wa, mk1, q1, 3)


def test_funnel_wave_kernel_does_not_read_stale_lds(wa):
    rng = np.random.default_rng(3)
    q0 = np.zeros((4096, 11)); q0[:, 0] = 3.0 * rng.normal(size=4096)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(4096, 10))
    mk = lambda r=0: wa.Engine(11, 4096, integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=ol.G_FIRSTK, gen_k=2, route=r).set_target(ol.T_FUNNEL10)
    assert "wn_orbit_kernel<funnel10" in _three_runs(wa, mk, q0, 6)  # the trajectory-first kernel: ring, slots, proposals in LDS
    assert "wn_wave_kernel<NE=1,funnel10" in _three_runs(wa, mk, q0, 6, route="ORBIT_LEGACY")  # the one-stream engine
    mkd = lambda r=0: wa.Engine(11, 4096, integrator=ol.I_ADAPT_D, H0=0.3, delta0=0.3, M=8, route=r).set_target(ol.T_FUNNEL10)
    assert "funnel10" in _three_runs(wa, mkd, q0, 4)
    assert "wn_wave_kernel" in _three_runs(wa, mkd, q0, 4, route="ORBIT_LEGACY")


def test_stockwatson_wave_kernel_does_not_read_stale_lds(wa):
    sw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stockwatson_data.npz"))
    T = int(sw["T"][0])
    tpar = np.concatenate([[float(T)], sw["y"]])
    q0 = np.zeros((256, 3 * T))
    for ig, H0, minC in ((ol.I_ADAPT_D, 0.1, 3), (ol.I_FIXED, 0.002, 0)):
        mk = lambda r=0: wa.Engine(3 * T, 256, integrator=ig, H0=H0, delta0=0.3, M=9, minC=minC, generated=ol.G_FIRSTK, gen_k=2, route=r).set_target(ol.T_STOCKWATSON, tpar)
        assert "stockwatson" in _three_runs(wa, mk, q0, 2)


def test_split_mode_and_wide_wave_kernels_do_not_read_stale_lds(wa):
    A, q0 = _dense(300, 256)
    mk = lambda r=0: wa.Engine(300, 256, integrator=ol.I_ADAPT_D, H0=0.4, delta0=0.3, M=6, generated=ol.G_Q0_SUMSQ, route=r).set_target(ol.T_DENSEGAUSS, A)
    assert "split" in _three_runs(wa, mk, q0, 2)
    lam = np.logspace(0, 2, 2000)
    qd = np.random.default_rng(1).normal(size=(64, 2000)) * np.sqrt(lam)
    mkw = lambda r=0: wa.Engine(2000, 64, integrator=ol.I_FIXED, H0=0.3, M=6, generated=ol.G_Q0_SUMSQ, route=r).set_target(ol.T_DIAGGAUSS, 1.0 / lam)
    assert "W=4" in _three_runs(wa, mkw, qd, 2)
