"""CPU tests (no GPU) of the role-split headline path's bookkeeper (walnuts_amd/csrc/wn_fast.h): the
issue / retire automaton with its two-tick (wn_k_fast.hip) and three-tick (wn_k_fast4.hip) lag, run by
tests/hostsim/fastsim.cpp with a scalar stand-in for the T waves, must reproduce the oracle -- every diagnostics column, every draw -- on the reference goldens
that use fixedLeapFrog on a Gaussian and on seeded edge cases (M reached, non-finite energies, zero weights)."""
import numpy as np
import pytest

import hostsim_lib as hs
import oracle_lib as ol
import test_oracle_golden as tg

INT_DIAG = (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22)


def check(h, o, k=0, what=""):
    d, rd = h["diagnostics"][k], o["diagnostics"]
    for col in INT_DIAG:
        np.testing.assert_array_equal(d[:, col], rd[:, col], err_msg=f"{what} diag col {col}")
    tg.close(d, rd, rtol=1e-10, atol=1e-12)
    tg.close(h["samples"][k], o["samples"], rtol=1e-10, atol=1e-12)


LAGS = pytest.mark.parametrize("lag", [2, 3, 4])  # 4: lag 3 as two instruction streams (RetireBook / IssueBook)


@LAGS
@pytest.mark.parametrize("name", ["std3_fixed", "dense8_fixed", "std100_fixed", "std3_fixed_M3"])
def test_bookkeeper_reproduces_reference_goldens(name, lag):
    c, ref = tg.load_case(name)
    seed = int(tg.TR["seed"][0])
    o = tg.run_oracle(c, ref, tape=False)
    for nlc in (0, 1, 2, 3):
        h = hs.fast_walnuts(c["target"], c["q0"], tpar=c["tpar"] if c["target"] != ol.T_STDGAUSS else None, H0=c["H0"], delta0=c["delta0"],
                            numIter=c["numIter"], M=c["M"], generated=c["generated"], gen_k=c["gen_k"], seed=seed,
                            chain0=c["chain"], nlc=nlc, lag=lag)
        check(h, o, what=f"{name} nlc={nlc}")
        # the reference's own draws (golden fixture), not only the oracle's
        tg.close(h["samples"][0], ref["samples"], rtol=1e-9, atol=1e-11)


@LAGS
@pytest.mark.parametrize("M", [1, 2, 4, 9])
@pytest.mark.parametrize("gen", [0, 1, 2])
def test_bookkeeper_depths_and_generated_kinds(M, gen, lag):
    rng = np.random.default_rng(M * 10 + gen)
    D = 7
    lam = np.logspace(0, 2, D)
    q0 = rng.normal(size=(3, D)) * np.sqrt(lam)
    kw = dict(H0=0.45, numIter=9, M=M, generated=gen, gen_k=3 if gen == 2 else 0, seed=77 + M)
    h = hs.fast_walnuts(ol.T_DIAGGAUSS, q0, tpar=1.0 / lam, chain0=40, lag=lag, **kw)
    for k in range(3):
        o = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=1.0 / lam, chain=40 + k, **kw)
        check(h, o, k, what=f"M={M} gen={gen} chain {k}")
    assert 0.0 < h["idle_frac"] < 0.92  # M = 1: one leaf per iteration among the start-up ticks


@pytest.mark.parametrize("scale,H0,expect", [(3e153, 0.05, {4}), (3e153, 0.5, {4, 5}), (1e154, 0.5, {999}), (1e200, 0.05, {999})])
@LAGS
def test_bookkeeper_huge_energies_zero_and_infinite_weights(scale, H0, expect, lag):
    """Energies near the top of the double range: multinomial weights exp(-H + H0) are exactly 0 or inf, so
    WnewSum stays 0 (the uniform of WALNUTS.py:426 is then NOT drawn, which shifts every later draw of the
    iteration: the bookkeeper may not issue step sizes ahead) or becomes inf / NaN; beyond 1e154 the energy
    itself overflows (stop code 999, Q3/Q4)."""
    D = 5
    q0 = np.random.default_rng(3).normal(size=(3, D)) * scale
    kw = dict(H0=H0, numIter=8, M=8, generated=1, seed=11)
    h = hs.fast_walnuts(ol.T_STDGAUSS, q0, chain0=3, lag=lag, **kw)
    seen = set()
    for k in range(3):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ol.I_FIXED, chain=3 + k, **kw)
        with np.errstate(all="ignore"):
            check(h, o, k, what=f"scale={scale} H0={H0} chain {k}")
        seen |= set(o["diagnostics"][:, 19].astype(int).tolist())
    assert expect <= seen, seen


@pytest.mark.parametrize("amax", [30.0, 1e6])
@LAGS
def test_bookkeeper_unstable_step_sizes(amax, lag):
    """Step sizes beyond the leapfrog stability limit in one coordinate (h*sqrt(a) > 2)."""
    a = np.array([1.0, 1.0, 1.0, amax])
    q0 = np.random.default_rng(3).normal(size=(4, 4)) / np.sqrt(a)
    kw = dict(H0=0.5, numIter=14, M=10, generated=1, seed=11)
    h = hs.fast_walnuts(ol.T_DIAGGAUSS, q0, tpar=a, chain0=3, lag=lag, **kw)
    for k in range(4):
        o = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=a, chain=3 + k, **kw)
        check(h, o, k, what=f"amax={amax} chain {k}")


@LAGS
def test_bookkeeper_config2_shape_many_chains(lag):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    A, q0 = make_dense_problem(100, 1e4, 3, 20251121)
    kw = dict(H0=0.5, numIter=6, M=10, seed=20251121, generated=1)
    h = hs.fast_walnuts(ol.T_DENSEGAUSS, q0, tpar=A.ravel(), lag=lag, **kw)
    for k in range(3):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), chain=k, **kw)
        check(h, o, k)
    assert h["idle_frac"] < (0.06 if lag == 2 else 0.085)  # two idle ticks per doubling + one per iteration start; lag 3: two more per doubling


@LAGS
@pytest.mark.parametrize("M,scale", [(6, 1.0), (3, 1.0), (8, 3e153)])
def test_bookkeeper_trace_rows_match_the_oracle(M, scale, lag):
    """Traced runs of the pair kernel (wn_k_fast4.hip): the T wave writes (q', v') of every leaf it integrates at its own
    row count and takes the row back on OP_UNDO (the leaf in flight when a sub-U-turn / non-finite energy ended the
    iteration); the bookkeeper writes the row's scalars when it retires the leaf.  Row for row the oracle's trace
    (one row per integrator call of the reference, adaptiveIntegrators.py:49-59)."""
    rng = np.random.default_rng(5 + M)
    D = 6
    lam = np.logspace(0, 2, D)
    q0 = rng.normal(size=(3, D)) * np.sqrt(lam) * scale
    kw = dict(H0=0.45, numIter=7, M=M, generated=0, seed=123 + M)
    h = hs.fast_walnuts(ol.T_DIAGGAUSS, q0, tpar=1.0 / lam, chain0=9, trace_cap=2000, lag=lag, **kw)
    for k in range(3):
        o = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=1.0 / lam, chain=9 + k, trace_cap=2000, **kw)
        n = o["ncalls"]
        assert h["trace_n"][k] == n
        cols = [0, 1, 2, 6, 7, 8, 10, 11]
        np.testing.assert_array_equal(h["trace"][k, :n][:, cols], o["trace"][:n][:, cols])
        tg.close(h["trace"][k, :n], o["trace"][:n], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("adaptH,adaptDelta", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("D,warm,iters", [(7, 25, 32), (40, 60, 64)])
def test_bookkeeper_warmup_adaptation_matches_the_oracle(D, warm, iters, adaptH, adaptDelta):
    """Round 5: the headline kernel's bookkeeper runs the reference's DEFAULT call shape -- fixedLeapFrog with warm-up adaptation of
    the step size H (P2 quantile of log igrConst, WALNUTS.py:313, P2quantile.py) and of the tolerance delta (np.quantile of the
    energy-error factors, :701-712).  The retire side pushes and adapts, the issue side draws the next iteration's step sizes around
    the new H: every draw, every diagnostics column (15 = H, 18 = delta of each iteration) as the oracle's."""
    rng = np.random.default_rng(D + warm)
    lam = np.logspace(0, 1.5, D)
    q0 = rng.normal(size=(3, D)) * np.sqrt(lam)
    kw = dict(H0=0.3, numIter=iters, M=6, generated=0, seed=31 + D, warmupIter=warm, adaptH=adaptH, adaptDelta=adaptDelta)
    h = hs.fast_walnuts(ol.T_DIAGGAUSS, q0, tpar=1.0 / lam, chain0=5, lag=4, **kw)
    for k in range(3):
        o = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=1.0 / lam, chain=5 + k, **kw)
        check(h, o, k, what=f"warm-up D={D} chain {k}")
        if adaptH:
            assert len(set(np.round(o["diagnostics"][:, 15], 12))) > 5  # H did adapt


@pytest.mark.parametrize("bits", [20, 12])
def test_adapt_quant_bits_rounds_the_adapted_step_size_like_the_oracle(bits):
    """ABI 5's opt-in `adapt_quant_bits` (NOT the reference's arithmetic; DESIGN 9, warm-up tolerance contract): the adapted H and delta are
    rounded to `bits` mantissa bits whenever the warm-up recomputes them -- by the engine's wn_quant_bits (here: the headline bookkeeper and
    the generic state machine, compiled for the CPU) and by the oracle's quant_bits, the same integer operation on the bit pattern.  Every
    adapted H / delta is a multiple of 2^(exponent - bits), equals the oracle's, and differs from the unrounded run's."""
    D, warm, iters = 12, 30, 36
    rng = np.random.default_rng(bits)
    lam = np.logspace(0, 1.2, D)
    q0 = rng.normal(size=(2, D)) * np.sqrt(lam)
    kw = dict(H0=0.3, numIter=iters, M=6, generated=0, seed=77, warmupIter=warm)
    h = hs.fast_walnuts(ol.T_DIAGGAUSS, q0, tpar=1.0 / lam, chain0=3, lag=4, adaptQuantBits=bits, **kw)
    g = hs.walnuts(ol.T_DIAGGAUSS, q0, tpar=1.0 / lam, chain0=3, adaptQuantBits=bits, **kw)
    for k in range(2):
        o = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=1.0 / lam, chain=3 + k, adaptQuantBits=bits, **kw)
        o0 = ol.walnuts(ol.T_DIAGGAUSS, q0[k], integrator=ol.I_FIXED, tpar=1.0 / lam, chain=3 + k, **kw)
        check(h, o, k, what=f"quantised warm-up, bookkeeper, chain {k}")
        np.testing.assert_array_equal(g["diagnostics"][k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        np.testing.assert_allclose(g["diagnostics"][k][:, [15, 18]], o["diagnostics"][:, [15, 18]], rtol=1e-12)
        for col in (15, 18):
            v = o["diagnostics"][12:warm, col]
            m, e = np.frexp(v)
            assert np.all(np.ldexp(m, bits + 1) == np.round(np.ldexp(m, bits + 1))), (col, v[:3])  # (frexp's mantissa is in [0.5, 1): bits + 1 binary places)
            assert np.any(v != o0["diagnostics"][12:warm, col])
