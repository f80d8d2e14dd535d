"""The trajectory-first kernel (wn_k_orbit.hip: integrator calls of up to 16 leaves back to back, their NUTS bookkeeping
lane-parallel) against the CPU oracle and against the one-stream engine of the same layout (WN_ORBIT_LEGACY=1), through the C ABI.

The reference path it stands for: WALNUTSpy/WALNUTS.py:189-695 with adaptiveIntegrators.py:49-59 / 65-137 / 361-475."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol  # noqa: E402

pytestmark = pytest.mark.gpu

SEED = 20251121
INT_COLS = [0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22]


@pytest.fixture(scope="module")
def wa():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import walnuts_amd
    return walnuts_amd


def funnel_q0(C, seed=3):
    rng = np.random.default_rng(seed)
    q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    return q0


def run_engine(wa, target, q0, iters, legacy=False, exact=True, seed=SEED, tpar=None, onewave=False, route=(), **kw):
    """Kernel routing is the engine's `route` argument (wn_config.route, include/walnuts_amd.h), never the environment.
    exact: the orbit kernel with the micro-step arithmetic of the one-stream engine (WN_ROUTE_ORBIT_EXACT; funnel10's default is a
    micro step of its own with one row reduction instead of two -- the same values, not the same bits); legacy: the one-stream engine
    (WN_ROUTE_ORBIT_LEGACY); onewave: StockWatson's runs on the one-wave form of the wide mode instead of wn_k_orbit2.hip's two waves;
    route: further WN_ROUTE_* names."""
    r = ["ORBIT_WIDE_ADAPT"]  # (StockWatson: the adaptive integrators' plain runs stay on the one-stream engine by default)
    r += ["ORBIT_LEGACY"] if legacy else []
    r += ["ORBIT_ONEWAVE"] if onewave else []
    r += ["ORBIT_EXACT"] if exact else []
    e = wa.Engine(q0.shape[1], q0.shape[0], route=r + list(route), **kw)
    e.set_target(target, tpar)
    name = e.kernel_name
    s, d = e.run(q0, seed, iters)
    s = s.cpu().numpy(); d = d.cpu().numpy()
    e.close()
    return s, d, name


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def near(a, b, rtol):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return bool(np.all((np.isnan(a) & np.isnan(b)) | (a == b) | (np.abs(a - b) <= rtol * np.maximum(1e-3, np.abs(b)))))


def same_run(s_new, d_new, s_old, d_old, rtol=0.0):
    """rtol = 0: bit for bit, except the orbit energy error (diagnostics column 17 = Hmax - Hmin), held to 1e-12: in the FIRST
    iteration its H0 comes from the log density of q0, which each kernel evaluates at a code site of its own (the compiler
    contracts the funnel's lp expression differently at different inlined sites: last-bit differences that nothing else sees).
    rtol > 0 (the two kernels are different instantiations with differently associated sums): integer columns exactly, reals to rtol."""
    if rtol > 0.0:
        return same(d_new[..., INT_COLS], d_old[..., INT_COLS]) and near(s_new, s_old, rtol) and near(d_new, d_old, rtol)
    k = [c for c in range(24) if c != 17]
    if not (same(d_new[..., k], d_old[..., k]) and same(s_new, s_old)):
        return False
    return near(d_new[..., 17], d_old[..., 17], 1e-12)


CASES = [
    # integrator, M, H0, delta0, minC, iters, generated, gen_k, rtol of the reals (0 = bit for bit)
    (ol.I_ADAPT_R2P, 12, 0.3, 0.3, 0, 40, ol.G_IDENTITY, 0, 0.0),
    (ol.I_ADAPT_R2P, 8, 0.3, 0.3, 0, 60, ol.G_FIRSTK, 2, 0.0),
    (ol.I_ADAPT_D, 10, 0.3, 0.3, 0, 40, ol.G_Q0_SUMSQ, 0, 1e-13),  # sum(q*q) of the draw: one reduction here, a deferred packed one there
    (ol.I_ADAPT_D, 6, 0.5, 0.1, 1, 40, ol.G_IDENTITY, 0, 0.0),
    (ol.I_FIXED, 9, 0.2, 0.3, 0, 8, ol.G_IDENTITY, 0, 1e-7),      # the one-stream fixedLeapFrog engine is another instantiation (tick loop)
    (ol.I_ADAPT_R2P, 3, 0.3, 0.3, 0, 80, ol.G_IDENTITY, 0, 0.0),
    (ol.I_ADAPT_R2P, 1, 0.3, 0.3, 0, 50, ol.G_IDENTITY, 0, 0.0),
    (ol.I_ADAPT_R2P, 5, 1.5, 0.05, 0, 40, ol.G_IDENTITY, 0, 0.0),  # large steps: deep refinement, passive ends, stop code -4
]


@pytest.mark.parametrize("case", CASES, ids=[f"integ{c[0]}_M{c[1]}_H{c[2]}_gen{c[6]}" for c in CASES])
def test_orbit_kernel_equals_the_one_stream_engine(wa, case):
    """Same arithmetic per micro step and per sum, different control structure: draws and all 24 diagnostics bit for bit."""
    integ, M, H0, delta0, minC, iters, gen, gk, rtol = case
    C = 384
    q0 = funnel_q0(C, seed=11 + M)
    kw = dict(integrator=integ, H0=H0, delta0=delta0, M=M, minC=minC, generated=gen, gen_k=gk)
    s_new, d_new, name = run_engine(wa, ol.T_FUNNEL10, q0, iters, **kw)
    assert "wn_orbit_kernel" in name, name
    s_old, d_old, name_old = run_engine(wa, ol.T_FUNNEL10, q0, iters, legacy=True, **kw)
    assert "wn_wave_kernel" in name_old, name_old
    bad = [c for c in range(C) if not same_run(s_new[c], d_new[c], s_old[c], d_old[c], rtol)]
    if bad:
        c = bad[0]
        it = next(t for t in range(iters) if not same_run(s_new[c, t + 1], d_new[c, t], s_old[c, t + 1], d_old[c, t], rtol))
        cols = [k for k in range(24) if not same(d_new[c, it, k], d_old[c, it, k])]
        raise AssertionError(f"{len(bad)} of {C} chains differ; first: chain {c} iteration {it} diag columns {cols}\n new {d_new[c, it]}\n old {d_old[c, it]}\n"
                             f" draw new {s_new[c, it + 1]}\n draw old {s_old[c, it + 1]}")
    codes = set(np.unique(d_new[:, :, 19]).astype(int).tolist())
    assert 4 in codes or M <= 1


@pytest.mark.parametrize("integ,M", [(ol.I_ADAPT_R2P, 12), (ol.I_ADAPT_D, 8), (ol.I_FIXED, 10), (ol.I_ADAPT_R2P, 4)])
def test_orbit_kernel_matches_oracle(wa, integ, M):
    """Per-chain draws and diagnostics against the CPU oracle (the restatement pinned to the Python reference)."""
    C, iters = 64, 12
    q0 = funnel_q0(C, seed=5)
    kw = dict(integrator=integ, H0=0.3, delta0=0.3, M=M)
    s, d, name = run_engine(wa, ol.T_FUNNEL10, q0, iters, exact=False, **kw)  # the default: funnel10's own micro step
    assert "wn_orbit_kernel" in name
    for c in (0, 7, 19, 33, 63):
        o = ol.walnuts(ol.T_FUNNEL10, q0[c], integrator=integ, H0=0.3, delta0=0.3, numIter=iters, M=M, seed=SEED, chain=c)
        np.testing.assert_array_equal(d[c][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {c}")
        np.testing.assert_allclose(s[c].T, o["samples"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
        np.testing.assert_allclose(d[c], o["diagnostics"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")


@pytest.mark.parametrize("target", [ol.T_CORRGAUSS, ol.T_SMILE, ol.T_FUNNEL1, ol.T_MODFUNNEL])
def test_orbit_kernel_two_dimensional_targets(wa, target):
    """The 2-D targets of targetDistr.py in the one-chain-per-wave layout: oracle parity and identity with the one-stream engine."""
    C, iters, M = 128, 25, 7
    q0 = np.random.default_rng(9).normal(size=(C, 2))
    kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.4, delta0=0.2, M=M)
    s, d, name = run_engine(wa, target, q0, iters, **kw)
    assert "wn_orbit_kernel" in name, name
    s_old, d_old, _ = run_engine(wa, target, q0, iters, legacy=True, **kw)
    # (the pair targets' gradient is inlined at several code sites of each kernel: last-bit differences, amplified by the dynamics)
    bad = np.argwhere(d[..., INT_COLS] != d_old[..., INT_COLS])
    assert len(bad) == 0, (len(bad), bad[:5])
    assert near(s[:, :6], s_old[:, :6], 1e-9), np.abs(s[:, :6] - s_old[:, :6]).max()
    for c in (0, 50, 127):
        o = ol.walnuts(target, q0[c], integrator=ol.I_ADAPT_R2P, H0=0.4, delta0=0.2, numIter=iters, M=M, seed=SEED, chain=c)
        np.testing.assert_array_equal(d[c][:, INT_COLS], o["diagnostics"][:, INT_COLS])
        np.testing.assert_allclose(s[c].T, o["samples"], rtol=1e-6, atol=1e-9)


def test_orbit_kernel_config3_size(wa):
    """BASELINE config 3 at its size for a shorter run: 4096 chains, adaptLeapFrogR2P, M = 12; every chain's gradient-evaluation
    counts equal the one-stream engine's, a few chains against the oracle."""
    C, iters = 4096, 20
    q0 = funnel_q0(C, seed=21)
    kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=ol.G_FIRSTK, gen_k=2)
    s, d, name = run_engine(wa, ol.T_FUNNEL10, q0, iters, **kw)
    s_old, d_old, _ = run_engine(wa, ol.T_FUNNEL10, q0, iters, legacy=True, **kw)
    assert same_run(s, d, s_old, d_old)
    busiest = int(np.argmax((d[:, :, 6] + d[:, :, 7]).sum(1)))
    s_f, d_f, _ = run_engine(wa, ol.T_FUNNEL10, q0, iters, exact=False, **kw)  # the default arithmetic (one reduction per micro step)
    # every chain: the two arithmetics take the same decisions (round 5 budgeted up to 2 of the 4096 chains here; tests/studies/cfg3_edge_chains.py
    # finds none, for this and two more sets of starting points: profiles/r06_cfg3_edge_chains.txt -- no budget)
    assert same(d_f[..., [6, 7, 19, 20]], d[..., [6, 7, 19, 20]]), np.flatnonzero((d_f[..., [6, 7, 19, 20]] != d[..., [6, 7, 19, 20]]).any(axis=(1, 2)))
    for c in (0, 1234, busiest):
        o = ol.walnuts(ol.T_FUNNEL10, q0[c], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=iters, M=12, seed=SEED, chain=c,
                       generated=ol.G_FIRSTK, gen_k=2)
        for dd, ss in ((d, s), (d_f, s_f)):
            np.testing.assert_array_equal(dd[c][:, INT_COLS], o["diagnostics"][:, INT_COLS])
            np.testing.assert_allclose(ss[c].T, o["samples"], rtol=1e-6, atol=1e-9)


# ---- wide mode (round 5): StockWatson, vectors of 12 registers per lane, ring / slots / proposals in the chain's workspace ----
def sw_tpar(T):
    sw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stockwatson_data.npz"))
    y = sw["y"][:T]
    return np.concatenate([[float(T)], y])


SW_CASES = [
    # T, integrator, kwargs, iterations, rtol against the one-stream engine (0 = bit for bit)
    (252, ol.I_ADAPT_D, dict(H0=0.1, delta0=0.3, M=8, minC=3), 4, 0.0),        # mainSW.py:41-47
    (252, ol.I_ADAPT_R2P, dict(H0=0.1, delta0=0.3, M=8, minC=3), 4, 0.0),      # :64-71
    (252, ol.I_FIXED, dict(H0=0.1 / 8, M=7), 4, 1e-7),                         # :56-62 (the one-stream fixedLeapFrog engine works in place: another instantiation)
    (30, ol.I_ADAPT_R2P, dict(H0=0.25, delta0=0.2, M=9, minC=0), 12, 0.0),     # short series: long orbits, deep sub-tree levels, refinements
    (30, ol.I_ADAPT_D, dict(H0=0.6, delta0=0.05, M=6, minC=0), 12, 0.0),       # large steps: refinement, backward searches, passive ends
    (101, ol.I_FIXED, dict(H0=0.02, M=10), 5, 1e-7),                           # T - 1 a multiple of 4: tSigma alone in its lane
]


SW_FORMS = [(c, w) for c in SW_CASES for w in (False, True)]


@pytest.mark.parametrize("case,onewave", SW_FORMS, ids=[f"T{c[0]}_integ{c[1]}_M{c[2]['M']}_{'one_wave' if w else 'two_waves'}" for c, w in SW_FORMS])
def test_wide_orbit_kernel_equals_the_one_stream_engine_and_the_oracle(wa, case, onewave):
    """The trajectory-first kernel's wide mode -- wn_orbit_kernel<stockwatson,NE=12,...> on one wave, and wn_orbit2_kernel (trajectory and
    bookkeeping on a wave each: BASELINE config 5's plain sampling runs) -- against the one-stream engine of the same layout
    (WN_ORBIT_LEGACY=1: same micro-step arithmetic, same wave sums -- draws and diagnostics bit for bit for the adaptive integrators)
    and against the CPU oracle (integers exactly, reals to 1e-6)."""
    T, integ, kw, iters, rtol = case
    C = 24
    tpar = sw_tpar(T)
    rng = np.random.default_rng(100 + T)
    q0 = 0.1 * rng.normal(size=(C, 3 * T)); q0[0] = 0.0
    s_new, d_new, name = run_engine(wa, ol.T_STOCKWATSON, q0, iters, tpar=tpar, integrator=integ, onewave=onewave, **kw)
    assert ("wn_orbit_kernel<stockwatson" if onewave else "wn_orbit2_kernel<stockwatson") in name, name
    s_old, d_old, name_old = run_engine(wa, ol.T_STOCKWATSON, q0, iters, legacy=True, tpar=tpar, integrator=integ, **kw)
    assert "wn_wave_kernel" in name_old, name_old
    bad = [c for c in range(C) if not same_run(s_new[c], d_new[c], s_old[c], d_old[c], rtol)]
    if bad:
        c = bad[0]
        it = next(t for t in range(iters) if not same_run(s_new[c, t + 1], d_new[c, t], s_old[c, t + 1], d_old[c, t], rtol))
        cols = [k for k in range(24) if not same(d_new[c, it, k], d_old[c, it, k])]
        raise AssertionError(f"{len(bad)} of {C} chains differ; first: chain {c} iteration {it} diag columns {cols}\n new {d_new[c, it]}\n old {d_old[c, it]}")
    for c in (0, C - 1):
        o = ol.walnuts(ol.T_STOCKWATSON, q0[c], integrator=integ, tpar=tpar, H0=kw["H0"], delta0=kw.get("delta0", 0.05), numIter=iters,
                       M=kw["M"], minC=kw.get("minC", 0), seed=SEED, chain=c)
        np.testing.assert_array_equal(d_new[c][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {c}")
        np.testing.assert_allclose(s_new[c].T, o["samples"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
        np.testing.assert_allclose(d_new[c], o["diagnostics"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
    assert (d_new[:, :, 6] + d_new[:, :, 7]).sum() > 0


@pytest.mark.parametrize("gen,gk", [(ol.G_FIRSTK, 5), (ol.G_Q0_SUMSQ, 0)])
def test_wide_orbit_kernel_generated_forms(wa, gen, gk):
    """generated = first k / [q0, sum q^2] in the time-major register layout (Tgt::map_gidx says which register holds which parameter)."""
    T, C, iters = 60, 8, 6
    tpar = sw_tpar(T)
    q0 = 0.1 * np.random.default_rng(4).normal(size=(C, 3 * T))
    kw = dict(integrator=ol.I_ADAPT_D, H0=0.2, delta0=0.3, M=6, minC=1, generated=gen, gen_k=gk)
    s_old, d_old, _ = run_engine(wa, ol.T_STOCKWATSON, q0, iters, legacy=True, tpar=tpar, **kw)
    for onewave in (False, True):
        s_new, d_new, name = run_engine(wa, ol.T_STOCKWATSON, q0, iters, tpar=tpar, onewave=onewave, **kw)
        assert ("wn_orbit_kernel<stockwatson" if onewave else "wn_orbit2_kernel<stockwatson") in name, name
        assert same(d_new[..., INT_COLS], d_old[..., INT_COLS])
        assert near(s_new, s_old, 1e-12), np.abs(s_new - s_old).max()
        assert s_new.shape[-1] == (gk if gen == ol.G_FIRSTK else 2)


def test_two_wave_kernel_is_the_one_wave_kernel_bit_for_bit(wa):
    """wn_k_orbit2.hip against wn_k_orbit.hip's wide mode on a longer fixedLeapFrog run (mainSW.py:56-62 shape: M = 11, orbits of up to 2047
    leaves, many ring wrap-arounds, sub-tree levels up to 10): the same arithmetic on another schedule -- every draw and diagnostic equal."""
    T, C, iters = 252, 64, 6
    tpar = sw_tpar(T)
    q0 = np.zeros((C, 3 * T))
    kw = dict(integrator=ol.I_FIXED, H0=0.1 / 8, M=11)
    s2, d2, name2 = run_engine(wa, ol.T_STOCKWATSON, q0, iters, tpar=tpar, **kw)
    s1, d1, name1 = run_engine(wa, ol.T_STOCKWATSON, q0, iters, tpar=tpar, onewave=True, **kw)
    assert "wn_orbit2_kernel" in name2 and "wn_orbit_kernel" in name1, (name2, name1)
    assert same(d2, d1) and same(s2, s1), (np.argwhere(d2 != d1)[:5], np.abs(s2 - s1).max())
    assert d2[:, :, 6].max() > 500  # long orbits


def test_two_wave_kernel_is_chosen_only_while_the_chains_leave_simds_idle(wa):
    """wn_k_orbit2.hip spends 2 (fixedLeapFrog) or 4 (adaptive integrators) wavefronts and up to 107 KB of LDS on a chain to cut its
    LATENCY -- BASELINE config 5 is one chain per CU.  With more chains than CUs the one-wave kernels fill the SIMDs themselves and win on
    throughput (tools/sw_chains_ab.sh), so the choice depends on the number of chains; WN_ORBIT2_ANYC=1 lifts the bound (same draws)."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    T = 30
    tpar = sw_tpar(T)
    names = {}
    for integ in (ol.I_ADAPT_D, ol.I_FIXED):
        for C in (cus, cus + 1, 2 * cus, 2 * cus + 1):
            e = wa.Engine(3 * T, C, integrator=integ, H0=0.2, delta0=0.3, M=4, minC=1)
            e.set_target(ol.T_STOCKWATSON, tpar)
            names[(integ, C)] = e.kernel_name
            e.close()
    assert "wn_orbit2_kernel" in names[(ol.I_ADAPT_D, cus)] and "wn_wave_kernel" in names[(ol.I_ADAPT_D, cus + 1)], names
    assert "wn_orbit2_kernel" in names[(ol.I_FIXED, 2 * cus)] and "wn_orbit_kernel" in names[(ol.I_FIXED, 2 * cus + 1)], names
    # the same draws either way
    C = cus + 40
    q0 = 0.1 * np.random.default_rng(8).normal(size=(C, 3 * T))
    kw = dict(integrator=ol.I_ADAPT_D, H0=0.2, delta0=0.3, M=5, minC=1)
    s1, d1, n1 = run_engine(wa, ol.T_STOCKWATSON, q0, 3, tpar=tpar, legacy=True, **kw)
    s2, d2, n2 = run_engine(wa, ol.T_STOCKWATSON, q0, 3, tpar=tpar, route=["ORBIT2_ANYC"], **kw)
    assert "wn_wave_kernel" in n1 and "wn_orbit2_kernel" in n2, (n1, n2)
    assert same_run(s2, d2, s1, d1)


# ---- the adaptive integrators on the headline geometry (round 5): dense Gaussian 17 <= D <= 128, one chain per wavefront, matrix in LDS ----
def dense_case(D, C, kappa=1e3, seed=5):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    return make_dense_problem(D, kappa, C, seed)


@pytest.mark.parametrize("D,integ,C", [(100, ol.I_ADAPT_R2P, 37), (100, ol.I_ADAPT_D, 24), (72, ol.I_ADAPT_R2P, 16), (128, ol.I_ADAPT_D, 9), (65, ol.I_ADAPT_R2P, 8),
                                       (40, ol.I_ADAPT_D, 19), (64, ol.I_ADAPT_R2P, 10), (17, ol.I_ADAPT_R2P, 12), (8, ol.I_ADAPT_R2P, 25), (3, ol.I_ADAPT_D, 11)])
def test_dense_adaptive_runs_one_chain_per_wave_matrix_in_lds(wa, D, integ, C):
    """adaptLeapFrogD / adaptLeapFrogR2P (adaptiveIntegrators.py:65-137 / 361-475) on the dense Gaussian of BASELINE config 2's geometry:
    plain sampling runs on the trajectory-first kernel's wide mode (wn_orbit_kernel<dense,NE=2,K=8>: eight chains per workgroup share the
    matrix in LDS), everything else on the one-stream engine of the same layout (wn_wavelds_kernel).  The two against each other bit for
    bit, both against the oracle (integers exactly, reals 1e-6), and against the cooperative MFMA layout they replace (route WN_ROUTE_DENSE_COOP:
    another summation order -- integers exactly, reals 1e-6)."""
    A, q0 = dense_case(D, C)
    iters = 4
    kw = dict(integrator=integ, H0=0.8, delta0=0.2, M=6)
    s_new, d_new, name = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, **kw)
    assert "wn_orbit_kernel<dense,NE=2,K=8" in name, name
    s_eng, d_eng, name_eng = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, legacy=True, **kw)
    assert "wn_wavelds_kernel" in name_eng, name_eng
    assert same_run(s_new, d_new, s_eng, d_eng), (np.argwhere(d_new != d_eng)[:4], np.abs(s_new - s_eng).max())
    s_co, d_co, name_co = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, route=["DENSE_COOP"], **kw)
    assert "wn_coop_kernel" in name_co, name_co
    assert same(d_new[..., INT_COLS], d_co[..., INT_COLS]) and near(s_new, s_co, 1e-6)
    for c in sorted(set([0, C // 2, C - 1])):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[c], integrator=integ, tpar=A.ravel(), H0=0.8, delta0=0.2, numIter=iters, M=6, seed=SEED, chain=c)
        np.testing.assert_array_equal(d_new[c][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {c}")
        np.testing.assert_allclose(s_new[c].T, o["samples"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
        np.testing.assert_allclose(d_new[c], o["diagnostics"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
    assert (d_new[:, :, 6] + d_new[:, :, 7]).sum() > 0


@pytest.mark.parametrize("D,integ,warm", [(40, ol.I_ADAPT_R2P, 0), (100, ol.I_ADAPT_D, 0), (24, ol.I_ADAPT_D, 3)])
def test_dense_adaptive_chain_queue_of_the_persistent_launch(wa, D, integ, warm):
    """Round 6: the dense Gaussian's trajectory-first launch is a persistent grid (one workgroup per CU, its matrix staged once) with a chain
    queue -- a wave whose chain has finished takes the next chain of the launch.  More chains than the chip holds at once (8 per CU): the chains
    beyond the first grid's are dealt through the queue, in whatever order waves come free (runs with warm-up keep one chain per wave: their
    instantiations have no register to spare for the loop); every chain's draws and diagnostics are the
    one-stream engine's (one chain per wave, no queue) bit for bit, whichever wave ran it, twice in a row (the queue head is reset per launch),
    and the oracle's for the last chain."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    C = 8 * cus + 8 * 37 + 3  # (ragged: the last workgroup's worth of chains is incomplete)
    A, q0 = dense_case(D, C)
    iters = 3
    kw = dict(integrator=integ, H0=0.8, delta0=0.2, M=4, warmupIter=warm)
    s_new, d_new, name = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, **kw)
    assert "wn_orbit_kernel<dense,NE=2,K=8" in name, name
    s_again, d_again, _ = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, **kw)
    assert same(s_new, s_again) and same(d_new, d_again)
    s_eng, d_eng, name_eng = run_engine(wa, ol.T_DENSEGAUSS, q0, iters, tpar=A, legacy=True, **kw)
    assert "wn_wavelds_kernel" in name_eng, name_eng
    ne = lambda a, b: ~((a == b) | (np.isnan(a) & np.isnan(b)))
    dcols = [k for k in range(24) if k != 17]  # (column 17: same_run's note -- held to 1e-12 below)
    bad = np.flatnonzero(ne(d_new[..., dcols], d_eng[..., dcols]).any(axis=(1, 2)) | ne(s_new, s_eng).any(axis=(1, 2)))
    assert bad.size == 0, (bad[:10], C)
    assert same_run(s_new, d_new, s_eng, d_eng)
    c = C - 1
    o = ol.walnuts(ol.T_DENSEGAUSS, q0[c], integrator=integ, tpar=A.ravel(), H0=0.8, delta0=0.2, numIter=iters, M=4, seed=SEED, chain=c, warmupIter=warm)
    np.testing.assert_array_equal(d_new[c][:, INT_COLS], o["diagnostics"][:, INT_COLS])
    np.testing.assert_allclose(s_new[c].T, o["samples"], rtol=1e-6, atol=1e-9)


def test_dense_adaptive_warmup_and_yoshida_on_the_lds_engine(wa):
    """What the trajectory-first kernel does not take (adaptYoshidaD; traced runs, orbit statistics) runs on wn_wavelds_kernel, the one-stream
    engine with the matrix in LDS -- and so does everything with route WN_ROUTE_ORBIT_LEGACY, e.g. warm-up adaptation: against the cooperative layout
    (integers exactly, reals 1e-6)."""
    A, q0 = dense_case(100, 20)
    for kw, it in ((dict(integrator=ol.I_ADAPT_R2P, H0=0.8, delta0=0.2, M=5, warmupIter=6), 8),
                   (dict(integrator=ol.I_ADAPT_YOSHIDA, H0=0.8, delta0=0.1, M=5), 4)):
        # (legacy: warm-up iterations of plain runs are the trajectory-first kernel's since its WARM instantiations -- the test below)
        s_new, d_new, name = run_engine(wa, ol.T_DENSEGAUSS, q0, it, tpar=A, legacy=True, **kw)
        assert "wn_wavelds_kernel" in name, name
        s_co, d_co, name_co = run_engine(wa, ol.T_DENSEGAUSS, q0, it, tpar=A, route=["DENSE_COOP"], **kw)
        assert "wn_coop_kernel" in name_co, name_co
        assert same(d_new[..., INT_COLS], d_co[..., INT_COLS]) and near(s_new, s_co, 1e-6) and near(d_new, d_co, 1e-6)


@pytest.mark.parametrize("target,D,integ,C", [(ol.T_STDGAUSS, 60, ol.I_ADAPT_R2P, 21), (ol.T_DIAGGAUSS, 100, ol.I_ADAPT_D, 13), (ol.T_DIAGGAUSS, 17, ol.I_ADAPT_R2P, 9),
                                              (ol.T_STDGAUSS, 128, ol.I_ADAPT_D, 6), (ol.T_DIAGGAUSS, 64, ol.I_ADAPT_R2P, 5),
                                              (ol.T_STDGAUSS, 200, ol.I_ADAPT_R2P, 7), (ol.T_DIAGGAUSS, 256, ol.I_ADAPT_D, 5), (ol.T_DIAGGAUSS, 300, ol.I_ADAPT_R2P, 6),
                                              (ol.T_STDGAUSS, 512, ol.I_ADAPT_D, 3), (ol.T_STDGAUSS, 10, ol.I_ADAPT_R2P, 33), (ol.T_DIAGGAUSS, 16, ol.I_ADAPT_D, 20),
                                              (ol.T_DIAGGAUSS, 3, ol.I_ADAPT_R2P, 17), (ol.T_STDGAUSS, 200, ol.I_FIXED, 9), (ol.T_DIAGGAUSS, 256, ol.I_FIXED, 4)])
def test_std_and_diagonal_gaussians_adaptive_runs_one_chain_per_wave(wa, target, D, integ, C):
    """The reference's Gaussian scaling experiments (mainGaussESS.py / mainGaussTransient.py: stdGauss with the adaptive integrators over D):
    adaptLeapFrogD / R2P runs of std / diagonal Gaussians of up to 512 dimensions on the trajectory-first kernel (one register per lane up to
    16 dimensions: 3-5x the cooperative layout; beyond it the wide mode, wn_orbit_kernel<diag,NE=2|4|8,K=4>; 2.4-3.5x the cooperative layout up to 128 dimensions, 2-2.5x / 1.1-1.3x the wave layout's one-stream
    engine up to 256 / 512, tools/adapt_std.py) -- against that layout (WN_DIAG_COOP=1: another
    summation order: integers exactly, reals 1e-6) and against the oracle."""
    rng = np.random.default_rng(D)
    tpar = None if target == ol.T_STDGAUSS else np.geomspace(0.25, 6.0, D)
    q0 = rng.normal(size=(C, D)) / (1.0 if tpar is None else np.sqrt(tpar))
    iters = 5
    kw = dict(integrator=integ, H0=1.3 * D ** -0.25, delta0=0.2, M=6)
    s_new, d_new, name = run_engine(wa, target, q0, iters, tpar=tpar, **kw)
    assert ("wn_orbit_kernel<diag,NE=1," if D <= 16 else "wn_orbit_kernel<diag,NE=%d,K=4" % (2 if D <= 128 else 4 if D <= 256 else 8)) in name, name
    s_co, d_co, name_co = run_engine(wa, target, q0, iters, tpar=tpar, route=["DIAG_COOP"], **kw)
    # (adaptive runs of 129 .. 512 dimensions were the wave layout's one-stream engine's; fixedLeapFrog on 129 .. 256 -- also moved in round 5 -- the cooperative layout's)
    assert ("wn_coop_kernel" if D <= 128 or integ == ol.I_FIXED else "wn_wave_kernel<NE=8,diag>") in name_co, name_co
    assert same(d_new[..., INT_COLS], d_co[..., INT_COLS]) and near(s_new, s_co, 1e-6) and near(d_new, d_co, 1e-6)
    for c in sorted(set([0, C - 1])):
        o = ol.walnuts(target, q0[c], integrator=integ, tpar=tpar, H0=kw["H0"], delta0=0.2, numIter=iters, M=6, seed=SEED, chain=c)
        np.testing.assert_array_equal(d_new[c][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {c}")
        np.testing.assert_allclose(s_new[c].T, o["samples"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")
        np.testing.assert_allclose(d_new[c], o["diagnostics"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")


# ---- warm-up iterations on the trajectory-first kernel (round 5: WARM instantiations; mainFunnel.py:27-29, example.py:25-27 run
# adaptLeapFrogR2P with warmupIter = 1000) ----
@pytest.mark.parametrize("bits", [0, 16], ids=["reference_arithmetic", "adapt_quant_bits_16"])
def test_orbit_kernel_runs_the_warmup_iterations_funnel(wa, bits):
    """funnel10 / adaptLeapFrogR2P with warm-up adaptation of H and delta (WALNUTS.py:701-712; mainFunnel.py:27-29): the trajectory-first
    kernel against the one-stream engine with the engine's micro-step arithmetic (WN_ORBIT_EXACT).  Not bit for bit: the first iteration's
    energy error differs in its last bit between the two kernels (H0 comes from lp(q0), evaluated at another inlined site: same_run), the
    warm-up feeds it into delta and H, and the funnel's dynamics amplify it -- 5e-13 in H at iteration 4, 1e-6 at iteration 17, with the
    oracle's value in between.  So, with the reference's arithmetic: integer diagnostics exactly over the whole run, the adapted H / delta to 1e-4
    throughout; the default arithmetic (one reduction per micro step) against the oracle over the first 12 iterations (the warm-up contract: DESIGN 9,
    tests/test_gpu_warmup.py, profiles/r06_warmup_curve.txt).  With `adapt_quant_bits = 16` the loop through H / delta is cut: the two kernels agree
    to 1e-6 throughout and the default arithmetic is the oracle's over ALL 40 iterations."""
    C, iters, W = 96, 40, 25
    q0 = funnel_q0(C, seed=31)
    kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=8, warmupIter=W, adaptQuantBits=bits)
    s_new, d_new, name = run_engine(wa, ol.T_FUNNEL10, q0, iters, **kw)
    assert "wn_orbit_kernel<funnel10" in name, name
    s_old, d_old, name_old = run_engine(wa, ol.T_FUNNEL10, q0, iters, legacy=True, **kw)
    assert "wn_wave_kernel" in name_old, name_old
    assert same(d_new[..., INT_COLS], d_old[..., INT_COLS]), (np.argwhere(d_new[..., INT_COLS] != d_old[..., INT_COLS])[:5])
    assert near(d_new[..., [15, 18]], d_old[..., [15, 18]], 1e-6 if bits else 1e-4)
    assert len(np.unique(d_new[:, :W, 15])) > 10 and len(np.unique(d_new[:, :W, 18])) > 10  # H and delta do adapt
    assert (d_new[:, W:, 15] == d_new[:, W:W + 1, 15]).all()                                 # and stay put behind the warm-up
    s_f, d_f, _ = run_engine(wa, ol.T_FUNNEL10, q0, iters, exact=False, **kw)
    for c in (0, 50, 95):
        o = ol.walnuts(ol.T_FUNNEL10, q0[c], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=iters, warmupIter=W, M=8, seed=SEED, chain=c, adaptQuantBits=bits)
        n = iters if bits else 12
        np.testing.assert_array_equal(d_f[c][:n, INT_COLS], o["diagnostics"][:n, INT_COLS], err_msg=f"chain {c}")
        np.testing.assert_allclose(d_f[c][:n, [15, 18]], o["diagnostics"][:n, [15, 18]], rtol=1e-6, err_msg=f"chain {c}")
        np.testing.assert_allclose(s_f[c].T[:, :(n + 1 if bits else 4)], o["samples"][:, :(n + 1 if bits else 4)], rtol=1e-6, atol=1e-9, err_msg=f"chain {c}")


@pytest.mark.parametrize("what", ["std60_R2P", "diag100_D", "dense100_R2P", "std300_D", "smile_R2P"])
def test_orbit_kernel_runs_the_warmup_iterations_wide_and_2d(wa, what):
    """Warm-up on the wide-mode geometries (std / diagonal / dense Gaussians) and a 2-D target: against the kernels these
    runs used before (cooperative layout / one-stream engines: integers exactly, reals 1e-6) and the oracle."""
    rng = np.random.default_rng(7)
    iters, W = 24, 16
    env = []
    if what == "std60_R2P":
        target, tpar, q0, kw, env, old = ol.T_STDGAUSS, None, rng.normal(size=(18, 60)), dict(integrator=ol.I_ADAPT_R2P, H0=0.5, delta0=0.2, M=6), ["DIAG_COOP"], "wn_coop_kernel"
    elif what == "diag100_D":
        tpar = np.geomspace(0.3, 5.0, 100)
        target, q0, kw, env, old = ol.T_DIAGGAUSS, rng.normal(size=(11, 100)) / np.sqrt(tpar), dict(integrator=ol.I_ADAPT_D, H0=0.4, delta0=0.2, M=6), ["DIAG_COOP"], "wn_coop_kernel"
    elif what == "dense100_R2P":
        A, q0 = dense_case(100, 14)
        target, tpar, kw, env, old = ol.T_DENSEGAUSS, A, dict(integrator=ol.I_ADAPT_R2P, H0=0.8, delta0=0.2, M=6), ["ORBIT_LEGACY"], "wn_wavelds_kernel"
    elif what == "std300_D":
        target, tpar, q0, kw, env, old = ol.T_STDGAUSS, None, rng.normal(size=(6, 300)), dict(integrator=ol.I_ADAPT_D, H0=0.35, delta0=0.2, M=5), ["DIAG_COOP"], "wn_wave_kernel"
    else:
        target, tpar, q0, kw, env, old = ol.T_SMILE, None, rng.normal(size=(40, 2)), dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.2, M=7), ["ORBIT_LEGACY"], "wn_wave_kernel"
    kw = dict(kw, warmupIter=W)
    s_new, d_new, name = run_engine(wa, target, q0, iters, tpar=tpar, **kw)
    assert "wn_orbit_kernel" in name, name
    s_old, d_old, name_old = run_engine(wa, target, q0, iters, tpar=tpar, route=env, **kw)
    assert old in name_old and "wn_orbit_kernel" not in name_old, name_old
    assert same(d_new[..., INT_COLS], d_old[..., INT_COLS]), np.argwhere(d_new[..., INT_COLS] != d_old[..., INT_COLS])[:5]
    assert near(s_new, s_old, 1e-6) and near(d_new, d_old, 1e-6)
    assert len(np.unique(d_new[:, :W, 15])) > 5
    c = q0.shape[0] - 1
    o = ol.walnuts(target, q0[c], integrator=kw["integrator"], tpar=None if tpar is None else np.asarray(tpar).ravel(), H0=kw["H0"], delta0=kw.get("delta0", 0.05),
                   numIter=iters, warmupIter=W, M=kw["M"], seed=SEED, chain=c)
    np.testing.assert_array_equal(d_new[c][:, INT_COLS], o["diagnostics"][:, INT_COLS])
    np.testing.assert_allclose(d_new[c], o["diagnostics"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("target,integ,gen,gk", [(ol.T_FUNNEL10, ol.I_ADAPT_R2P, ol.G_FIRSTK, 2), (ol.T_FUNNEL10, ol.I_ADAPT_D, ol.G_IDENTITY, 0),
                                                  (ol.T_FUNNEL10, ol.I_FIXED, ol.G_IDENTITY, 0), (ol.T_SMILE, ol.I_ADAPT_R2P, ol.G_IDENTITY, 0)])
def test_orbit_kernel_records_orbit_statistics(wa, target, integ, gen, gk):
    """recordOrbitStats = True (WALNUTS.py:182-184, 274-276, 331-333; the funnel experiments mainFunnel.py:40, mainFunnelOrbitLen.py:40,
    mainFunnelTransient.py:32 run this way) on the trajectory-first kernel: orbitMin / orbitMax of generated(q) over every state of the orbit,
    against the one-stream engine (bit for bit: the same states) and -- through the recordOrbitStats goldens' entry point -- the oracle."""
    from walnuts_amd import _lib as L
    C, iters = 48, 10
    q0 = funnel_q0(C, seed=23) if target == ol.T_FUNNEL10 else np.random.default_rng(2).normal(size=(C, 2))
    kw = dict(integrator=integ, H0=0.3, delta0=0.3, M=8, generated=gen, gen_k=gk)
    out = {}
    for legacy in (False, True):
        e = wa.Engine(q0.shape[1], C, route=["ORBIT_EXACT"] + (["ORBIT_LEGACY"] if legacy else []), **kw).set_target(target, None)
        name = e.kernel_name
        s, d, omin, omax = e.run_orbit_stats(q0, SEED, iters)
        out[legacy] = (s.cpu().numpy(), d.cpu().numpy(), omin.cpu().numpy(), omax.cpu().numpy(), name)
        e.close()
    (s1, d1, mn1, mx1, n1), (s2, d2, mn2, mx2, n2) = out[False], out[True]
    assert "wn_orbit_kernel" in n1 and "wn_wave_kernel" in n2, (n1, n2)
    if integ == ol.I_FIXED:  # (another instantiation of the engine: integers exactly, reals to 1e-7)
        assert same(d1[..., INT_COLS], d2[..., INT_COLS]) and near(s1, s2, 1e-7) and near(mn1, mn2, 1e-7) and near(mx1, mx2, 1e-7)
    else:
        assert same(mn1, mn2) and same(mx1, mx2) and same(s1, s2)
    assert (mx1 >= mn1).all() and (mx1 > mn1).any()
    for c in (0, C - 1):
        o = ol.walnuts(target, q0[c], integrator=integ, H0=0.3, delta0=0.3, numIter=iters, M=8, seed=SEED, chain=c, generated=gen, gen_k=gk, recordOrbitStats=True)
        np.testing.assert_allclose(mn1[c].T, o["orbitMin"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c} orbitMin")
        np.testing.assert_allclose(mx1[c].T, o["orbitMax"], rtol=1e-6, atol=1e-9, err_msg=f"chain {c} orbitMax")
