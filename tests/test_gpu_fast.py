"""GPU tests of the headline kernels through the C ABI -- wn_k_fast4.hip (dense Gaussian, D <= 100: one T wave per four chains, a
retire and an issue wave per eight) and wn_k_fast.hip (std / diagonal Gaussians, D <= 112: seven T waves + one bookkeeper wave per
16-chain group): against the oracle on seeded inputs, per-step states of traced runs, against the one-stream kernel of the same layout
(WN_COOP_LEGACY=1: identical integers, draws to 1e-9) at full BASELINE size, the edge cases of tests/test_fast_book.py on the device,
and random configurations."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SEED = 20251121
INT_DIAG = (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22)


@pytest.fixture(scope="module")
def wa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import walnuts_amd
    return walnuts_amd


def close(a, b, rtol=1e-6, atol=1e-9, what=""):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    with np.errstate(all="ignore"):
        ok = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
             (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), f"{what}: {np.count_nonzero(~ok)} mismatches, first at {np.argwhere(~ok)[:4].tolist()}, " \
                     f"got {a[~ok][:4]}, want {b[~ok][:4]}"


def run(wa, target, tpar, q0, legacy=False, seven=False, whole=False, **kw):
    """legacy / seven / whole: the engine's `route` argument (wn_config.route, include/walnuts_amd.h) -- the one-stream cooperative kernel,
    the seven-T-wave kernel (wn_k_fast.hip) for a dense target, wn_k_fast4.hip's whole-group form; the default (route 0) is the library's choice."""
    C, D = q0.shape
    route = (["COOP_LEGACY"] if legacy else []) + (["FAST_LEGACY"] if seven else []) + (["FAST4_WHOLE"] if whole else [])
    e = wa.Engine(D, C, integrator=ol.I_FIXED, route=route, **{k: v for k, v in kw.items() if k not in ("numIter", "seed")})
    e.set_target(target, tpar)
    name = e.kernel_name
    s, d = e.run(q0, kw["seed"], kw["numIter"])
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy()
    ms = e.last_run_ms()
    e.close()
    return s, d, name, ms


def dense_problem(D, kappa, C, seed=SEED):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    return make_dense_problem(D, kappa, C, seed)


@pytest.mark.parametrize("D,C,M,gen", [(100, 48, 10, ol.G_Q0_SUMSQ), (100, 21, 6, ol.G_IDENTITY), (40, 16, 8, ol.G_FIRSTK),
                                         (64, 17, 7, ol.G_Q0_SUMSQ), (20, 33, 11, ol.G_Q0_SUMSQ), (97, 5, 3, ol.G_IDENTITY)])
def test_fast_dense_matches_oracle(wa, D, C, M, gen):
    A, q0 = dense_problem(D, 1e3, C)
    kw = dict(H0=0.5, M=M, generated=gen, gen_k=3 if gen == ol.G_FIRSTK else 0, chain_offset=11, numIter=5, seed=SEED + D)
    s, d, name, _ = run(wa, ol.T_DENSEGAUSS, A, q0, **kw)
    assert "wn_fast4_kernel" in name and "dense" in name, name
    for k in sorted({0, C - 1, C // 2, min(15, C - 1), min(16, C - 1)}):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=5, M=M, seed=SEED + D,
                       chain=11 + k, generated=gen, gen_k=kw["gen_k"])
        for col in INT_DIAG:
            np.testing.assert_array_equal(d[k][:, col], o["diagnostics"][:, col], err_msg=f"chain {k} diag col {col}")
        close(d[k], o["diagnostics"], what=f"chain {k} diagnostics")
        close(s[k], o["samples"], what=f"chain {k} draws")


@pytest.mark.parametrize("D,tk", [(100, ol.T_DIAGGAUSS), (30, ol.T_STDGAUSS), (112, ol.T_DIAGGAUSS), (65, ol.T_STDGAUSS)])
def test_fast_diag_matches_oracle(wa, D, tk):
    lam = np.logspace(0, 3, D)
    a = 1.0 / lam if tk == ol.T_DIAGGAUSS else None
    q0 = np.random.default_rng(D).normal(size=(35, D)) * (np.sqrt(lam) if a is not None else 1.0)
    H0 = 0.5 if a is not None else 1.4 * D ** -0.25
    kw = dict(H0=H0, M=9, generated=ol.G_Q0_SUMSQ, chain_offset=0, numIter=6, seed=SEED)
    for seven in (False, True):  # the T / R / I kernel (default) and the seven-T-wave kernel (WN_FAST_LEGACY=1)
        s, d, name, _ = run(wa, tk, a, q0, seven=seven, **kw)
        assert ("wn_fast_kernel" if seven else "wn_fast4_kernel") in name and "diag" in name, name
        for k in (0, 16, 34):
            o = ol.walnuts(tk, q0[k], integrator=ol.I_FIXED, tpar=a, H0=H0, numIter=6, M=9, seed=SEED, chain=k, generated=ol.G_Q0_SUMSQ)
            for col in INT_DIAG:
                np.testing.assert_array_equal(d[k][:, col], o["diagnostics"][:, col], err_msg=f"{name} chain {k} diag col {col}")
            close(d[k], o["diagnostics"], what=f"{name} chain {k} diagnostics")
            close(s[k], o["samples"], what=f"{name} chain {k} draws")


def test_fast_equals_one_stream_kernel_at_baseline_size(wa):
    """BASELINE configs[1] at the bench's size (4096 chains, D = 100, kappa = 1e4, M = 10), 12 iterations: the two
    kernels differ only in how the D-term sums are associated (per 16-row tile here, per pair of tiles there), so
    every integer column is identical and draws agree to 1e-9; 16 random chains against the oracle."""
    A, q0 = dense_problem(100, 1e4, 4096)
    kw = dict(H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=0, numIter=12, seed=SEED)
    s, d, name, ms = run(wa, ol.T_DENSEGAUSS, A, q0, **kw)
    s1, d1, name1, ms1 = run(wa, ol.T_DENSEGAUSS, A, q0, legacy=True, **kw)
    assert "wn_fast4_kernel" in name and "wn_coop_kernel" in name1, (name, name1)
    # the seven-T-wave kernel at the same size: identical integers too (each kernel sums the D terms of a dot product in its own order)
    s4, d4, name4, ms4 = run(wa, ol.T_DENSEGAUSS, A, q0, seven=True, **kw)
    assert "wn_fast_kernel" in name4, name4
    for col in INT_DIAG:
        np.testing.assert_array_equal(d4[:, :, col], d1[:, :, col], err_msg=f"seven-wave kernel diag col {col}")
    close(d4, d1, rtol=1e-9, atol=1e-10, what="seven-wave kernel diagnostics vs one-stream kernel")
    close(s4, s1, rtol=1e-9, atol=1e-10, what="seven-wave kernel draws vs one-stream kernel")
    print(f"seven-wave kernel {ms4:.1f} ms")
    for col in INT_DIAG:
        np.testing.assert_array_equal(d[:, :, col], d1[:, :, col], err_msg=f"diag col {col}")
    close(d, d1, rtol=1e-9, atol=1e-10, what="diagnostics vs one-stream kernel")
    close(s, s1, rtol=1e-9, atol=1e-10, what="draws vs one-stream kernel")
    steps = d[:, :, 6] + d[:, :, 7]
    assert (d[:, :, 2] > 0).all() and steps.sum() > 4096 * 12 * 100
    rng = np.random.default_rng(0)
    for k in sorted(set(rng.integers(0, 4096, size=14).tolist() + [0, 4095])):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=12, M=10, seed=SEED,
                       chain=k, generated=ol.G_Q0_SUMSQ)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k}")
        close(d[k], o["diagnostics"], what=f"diagnostics chain {k}")
    print(f"fast {ms:.1f} ms vs one-stream {ms1:.1f} ms")


@pytest.mark.parametrize("scale,H0", [(3e153, 0.05), (3e153, 0.5), (1e154, 0.5), (1e200, 0.05)])
def test_fast_huge_energies_on_device(wa, scale, H0):
    """Zero / infinite multinomial weights and overflowing energies (tests/test_fast_book.py) on the device."""
    D = 20
    q0 = np.random.default_rng(3).normal(size=(6, D)) * scale
    q0[4:] = np.random.default_rng(4).normal(size=(2, D))  # ordinary chains in the same group are not disturbed
    kw = dict(H0=H0, M=8, generated=ol.G_Q0_SUMSQ, chain_offset=3, numIter=8, seed=11)
    s, d, name, _ = run(wa, ol.T_STDGAUSS, None, q0, **kw)
    assert "wn_fast4_kernel" in name, name
    for k in range(6):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ol.I_FIXED, H0=H0, numIter=8, M=8, seed=11, chain=3 + k, generated=ol.G_Q0_SUMSQ)
        for col in INT_DIAG:
            np.testing.assert_array_equal(d[k][:, col], o["diagnostics"][:, col], err_msg=f"chain {k} diag col {col}")
        close(d[k], o["diagnostics"], what=f"chain {k} diagnostics")
        close(s[k], o["samples"], what=f"chain {k} draws")


def test_fast_one_iteration_and_ragged_groups(wa):
    A, q0 = dense_problem(100, 1e4, 19)
    s, d, name, _ = run(wa, ol.T_DENSEGAUSS, A, q0, H0=0.5, M=5, generated=ol.G_IDENTITY, numIter=1, seed=1)
    o = ol.walnuts(ol.T_DENSEGAUSS, q0[18], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=1, M=5, seed=1, chain=18)
    np.testing.assert_array_equal(s[:, :, 0], q0)
    close(s[18], o["samples"]); close(d[18], o["diagnostics"])


def test_fast_result_does_not_depend_on_the_previous_launch(wa):
    """LDS keeps what the previous kernel on the CU left there.  The bookkeeper reads the service wave's uniform entries
    of tick 0 (never written) at tick 1; a well-formed entry {iteration 1, block b, u0, u1} left by an EARLIER launch of
    this kernel (any numIter = 1 run) was taken for the chain's own uniforms when (iteration, block) matched.  The
    control areas are zeroed at kernel start now: the same problem gives the same bits whatever ran before it."""
    A, q0 = dense_problem(100, 1e4, 4096)
    kw = dict(H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=0, numIter=3, seed=SEED)
    s0, d0, name, _ = run(wa, ol.T_DENSEGAUSS, A, q0, seven=True, **kw)
    assert "wn_fast_kernel" in name
    for rep in range(4):
        # launches that leave iteration-1 entries of OTHER streams in every CU's LDS
        run(wa, ol.T_DENSEGAUSS, A, q0, seven=True, H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=7 + rep, numIter=1, seed=SEED + 1 + rep)
        s1, d1, _, _ = run(wa, ol.T_DENSEGAUSS, A, q0, seven=True, **kw)
        np.testing.assert_array_equal(d1, d0, err_msg=f"diagnostics, repetition {rep}")
        np.testing.assert_array_equal(s1, s0, err_msg=f"draws, repetition {rep}")


def _traced(wa, A, q0, pair, keep=None, **kw):
    """(samples, diagnostics, trace, trace_n) of a traced fixedLeapFrog run through the C ABI"""
    import torch
    C, D = q0.shape
    try:
        e = wa.Engine(D, C, integrator=ol.I_FIXED, **{k: v for k, v in kw.items() if k not in ("numIter", "seed", "trace_cap")})
        e.set_target(ol.T_DENSEGAUSS, A)
        name = e.kernel_name
        out = e.run(q0, kw["seed"], kw["numIter"], trace_cap=kw["trace_cap"])
        e.close()
    finally:
        pass
    s, d, tr, tn = out
    if keep is not None:  # (a 4096-chain trace is gigabytes: only the checked chains leave the device)
        tr = tr[keep]
    return s.transpose(1, 2).cpu().numpy(), d.cpu().numpy(), tr.cpu().numpy(), tn.cpu().numpy(), name


@pytest.mark.parametrize("D,C,M", [(100, 24, 7), (64, 9, 6), (97, 5, 8), (20, 13, 5)])
def test_pair_kernel_per_step_states_match_oracle(wa, D, C, M):
    """north_star: per-step (q, p) parity.  A traced run of the wave-pair kernel writes one row per integrator call
    (adaptiveIntegrators.py:49-59): the T wave the state (q', v'), the bookkeeper the scalars; every row of every checked
    chain against the oracle's trace to rtol 1e-6, row counts and integer columns exact."""
    A, q0 = dense_problem(D, 1e3, C)
    kw = dict(H0=0.5, M=M, generated=ol.G_IDENTITY, chain_offset=5, numIter=4, seed=SEED + D, trace_cap=4 * (1 << M))
    s, d, tr, tn, name = _traced(wa, A, q0, False, **kw)
    assert "wn_fast4_kernel" in name, name
    for k in sorted({0, C - 1, C // 2, min(3, C - 1), min(4, C - 1)}):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=4, M=M, seed=SEED + D,
                       chain=5 + k, trace_cap=kw["trace_cap"])
        n = o["ncalls"]
        assert tn[k] == n, (k, tn[k], n)
        cols = [0, 1, 2, 6, 7, 8, 10, 11]
        np.testing.assert_array_equal(tr[k, :n][:, cols], o["trace"][:n][:, cols], err_msg=f"chain {k} integer columns")
        close(tr[k, :n], o["trace"][:n], what=f"chain {k} trace rows")
        close(s[k], o["samples"], what=f"chain {k} draws")
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])


def test_pair_kernel_per_step_states_at_4096_chains(wa):
    """The same at BASELINE size: 4096 chains, D = 100, kappa = 1e4, M = 10, two iterations traced; 8 chains against
    the oracle step by step, and the traced run's draws equal the plain run's bit for bit (tracing changes nothing)."""
    A, q0 = dense_problem(100, 1e4, 4096)
    kw = dict(H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=0, numIter=2, seed=SEED)
    rng = np.random.default_rng(4)
    keep = sorted(set(rng.integers(0, 4096, size=6).tolist() + [0, 4095]))
    s, d, tr, tn, name = _traced(wa, A, q0, True, keep=keep, trace_cap=1400, **kw)
    assert "wn_fast4_kernel" in name, name
    sp, dp, namep, _ = run(wa, ol.T_DENSEGAUSS, A, q0, **kw)
    np.testing.assert_array_equal(s, sp); np.testing.assert_array_equal(d, dp)
    assert (tn == (d[:, :, 6] + d[:, :, 7]).sum(1)).all()
    for j, k in enumerate(keep):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=2, M=10, seed=SEED, chain=k,
                       generated=ol.G_Q0_SUMSQ, trace_cap=2100)
        n = o["ncalls"]
        assert tn[k] == n
        m = min(n, 1400)
        close(tr[j, :m], o["trace"][:m], what=f"chain {k} trace rows")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pair_kernel_random_configurations(wa, seed):
    """A short run of tools/fuzz_fast4.py inside the suite: random dimension (all four register geometries), chain count (one, two,
    four T waves per workgroup; ragged last quads and groups), depth (incl. M = 1, 2: orbits that end inside the bookkeeper's lag),
    step size, conditioning, generated kind, overflowing energies (stop code 999, zero / infinite weights), traced or not -- every
    diagnostics column, every draw and every trace row of the checked chains against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    for it in range(14):
        D = int(rng.choice([5, 16, 17, 32, 33, 64, 65, 96, 97, 100]))
        C = int(rng.choice([1, 3, 4, 5, 17, 70, 300, 1100, 2500]))
        M = int(rng.integers(1, 8))
        numIter = int(rng.integers(1, 5))
        Q, _ = np.linalg.qr(rng.normal(size=(D, D))); lam = np.logspace(0, rng.uniform(0.0, 4.0), D)
        A = (Q * (1.0 / lam)) @ Q.T; A = 0.5 * (A + A.T)
        q0 = rng.normal(size=(C, D)) @ (Q * np.sqrt(lam)).T
        if rng.random() < 0.15:
            q0[: max(1, C // 7)] *= 10.0 ** rng.uniform(100, 200)
        H0 = float(rng.uniform(0.2, 1.1))
        gen, gk = (ol.G_IDENTITY, 0) if rng.random() < 0.4 else ((ol.G_Q0_SUMSQ, 0) if rng.random() < 0.5 else (ol.G_FIRSTK, int(rng.integers(1, 4))))
        traced = rng.random() < 0.4 and C <= 300
        cap = numIter * (1 << M) + 8
        off = int(rng.integers(0, 1 << 20))
        e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=H0, M=M, generated=gen, gen_k=gk, chain_offset=off)
        e.set_target(ol.T_DENSEGAUSS, A)
        assert "wn_fast4_kernel" in e.kernel_name, e.kernel_name
        out = e.run(q0, 99 + it, numIter, trace_cap=cap if traced else 0)
        s = out[0].transpose(1, 2).cpu().numpy(); d = out[1].cpu().numpy()
        tr, tn = (out[2].cpu().numpy(), out[3].cpu().numpy()) if traced else (None, None)
        e.close()
        what = f"seed {seed} it {it}: D={D} C={C} M={M} iters={numIter} gen={gen} traced={traced}"
        for k in sorted({0, C - 1, C // 2, min(C - 1, 4), int(rng.integers(0, C))}):
            o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=H0, numIter=numIter, M=M, seed=99 + it,
                           chain=off + k, generated=gen, gen_k=gk, trace_cap=cap if traced else 0)
            for col in INT_DIAG:
                np.testing.assert_array_equal(d[k][:, col], o["diagnostics"][:, col], err_msg=f"{what} chain {k} diag col {col}")
            with np.errstate(all="ignore"):
                sc = np.nanmax(np.abs(o["samples"])) if np.isfinite(o["samples"]).any() else 1.0
            close(d[k][:, [2, 3, 15, 23]], o["diagnostics"][:, [2, 3, 15, 23]], rtol=1e-6, atol=1e-9, what=what + f" chain {k} diag")
            close(s[k], o["samples"], rtol=1e-6, atol=1e-9 + 1e-6 * (sc if np.isfinite(sc) else 1.0), what=what + f" chain {k} draws")
            if traced:
                m = o["ncalls"]
                assert tn[k] == m, (what, k, tn[k], m)
                np.testing.assert_array_equal(tr[k, :m][:, [0, 1, 2, 10]], o["trace"][:m][:, [0, 1, 2, 10]], err_msg=what + f" chain {k} trace integers")


@pytest.mark.parametrize("D,C", [(100, 1024), (96, 1000), (80, 517), (100, 3)])
def test_pair_kernel_two_position_waves_per_group(wa, D, C):
    """Fewer than two groups per CU (BASELINE configs[1] at its literal 1024 chains): a group's register sets are cut over TWO T waves
    (wn_k_fast4.hip, halves mode).  Against the whole-group form of the same kernel (route WN_ROUTE_FAST4_WHOLE: the second wave sums its k-steps
    in another order -- integers identical, reals to 1e-9), against the one-stream kernel and, chain by chain, against the oracle."""
    A, q0 = dense_problem(D, 1e4, C)
    kw = dict(H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=0, numIter=10, seed=SEED)
    s, d, name, ms = run(wa, ol.T_DENSEGAUSS, A, q0, **kw)
    assert "wn_fast4_kernel" in name, name
    s0, d0, name0, ms0 = run(wa, ol.T_DENSEGAUSS, A, q0, whole=True, **kw)
    s1, d1, name1, _ = run(wa, ol.T_DENSEGAUSS, A, q0, legacy=True, **kw)
    assert "wn_coop_kernel" in name1, name1
    for ref_d, ref_s, what in ((d0, s0, "whole-group form"), (d1, s1, "one-stream kernel")):
        for col in INT_DIAG:
            np.testing.assert_array_equal(d[:, :, col], ref_d[:, :, col], err_msg=f"diag col {col} vs {what}")
        close(d, ref_d, rtol=1e-9, atol=1e-10, what=f"diagnostics vs {what}")
        close(s, ref_s, rtol=1e-9, atol=1e-10, what=f"draws vs {what}")
    for k in sorted({0, C - 1, C // 2, C // 3}):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=10, M=10, seed=SEED, chain=k,
                       generated=ol.G_Q0_SUMSQ)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k}")
    print(f"two T waves per group {ms:.1f} ms, one {ms0:.1f} ms")


@pytest.mark.parametrize("bits", [0, 16], ids=["reference_arithmetic", "adapt_quant_bits_16"])
@pytest.mark.parametrize("kind,D,C", [("dense", 100, 4096), ("dense", 100, 1024), ("dense", 40, 300), ("diag", 112, 2100), ("std", 10, 64)])
def test_pair_kernel_runs_the_warmup_iterations(wa, kind, D, C, bits):
    """Round 5: the reference's DEFAULT call -- integrator=fixedLeapFrog with warm-up adaptation of H and delta
    (WALNUTS.py:111-125, 313, 701-712) -- runs on the headline kernel: the retire wave pushes log(igrConst) into the P2 markers and adapts
    H / delta at every warm-up iteration's end, the issue wave draws the next iteration's step sizes around the new H.  Per chain against
    the oracle: H (col 15) and delta (col 18) of every iteration, the integer columns exactly, the draws.
    Tolerance (the warm-up contract, DESIGN 9 / tests/test_gpu_warmup.py): with `adapt_quant_bits = 16` (the feedback loop through H / delta cut,
    both sides) rtol 1e-6 on H, delta and every draw over the whole run.  With the reference's arithmetic the loop amplifies the last-bit
    differences of the two sides' D-term sums (the reference's own result moves as much under a 1-ulp change of its input:
    profiles/r06_warmup_curve.txt), so: rtol 1e-6 over the first 8 iterations, 1e-5 over these 20 (4 chains; kappa = 1e2, M = 6)."""
    rng = np.random.default_rng(D + C)
    if kind == "dense":
        A, q0 = dense_problem(D, 1e2, C, 5)
        tk, tp = ol.T_DENSEGAUSS, A.ravel()
    elif kind == "diag":
        lam = np.logspace(0, 1.5, D)
        q0 = rng.normal(size=(C, D)) * np.sqrt(lam)
        tk, tp = ol.T_DIAGGAUSS, 1.0 / lam
    else:
        q0 = rng.normal(size=(C, D))
        tk, tp = ol.T_STDGAUSS, None
    kw = dict(H0=0.3, delta0=0.05, M=6)
    e = wa.Engine(D, C, integrator=ol.I_FIXED, warmupIter=14, chain_offset=50, adaptQuantBits=bits, **kw)
    e.set_target(tk, tp)
    assert "wn_fast4_kernel" in e.kernel_name, e.kernel_name
    s, d = e.run(q0, SEED, 20)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy()
    e.close()
    for k in (0, 1, C // 2, C - 1):
        o = ol.walnuts(tk, q0[k], integrator=ol.I_FIXED, tpar=tp, numIter=20, seed=SEED, chain=50 + k, warmupIter=14, adaptQuantBits=bits, **kw)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]], err_msg=f"{kind} chain {k}")
        assert len(set(np.round(d[k][:, 15], 12))) > 3  # H did adapt
        if bits:
            np.testing.assert_allclose(d[k][:, [15, 18]], o["diagnostics"][:, [15, 18]], rtol=1e-6, err_msg=f"H / delta, {kind} chain {k}")
            np.testing.assert_allclose(s[k], o["samples"], rtol=1e-6, atol=1e-9, err_msg=f"draws, {kind} chain {k}")
        else:
            np.testing.assert_allclose(d[k][:8, [15, 18]], o["diagnostics"][:8, [15, 18]], rtol=1e-6, err_msg=f"H / delta, first iterations, {kind} chain {k}")
            np.testing.assert_allclose(s[k][:, :9], o["samples"][:, :9], rtol=1e-6, atol=1e-9, err_msg=f"draws, first iterations, {kind} chain {k}")
            np.testing.assert_allclose(d[k][:, [15, 18]], o["diagnostics"][:, [15, 18]], rtol=1e-5, err_msg=f"H / delta, {kind} chain {k}")
            np.testing.assert_allclose(s[k], o["samples"], rtol=1e-5, atol=1e-7, err_msg=f"draws, {kind} chain {k}")
