"""The warm-up tolerance contract (Next #3), on the GPU through the C ABI.

With `warmupIter > 0` a WALNUTS run is a feedback loop: every iteration's energies adapt the step size H and the tolerance delta
(WALNUTS.py:701-712, P2quantile.py:41-86), which shape the next iteration's energies.  On ill-conditioned targets the loop amplifies a
last-bit difference by ~1e3 per iteration -- the REFERENCE'S OWN result leaves 1e-6 within ~15 iterations when one input coordinate
moves by one ulp (tests/studies/warmup_sensitivity.py; profiles/r06_warmup_curve.txt has the engine's curve beside the oracle's own).
So, with warm-up:
  (1) what holds for the reference's arithmetic as it is (`adapt_quant_bits = 0`): every chain is the oracle's to 1e-6 (integers
      exactly) until the loop has amplified the first last-bit difference -- the engine leaves no earlier than the oracle leaves
      ITSELF under a 1-ulp perturbation -- and the adapted H / delta have the oracle's DISTRIBUTION (two-sample KS below);
  (2) with the opt-in `adapt_quant_bits = 16` (ABI 5; the adapted H / delta rounded to 16 mantissa bits on both sides, which cuts the
      loop): >= 97 % of the chains are the oracle's to 1e-6, integers exactly, through the WHOLE warm-up.
"""
import os
import sys
from multiprocessing import get_context

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as ol  # noqa: E402

pytestmark = pytest.mark.gpu
SEED = 20251121
INT_COLS = [0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22]


@pytest.fixture(scope="module")
def wa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import walnuts_amd
    return walnuts_amd


def _oracle_one(a):
    target, tpar, q0, chain, iters, warm, kw = a
    o = ol.walnuts(target, q0, tpar=tpar, numIter=iters, warmupIter=warm, seed=SEED, chain=chain, **kw)
    return o["diagnostics"], o["samples"].T


def oracle_many(target, tpar, q0, iters, warm, kw):
    """every chain through the oracle, on the host cores (spawned workers: the parent holds a GPU context)"""
    C = q0.shape[0]
    args = [(target, tpar, q0[c], c, iters, warm, kw) for c in range(C)]
    with get_context("spawn").Pool(min(os.cpu_count() or 1, 64)) as pool:
        r = pool.map(_oracle_one, args, chunksize=max(1, C // 256))
    return np.stack([x[0] for x in r]), np.stack([x[1] for x in r])


def engine_run(wa, target, tpar, q0, iters, warm, kw, bits=0):
    e = wa.Engine(q0.shape[1], q0.shape[0], warmupIter=warm, adaptQuantBits=bits, **kw).set_target(target, tpar)
    s, d = e.run(q0, SEED, iters)
    s = s.cpu().numpy(); d = d.cpu().numpy(); name = e.kernel_name; e.close()
    return d, s, name


def within(dg, sg, do, so, rtol=1e-6, atol=1e-9):
    """[C] bool: H (col 15), delta (col 18) and every draw of the chain agree with the oracle over the whole run"""
    rel = lambda x, y: np.abs(x - y) <= atol + rtol * np.abs(y)
    return rel(dg[..., [15, 18]], do[..., [15, 18]]).all(axis=(1, 2)) & rel(sg, so).all(axis=(1, 2))


def problem(name, C):
    from bench import make_dense_problem
    rng = np.random.default_rng(11)
    if name == "std100_fixed":
        return ol.T_STDGAUSS, None, rng.normal(size=(C, 100)), dict(integrator=ol.I_FIXED, H0=0.2, delta0=0.05, M=10), "wn_fast4_kernel"
    if name == "dense100_fixed":
        A, q0 = make_dense_problem(100, 1e4, C, SEED)
        return ol.T_DENSEGAUSS, A.ravel(), q0, dict(integrator=ol.I_FIXED, H0=0.5, delta0=0.05, M=10), "wn_fast4_kernel"
    if name == "dense32_default_call":  # WALNUTS(lpFun, q0) with every default of WALNUTS.py:111-129
        A, q0 = make_dense_problem(32, 1e3, C, SEED)
        return ol.T_DENSEGAUSS, A.ravel(), q0, dict(integrator=ol.I_FIXED, H0=0.2, delta0=0.05, M=10), "wn_fast4_kernel"
    qf = np.zeros((C, 11)); qf[:, 0] = 3.0 * rng.normal(size=C); qf[:, 1:] = np.exp(0.5 * qf[:, :1]) * rng.normal(size=(C, 10))
    return ol.T_FUNNEL10, None, qf, dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=10), "wn_orbit_kernel"


@pytest.mark.parametrize("name,C,iters", [("std100_fixed", 256, 120), ("dense100_fixed", 192, 100), ("funnel10_R2P", 256, 120)])
def test_warmup_with_adapt_quant_bits_is_the_oracle_through_the_whole_warmup(wa, name, C, iters):
    """Contract (2): adapt_quant_bits = 16, every iteration a warm-up iteration.  Measured at 1000 chains x 200 iterations
    (profiles/r06_warmup_curve.txt): 100 % / 99.0 % / 99.6 % of the chains within 1e-6 -- the same as the oracle against itself under a
    1-ulp perturbation (100 % / 98.7 % / 99.6 %).  Here: >= 97 %, and EVERY chain inside has the oracle's integer columns exactly."""
    target, tpar, q0, kw, kernel = problem(name, C)
    dg, sg, kname = engine_run(wa, target, tpar, q0, iters, iters, kw, bits=16)
    assert kernel in kname, kname
    do, so = oracle_many(target, tpar, q0, iters, iters, dict(kw, adaptQuantBits=16))
    ok = within(dg, sg, do, so)
    assert ok.mean() >= 0.97, (name, ok.mean(), np.flatnonzero(~ok)[:10])
    np.testing.assert_array_equal(dg[ok][..., INT_COLS], do[ok][..., INT_COLS])
    # the adapted values really are rounded: 16 explicit mantissa bits, and the adaptation did run
    m, _ = np.frexp(dg[:, 20:, 15])
    assert np.all(np.ldexp(m, 17) == np.round(np.ldexp(m, 17)))
    assert len(np.unique(dg[0, :, 15])) > 20


@pytest.mark.parametrize("name,C,iters", [("std100_fixed", 128, 60), ("dense100_fixed", 128, 40), ("funnel10_R2P", 256, 60)])
def test_warmup_reference_arithmetic_leaves_the_oracle_no_earlier_than_the_oracle_leaves_itself(wa, name, C, iters):
    """Contract (1), chain by chain: with the reference's arithmetic (adapt_quant_bits = 0) the engine's chains leave the oracle's 1e-6
    only as the loop amplifies last-bit differences -- measured against the oracle run a second time with q0[0] moved by ONE ulp (what
    NumPy's pairwise `np.sum` against the builtin `sum` of adaptiveIntegrators.py:55 would do to the reference itself).  The engine's
    differences start larger than one ulp (its D-term sums are associated differently: ~1e-13 relative in an energy), so its curve
    sits a few iterations ahead of the perturbed oracle's: the fraction of chains still inside after k iterations must be at least the
    perturbed oracle's fraction after 1.5 k + 5.  And while a chain IS inside, its integer columns are the oracle's exactly."""
    target, tpar, q0, kw, kernel = problem(name, C)
    dg, sg, kname = engine_run(wa, target, tpar, q0, iters, iters, kw)
    assert kernel in kname, kname
    do, so = oracle_many(target, tpar, q0, iters, iters, kw)
    qb = q0.copy(); qb[:, 0] = np.nextafter(qb[:, 0], np.inf)
    dp, sp = oracle_many(target, tpar, qb, iters, iters, kw)

    def inside_until(d, s):  # [C, iters] bool: the chain agrees with the oracle up to and including iteration k
        rel = lambda x, y: np.abs(x - y) <= 1e-9 + 1e-6 * np.abs(y)
        ok = rel(d[..., 15], do[..., 15]) & rel(d[..., 18], do[..., 18]) & rel(s[:, 1:], so[:, 1:]).all(-1)
        return np.logical_and.accumulate(ok, axis=1)

    ig, ip = inside_until(dg, sg), inside_until(dp, sp)
    fg, fp = ig.mean(0), ip.mean(0)
    for k in range(iters):
        k2 = min(iters - 1, int(1.5 * k) + 5)
        assert fg[k] >= fp[k2] - 0.05, (name, k, fg[k], k2, fp[k2])
    same_int = (dg[..., INT_COLS] == do[..., INT_COLS]).all(-1)
    assert same_int[ig].all(), np.argwhere(ig & ~same_int)[:5]
    assert fg[4] >= 0.85  # (nobody leaves before the loop has had time to amplify anything)


def test_warmup_default_call_adapted_step_size_has_the_oracles_distribution(wa):
    """Contract (1), in distribution: the reference's DEFAULT call -- fixedLeapFrog, H0 = 0.2, delta0 = 0.05, M = 10, warmupIter = 1000,
    adaptH, adaptDelta (WALNUTS.py:111-129) -- 1000 chains x 1000 warm-up iterations on a 32-D dense Gaussian (kappa = 1e3), engine
    against oracle on the same random streams.  After 1000 iterations of the feedback loop few chains are still within 1e-6 of their
    oracle twin -- nor would the reference be of itself -- but the adapted step size H and tolerance delta at the end of the warm-up
    are draws from the same distribution: two-sample Kolmogorov-Smirnov, p > 0.01, on both; the chains that ARE still inside have the
    oracle's integer columns exactly; and the sampler's output is the same in distribution (mean / variance of q[0] over chains)."""
    from scipy.stats import ks_2samp
    C, iters = 1000, 1000
    target, tpar, q0, kw, kernel = problem("dense32_default_call", C)
    dg, sg, kname = engine_run(wa, target, tpar, q0, iters, iters, kw)
    assert kernel in kname, kname
    do, so = oracle_many(target, tpar, q0, iters, iters, kw)
    pH = ks_2samp(dg[:, -1, 15], do[:, -1, 15]).pvalue
    pd = ks_2samp(dg[:, -1, 18], do[:, -1, 18]).pvalue
    ok = within(dg, sg, do, so)
    print(f"KS p-values: H {pH:.3f}, delta {pd:.3f}; chains within 1e-6 of the oracle after {iters} warm-up iterations: {ok.mean():.3f}; "
          f"H median {np.median(dg[:, -1, 15]):.4f} / {np.median(do[:, -1, 15]):.4f}")
    assert pH > 0.01 and pd > 0.01, (pH, pd)
    np.testing.assert_array_equal(dg[ok][..., INT_COLS], do[ok][..., INT_COLS])
    assert len(np.unique(np.round(dg[:, -1, 15], 6))) > 100  # H did adapt, chain by chain
    # total leapfrog steps: the same work to 2 % (each chain's count depends on its own H)
    tg, to = (dg[..., 6] + dg[..., 7]).sum(), (do[..., 6] + do[..., 7]).sum()
    assert abs(tg - to) <= 0.02 * to, (tg, to)


@pytest.mark.parametrize("what", ["lane_smile_R2P", "coop_dense40_FlowD", "wave_std300_D", "wavelds_dense100_Yoshida", "split_dense200_fixed", "orbit_std60_R2P",
                                  "fast_diag112_fixed"])
def test_adapt_quant_bits_on_every_kernel_family(wa, what):
    """`adapt_quant_bits` lives in the one function every kernel family adapts through (wn_adapt_update, wn_chain.h): a short run with
    warm-up on each family -- one chain per lane, the cooperative MFMA layout, one chain per wavefront, the dense-LDS engine, split mode,
    the trajectory-first kernel, the headline kernel's diagonal form -- with 16 bits, against the oracle with the same switch: integers
    exactly, H / delta / draws to 1e-6 over the whole run, and every adapted value rounded."""
    from bench import make_dense_problem
    rng = np.random.default_rng(23)
    iters, W, bits = 24, 24, 16
    if what == "lane_smile_R2P":
        tk, tp, q0, kw, kern = ol.T_SMILE, None, rng.normal(size=(20000, 2)), dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.2, M=6), "wn_lane_kernel"
    elif what == "coop_dense40_FlowD":
        A, q0 = make_dense_problem(40, 1e2, 24, 5)
        tk, tp, kw, kern = ol.T_DENSEGAUSS, A.ravel(), dict(integrator=ol.I_ADAPT_FLOWD, H0=0.5, delta0=0.05, M=6), "wn_coop_kernel"
    elif what == "wave_std300_D":
        tk, tp, q0, kw, kern = ol.T_STDGAUSS, None, rng.normal(size=(8, 300)), dict(integrator=ol.I_ADAPT_YOSHIDA, H0=0.4, delta0=0.1, M=5), "wn_wave_kernel"
    elif what == "wavelds_dense100_Yoshida":
        A, q0 = make_dense_problem(100, 1e2, 10, 5)
        tk, tp, kw, kern = ol.T_DENSEGAUSS, A.ravel(), dict(integrator=ol.I_ADAPT_YOSHIDA, H0=0.8, delta0=0.1, M=5), "wn_wavelds_kernel"
    elif what == "split_dense200_fixed":
        A, q0 = make_dense_problem(200, 1e2, 6, 5)
        tk, tp, kw, kern = ol.T_DENSEGAUSS, A.ravel(), dict(integrator=ol.I_FIXED, H0=0.4, delta0=0.05, M=5), "split"
    elif what == "orbit_std60_R2P":
        tk, tp, q0, kw, kern = ol.T_STDGAUSS, None, rng.normal(size=(18, 60)), dict(integrator=ol.I_ADAPT_R2P, H0=0.5, delta0=0.2, M=6), "wn_orbit_kernel"
    else:
        lam = np.logspace(0, 1.5, 112)
        tk, tp, q0, kw, kern = ol.T_DIAGGAUSS, 1.0 / lam, rng.normal(size=(40, 112)) * np.sqrt(lam), dict(integrator=ol.I_FIXED, H0=0.3, delta0=0.05, M=6), "wn_fast4_kernel"
    dg, sg, kname = engine_run(wa, tk, tp, q0, iters, W, kw, bits=bits)
    assert kern in kname, kname
    chains = sorted(set([0, q0.shape[0] // 2, q0.shape[0] - 1]))
    for c in chains:
        o = ol.walnuts(tk, q0[c], tpar=tp, numIter=iters, warmupIter=W, seed=SEED, chain=c, adaptQuantBits=bits, **kw)
        np.testing.assert_array_equal(dg[c][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"{what} chain {c}")
        np.testing.assert_allclose(dg[c][:, [15, 18]], o["diagnostics"][:, [15, 18]], rtol=1e-6, err_msg=f"{what} chain {c}")
        np.testing.assert_allclose(sg[c], o["samples"].T, rtol=1e-6, atol=1e-9, err_msg=f"{what} chain {c}")
    m, _ = np.frexp(dg[chains, 12:, 15])
    assert np.all(np.ldexp(m, bits + 1) == np.round(np.ldexp(m, bits + 1)))
    assert len(np.unique(dg[chains[0], :, 15])) > 5
