"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, and the
engine's chain state machine (walnuts_amd/csrc/wn_chain.h), instantiated on the host by
tests/hostsim, reproduces the oracle on every golden configuration."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import hostsim_lib as hs
import oracle_lib as ol
import test_oracle_golden as tg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    from walnuts_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "walnuts_amd.h")).read()
    declared = set(re.findall(r"\b(wn_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()  # raises ImportError if the HIP library is not built (no fallback)
    for s in declared:
        assert hasattr(L, s), s
    assert L.wn_abi_version() == _lib.WN_ABI_VERSION
    assert C.sizeof(_lib.WnConfig) == 136  # ABI 5: + route, split_parts, split_ks, adapt_quant_bits


def test_routing_is_configuration_not_environment():
    """ABI 5 (Next #6): which of two equivalent kernels runs a configuration is `wn_config.route`, as the reference's
    boundary is keyword arguments (WALNUTS.py:111-129).  The default build of the library reads NO environment variable: every getenv
    left in walnuts_amd/csrc sits inside an `#ifdef WN_DIAG` region (diagnostic builds of tools/build_variant.sh), the shared object
    does not import getenv, and wn_create validates the new fields before any HIP call."""
    import subprocess
    from walnuts_amd import _lib
    src = os.path.join(ROOT, "walnuts_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if not fn.endswith((".hip", ".h")):
            continue
        depth_diag = []
        for ln, line in enumerate(open(os.path.join(src, fn)), 1):
            t = line.strip()
            if t.startswith("#if"):
                depth_diag.append("WN_DIAG" in t and "ifndef" not in t)
            elif t.startswith("#else") and depth_diag:
                depth_diag[-1] = False
            elif t.startswith("#endif") and depth_diag:
                depth_diag.pop()
            elif "getenv(" in t and not t.startswith("//"):
                assert any(depth_diag), f"{fn}:{ln}: getenv outside an #ifdef WN_DIAG region: {t[:90]}"
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in nm, "libwalnuts_amd.so imports getenv"
    hdr = open(os.path.join(ROOT, "include", "walnuts_amd.h")).read()
    names = re.findall(r"WN_ROUTE_([A-Z0-9_]+) = 1 << (\d+)", hdr)
    assert {n: 1 << int(k) for n, k in names} == _lib.ROUTE and len(names) == 17
    assert _lib.route_bits(["ORBIT_LEGACY", "WN_ROUTE_DIAG_COOP"]) == (1 | 1 << 7) and _lib.route_bits(None) == 0 and _lib.route_bits(5) == 5
    L = _lib.lib()
    cfg = _lib.WnConfig()
    cfg.abi_version = _lib.WN_ABI_VERSION
    cfg.dim, cfg.n_chains, cfg.M, cfg.max_c = 3, 8, 5, 10
    cfg.H0, cfg.delta0, cfg.step_size_rand_scale, cfg.r2p_prob0 = 0.2, 0.05, 0.2, 2.0 / 3.0
    h = C.c_void_p()
    for field, bad, msg in (("route", 1 << 17, b"WN_ROUTE"), ("adapt_quant_bits", 5, b"adapt_quant_bits"), ("adapt_quant_bits", 53, b"adapt_quant_bits"),
                            ("split_parts", 5, b"split_parts"), ("split_ks", 3, b"split_parts")):
        setattr(cfg, field, bad)
        assert L.wn_create(C.byref(cfg), C.byref(h)) == _lib.E_ARG, field
        assert msg in L.wn_last_error(None), (field, L.wn_last_error(None))
        setattr(cfg, field, 0)
    cfg.abi_version = 4  # a caller built against the previous header is refused, not misread
    assert L.wn_create(C.byref(cfg), C.byref(h)) == _lib.E_ARG and b"abi_version" in L.wn_last_error(None)


def test_wn_create_rejects_chain_ids_beyond_32_bits():
    """Argument checks run before any HIP call (no GPU needed): global chain ids are one 32-bit Philox word."""
    from walnuts_amd import _lib
    L = _lib.lib()
    cfg = _lib.WnConfig()
    cfg.abi_version = _lib.WN_ABI_VERSION
    cfg.dim, cfg.n_chains, cfg.M, cfg.max_c = 3, 8, 5, 10
    cfg.H0, cfg.delta0, cfg.step_size_rand_scale, cfg.r2p_prob0 = 0.2, 0.05, 0.2, 2.0 / 3.0
    h = C.c_void_p()
    for off in (2 ** 32 - 7, -1, 2 ** 40):
        cfg.chain_offset = off
        assert L.wn_create(C.byref(cfg), C.byref(h)) == _lib.E_ARG
        assert b"chain_offset" in L.wn_last_error(None)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import walnuts_amd as wa
    with pytest.raises(wa.WalnutsError):
        wa.Engine(3, 4)
    from walnuts_amd import targetDistr as td
    with pytest.raises(wa.WalnutsError):
        wa.WALNUTS(td.stdGauss, np.zeros(3), warmupIter=0, numIter=2)


def test_package_never_imports_oracle():
    """The product path must not route through oracle/ or tests/hostsim."""
    pkg = os.path.join(ROOT, "walnuts_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in txt and "walnuts_oracle" not in txt and "hostsim" not in txt, f


@pytest.mark.parametrize("name", tg.CASES)
def test_state_machine_on_host_matches_oracle(name):
    c, ref = tg.load_case(name)
    seed = int(tg.TR["seed"][0])
    o = tg.run_oracle(c, ref, tape=False)
    h = hs.walnuts(c["target"], c["q0"], integrator=c["integrator"], tpar=c["tpar"], H0=c["H0"],
                   delta0=c["delta0"], numIter=c["numIter"], M=c["M"], minC=c["minC"], maxC=c["maxC"],
                   generated=c["generated"], gen_k=c["gen_k"], seed=seed, chain0=c["chain"],
                   trace_cap=len(ref["trace"]) + 8, warmupIter=c["warmupIter"], adaptH=c["adaptH"],
                   adaptHtarget=c["adaptHtarget"], adaptDelta=c["adaptDelta"],
                   adaptDeltaTarget=c["adaptDeltaTarget"], adaptDeltaQuantile=c["adaptDeltaQuantile"],
                   recordOrbitStats="orbitMin" in ref)
    if "orbitMin" in ref:
        tg.close(h["orbitMin"][0], o["orbitMin"], rtol=1e-12, atol=1e-13)
        tg.close(h["orbitMax"][0], o["orbitMax"], rtol=1e-12, atol=1e-13)
    n = o["ncalls"]
    assert int(h["trace_n"][0]) == n
    tr = h["trace"][0, :n]
    for col in (0, 1, 2, 6, 7, 8, 10, 11):
        np.testing.assert_array_equal(tr[:, col], o["trace"][:, col], err_msg=f"col {col}")
    # the engine's funnel log-density is algebraically simplified (last-bit differences from the
    # oracle's literal scipy form); with warm-up those feed the adapted step size and are amplified
    rt, at = (1e-7, 1e-9) if (c["warmupIter"] > 0 and c["target"] in tg.EXP_TARGETS) else (1e-12, 1e-13)
    tg.close(tr[:, 3:6], o["trace"][:, 3:6], rtol=rt, atol=at)
    tg.close(tr[:, 12:], o["trace"][:, 12:], rtol=rt, atol=at)
    tg.close(h["samples"][0], o["samples"], rtol=rt, atol=at)
    d, rd = h["diagnostics"][0], o["diagnostics"]
    for col in (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22):
        np.testing.assert_array_equal(d[:, col], rd[:, col], err_msg=f"diag col {col}")
    tg.close(d, rd, rtol=max(rt, 1e-10), atol=max(at, 1e-12))


def test_state_machine_many_chains_host():
    rng = np.random.default_rng(1)
    q0 = rng.normal(size=(6, 10))
    h = hs.walnuts(ol.T_STDGAUSS, q0, integrator=ol.I_ADAPT_R2P, H0=0.9, delta0=0.2, numIter=15, M=7,
                   seed=99, chain0=500, generated=1)
    for k in range(6):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ol.I_ADAPT_R2P, H0=0.9, delta0=0.2, numIter=15,
                       M=7, seed=99, chain=500 + k, generated=1)
        tg.close(h["samples"][k], o["samples"], rtol=1e-12, atol=1e-13)
        np.testing.assert_array_equal(h["diagnostics"][k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])


def test_walnuts_argument_errors_match_reference():
    """ValueError cases of walnuts_step (walnuts/walnuts.py:309-320) are raised before any device work."""
    import walnuts_amd as wa
    from walnuts_amd import targets as tt
    rng = np.random.default_rng(0)
    sn = (tt.standard_normal_lpdf, tt.standard_normal_grad)
    for theta, im, ms, depth, err in [
        (np.zeros((2, 2)), np.ones(2), 1.0, 5, 0.1),
        (np.zeros(2), np.ones((2, 1)), 1.0, 5, 0.1),
        (np.zeros(2), np.ones(3), 1.0, 5, 0.1),
        (np.zeros(2), np.ones(2), 0.0, 5, 0.1),
        (np.zeros(2), np.ones(2), 1.0, 0, 0.1),
        (np.zeros(2), np.ones(2), 1.0, 5, -1.0),
    ]:
        with pytest.raises(ValueError):
            wa.walnuts_step(rng, theta, *sn, im, ms, depth, err)
    with pytest.raises(TypeError):
        wa.walnuts_step(rng, np.zeros(2), 3.0, lambda q: -q, np.ones(2), 1.0, 5, 0.1)  # not callable
    with pytest.raises(ValueError):
        wa.walnuts_step(rng, np.zeros(2), tt.standard_normal_lpdf, tt.funnel_grad, np.ones(2), 1.0, 5, 0.1)
    with pytest.raises(ValueError):
        wa.walnuts_batch([rng], np.zeros((2, 3)), *sn, np.ones(3), 1.0, 5, 0.1, 0, 1)  # one Generator per chain
    import torch
    if not torch.cuda.is_available():  # plain (logp, grad) callables are accepted as in the reference: no GPU -> loud failure
        with pytest.raises(wa.WalnutsError):
            wa.walnuts_step(rng, np.zeros(2), lambda q: 0.0, lambda q: -q, np.ones(2), 1.0, 5, 0.1)
        with pytest.raises(ImportError):
            wa.walnuts_stan(rng, np.zeros(2), "model.stan", "data.json", np.ones(2), 1.0, 5, 0.1, 0, 1)  # no bridgestan here


def test_hostrng_matches_generator_methods():
    """walnuts_batch() consumes the callers' numpy Generators through numpy's own C distributions (csrc/wn_hostrng.c on libnpyrandom).
    The contract (walnuts/walnuts.py:325,330,194,346,350): exactly the draws -- and the stream positions -- of per-chain
    Generator.normal / binomial / choice / uniform / random calls in the reference's order, with the data-dependent skips (inactive
    chains, no uniform after a sub-U-turn, random() only after an accept)."""
    import ctypes as C
    import importlib
    W = importlib.import_module("walnuts_amd.walnuts")  # (the package exports the function `walnuts` under the same name)
    hr = W._hostrng_lib()
    assert hr is not None, "libwalnuts_hostrng.so is not built (make -C walnuts_amd/csrc)"
    B, D = 37, 10
    a = [np.random.default_rng(5 + c) for c in range(B)]
    b = [np.random.default_rng(5 + c) for c in range(B)]
    bg = W._bitgen_pointers(b)
    assert bg is not None
    assert W._bitgen_pointers([object()]) is None  # a stand-in without numpy's C interface: the per-chain Python calls are used
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    lm = np.geomspace(0.5, 2.0, D)
    rho_py = np.stack([lm * a[c].normal(size=D) for c in range(B)])
    rho_c = np.zeros((B, D))
    hr.wn_hr_momentum(bg, B, D, vp(lm), vp(rho_c))
    np.testing.assert_array_equal(rho_py, rho_c)
    rng = np.random.default_rng(0)
    for _ in range(120):
        n = 1 << int(rng.integers(0, 7))
        active = (rng.random(B) < 0.8).astype(np.int32)
        bw_py = np.zeros(B, np.int32); lo_py = np.zeros((B, n), np.int32)
        for c in range(B):
            if active[c]:
                bw_py[c] = int(bool(a[c].binomial(1, 0.5)))
                lo_py[c] = [[0, 1, 2].index(int(a[c].choice([0, 1, 2]))) for _ in range(n)]  # the reference's call shape (:194)
        bw_c = np.zeros(B, np.int32); lo_c = np.zeros((B, n), np.int32); scratch = np.zeros(n, np.uint64)
        hr.wn_hr_extend_draws(bg, vp(active), B, n, vp(bw_c), vp(lo_c), vp(scratch))
        np.testing.assert_array_equal(bw_py, bw_c)
        np.testing.assert_array_equal(lo_py, lo_c)
        sub = (rng.random(B) < 0.3).astype(np.int32)
        u_py = np.zeros(B); u_c = np.zeros(B)
        for c in range(B):
            if active[c] and not sub[c]:
                u_py[c] = a[c].uniform(0, 1)
        hr.wn_hr_accept_uniforms(bg, vp(active), vp(sub), B, vp(u_c))
        np.testing.assert_array_equal(u_py, u_c)
        acc = ((rng.random(B) < 0.5) & (active == 1) & (sub == 0)).astype(np.int32)
        r_py = np.zeros(B); r_c = np.zeros(B)
        for c in range(B):
            if acc[c]:
                r_py[c] = a[c].random()
        hr.wn_hr_choice_uniforms(bg, vp(acc), B, vp(r_c))
        np.testing.assert_array_equal(r_py, r_c)
    assert all(a[c].uniform() == b[c].uniform() for c in range(B))  # every stream at the same position
