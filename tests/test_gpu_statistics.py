"""Distributional checks at scale (a size-independent property of the domain): thousands of chains must
reproduce the known moments of their targets.  These would catch an error common to the oracle and the engine."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_parity import dense_problem, wa  # noqa: F401

pytestmark = pytest.mark.gpu


def _run(wa, D, C, ig, tk, tpar, q0, H0, M, n_iter, generated=ol.G_IDENTITY, gen_k=0, **kw):
    e = wa.Engine(D, C, integrator=ig, H0=H0, M=M, generated=generated, gen_k=gen_k, **kw).set_target(tk, tpar)
    s, d = e.run(q0, 4242, n_iter)
    s = s.cpu().numpy(); d = d.cpu().numpy(); e.close()
    return s, d


def test_standard_normal_moments(wa):
    C, D = 4096, 10
    q0 = np.zeros((C, D))
    s, d = _run(wa, D, C, ol.I_FIXED, ol.T_STDGAUSS, None, q0, 0.9, 8, 60)
    x = s[:, 20:, :].reshape(-1, D)  # 4096 x 41 draws after burn-in
    assert np.abs(x.mean(0)).max() < 0.02
    assert np.abs(x.var(0) - 1.0).max() < 0.03
    assert np.abs(np.corrcoef(x.T) - np.eye(D)).max() < 0.03


def test_dense_gaussian_covariance(wa):
    """The headline target: draws must have covariance A^-1 (checked through the quadratic form q'Aq ~ chi2_D and the
    largest-variance direction)."""
    C, D = 2048, 100
    A, q0 = dense_problem(D, 1e4, C, 3)
    s, d = _run(wa, D, C, ol.I_FIXED, ol.T_DENSEGAUSS, A.ravel(), q0, 0.5, 10, 40)
    x = s[:, 15:, :].reshape(-1, D)
    qf = np.einsum("ni,ij,nj->n", x, A, x)
    assert abs(qf.mean() - D) < 0.6 and abs(qf.var() - 2 * D) < 12
    w, V = np.linalg.eigh(np.linalg.inv(A))
    proj = x @ V[:, -1]
    assert abs(proj.var() / w[-1] - 1.0) < 0.08  # slowest direction (variance 1e4) is the hard one


def test_funnel_marginal_of_q0(wa):
    """funnel10 with adaptLeapFrogR2P and warm-up: q0 ~ N(0, 3^2) (targetDistr.py:74-78)."""
    C = 2048
    rng = np.random.default_rng(1)
    q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    s, d = _run(wa, 11, C, ol.I_ADAPT_R2P, ol.T_FUNNEL10, None, q0, 0.3, 10, 120, generated=ol.G_FIRSTK, gen_k=2,
                delta0=0.3, warmupIter=40)
    x = s[:, 60:, 0].ravel()
    assert abs(x.mean()) < 0.25
    assert abs(x.std() - 3.0) < 0.25
    # the conditional scale of q1: q1 * exp(-q0/2) ~ N(0, 1)
    z = (s[:, 60:, 1] * np.exp(-0.5 * s[:, 60:, 0])).ravel()
    assert abs(z.std() - 1.0) < 0.05
