"""`ska cov` mixture fit (coverage.rs:151-226) pinned by answers that do not come from the restated optimiser: for histograms drawn
from known two-component mixtures, the fitted (w0, c) must maximise the reference's log-likelihood (coverage.rs:310-326), which
this test evaluates on its own in numpy over a dense grid, and the cutoff must be the one find_cutoff (coverage.rs:349-363)
gives at the grid's optimum.  The BFGS iterates are free to differ from argmin's; the argmax is not.  Runs without a GPU (the fit
is host code on both sides: the oracle's ora_cov_fit and the engine's skh_cov_fit)."""
import math

import numpy as np
import pytest

import ora


def _ln_dpois(x, lam):
    return x * math.log(lam) - math.lgamma(x + 1.0) - lam


def _loglik(w0, c, counts):
    x = np.arange(1, len(counts) + 1, dtype=np.float64)
    lg = np.array([math.lgamma(v + 1.0) for v in x])
    a = math.log(w0) + (x * 0.0 - lg - 1.0)                       # ln(w0) + ln dpois(x, 1)
    b = math.log(1.0 - w0) + (x * math.log(c) - lg - c)
    m = np.maximum(a, b)
    return float((counts * (m + np.log(np.exp(a - m) + np.exp(b - m)))).sum())


def _cutoff(w0, c, n):
    cut = 1
    while cut < n:
        a = math.log(w0) + _ln_dpois(cut, 1.0)
        b = math.log(1.0 - w0) + _ln_dpois(cut, c)
        if a - b < 0.0:
            break
        cut += 1
    return cut


def _mixture_hist(w0, c, total, n_bins, seed):
    rng = np.random.default_rng(seed)
    n_err = rng.binomial(total, w0)
    v = np.concatenate([rng.poisson(1.0, n_err), rng.poisson(c, total - n_err)])
    h = np.bincount(v, minlength=n_bins + 1)[1:n_bins + 1].astype(np.float64)
    while len(h) and h[-1] < 50:                                   # MIN_FREQ truncation, coverage.rs:166-173
        h = h[:-1]
    return h


@pytest.mark.parametrize("w0,c,seed", [(0.7, 25.0, 1), (0.85, 40.0, 2), (0.5, 12.0, 3), (0.9, 60.0, 4)])
def test_fit_is_the_argmax_of_the_reference_likelihood(w0, c, seed):
    import skx_engine as E
    h = _mixture_hist(w0, c, 3_000_000, 400, seed)
    # independent optimum: dense grid around the generating parameters, refined twice
    lo_w, hi_w, lo_c, hi_c = max(0.01, w0 - 0.2), min(0.99, w0 + 0.2), max(1.5, c * 0.6), c * 1.4
    for _ in range(3):
        ws, cs = np.linspace(lo_w, hi_w, 41), np.linspace(lo_c, hi_c, 41)
        ll = np.array([[_loglik(a, b, h) for b in cs] for a in ws])
        i, j = np.unravel_index(np.argmax(ll), ll.shape)
        dw, dc = (hi_w - lo_w) / 10, (hi_c - lo_c) / 10
        lo_w, hi_w, lo_c, hi_c = max(0.001, ws[i] - dw), min(0.999, ws[i] + dw), max(1.01, cs[j] - dc), cs[j] + dc
    gw, gc, gll = ws[i], cs[j], ll[i, j]
    for name, (fw, fc, fcut) in (("oracle", ora.cov_fit(h)), ("engine", E.cov_fit(h))):
        assert abs(fw - gw) < 2e-3 and abs(fc - gc) / gc < 2e-3, (name, fw, fc, gw, gc)
        assert _loglik(fw, fc, h) >= gll - 1e-6 * abs(gll), name           # at least as good as the best grid point
        assert fcut == _cutoff(gw, gc, len(h)) == _cutoff(fw, fc, len(h)), name
    # and the two fits agree with each other to the optimiser's tolerance
    (ow, oc, _), (ew, ec, _) = ora.cov_fit(h), E.cov_fit(h)
    assert abs(ow - ew) < 1e-9 and abs(oc - ec) < 1e-7
