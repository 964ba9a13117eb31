"""GP predictor: oracle restatement on the reference's own predictor data; HIP kernel (emulator) vs it."""
import ctypes

import numpy as np
import pytest

from oracle import gp as ogp
from tests.helpers import load_npz


def _split(fix):
    x, y = fix["x"].astype(np.float64), fix["y"]
    ntr = int(np.floor(0.8 * len(x)))            # experiments/01_bn_asia/main.py:322-327
    return x[:ntr], y[:ntr], x[ntr:], y[ntr:]


def test_oracle_reproduces_the_shipped_predictors_behaviour():
    """PARITY UNPINNED (gpytorch absent).  Sanity pinned by data: with the shipped hyper-parameters the SGPR predictive mean
    on the reference's own 80/20 split has the constant predictor's test MAE (686.1 vs 686.3; MAPE 0.0515) — the trained
    GP of experiments/01_bn_asia carries no information beyond the mean, and a restatement must reproduce exactly that."""
    fix = load_npz("asia_predictor.npz")
    h = ogp.hyper(fix)
    assert h["lengthscale"] == pytest.approx(8.7059, abs=1e-3) and h["outputscale"] == pytest.approx(19.156, abs=1e-2)
    xtr, ytr, xte, yte = _split(fix)
    alpha = ogp.fit_alpha(xtr, ytr, fix["inducing_points"], h)
    pred = ogp.predict_mean(xte, fix["inducing_points"], alpha, h)
    mae = float(np.abs(pred - yte).mean())
    const_mae = float(np.abs(ytr.mean() - yte).mean())
    assert mae == pytest.approx(686.1, abs=1.0) and abs(mae - const_mae) < 2.0
    assert float(np.abs((pred - yte) / yte).mean()) == pytest.approx(0.0515, abs=1e-3)


def test_emu_gp_predict_kernel_matches_oracle():
    from tests.emu.harness import emu, ptr
    fix = load_npz("asia_predictor.npz")
    h = ogp.hyper(fix)
    xtr, ytr, xte, yte = _split(fix)
    Z = np.ascontiguousarray(fix["inducing_points"], np.float32)
    alpha = np.ascontiguousarray(ogp.fit_alpha(xtr, ytr, Z, h))
    x = np.ascontiguousarray(fix["x"][-37:], np.float32)
    out = np.zeros(len(x), np.float64)
    lib = emu()
    assert lib.dvs_gp_predict(len(x), Z.shape[0], 32, ptr(x), ptr(Z), ptr(alpha), h["outputscale"], h["lengthscale"],
                              h["constant"], ptr(out), None) == 0
    ref = ogp.predict_mean(x, Z, alpha, h)
    assert np.abs(out - ref).max() < 1e-6 * np.abs(ref).max()
    assert lib.dvs_gp_predict(len(x), Z.shape[0], 32, ptr(x), ptr(Z), ptr(alpha), h["outputscale"], 0.0,
                              h["constant"], ptr(out), None) != 0
