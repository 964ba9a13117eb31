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


def _emu_lib():
    from tests.emu.harness import emu
    return emu()


def test_sgpr_objective_and_hand_derived_gradient_match_autograd_of_the_oracle():
    """Hyper-parameter training (gp.py:55-81): the product's -ExactMarginalLogLikelihood and its hand-derived gradient
    (predictor.vfe_loss_and_grad: dvs_gp_kernel / dvs_gp_kernel_backward on the emulator build + dense float64 algebra)
    against autograd through oracle/gp.vfe_loss_torch, on a slice of the reference's own predictor data set."""
    import torch
    from dags_vae_search_amd import predictor as P
    fix = load_npz("asia_predictor.npz")
    n, M, D = 160, 48, 32
    X = torch.from_numpy(fix["x"][:n].copy())
    y = torch.from_numpy(fix["y"][:n].copy())
    flat = torch.zeros(M * D + 4)
    flat[:M * D] = (X[:M] + 0.05 * torch.randn(M, D, generator=torch.Generator().manual_seed(1))).reshape(-1)
    flat[M * D:] = torch.tensor([1.3, -20.0, 2.1, 0.4])
    loss, grad = P.vfe_loss_and_grad(_emu_lib(), None, X, y, flat, M)
    raw = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in zip(P._RAW, flat[M * D:])}
    Z = flat[:M * D].view(M, D).double().clone().requires_grad_(True)
    ref = ogp.vfe_loss_torch(raw, Z, X.double(), y)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-9 * abs(float(ref))
    gref = torch.cat([Z.grad.reshape(-1)] + [raw[k].grad.reshape(1) for k in P._RAW])
    err = (grad.double() - gref).abs()
    assert float(err[:M * D].max()) < 1e-5 * float(gref[:M * D].abs().max())
    assert all(float(err[M * D + i]) < 1e-5 * abs(float(gref[M * D + i])) + 1e-12 for i in range(4))


def test_sgpr_training_trajectory_matches_the_oracles_adam_loop():
    """20 iterations of the product's loop body (objective + gradient above, then the fused Adam kernel dvs_clip_adam) against
    the oracle's restatement of the reference loop (torch.optim.Adam(lr 0.01) on float32 parameters), from gpytorch's default
    initialisation: same parameters to float32 rounding."""
    import torch
    from dags_vae_search_amd import predictor as P
    from dags_vae_search_amd import _lib as dl
    from tests.emu.harness import ptr
    fix = load_npz("asia_predictor.npz")
    n, M, D, iters = 128, 32, 32, 20
    X = torch.from_numpy(fix["x"][:n].copy())
    y = torch.from_numpy(fix["y"][:n].copy())
    lib = _emu_lib()
    flat = torch.zeros(M * D + 4)
    flat[:M * D] = X[:M].reshape(-1)
    m, v, scratch = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(4096)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    losses = []
    for it in range(1, iters + 1):
        loss, grad = P.vfe_loss_and_grad(lib, None, X, y, flat, M)
        losses.append(float(loss))
        dl.check(lib, lib.dvs_clip_adam(flat.numel(), p(flat), p(grad), p(m), p(v), 0.01, 0.9, 0.999, 1e-8, it, -1.0, p(scratch),
                                        None, None), "dvs_clip_adam")
    ref, hist = ogp.train_torch(fix["x"][:n], fix["y"][:n], M=M, iterations=iters, lr=0.01, log_every=1)
    assert losses[-1] < losses[0]
    assert all(abs(a - b[1]) < 1e-6 * abs(b[1]) for a, b in zip(losses, hist))
    assert float((flat[:M * D].view(M, D) - torch.from_numpy(ref["inducing_points"])).abs().max()) < 2e-5
    for i, k in enumerate(P._RAW):
        assert abs(float(flat[M * D + i]) - float(ref[k])) < 2e-5, k
