"""GP predictor of the reference at prediction time, on the GPU.

Mirror of ``GPRegressionModel`` (src/predictors/gp.py:13-32; trained in experiments/01_bn_asia/main.py:315-393): an
SGPR model — ConstantMean, RBF kernel with an output scale, 500 learned inducing points, Gaussian noise.  This class
loads the reference's ``predictor.pth`` state dict (same keys), solves the M x M system for the mean weights once
(float64, ``torch.linalg`` on the device: a one-off library solve, not a hot path) and evaluates ``predict(x)`` — the
``model(x).mean`` of the reference — in a HIP kernel (csrc/k_bic.hip: k_gp_predict), so that the latent-space search loop
encode -> predict -> decode stays on the device.

``train_hyperparameters`` is the reference's training loop (src/predictors/gp.py:55-81 = experiments/01_bn_asia/main.py:
329-365): Adam(lr 0.01) on ``-ExactMarginalLogLikelihood(likelihood, model)``, which for this model is the collapsed SGPR
bound of Titsias (2009) divided by the number of training points.  One iteration on the device: RBF kernel matrices
K_uu, K_uf in float64 (``dvs_gp_kernel``), two Cholesky factorisations + triangular solves + M x n products (dense float64
library calls through torch.linalg — plumbing, like ``fit``), the HAND-DERIVED gradient with respect to K_uu / K_uf pulled
back to the inducing points, lengthscale and outputscale by ``dvs_gp_kernel_backward``, and the fused Adam kernel of the
train step (``dvs_clip_adam``) over the flat float32 parameter vector [inducing points | raw_noise, raw_constant,
raw_outputscale, raw_lengthscale].  No autograd graph.  Parity with gpytorch is unpinned (not installed; oracle/gp.py restates
the objective and checks this gradient by autograd; tests pin the training dynamics against the reference's SHIPPED
hyper-parameters, which this loop reproduces from gpytorch's default initialisation).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict

import torch

from . import _lib as dl


def _softplus(v: torch.Tensor) -> float:
    return float(torch.nn.functional.softplus(v.double().reshape(-1)[0]))


def _sp(v: float) -> float:
    return v if v > 30.0 else math.log1p(math.exp(v))


def _sig(v: float) -> float:
    return 1.0 / (1.0 + math.exp(-v))


_RAW = ("raw_noise", "raw_constant", "raw_outputscale", "raw_lengthscale")       # tail of the flat parameter vector


def vfe_loss_and_grad(lib, stream, X: torch.Tensor, y: torch.Tensor, flat: torch.Tensor, M: int, raw=None,
                      jitter: float = 1e-6):
    """-ExactMarginalLogLikelihood of the reference's SGPR model and its gradient with respect to the flat float32
    parameter vector ``flat`` = [Z (M x D) | raw_noise, raw_constant, raw_outputscale, raw_lengthscale].

    F = n/2 log 2pi + 1/2 logdet(Q + s2 I) + 1/2 r^T (Q + s2 I)^-1 r + (n o - tr Q) / (2 s2),  Q = K_fu K_uu^-1 K_uf,
    loss = F / n (gpytorch divides by the number of data points).  With C = s2 K_uu + K_uf K_fu, V = C^-1 K_uf,
    W = K_uu^-1 K_uf, beta = V r, alpha = (r - K_fu beta) / s2:
        dF/dK_uf = V - beta alpha^T - W / s2            dF/dK_uu = -1/2 (V W^T - beta beta^T - W W^T / s2)
        dF/ds2   = 1/2 tr S - 1/2 |alpha|^2 - (n o - tr Q) / (2 s2^2),  tr S = (n - M + s2 tr(C^-1 K_uu)) / s2
        dF/do    = n / (2 s2) + (kernel terms)          dF/dc = -sum alpha
    (checked against autograd of oracle/gp.vfe_loss_torch in tests/test_predictor.py).  X [n, D] float32, y [n] float64,
    on the device the library `lib` computes on.  Returns (loss as a 0-d float64 tensor, gradient float32 like flat)."""
    n, D = X.shape
    dev = X.device
    Z = flat[:M * D].view(M, D)
    rn, rc, ro, rl = raw if raw is not None else flat[M * D:].tolist()
    s2, o, l, c = _sp(rn) + 1e-4, _sp(ro), _sp(rl), rc
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    f64 = dict(dtype=torch.float64, device=dev)
    Kuu = torch.empty(M, M, **f64)
    Kuf = torch.empty(M, n, **f64)
    dl.check(lib, lib.dvs_gp_kernel(M, M, D, p(Z), p(Z), o, l, p(Kuu), stream), "dvs_gp_kernel")
    dl.check(lib, lib.dvs_gp_kernel(M, n, D, p(Z), p(X), o, l, p(Kuf), stream), "dvs_gp_kernel")
    Kuu.diagonal().add_(jitter)
    L = torch.linalg.cholesky(Kuu)
    C = torch.addmm(Kuu, Kuf, Kuf.T, beta=s2)
    LC = torch.linalg.cholesky(C)
    r = y - c
    V = torch.cholesky_solve(Kuf, LC)
    W = torch.cholesky_solve(Kuf, L)
    beta = V @ r
    alpha = (r - Kuf.T @ beta) / s2
    trQ = (Kuf * W).sum()
    logdet = 2.0 * (torch.log(LC.diagonal()).sum() - torch.log(L.diagonal()).sum()) + (n - M) * math.log(s2)
    F = 0.5 * n * math.log(2.0 * math.pi) + 0.5 * logdet + 0.5 * (r @ alpha) + 0.5 * (n * o - trQ) / s2
    Guf = (V - torch.outer(beta, alpha) - W / s2).contiguous()
    Guu = (-0.5 * (V @ W.T - torch.outer(beta, beta) - (W @ W.T) / s2)).contiguous()
    trS = (n - M + s2 * torch.cholesky_solve(Kuu, LC).diagonal().sum()) / s2
    ds2 = 0.5 * trS - 0.5 * (alpha @ alpha) - 0.5 * (n * o - trQ) / (s2 * s2)
    dZ1 = torch.empty(M, D, **f64)
    dZ2 = torch.empty(M, D, **f64)
    rows1 = torch.empty(M, 2, **f64)
    rows2 = torch.empty(M, 2, **f64)
    dl.check(lib, lib.dvs_gp_kernel_backward(M, n, D, 0, p(Z), p(X), o, l, p(Guf), p(dZ1), p(rows1), stream),
             "dvs_gp_kernel_backward")
    dl.check(lib, lib.dvs_gp_kernel_backward(M, M, D, 1, p(Z), p(Z), o, l, p(Guu), p(dZ2), p(rows2), stream),
             "dvs_gp_kernel_backward")
    sc = rows1.sum(0) + rows2.sum(0)                       # [d/dl, d/do] through the kernels
    # the jitter on K_uu's diagonal is a constant, not o * 1: take its share back out of d/do
    do_ = 0.5 * n / s2 + sc[1] - jitter * Guu.diagonal().sum() / o
    grad = torch.empty_like(flat)
    grad[:M * D] = ((dZ1 + dZ2) / n).reshape(-1).to(torch.float32)
    tail = torch.stack([ds2 * _sig(rn), -alpha.sum(), do_ * _sig(ro), sc[0] * _sig(rl)]) / n
    grad[M * D:] = tail.to(torch.float32)
    return F / n, grad


class GPRegressionModel:
    def __init__(self, train_x: torch.Tensor, train_y: torch.Tensor, likelihood=None, device="cuda"):
        self.device = torch.device(device)
        self.train_x = train_x.to(self.device, torch.float64)
        self.train_y = train_y.to(self.device, torch.float64)
        self.inducing_points = train_x[:500].to(self.device, torch.float32).contiguous()      # gp.py:24
        # gpytorch defaults: softplus(0) for the positive parameters, zero mean constant
        self.noise, self.outputscale, self.lengthscale, self.constant = 0.6932 + 1e-4, 0.6931, 0.6931, 0.0
        self._alpha = None
        self.lib = dl.load()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Reference keys (predictor_results/predictor.pth): likelihood.noise_covar.raw_noise, mean_module.raw_constant,
        base_covar_module.raw_outputscale, base_covar_module.base_kernel.raw_lengthscale, covar_module.inducing_points."""
        self.noise = _softplus(sd["likelihood.noise_covar.raw_noise"]) + 1e-4         # GreaterThan(1e-4) constraint
        self.outputscale = _softplus(sd["base_covar_module.raw_outputscale"])
        self.lengthscale = _softplus(sd["base_covar_module.base_kernel.raw_lengthscale"])
        self.constant = float(sd["mean_module.raw_constant"].double().reshape(-1)[0])
        self.inducing_points = sd["covar_module.inducing_points"].to(self.device, torch.float32).contiguous()
        self._alpha = None
        return self

    def _kernel64(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        d2 = torch.cdist(a, b).pow(2)
        return self.outputscale * torch.exp(-0.5 * d2 / self.lengthscale ** 2)

    def fit(self, jitter: float = 1e-6):
        """Mean weights alpha = (K_uu + K_uf K_fu / s^2)^-1 K_uf (y - c) / s^2 (SGPR / DTC predictive mean)."""
        Z = self.inducing_points.double()
        Kuu = self._kernel64(Z, Z) + jitter * torch.eye(Z.shape[0], dtype=torch.float64, device=self.device)
        Kuf = self._kernel64(Z, self.train_x)
        A = Kuu + Kuf @ Kuf.T / self.noise
        self._alpha = (torch.linalg.solve(A, Kuf @ (self.train_y - self.constant)) / self.noise).contiguous()
        return self

    def eval(self):
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference's predictor.pth keys / shapes (float32, raw = pre-softplus values)."""
        inv = lambda v: math.log(math.expm1(v)) if v < 30.0 else v
        t = lambda v, shape: torch.full(shape, v, dtype=torch.float32)
        return {"likelihood.noise_covar.raw_noise": t(inv(self.noise - 1e-4), (1,)),
                "mean_module.raw_constant": t(self.constant, ()),
                "base_covar_module.raw_outputscale": t(inv(self.outputscale), ()),
                "base_covar_module.base_kernel.raw_lengthscale": t(inv(self.lengthscale), (1, 1)),
                "covar_module.inducing_points": self.inducing_points.detach().cpu().clone()}

    def train_hyperparameters(self, iterations: int = 10000, lr: float = 0.01, log_every: int = 0, log=print,
                              from_defaults: bool = True, checkpoints=(), on_checkpoint=None):
        """The reference's training loop (gp.py:55-81 / main.py:329-365): `iterations` full-batch Adam(lr) steps on
        -ExactMarginalLogLikelihood, all on the device (module docstring).  ``from_defaults`` starts from gpytorch's
        initial values (raw parameters 0, constant 0, inducing points = train_x[:M]) like a freshly constructed reference
        model; otherwise from the current hyper-parameters.  Returns [(iteration, loss)] at the logging points.
        ``checkpoints`` / ``on_checkpoint``: at these iteration counts the model's hyper-parameters are brought up to date and
        ``on_checkpoint(iteration, self)`` is called INSIDE the one run (optimiser moments untouched) — how ``train_predictor``
        gets the reference's MAE / MAPE-versus-iterations comments (gp.py:95-106) from a single 10 000-iteration run."""
        if self.device.type != "cuda":
            raise RuntimeError("dags_vae_search_amd.predictor trains only on the GPU (there is no CPU path)")
        X = self.train_x.to(torch.float32).contiguous()
        y = self.train_y.contiguous()
        M, D = self.inducing_points.shape
        flat = torch.zeros(M * D + 4, dtype=torch.float32, device=self.device)
        if from_defaults:
            flat[:M * D] = X[:M].reshape(-1)
        else:
            sd = self.state_dict()
            flat[:M * D] = self.inducing_points.reshape(-1)
            flat[M * D:] = torch.stack([sd["likelihood.noise_covar.raw_noise"].reshape(()), sd["mean_module.raw_constant"],
                                        sd["base_covar_module.raw_outputscale"],
                                        sd["base_covar_module.base_kernel.raw_lengthscale"].reshape(())]).to(self.device)
        m = torch.zeros_like(flat)
        v = torch.zeros_like(flat)
        scratch = torch.zeros(dl.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        hist = []
        for it in range(1, iterations + 1):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            loss, grad = vfe_loss_and_grad(self.lib, stream, X, y, flat, M)
            dl.check(self.lib, self.lib.dvs_clip_adam(flat.numel(), p(flat), p(grad), p(m), p(v), lr, 0.9, 0.999, 1e-8, it, -1.0,
                                                      p(scratch), None, stream), "dvs_clip_adam")
            if log_every and it % log_every == 0:
                hist.append((it, float(loss)))
                log("Iter %d/%d - Loss: %.3f" % (it, iterations, hist[-1][1]))
            if on_checkpoint is not None and it in checkpoints:
                self._adopt(flat, M, D)
                on_checkpoint(it, self)
        self._adopt(flat, M, D)
        return hist

    def _adopt(self, flat: torch.Tensor, M: int, D: int):
        """hyper-parameters <- the flat training vector [inducing points | raw noise, constant, raw outputscale, raw lengthscale]"""
        rn, rc, ro, rl = flat[M * D:].tolist()
        self.noise, self.constant, self.outputscale, self.lengthscale = _sp(rn) + 1e-4, rc, _sp(ro), _sp(rl)
        self.inducing_points = flat[:M * D].view(M, D).clone()
        self._alpha = None

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        """``model(x).mean`` of the reference for latent vectors x [B, dim] -> float64 [B], computed by k_gp_predict."""
        if self._alpha is None:
            self.fit()
        if not x.is_cuda:
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        B, D = x.shape
        if D != self.inducing_points.shape[1]:
            raise AssertionError(f"Expected latent vectors of size {self.inducing_points.shape[1]}, got {D}")
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        dl.check(self.lib, self.lib.dvs_gp_predict(B, self.inducing_points.shape[0], D, p(x), p(self.inducing_points),
                                                   p(self._alpha), self.outputscale, self.lengthscale, self.constant,
                                                   p(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                 "dvs_gp_predict")
        return out


def train_predictor(X: torch.Tensor, y: torch.Tensor, iterations: int = 10000, lr: float = 0.01,
                    checkpoints=(100, 1000, 2000, 5000, 10000), log=print, device="cuda"):
    """The reference's predictor driver (experiments/01_bn_asia/main.py:315-393; src/predictors/gp.py:35-106): first 80 % of
    the (vector, target) rows train, the rest test; SGPR with 500 inducing points = the first 500 training rows; ``iterations``
    full-batch Adam(lr) steps on the negative exact marginal log-likelihood; then ``Test MAE`` / ``Test MAPE`` of the
    predictive mean.  The reference's source carries those two figures after 100 / 1 000 / 2 000 / 5 000 iterations as
    comments of separate runs (gp.py:95-106); here they come from ONE run through ``checkpoints``.
    Returns (model, [{"iterations", "mae", "mape", "noise", "outputscale", "lengthscale", "constant"}]).
    Parity against gpytorch itself is UNPINNED (gpytorch is absent; the reference ships neither predictions nor a loss curve)."""
    import math as _m
    n_train = int(_m.floor(0.8 * len(X)))                              # main.py:323
    train_x, train_y = X[:n_train].contiguous(), y[:n_train].contiguous()
    test_x, test_y = X[n_train:].contiguous(), y[n_train:].to(torch.float64)
    model = GPRegressionModel(train_x, train_y, device=device)
    rows = []

    def report(it, m):
        pred = m.predict(test_x).cpu()
        err = (pred - test_y).abs()
        rows.append({"iterations": it, "mae": float(err.mean()), "mape": float((err / test_y).mean()),     # main.py:373-374 (signed
                     "noise": m.noise, "outputscale": m.outputscale, "lengthscale": m.lengthscale,         # denominator, as there)
                     "constant": m.constant})
        if log is not None:
            log("%d\nTest MAE: %s\nTest MAPE: %s" % (it, rows[-1]["mae"], rows[-1]["mape"]))

    cps = sorted({int(c) for c in checkpoints if 0 < int(c) <= iterations} | {iterations})
    model.train_hyperparameters(iterations, lr=lr, checkpoints=cps, on_checkpoint=report)
    return model, rows
