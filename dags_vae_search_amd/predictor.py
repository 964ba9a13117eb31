"""GP predictor of the reference at prediction time, on the GPU.

Mirror of ``GPRegressionModel`` (src/predictors/gp.py:13-32; trained in experiments/01_bn_asia/main.py:315-393): an
SGPR model — ConstantMean, RBF kernel with an output scale, 500 learned inducing points, Gaussian noise.  This class
loads the reference's ``predictor.pth`` state dict (same keys), solves the M x M system for the mean weights once
(float64, ``torch.linalg`` on the device: a one-off library solve, not a hot path) and evaluates ``predict(x)`` — the
``model(x).mean`` of the reference — in a HIP kernel (csrc/k_bic.hip: k_gp_predict), so that the latent-space search loop
encode -> predict -> decode stays on the device.  Training the hyper-parameters (10 000 Adam steps on the exact marginal
likelihood) is not built; parity with gpytorch is unpinned (not installed; see oracle/gp.py).
"""
from __future__ import annotations

import ctypes
from typing import Dict

import torch

from . import _lib as dl


def _softplus(v: torch.Tensor) -> float:
    return float(torch.nn.functional.softplus(v.double().reshape(-1)[0]))


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
