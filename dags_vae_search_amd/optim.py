"""Fused clip_grad_norm_ + Adam for a flat-buffer PaceVaeV3 (one HIP kernel pair over all 108 tensors).

``Adam(model.parameters(), lr=1e-4)`` has the constructor of ``torch.optim.Adam`` (the optimiser the reference uses,
experiments/03_synthetic_12/main.py:165) and is a ``torch.optim.Optimizer``, so ``ReduceLROnPlateau`` and
``state_dict`` keep working.  ``step(max_grad_norm=...)`` folds ``clip_grad_norm_(params, max_grad_norm)``
(main.py:115) into the same launch sequence: global L2 norm -> clip coefficient -> Adam update, no host sync.
"""
from __future__ import annotations

import torch


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("fused Adam implements the reference's configuration (no weight decay, no amsgrad)")
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = None
        self._exp_avg = None
        self._exp_avg_sq = None
        self._scratch = None
        self._steps = 0

    def attach(self, model):
        """Bind to the PaceVaeV3 whose flat buffers hold exactly this optimiser's parameters."""
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        theirs = {id(p) for p in model.parameters()}
        if mine != theirs or len(self.param_groups) != 1:
            raise ValueError("fused Adam must own exactly the parameters of one PaceVaeV3 in a single group")
        self._model = model
        return self

    @property
    def grad_norm(self) -> torch.Tensor:
        """L2 norm of the (unclipped) gradient of the last step, as a device scalar."""
        return self._scratch[0].sqrt()

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: float = -1.0):
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        model = self._model
        if model is None:
            raise RuntimeError("call optimizer.attach(model) first")
        flat, grads = model.flat_params, model.bind_flat_grads()
        if self._exp_avg is None or self._exp_avg.device != flat.device:
            self._exp_avg = torch.zeros_like(flat)
            self._exp_avg_sq = torch.zeros_like(flat)
            self._scratch = torch.zeros(320, dtype=torch.float32, device=flat.device)   # DVS_CLIP_SCRATCH_FLOATS
        g = self.param_groups[0]
        self._steps += 1
        model._eng().clip_adam(flat, grads, self._exp_avg, self._exp_avg_sq, float(g["lr"]), float(g["betas"][0]),
                               float(g["betas"][1]), float(g["eps"]), self._steps, float(max_grad_norm), self._scratch)

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in model.flat_grads and are overwritten by every backward: nothing to clear
        return None
