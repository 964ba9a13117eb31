"""Fused clip_grad_norm_ + Adam for a flat-buffer PaceVaeV3 (one HIP kernel pair over all 108 tensors).

``Adam(model.parameters(), lr=1e-4)`` has the constructor of ``torch.optim.Adam`` (the optimiser the reference uses,
experiments/03_synthetic_12/main.py:165) and is a ``torch.optim.Optimizer``, so ``ReduceLROnPlateau`` and
``state_dict`` / ``load_state_dict`` keep working (the flat moments are ordinary optimiser state).  ``step(max_grad_norm=...)`` folds ``clip_grad_norm_(params, max_grad_norm)``
(main.py:115) into the same launch sequence: global L2 norm -> clip coefficient -> Adam update, no host sync.
"""
from __future__ import annotations

import torch


class Adam(torch.optim.Optimizer):
    """State lives where ``torch.optim.Optimizer`` expects it — ``self.state`` — keyed on the model's FIRST parameter:
    ``{"step": int, "exp_avg": flat [P], "exp_avg_sq": flat [P]}`` (the flat moments cover all 108 tensors).  So the stock
    ``state_dict()`` / ``load_state_dict()`` save and restore the moments and the bias-correction step, and a resumed
    run continues exactly where it stopped (tests/test_gpu_module.py: save -> load -> step round trip vs torch.optim.Adam)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("fused Adam implements the reference's configuration (no weight decay, no amsgrad)")
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = None
        self._scratch = None

    def attach(self, model):
        """Bind to the PaceVaeV3 whose flat buffers hold exactly this optimiser's parameters."""
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        theirs = {id(p) for p in model.parameters()}
        if mine != theirs or len(self.param_groups) != 1:
            raise ValueError("fused Adam must own exactly the parameters of one PaceVaeV3 in a single group")
        self._model = model
        return self

    def _flat_state(self, flat: torch.Tensor) -> dict:
        st = self.state[self.param_groups[0]["params"][0]]
        if "exp_avg" not in st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(flat)
            st["exp_avg_sq"] = torch.zeros_like(flat)
        for k in ("exp_avg", "exp_avg_sq"):
            t = st[k]
            if t.numel() != flat.numel():
                raise ValueError(f"fused Adam: {k} has {t.numel()} elements, the model has {flat.numel()} parameters")
            if t.device != flat.device or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
                # model.to(other device) after some steps, or a state restored by load_state_dict (which casts state
                # tensors to the key parameter's device/dtype but keeps their shape): the moments follow the parameters
                st[k] = t.to(device=flat.device, dtype=torch.float32).reshape(-1).contiguous()
        st["step"] = int(st["step"])
        return st

    @property
    def grad_norm(self) -> torch.Tensor:
        """L2 norm of the (unclipped) gradient of the last step, as a device scalar."""
        return self._scratch[0].sqrt()

    @property
    def _steps(self) -> int:
        return int(self.state[self.param_groups[0]["params"][0]].get("step", 0))

    def step_skipped(self):
        """The last step's guard fired on the device (non-finite loss / invalid batch): nothing was updated, so the
        bias-correction step count goes back too."""
        st = self.state[self.param_groups[0]["params"][0]]
        if st.get("step", 0) > 0:
            st["step"] -= 1

    def clip_scratch(self, device) -> torch.Tensor:
        """The device scratch of the clip (DVS_CLIP_SCRATCH_FLOATS): [0] sum of squares, [1] clip coefficient, then partial sums.
        ``train_batch`` hands it to the backward (``loss_and_grad(clip_scratch=...)``) so that the kernel that sums the gradient
        slabs leaves the partials behind and ``step(from_partials=True)`` needs no pass over the gradient for its norm."""
        if self._scratch is None or self._scratch.device != device:
            from . import _lib as dl
            self._scratch = torch.zeros(dl.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=device)
        return self._scratch

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: float = -1.0, guard: torch.Tensor = None, from_partials: bool = False):
        """guard: optional device f32[2] [non-finite flag, invalid-features flag] of the step's forward; when either is
        set the kernels leave parameters and moments untouched (``train_batch`` then calls ``step_skipped``).
        from_partials: the backward of THIS step wrote the gradient's partial sums of squares into ``clip_scratch`` (and
        nothing has changed the gradient since — no all-reduce)."""
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        model = self._model
        if model is None:
            raise RuntimeError("call optimizer.attach(model) first")
        flat, grads = model.flat_params, model.bind_flat_grads()
        st = self._flat_state(flat)
        scratch = self.clip_scratch(flat.device)
        g = self.param_groups[0]
        st["step"] += 1
        model._eng().clip_adam(flat, grads, st["exp_avg"], st["exp_avg_sq"], float(g["lr"]), float(g["betas"][0]),
                               float(g["betas"][1]), float(g["eps"]), st["step"], float(max_grad_norm), scratch,
                               guard, from_partials=from_partials)

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in model.flat_grads and are overwritten by every backward: nothing to clear
        return None
