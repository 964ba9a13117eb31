"""Thin host-side driver of the C ABI (include/dvs.h) over torch device memory and streams.

PyTorch is plumbing here: it owns the device buffers and the stream; all arithmetic happens inside
libdvs_hip.so.  Every method requires CUDA (ROCm) tensors and raises otherwise.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib as dl


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"dags_vae_search_amd: {what} must live on the GPU (got device {t.device}); "
                           f"this package has no CPU path")


class PaceEngine:
    """Workspace + record buffers for one (n_tokens, n_classes) model; sized lazily per batch size."""

    def __init__(self, n_tokens: int, n_classes: int):
        self.lib = dl.load()
        self.n_tokens = int(n_tokens)
        self.n_classes = int(n_classes)
        shape = dl.make_shape(1, n_tokens, n_classes)
        if self.lib.dvs_param_count(ctypes.byref(shape)) < 0:
            dl.check(self.lib, 1, "dvs_param_count")
        self.table, self.param_floats = dl.param_table(self.lib, shape)
        self.record_bytes = dl.record_bytes(self.lib, shape)
        self.wide = dl.is_wide(self.n_tokens, self.n_classes)
        self.tiles = (self.n_tokens + dl.TILE_TOKENS - 1) // dl.TILE_TOKENS
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0
        self._records: Optional[torch.Tensor] = None
        self._status: Optional[torch.Tensor] = None
        self._status_external = False

    # ---- buffers -------------------------------------------------------------------------------------
    def shape(self, batch, training=False, dropout=0.15, beta=0.005, eps_scale=0.01, dag_offset=0, seed=0):
        return dl.make_shape(batch, self.n_tokens, self.n_classes, training, dropout, beta, eps_scale, dag_offset, seed)

    def workspace(self, batch: int, device) -> torch.Tensor:
        if self._ws is None or self._ws_batch != batch or self._ws.device != device:
            shape = self.shape(batch)
            nbytes = self.lib.dvs_workspace_bytes(ctypes.byref(shape))
            if nbytes == 0:
                dl.check(self.lib, 1, "dvs_workspace_bytes")
            self._ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)   # slab padding must stay 0
            self._ws_batch = batch
            self._records = torch.empty(batch * self.record_bytes, dtype=torch.uint8, device=device)
            if not self._status_external or self._status.device != device:
                self._status = torch.zeros(1, dtype=torch.int32, device=device)
                self._status_external = False
        return self._ws

    def use_status(self, status: torch.Tensor):
        """Let the caller own the int32[1] feature-validation word (PaceVaeV3 keeps it next to the step's loss scalars so
        that one small device->host copy per train step reads everything).  It must be zero before a pack/build call
        with ``zero_status=False``."""
        self._status = status
        self._status_external = True

    def _read_status(self) -> int:
        st = int(self._status.item())
        if st:
            self._status.zero_()          # never leave stale bits behind an exception
        return st

    # ---- entry points ----------------------------------------------------------------------------------
    def pack(self, features: Dict, check: bool = True, zero_status: bool = True) -> torch.Tensor:
        """dvs_pack_features: reference-layout dense feature tensors (already on the GPU) -> records."""
        lab = features["vertex_label_features"]
        pos = features["vertex_position_features"]
        adj = features["adjacency_matrices"]
        tm = features["target_masks"]
        for name, t in (("vertex_label_features", lab), ("vertex_position_features", pos),
                        ("adjacency_matrices", adj), ("target_masks", tm)):
            _require_cuda(t, name)
        B, N, C = lab.shape
        if N != self.n_tokens or C != self.n_classes:
            raise AssertionError(f"Expected [B,{self.n_tokens},{self.n_classes}] label features, got {tuple(lab.shape)}")
        if tuple(pos.shape) != (B, N, N) or tuple(adj.shape) != (B, N, N) or tuple(tm.shape) != (8 * B, N, N):
            raise AssertionError("feature tensors have inconsistent shapes")
        def aligned(t, dtype):              # dvs_pack_features streams with 16-byte loads
            if t.dtype != dtype:
                t = t.to(dtype)
            if not t.is_contiguous():
                t = t.contiguous()
            return t if t.data_ptr() % 16 == 0 else t.clone()
        lab = aligned(lab, torch.float32)
        pos = aligned(pos, torch.float32)
        adj = aligned(adj, torch.float32)
        tm = aligned(tm, tm.dtype)
        tm = tm.view(torch.uint8) if tm.dtype == torch.bool else aligned(tm, torch.uint8)
        self.workspace(B, lab.device)
        if zero_status:
            self._status.zero_()
        shape = self.shape(B)
        dl.check(self.lib, self.lib.dvs_pack_features(ctypes.byref(shape), _ptr(lab), _ptr(pos), _ptr(adj), _ptr(tm),
                                                      _ptr(self._records), _nbytes(self._records), _ptr(self._status),
                                                      _stream()),
                 "dvs_pack_features")
        if check:
            st = self._read_status()
            if st:
                raise ValueError(f"features violate the prepare_features invariants (status bits {st:#x}: "
                                 f"1 = label/position row not one-hot, 2 = per-head masks differ, 4 = self masked)")
        return self._records

    def build_records(self, labels: torch.Tensor, preds: torch.Tensor, check: bool = True,
                      zero_status: bool = True) -> torch.Tensor:
        """dvs_build_records: row codec (labels u8 [B,n], preds [B,n]: i16 bit pattern of u16 masks on the one-tile
        path, i64 on the wide path) -> records, all on the device."""
        _require_cuda(labels, "labels")
        _require_cuda(preds, "preds")
        B, n = labels.shape
        if n != self.n_tokens - 3 or tuple(preds.shape) != (B, n):
            raise AssertionError(f"Expected {self.n_tokens - 3}, got instead {n}")
        labels = labels.contiguous().to(torch.uint8)
        preds = preds.contiguous().to(torch.int64 if self.wide else torch.int16)
        self.workspace(B, labels.device)
        if zero_status:
            self._status.zero_()
        shape = self.shape(B)
        dl.check(self.lib, self.lib.dvs_build_records(ctypes.byref(shape), _ptr(labels), _ptr(preds), _ptr(self._records),
                                                      _nbytes(self._records), _ptr(self._status), _stream()),
                 "dvs_build_records")
        if check:
            st = self._read_status()
            if st:
                raise ValueError(f"invalid compact DAG batch (status bits {st:#x}: 1 = label out of range, "
                                 f"8 = edge not from a lower to a higher vertex id)")
        return self._records

    def loss_forward(self, shape, params: torch.Tensor, eps: Optional[torch.Tensor], losses: torch.Tensor,
                     mu: Optional[torch.Tensor] = None, logvar: Optional[torch.Tensor] = None,
                     host_tail: Optional[torch.Tensor] = None, host_seq: int = 0):
        """host_tail (pinned float32 words, the first four are used) + host_seq (24 bits): the device writes [total, recon, kld,
        (seq << 8) | flags | validation bits] there with one 16-byte store (include/dvs.h: dvs_loss_forward_notify) and re-arms
        the validation word in the same kernel."""
        _require_cuda(params, "parameters")
        ws = self.workspace(shape.batch, params.device)
        if losses.numel() < dl.LOSS_FLOATS:
            raise ValueError(f"losses must hold {dl.LOSS_FLOATS} floats")
        if host_tail is None:
            dl.check(self.lib, self.lib.dvs_loss_forward(ctypes.byref(shape), _ptr(self._records), _nbytes(self._records),
                                                         _ptr(params), params.numel(), _ptr(ws), _nbytes(ws), _ptr(eps),
                                                         _ptr(self._status), _ptr(losses), _ptr(mu), _ptr(logvar), _stream()),
                     "dvs_loss_forward")
            return
        if not host_tail.is_pinned() or host_tail.numel() < 4 or host_tail.dtype != torch.float32 or host_tail.data_ptr() % 16:
            raise ValueError("host_tail must be at least 4 pinned float32 words, 16-byte aligned")
        dl.check(self.lib, self.lib.dvs_loss_forward_notify(
            ctypes.byref(shape), _ptr(self._records), _nbytes(self._records), _ptr(params), params.numel(), _ptr(ws), _nbytes(ws),
            _ptr(eps), _ptr(self._status), _ptr(losses), _ptr(mu), _ptr(logvar), host_tail.data_ptr(), int(host_seq) & 0xFFFFFF,
            _stream()), "dvs_loss_forward_notify")

    def loss_backward(self, shape, params: torch.Tensor, gcoef: torch.Tensor, grads: torch.Tensor,
                      clip_scratch: Optional[torch.Tensor] = None):
        """clip_scratch (device f32[CLIP_SCRATCH_FLOATS], the fused optimiser's scratch): dvs_loss_backward_sq — the kernel that
        sums the gradient slabs also leaves the partial sums of squares the optimiser clips with (single-process steps only)."""
        ws = self.workspace(shape.batch, params.device)
        if grads.numel() < params.numel():
            raise ValueError("gradient buffer is smaller than the parameter buffer")
        if clip_scratch is not None:
            if clip_scratch.numel() < dl.CLIP_SCRATCH_FLOATS or clip_scratch.dtype != torch.float32:
                raise ValueError(f"clip_scratch must hold {dl.CLIP_SCRATCH_FLOATS} float32 words")
            dl.check(self.lib, self.lib.dvs_loss_backward_sq(
                ctypes.byref(shape), _ptr(self._records), _nbytes(self._records), _ptr(params), params.numel(), _ptr(ws), _nbytes(ws),
                _ptr(gcoef), _ptr(grads), _ptr(clip_scratch), _stream()), "dvs_loss_backward_sq")
            return
        dl.check(self.lib, self.lib.dvs_loss_backward(ctypes.byref(shape), _ptr(self._records), _nbytes(self._records),
                                                      _ptr(params), params.numel(), _ptr(ws), _nbytes(ws), _ptr(gcoef),
                                                      _ptr(grads), _stream()), "dvs_loss_backward")

    def encode(self, shape, params: torch.Tensor, mu: torch.Tensor, logvar: torch.Tensor):
        ws = self.workspace(shape.batch, params.device)
        dl.check(self.lib, self.lib.dvs_encode(ctypes.byref(shape), _ptr(self._records), _nbytes(self._records),
                                               _ptr(params), params.numel(), _ptr(ws), _nbytes(ws), _ptr(mu), _ptr(logvar),
                                               _stream()), "dvs_encode")

    def decode(self, shape, params: torch.Tensor, z: torch.Tensor, uniforms: Optional[torch.Tensor]) -> torch.Tensor:
        """dvs_decode: the whole autoregressive generation loop on the device.  Returns the raw dvs_decode_state
        records as a uint8 tensor [B, DECODE_STATE_BYTES]."""
        _require_cuda(params, "parameters")
        _require_cuda(z, "z")
        B = z.shape[0]
        ws = self.workspace(B, params.device)
        state = torch.empty(B, dl.DECODE_STATE_BYTES, dtype=torch.uint8, device=params.device)
        dl.check(self.lib, self.lib.dvs_decode(ctypes.byref(shape), _ptr(params), params.numel(), _ptr(ws), _nbytes(ws),
                                               _ptr(self._records), _nbytes(self._records), _ptr(z), _ptr(uniforms),
                                               _ptr(state), _nbytes(state), _stream()), "dvs_decode")
        return state

    def clip_adam(self, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, max_norm, scratch, guard=None,
                  from_partials: bool = False):
        """guard: optional device f32[2] = [non-finite flag, invalid-features flag]; the update is skipped on the device
        when either is non-zero (include/dvs.h).  from_partials: `scratch` already holds the partial sums of squares of
        `grads` (loss_backward(clip_scratch=scratch) wrote them): dvs_clip_adam_from_partials, one launch instead of two."""
        fn = self.lib.dvs_clip_adam_from_partials if from_partials else self.lib.dvs_clip_adam
        dl.check(self.lib, fn(params.numel(), _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), lr, beta1, beta2, eps,
                              int(step), float(max_norm), _ptr(scratch), _ptr(guard), _stream()),
                 "dvs_clip_adam_from_partials" if from_partials else "dvs_clip_adam")

    def activation(self, batch: int, slot: int) -> torch.Tensor:
        out = torch.empty(batch, 16 * self.tiles, 64, dtype=torch.float32, device=self._ws.device)
        shape = self.shape(batch)
        dl.check(self.lib, self.lib.dvs_debug_activation(ctypes.byref(shape), _ptr(self._ws), slot, _ptr(out), _stream()),
                 "dvs_debug_activation")
        return out

    def flatten(self, params: Dict[str, torch.Tensor], device) -> torch.Tensor:
        flat = torch.zeros(self.param_floats, dtype=torch.float32, device=device)
        for name, off, shp in self.table:
            v = params[name].detach().to(device=device, dtype=torch.float32).reshape(-1)
            flat[off:off + v.numel()] = v
        return flat

    def unflatten(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for name, off, shp in self.table:
            n = 1
            for s in shp:
                n *= s
            out[name] = flat[off:off + n].view(shp)
        return out
