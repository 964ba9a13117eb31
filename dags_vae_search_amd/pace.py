"""Drop-in ``PaceVaeV3`` (reference: src/encoders/pace.py:1139-2046) whose train-step arithmetic runs in
hand-written HIP kernels (libdvs_hip.so) on MI355X.

Same constructor arguments, properties, state-dict keys/shapes (108 tensors), ``loss_direct`` /
``encode_direct`` / ``loss`` / ``encode`` / ``prepare_features`` / ``reparameterize`` signatures as the reference.
Differences a user can observe:
  * the model must live on a GPU for any compute (`.to("cuda")`); there is no CPU path;
  * only the BASELINE architecture is built (embedding 32, 8 heads, 3 layers, width 64, latent 32, fc_hidden 32,
    max_num_vertices <= 45, cardinality <= 45: up to 13 on the one-wave-per-DAG kernels, beyond that on the tiled
    workgroup-per-DAG kernels) — anything else raises NotImplementedError;
  * dropout masks / reparameterisation noise come from a counter-based generator keyed by (seed, step, DAG index)
    instead of torch's global generator (``model.seed(s)`` re-seeds it; ``eps=`` injects the noise);
  * ``decode`` (generation, pace.py:1666-1749) runs batched on the device and returns LabeledGraph objects; its
    random draws are the model's own counter-based ones (or injected ``uniforms``), see decode().
All 108 parameters are views into ONE flat fp32 buffer (``model.flat_params``) and their gradients views into
``model.flat_grads``: one RCCL all-reduce and one fused clip+Adam kernel cover the whole model.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Optional, Tuple

import os

import torch
from torch import nn

from . import _lib as dl
from . import features as feat
from .engine import PaceEngine
from .records import CompactBatch

LABEL_KEY = feat.LABEL_KEY
POSITION_KEY = feat.POSITION_KEY


# ---- parameter containers with the reference's module tree (names + init order => identical state dicts) -------------
class GnnPositionalEncoding(nn.Module):          # pace.py:186-199
    def __init__(self, ninp, dropout, max_n):
        super().__init__()
        self.ninp, self.max_n, self.dropout = ninp, max_n, dropout
        self.W1 = nn.Parameter(torch.zeros(2 * max_n, 2 * ninp))
        self.W2 = nn.Parameter(torch.zeros(2 * ninp, ninp))
        nn.init.xavier_uniform_(self.W1.data, gain=1.414)
        nn.init.xavier_uniform_(self.W2.data, gain=1.414)


class _EncoderLayerParams(nn.Module):           # pace.py:17-43
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)


class _DecoderLayerParams(nn.Module):           # pace.py:110-133
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm3 = nn.LayerNorm(d_model, eps=1e-5)


class _Stack(nn.Module):                        # pace.py:70-80 / 157-161: deep copies of one initialised layer
    def __init__(self, layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.num_layers = num_layers


class _PaceLoss(torch.autograd.Function):
    """loss_direct as one autograd node: forward = dvs_loss_forward, backward = dvs_loss_backward."""

    @staticmethod
    def forward(ctx, model, shape, eps, *params):
        losses = torch.zeros(dl.LOSS_FLOATS, dtype=torch.float32, device=model.flat_params.device)
        model._engine.loss_forward(shape, model.flat_params, eps, losses)
        model._fwd_generation += 1
        ctx.model, ctx.shape, ctx.generation = model, shape, model._fwd_generation
        model._last_losses = losses
        return losses[0].clone(), losses[1].clone(), losses[2].clone()

    @staticmethod
    def backward(ctx, g_total, g_recon, g_kld):
        model = ctx.model
        if ctx.generation != model._fwd_generation:
            raise RuntimeError("PaceVaeV3: the saved activations of this loss were overwritten by a later forward "
                               "(call backward() before the next loss_direct())")
        dev = model.flat_params.device
        z = torch.zeros((), device=dev)
        g_total = z if g_total is None else g_total
        g_recon = z if g_recon is None else g_recon
        g_kld = z if g_kld is None else g_kld
        gcoef = torch.stack([g_total + g_recon, ctx.shape.beta * g_total + g_kld]).float().contiguous()
        flat = torch.empty_like(model.flat_params)
        model._engine.loss_backward(ctx.shape, model.flat_params, gcoef, flat)
        grads = [flat[off:off + n].view(shp) for (_, off, shp), n in zip(model._engine.table, model._numels)]
        return (None, None, None, *grads)


_EARLY_READ_MODE = os.environ.get("DVS_EARLY_READ", "poll")      # "event": the side-stream copy behind an event (A/B, data parallel)


class PaceVaeV3(nn.Module):
    def __init__(
            self,
            max_num_vertices: int,
            vertex_label_cardinality: int,
            vertices_embedding_size: int = 256,
            num_heads: int = 8,
            num_layers: int = 6,
            ff_hidden_size: int = 512,
            latent_layer_size: int = 64,
            fc_hidden: int = 256,
            dropout: float = 0.25,
            graph_label_key: str = LABEL_KEY,
            graph_position_key: str = POSITION_KEY,
            graph_label_input: int = 0,
            graph_label_output: int = 1,
            graph_label_start: int = 2,
    ):
        super().__init__()
        built = dict(vertices_embedding_size=dl.EMB, num_heads=dl.HEADS, num_layers=dl.LAYERS,
                     ff_hidden_size=dl.D_MODEL, latent_layer_size=dl.LATENT, fc_hidden=dl.FC_HIDDEN)
        given = dict(vertices_embedding_size=vertices_embedding_size, num_heads=num_heads, num_layers=num_layers,
                     ff_hidden_size=ff_hidden_size, latent_layer_size=latent_layer_size, fc_hidden=fc_hidden)
        if given != built:
            raise NotImplementedError(f"this MI355X build implements the BASELINE architecture {built}; got {given}")
        if max_num_vertices + 3 > dl.MAX_TOKENS or vertex_label_cardinality + 3 > dl.MAX_TOKENS or max_num_vertices < 1:
            raise NotImplementedError("this build supports max_num_vertices <= 45 and vertex_label_cardinality <= 45 "
                                      "(up to three 16-token tiles per DAG)")
        self._max_num_vertices = max_num_vertices + 3          # pace.py:1159
        self._vertex_label_cardinality = vertex_label_cardinality + 3
        self.vertices_embedding_size = vertices_embedding_size
        self.num_heads = num_heads
        self.num_layers = num_layers
        self.ff_hidden_size = ff_hidden_size
        self.latent_layer_size = latent_layer_size
        self.dropout = dropout
        self._graph_label_key = graph_label_key
        self._graph_position_key = graph_position_key
        self._graph_label_input = graph_label_input
        self._graph_label_output = graph_label_output
        self._graph_label_start = graph_label_start

        # same construction order as pace.py:1176-1207 => same initial weights under the same torch seed
        self.vertex_position_embed = GnnPositionalEncoding(vertices_embedding_size, dropout, self.max_num_vertices)
        self.vertex_label_embed = nn.Sequential(nn.Linear(self._vertex_label_cardinality, vertices_embedding_size),
                                                nn.ReLU())
        self.encoder = _Stack(_EncoderLayerParams(ff_hidden_size, num_heads, ff_hidden_size, dropout), num_layers)
        hidden_size = self.ff_hidden_size * self.max_num_vertices
        self.hidden_size = hidden_size
        self.fc1 = nn.Linear(hidden_size, latent_layer_size)
        self.fc2 = nn.Linear(hidden_size, latent_layer_size)
        self.decoder = _Stack(_DecoderLayerParams(ff_hidden_size, num_heads, ff_hidden_size, dropout), num_layers)
        self.add_node = nn.Sequential(nn.Linear(ff_hidden_size, fc_hidden), nn.ReLU(),
                                      nn.Linear(fc_hidden, self._vertex_label_cardinality))
        self.add_edge = nn.Sequential(nn.Linear(ff_hidden_size * 2, ff_hidden_size), nn.ReLU(),
                                      nn.Linear(ff_hidden_size, 1))
        self.fc3 = nn.Linear(latent_layer_size, hidden_size)

        self._engine: Optional[PaceEngine] = None
        self._table = None
        self._numels: List[int] = []
        self.flat_params: Optional[torch.Tensor] = None
        self.flat_grads: Optional[torch.Tensor] = None
        self._fwd_generation = 0
        self._last_losses = None
        self._side_stream = None       # early loss read-back of the fused train step (_early_read)
        self._early_pending = False
        self._early_scalars = None
        self._seed = 0
        self._step = 0
        self.dag_offset = 0            # global index of the first DAG of the next batch (data-parallel shards)
        self.nan_check = True          # raise ValueError on non-finite loss (pace.py:97-98), costs one host sync
        self._flatten()

    # ---- properties (pace.py:1217-1243) ------------------------------------------------------------------------
    @property
    def max_num_vertices(self) -> int:
        return self._max_num_vertices

    @property
    def vertex_label_cardinality(self) -> int:
        return self._vertex_label_cardinality

    @property
    def graph_label_key(self) -> str:
        return self._graph_label_key

    @property
    def graph_position_key(self) -> str:
        return self._graph_position_key

    @property
    def graph_label_input(self) -> int:
        return self._graph_label_input

    @property
    def graph_label_output(self) -> int:
        return self._graph_label_output

    @property
    def graph_label_start(self) -> int:
        return self._graph_label_start

    # ---- flat parameter / gradient buffers ----------------------------------------------------------------------
    def _layout(self):
        if self._table is None:
            lib = dl.load()     # raises if libdvs_hip.so is missing: the model cannot exist without its kernels
            shape = dl.make_shape(1, self._max_num_vertices, self._vertex_label_cardinality)
            self._table, self._total = dl.param_table(lib, shape)
            names = [n for n, _ in self.named_parameters()]
            assert [t[0] for t in self._table] == names, "state-dict order mismatch with the C layout"
        return self._table, self._total

    def _flatten(self):
        table, total = self._layout()
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._numels = []
        for name, off, shp in table:
            p = params[name]
            n = p.numel()
            self._numels.append(n)
            flat[off:off + n] = p.data.reshape(-1).float()
            p.data = flat[off:off + n].view(shp)
            p.grad = None
        self.flat_params = flat
        self.flat_grads = None
        self._grad_params = None
        self._step_losses = None

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flatten()
        self._engine = None
        self._side_stream = None
        self._early_pending = False
        return out

    def bind_flat_grads(self) -> torch.Tensor:
        """Allocate the flat gradient buffer and point every parameter's .grad at its slice (a no-op when they already
        are: the check looks at the first and last parameter, which is what zero_grad(set_to_none=True) of a stock
        optimiser would have cleared)."""
        if self.flat_grads is None or self.flat_grads.device != self.flat_params.device:
            # gradient buffer + the 5 step scalars [total, recon, kld, non-finite flag, invalid-features flag] in ONE
            # allocation: the data-parallel step all-reduces both in one call (the two flags travel with the losses, so every
            # rank skips the same update and raises the same error) ... followed by the rank-local int32 feature-validation
            # word (last word of the tail), so that ONE small device->host copy ends a train step
            P = self.flat_params.numel()
            self._grads_and_losses = torch.zeros(P + 8, dtype=torch.float32, device=self.flat_params.device)
            self.flat_grads = self._grads_and_losses[:P]
            self._step_losses = self._grads_and_losses[P:P + dl.LOSS_FLOATS]
            self._fixed_guard = self._grads_and_losses[P + 3:P + 5]     # dvs_clip_adam skips the update when either is set
            self._step_guard = self._fixed_guard                        # (the notify variant of a step re-points it, loss_and_grad)
            self._step_tail = self._grads_and_losses[P:P + 8]
            self._step_status = self._grads_and_losses[P + 7:P + 8].view(torch.int32)
            self._host_tail = torch.zeros(8, dtype=torch.float32)
            if self.flat_params.is_cuda:
                self._host_tail = self._host_tail.pin_memory()
            self._grad_params = None
            if self._engine is not None:
                self._engine.use_status(self._step_status)
        if self._grad_params is None:
            params = dict(self.named_parameters())
            self._grad_params = [params[name] for name, _, _ in self._table]
        first, last = self._grad_params[0], self._grad_params[-1]
        base = self.flat_grads.data_ptr()
        if (first.grad is None or last.grad is None or first.grad.data_ptr() != base + 4 * self._table[0][1]
                or last.grad.data_ptr() != base + 4 * self._table[-1][1]):
            for p, (name, off, shp), n in zip(self._grad_params, self._table, self._numels):
                p.grad = self.flat_grads[off:off + n].view(shp)
        return self.flat_grads

    def _eng(self) -> PaceEngine:
        if not self.flat_params.is_cuda:
            raise RuntimeError("dags_vae_search_amd.PaceVaeV3 computes only on the GPU: call model.to('cuda') first "
                               "(there is no CPU fallback)")
        if self._engine is None:
            self._engine = PaceEngine(self._max_num_vertices, self._vertex_label_cardinality)
            assert self._engine.param_floats == self.flat_params.numel()
            if self.flat_grads is not None and self.flat_grads.device == self.flat_params.device:
                self._engine.use_status(self._step_status)
        return self._engine

    # ---- RNG ------------------------------------------------------------------------------------------------------
    def seed(self, seed: int):
        """Re-seed the counter-based dropout/noise generator (the analogue of torch.manual_seed for this model)."""
        self._seed = int(seed) & 0xFFFFFFFF
        self._step = 0

    def _next_seed(self) -> int:
        self._step += 1
        return (self._seed << 32) | (self._step & 0xFFFFFFFF)

    # ---- features (pace.py:1345-1478) ---------------------------------------------------------------------------
    def prepare_features(self, labeled_graphs_batch, fixed_memory_len: Optional[int] = None):
        device = self.flat_params.device
        return feat.prepare_features(labeled_graphs_batch, self._max_num_vertices, self._vertex_label_cardinality,
                                     self.num_heads, self._graph_label_key, self._graph_label_input,
                                     self._graph_label_output, self._graph_label_start, fixed_memory_len, device)

    def _pack(self, features, check: Optional[bool] = None, zero_status: bool = True):
        eng = self._eng()
        dev = self.flat_params.device
        check = self.nan_check if check is None else check
        if isinstance(features, CompactBatch):       # device-side front-end (records.py): no dense features at all
            eng.build_records(features.labels.to(dev), features.preds.to(dev), check=check, zero_status=zero_status)
            return len(features)
        f = {k: features[k].to(dev) for k in ("vertex_label_features", "vertex_position_features",
                                              "adjacency_matrices", "target_masks")}   # pace.py:1981-1984
        eng.pack(f, check=check, zero_status=zero_status)
        return f["vertex_label_features"].shape[0]

    def _early_read(self, exchange=None):
        """Called between the forward and the backward of a fused step: the [losses, flags | validation word] tail is final
        once the forward has run, so its device->host copy (and the re-arming of the validation word) goes to a side stream
        behind an event.  The host then blocks only until the FORWARD is done — where the reference's ``loss.item()``
        blocks (main.py:104) — and enqueues the next step while this step's backward and optimiser are still running.
        Data-parallel (``exchange``: a ``dist.DpExchange``): the five scalars are SUM-all-reduced on the side stream first
        (``exchange.scalars``: an 8-float collective that overlaps the backward), so the host reads GLOBAL losses / flags just
        as early, and the optimiser's guard (``exchange.guard``) sees the flags of every rank."""
        if self._side_stream is None:
            # HIGH priority: its five small operations (the scalars' collective, the copies) then run beside the backward's first
            # kernel; on a default-priority stream they ran BETWEEN the forward and the backward with nothing else on the device
            # (43 us per data-parallel step; same-call A/B on the world-1 RCCL path: 1.235 -> 1.211 ms)
            self._side_stream = torch.cuda.Stream(device=self.flat_params.device, priority=-1)
            self._ev_tail = torch.cuda.Event()
            self._ev_forward = torch.cuda.Event()
        # (a marker with DEVICE-scope release — hipEventDisableSystemFence through the C ABI — instead of this event was tried in
        # round 3 on the world-1 RCCL path: 1.325 ms per step either way, same-call A/B; the ~12 us the single-GPU path saved by
        # dropping its event are not the event's release scope)
        self._ev_forward.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._side_stream):
            self._side_stream.wait_event(self._ev_forward)
            src = self._step_losses
            if exchange is not None:
                src = exchange.scalars(src)
                self._host_tail[:dl.LOSS_FLOATS].copy_(src, non_blocking=True)
                self._host_tail[7:8].copy_(self._step_tail[7:8], non_blocking=True)
            else:
                self._host_tail.copy_(self._step_tail, non_blocking=True)
            self._step_status.zero_()
            self._early_scalars = src.clone()     # the caller's recon / kld tensors (ready once _ev_tail is)
            self._ev_tail.record(self._side_stream)
        self._early_pending = True

    def read_step(self):
        """The one host synchronisation of a fused train step: a pinned device->host copy of [total, recon, kld,
        non-finite flag, invalid-features flag, -, -, validation bits]; the validation word is re-armed behind the copy.  After ``_early_read`` the
        copy is already in flight on the side stream (wait for its event); otherwise it is issued here, at the end of the
        step (data-parallel steps: the loss scalars are only global after the all-reduce).
        Returns (list of 5 floats, rank-local status bits)."""
        if self._early_pending == "poll":
            self._early_pending = False
            words = self._host_tail.numpy().view("uint32")          # pinned memory, re-read on every access
            seq, spins = self._host_seq, 0
            while int(words[3]) >> 8 != seq:                        # the device's ONE 16-byte store carries the sequence number
                spins += 1
                if spins > 2_000_000:                               # ~0.3-0.5 s of polling: then fall back to a stream synchronise
                    torch.cuda.current_stream().synchronize()
                    if int(words[3]) >> 8 != seq:
                        raise RuntimeError("the device never signalled the end of the forward (dvs_loss_forward_notify)")
            # ONE 16-byte store on the device side, but the host reads four separate words: take the payload, then check that
            # the sequence word still says the same (include/dvs.h states the coherence requirement on this buffer; nothing
            # else writes it before the NEXT step's forward, which this thread has not enqueued yet — the re-read is the cheap
            # proof that payload and sequence number belong together)
            for _ in range(4):
                word = int(words[3])
                vals = self._host_tail[:3].tolist() + [float((word >> 6) & 1), float((word >> 7) & 1)]
                if int(words[3]) == word and word >> 8 == seq:
                    break
            else:
                raise RuntimeError("dvs_loss_forward_notify: the notification packet kept changing while it was read")
            # the caller's recon / kld: 0-d views of THIS step's own device tensor (main.py:111-118 returns device tensors)
            self._early_scalars = self._notify_losses
            return vals, word & 0x3F
        if self._early_pending:
            self._early_pending = False
            self._ev_tail.synchronize()
        else:
            self._host_tail.copy_(self._step_tail, non_blocking=True)
            self._step_status.zero_()
            torch.cuda.current_stream().synchronize()
        vals = self._host_tail.tolist()
        status = int(self._host_tail.view(torch.int32)[7])
        return vals[:5], status

    def _shape(self, batch: int, beta: float):
        return self._eng().shape(batch, training=self.training, dropout=self.dropout, beta=beta, eps_scale=0.01,
                                 dag_offset=self.dag_offset, seed=self._next_seed() if self.training else 0)

    # ---- encode (pace.py:1613-1647) -----------------------------------------------------------------------------
    def encode_direct(self, features: Dict) -> Tuple[torch.Tensor, torch.Tensor]:
        B = self._pack(features)
        dev = self.flat_params.device
        mu = torch.empty(B, self.latent_layer_size, device=dev)
        logvar = torch.empty(B, self.latent_layer_size, device=dev)
        shape = self._shape(B, 0.005)
        self._engine.encode(shape, self.flat_params, mu, logvar)
        self._fwd_generation += 1
        return mu, logvar

    def encode(self, labeled_graphs_batch) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.encode_direct(self.prepare_features(labeled_graphs_batch))

    def reparameterize(self, mu: torch.Tensor, log_var: torch.Tensor, epsilon_scale: float = 0.01) -> torch.Tensor:
        if self.training:                                    # pace.py:1659-1664
            std = torch.exp(0.5 * log_var)
            return mu + torch.randn_like(std) * epsilon_scale * std
        return mu

    def decode(self, z: torch.Tensor, uniforms: Optional[torch.Tensor] = None, strict: bool = True):
        """Generation (pace.py:1666-1749), batched on the device: returns one LabeledGraph per row of ``z``.

        The reference grows igraph objects on the host and samples with numpy's / torch's global generators; here the
        N-2 step loop runs in HIP (csrc/k_decode.hip) and draws from the model's counter-based generator
        (``model.seed``), or from ``uniforms`` ([B, N, N], see include/dvs.h) when given.  Quirks kept: no start->input
        edge in the grown graph, the last vertex is hooked to the loose ends only if its SAMPLED type was `output`,
        edges into the first user vertex are dropped by the PACE -> labelled conversion (pace.py:1298).  A graph that
        samples `output` early stops growing; the reference then fails with IndexError inside
        from_pace_graph_to_labeled_graph — so does this (``strict=True``); ``strict=False`` returns None for those."""
        import numpy as np
        eng = self._eng()
        dev = self.flat_params.device
        z = z.to(dev, torch.float32).contiguous()
        B = z.shape[0]
        if z.dim() != 2 or z.shape[1] != self.latent_layer_size:
            raise AssertionError(f"Expected z of shape [B, {self.latent_layer_size}], got {tuple(z.shape)}")
        N = self._max_num_vertices
        if uniforms is not None:
            uniforms = uniforms.to(dev, torch.float32).contiguous()
            if tuple(uniforms.shape) != (B, N, N):
                raise AssertionError(f"Expected uniforms of shape [{B}, {N}, {N}], got {tuple(uniforms.shape)}")
        self._step += 1
        shape = eng.shape(B, training=False, dropout=self.dropout, dag_offset=self.dag_offset,
                          seed=(self._seed << 32) | (self._step & 0xFFFFFFFF))
        raw = eng.decode(shape, self.flat_params, z, uniforms).cpu().numpy()
        self._fwd_generation += 1
        labels = raw[:, 384:432].astype(np.int64) - 3
        nv = raw[:, 432:436].copy().view(np.int32)[:, 0]
        if strict and (nv < N).any():
            raise IndexError("vertex index out of range")            # igraph's error at pace.py:1296
        # edges u -> v between user vertices (PACE ids 2 .. N-2), v == 2 skipped (pace.py:1298): unpack the little-endian
        # 64-bit parent rows into a [B, v, u] bit cube and keep its strict lower triangle
        cube = np.unpackbits(raw[:, :384].reshape(B, 48, 8), axis=2, bitorder="little")          # [B, 48 (v), 64 (u)]
        vs = np.arange(3, N - 1)
        us = np.arange(2, N - 2)
        bits = cube[:, 3:N - 1, 2:N - 2] & (us[None, None, :] < vs[None, :, None])
        bb, vi, ui = np.nonzero(bits)
        starts = np.searchsorted(bb, np.arange(B + 1))
        ev = (vs[vi] - 2).tolist()
        eu = (us[ui] - 2).tolist()
        lab_lists = labels[:, 2:N - 1].tolist()
        full = (nv >= N).tolist()
        out = []
        for b in range(B):
            if not full[b]:
                out.append(None)
                continue
            lo, hi = starts[b], starts[b + 1]
            out.append(feat.LabeledGraph(lab_lists[b], list(zip(eu[lo:hi], ev[lo:hi]))))
        return out

    # ---- loss (pace.py:1974-2046) -------------------------------------------------------------------------------
    def loss_direct(self, features: Dict, beta: float = 0.005, eps: Optional[torch.Tensor] = None):
        """(total, recon, kld) as 0-d tensors attached to the autograd graph of the 108 parameters.
        ``eps`` (optional, [B, latent], already multiplied by epsilon_scale) injects the reparameterisation noise."""
        B = self._pack(features)
        shape = self._shape(B, beta)
        if eps is not None:
            eps = eps.to(self.flat_params.device, torch.float32).contiguous()
        total, recon, kld = _PaceLoss.apply(self, shape, eps, *self.parameters())
        if self.nan_check and bool(self._last_losses[3].item() != 0.0):
            raise ValueError("NaN detected in the output of the PACE-VAE step")        # pace.py:98
        return total, recon, kld

    def loss(self, labeled_graphs_batch, beta: float = 0.005):
        return self.loss_direct(self.prepare_features(labeled_graphs_batch), beta)

    # ---- fused step pieces used by train.train_batch / bench.py (no autograd graph) ----------------------------------
    def loss_and_grad(self, features: Dict, beta: float = 0.005, eps: Optional[torch.Tensor] = None,
                      packed: bool = False, defer_check: bool = False, early_read: bool = False,
                      exchange=None, clip_scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward + backward straight into ``flat_grads`` (overwritten).  Returns the device tensor
        [total, recon, kld, non-finite flag, invalid-features flag]; nothing is synchronised."""
        eng = self._eng()
        if self.flat_grads is None or self.flat_grads.device != self.flat_params.device:
            self.bind_flat_grads()        # first step: allocate (also hands the validation word to the engine)
        if self._early_pending:           # a step whose read_step() never came (exception in between): drain it first, so
            pending, self._early_pending = self._early_pending, False   # that its re-arming of the validation word cannot land
            if pending != "poll":                                       # behind this step's pack (the polled variant re-arms
                self._ev_tail.synchronize()                             # on the main stream itself: ordered already)
        if not packed:
            # defer_check: the validation word is read (and re-armed) by read_step() at the end of the step
            self._pack(features, check=False if defer_check else None, zero_status=not defer_check)
        B = eng._ws_batch
        shape = self._shape(B, beta)
        grads = self.flat_grads
        losses = self._step_losses            # tail of the gradient allocation (see bind_flat_grads); rewritten each step
        self._step_guard = self._fixed_guard
        # early read, one GPU: the kernel that reduces the losses writes the scalars, the validation word and — last — this step's
        # sequence number into pinned host memory (dvs_loss_forward_notify); read_step() polls for it.  No event and no copy on
        # any stream: an event recorded between the forward and the backward cost the main stream ~12 us per step.
        # Data-parallel steps (``exchange``) always take the event + side-stream variant — their scalars pass through the
        # all-reduce first — whatever DVS_EARLY_READ says, so every rank issues the same collectives.
        notify = early_read and exchange is None and self._host_tail.is_pinned() and _EARLY_READ_MODE != "event"
        if notify:
            self._host_seq = (getattr(self, "_host_seq", 0) + 1) & 0xFFFFFF or 1
            # this step's scalars get a device tensor of their own: train_batch hands 0-d views of it to the caller (the
            # reference returns device tensors, main.py:111-118), and the next step must not overwrite them (caching allocator:
            # a host-side free-list pop, no device work)
            losses = self._notify_losses = torch.empty(8, dtype=torch.float32, device=grads.device)[:dl.LOSS_FLOATS]
            self._step_guard = losses[3:5]
            eng.loss_forward(shape, self.flat_params, eps, losses, host_tail=self._host_tail, host_seq=self._host_seq)
            self._early_pending = "poll"
        else:
            eng.loss_forward(shape, self.flat_params, eps, losses)
        self._fwd_generation += 1
        if early_read and not notify:
            self._early_read(exchange)
        if not hasattr(self, "_gcoef") or self._gcoef.device != grads.device or self._gcoef_beta != beta:
            self._gcoef = torch.tensor([1.0, beta], dtype=torch.float32, device=grads.device)
            self._gcoef_beta = beta
        eng.loss_backward(shape, self.flat_params, self._gcoef, grads, clip_scratch=clip_scratch)
        self.bind_flat_grads()            # host-only (re-points .grad views if an optimiser cleared them); GPU is busy
        return losses
