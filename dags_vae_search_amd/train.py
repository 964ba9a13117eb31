"""Train-step API of the reference, MI355X-native underneath.

Mirrors ``train_batch`` (experiments/03_synthetic_12/main.py:95-118; API twin src/train_model.py:18-28),
``load_model_state`` (src/train_utils.py:11-36), ``collate_graph_batch`` (src/train_utils.py:39-40),
``pace_collate_fn`` (main.py:75-92) and the epoch loop semantics of ``train_model`` (main.py:175-193).
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, Iterable, Optional

import torch
from torch.nn.utils import clip_grad_norm_

from .features import collate_graph_batch, pace_collate_fn  # noqa: F401  (re-exported, reference names)
from .optim import Adam as FusedAdam


def load_model_state(model, state_name):
    """src/train_utils.py:11-36: load the checkpoint entries whose keys exist in the model (weights only)."""
    pretrained = torch.load(state_name, map_location="cpu", weights_only=True)
    model_dict = model.state_dict()
    pretrained = {k: v for k, v in pretrained.items() if k in model_dict}
    model.load_state_dict(pretrained, strict=False) if len(pretrained) != len(model_dict) else \
        model.load_state_dict(pretrained)
    return


def _to_device(batch: Dict, device) -> Dict:
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def train_batch(batch, model, optimizer, max_grad_norm=1.0, group=None):
    """One train step: loss_direct -> backward -> clip_grad_norm_ -> optimizer.step (main.py:95-118).

    Returns ``(loss_value: float, recon, kld)`` like the reference.  With the fused optimiser
    (``dags_vae_search_amd.optim.Adam``) the step runs without an autograd graph: forward+backward kernels write the
    flat gradient, ``group`` (a torch.distributed process group, or True for the default one) SUM-all-reduces it over
    RCCL, and one fused kernel pair clips and applies Adam.  Either way the call returns once the step's FORWARD is done
    (where the reference's ``loss.item()`` returns) with the global loss; backward, all-reduce and optimiser are queued.  With any other optimiser the reference sequence runs on
    top of the autograd-wrapped kernels (parameters are ordinary leaf tensors with .grad)."""
    if not model.training:
        model.train()
    optimizer.zero_grad()
    if isinstance(optimizer, FusedAdam):
        if optimizer._model is None:
            optimizer.attach(model)
        # The step's only host sync is ONE small pinned read-back, taken right behind the FORWARD on a side stream
        # (PaceVaeV3._early_read): feature-validation word + loss scalars + non-finite flag.  Data-parallel: the five scalars
        # are all-reduced there too (a 20-byte collective overlapping the backward), the flat gradient is all-reduced on the
        # main stream after the backward (ONE collective of P floats, SUM, then clip: SURVEY 8e).
        dp = group is not None
        ex = None
        if dp:
            from .dist import DpExchange
            pg = None if group is True else group
            ex = getattr(model, "_dp_exchange", None)
            if ex is None or ex.group is not pg:
                ex = model._dp_exchange = DpExchange(pg)
        # forward -> [side stream: ex.scalars] -> backward   (dist.DpExchange documents the order of the two collectives)
        # single process: the kernel that sums the gradient slabs also leaves the partial sums of squares the clip needs
        # (one launch less); data-parallel: the gradient changes in the all-reduce, its norm is taken afterwards
        scratch = None if dp else optimizer.clip_scratch(model.flat_params.device)
        model.loss_and_grad(batch, defer_check=True, early_read=True, exchange=ex, clip_scratch=scratch)
        guard = model._step_guard
        if dp:
            ex.gradient(model.flat_grads)
            torch.cuda.current_stream().wait_event(model._ev_tail)     # the guard below reads the all-reduced flags
            guard = ex.guard
        # A non-finite loss or an invalid batch must leave the weights and the moments alone: the reference raises inside
        # loss_direct (pace.py:97-98), before backward / clip / step (main.py:111-116).  The optimiser kernels are already
        # enqueued when the host learns about it, so they carry the two flags as a device-side guard and skip the update
        # (data-parallel: the flags were all-reduced with the losses, every rank skips and raises alike).
        optimizer.step(max_grad_norm=max_grad_norm, guard=guard, from_partials=not dp)
        host, status = model.read_step()                            # waits for the forward's notification / the side stream's copy only
        scalars = model._early_scalars                              # 0-d views of a device tensor owned by this step
        recon, kld = scalars[1], scalars[2]
        try:
            from .dist import DpExchange as _Dp
            _Dp.decide(host, status)
        except ValueError:
            optimizer.step_skipped()
            raise
        return host[0], recon, kld
    loss, recon, kld = model.loss_direct(batch)
    loss_value = loss.item()
    loss.backward()
    if group is not None:
        import torch.distributed as dist
        pg = None if group is True else group
        for p in model.parameters():
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=pg)
    clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    return loss_value, recon, kld


def train_model(model, dataset, epochs: int = 1, batch_size: int = 32, lr: float = 1e-4, max_grad_norm: float = 1.0,
                checkpoint_dir: Optional[str] = None, seed: int = 42, fused: bool = True, log=print):
    """Epoch loop with the reference's semantics (main.py:121-198): seeds 42, shuffle every epoch, Adam lr 1e-4,
    ReduceLROnPlateau('min', 0.1, patience 10) stepped on the LAST batch's loss, state_dict saved per epoch."""
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    from torch.utils.data import DataLoader
    torch.manual_seed(seed)
    model.seed(seed)
    dl = DataLoader(dataset=dataset, batch_size=batch_size, collate_fn=pace_collate_fn, shuffle=True)
    optimizer = (FusedAdam(model.parameters(), lr=lr).attach(model) if fused
                 else torch.optim.Adam(model.parameters(), lr=lr))
    scheduler = ReduceLROnPlateau(optimizer, "min", factor=0.1, patience=10)
    device = model.flat_params.device
    loss_value = math.inf
    t0 = time.time()
    history = []
    for epoch in range(1, epochs + 1):
        model.train()
        for batch in dl:
            loss_value, recon, kld = train_batch(_to_device(batch, device), model, optimizer, max_grad_norm)
        scheduler.step(loss_value)
        history.append(loss_value / batch_size)
        log("====> Epoch: {0} loss: {1:.4f}, compute time: {2:.4f}".format(epoch, loss_value / batch_size,
                                                                         time.time() - t0))
        if checkpoint_dir:
            os.makedirs(checkpoint_dir, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(checkpoint_dir, "model_checkpoint_{}.pth".format(epoch)))
    return history


def batch_test(toolkit, batch, model, encode_times: int = 10, decode_times: int = 10):
    """Reconstruction metrics of one batch of labelled graphs (experiments/03_synthetic_12/main.py:200-217):
    returns (nll, n_valid, n_perfect) over encode_times x decode_times decodes of the posterior means.  Graphs that stop
    growing early (the reference's decode then raises inside its conversion) count as invalid."""
    n_valid = 0
    n_perfect = 0
    mu, logvar = model.encode(batch)
    _, nll, _ = model.loss(batch)
    for _ in range(encode_times):
        z = mu
        for _ in range(decode_times):
            rec = model.decode(z, strict=False)
            n_valid += sum(toolkit.is_valid_graph(g) for g in rec)
            n_perfect += sum(toolkit.graph_equals(g0, g1) for g0, g1 in zip(batch, rec))
    return nll, n_valid, n_perfect


def model_test(model, dataset, toolkit, batch_size: int = 32, encode_times: int = 10, decode_times: int = 10, shuffle: bool = True,
               seed: Optional[int] = None, log=None):
    """The reference's evaluation driver (experiments/03_synthetic_12/main.py:219-283): eval mode, batches of graph objects
    (list collate, shuffled), ``batch_test`` on each, running averages of the reconstruction loss per graph, the share of
    valid decodes and the share of exact (label-preserving isomorphic) reconstructions over encode_times x decode_times
    decodes per graph.  Returns {"recon_loss", "valid_ratio", "recon_accuracy", "graphs"}.  The reference divides by
    BATCH_SIZE * (batches so far), which over-counts a ragged last batch; this divides by the graphs actually seen.  With the
    batched on-device ``decode`` a 32-graph batch takes milliseconds instead of the reference's ~26 s (its progress-bar
    comments at main.py:236-239)."""
    from torch.utils.data import DataLoader
    model.eval()
    if seed is not None:
        torch.manual_seed(seed)
        model.seed(seed)
    loader = DataLoader(dataset=dataset, batch_size=batch_size, collate_fn=lambda data: [g for g in data], shuffle=shuffle)
    total_nll, n_valid, n_perfect, n_graphs = 0.0, 0, 0, 0
    for i, batch in enumerate(loader):
        nll, v, p = batch_test(toolkit, batch, model, encode_times, decode_times)
        total_nll += float(nll)
        n_valid += v
        n_perfect += p
        n_graphs += len(batch)
        if log is not None:
            decodes = n_graphs * encode_times * decode_times
            log(f"batch {i}: AVG recon loss: {total_nll / n_graphs}, valid ratio: {n_valid / decodes:.4f}, "
                f"recon accuracy: {n_perfect / decodes:.4f}")
    decodes = max(n_graphs * encode_times * decode_times, 1)
    return {"recon_loss": total_nll / max(n_graphs, 1), "valid_ratio": n_valid / decodes, "recon_accuracy": n_perfect / decodes,
            "graphs": n_graphs}
