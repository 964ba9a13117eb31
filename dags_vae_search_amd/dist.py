"""Data-parallel sharding of the DAG minibatch: one process per GPU, torch.distributed over RCCL.

DAGs are independent and the loss / every gradient are plain SUMs over DAGs (pace.py:1919,1965-1970,2030), so the
global batch is cut into contiguous rank slices; each rank runs forward+backward on its slice with the GLOBAL DAG
index feeding the counter-based RNG (identical masks to the single-GPU big batch), then ONE SUM all-reduce of the
flat gradient (P*4 bytes ~ 1.2 MB) + the 4 loss scalars, then clip on the reduced gradient and a replicated Adam.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

TENSOR_KEYS = ["vertex_label_features", "vertex_position_features", "adjacency_matrices", "source_masks",
               "target_masks", "memory_masks"]
HEAD_KEYS = {"source_masks", "target_masks", "memory_masks"}      # 8 rows per DAG (one per head)


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    per, rem = divmod(global_batch, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def shard_features(features: Dict, rank: int, world: int, num_heads: int = 8) -> Tuple[Dict, int]:
    """Contiguous slice of a collated feature dict for `rank`; returns (shard, dag_offset)."""
    B = features["vertex_label_features"].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    out = {}
    for k, v in features.items():
        if torch.is_tensor(v):
            out[k] = v[lo * num_heads:hi * num_heads] if k in HEAD_KEYS else v[lo:hi]
        else:
            out[k] = v[lo:hi]
    return out, lo


def allreduce_gradients(flat_grads: torch.Tensor, losses: torch.Tensor, group=None):
    """Single-collective form of the step's exchange (kept for callers that want the loss scalars only at the end of the
    step; ``train.train_batch`` all-reduces the five scalars right behind the forward on a side stream instead, so that the
    host reads the global loss early, and the gradient alone here on the main stream).  The step's ONE exchange: SUM all-reduce of the flat gradient and of the [total, recon, kld, flag] scalars
    (RCCL over xGMI on GPUs, gloo in the CPU tests).  Clipping and Adam run AFTER it, replicated on every rank.
    ``PaceVaeV3.loss_and_grad`` keeps the four scalars right behind the gradient in one allocation, so both travel in a
    single collective (the message is ~1.2 MB: latency-bound, a second call would cost as much as the first)."""
    import torch.distributed as dist
    if (losses.untyped_storage().data_ptr() == flat_grads.untyped_storage().data_ptr()
            and losses.storage_offset() == flat_grads.storage_offset() + flat_grads.numel()
            and flat_grads.is_contiguous() and losses.is_contiguous()):
        both = torch.as_strided(flat_grads, (flat_grads.numel() + losses.numel(),), (1,), flat_grads.storage_offset())
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=group)
    return flat_grads, losses
