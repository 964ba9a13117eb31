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
    """Single-collective form of the step's exchange: ONE SUM all-reduce of the flat gradient together with the
    [total, recon, kld, non-finite, invalid] scalars that ``PaceVaeV3.loss_and_grad`` keeps right behind it in the same
    allocation (RCCL over xGMI on GPUs, gloo in the CPU tests); clipping and Adam run AFTER it, replicated on every rank.
    Kept for callers that want the loss scalars only at the end of the step.  ``train.train_batch`` does NOT use it: it
    all-reduces the five scalars right behind the forward on a side stream (so the host reads the global loss early and
    the optimiser's guard sees every rank's flags) and the gradient alone on the main stream (``dp_exchange`` below)."""
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


class DpExchange:
    """The data-parallel step's exchanges, in the ONE order every rank issues them — shared by ``train.train_batch`` (GPU,
    RCCL) and by the 2-rank CPU test (gloo), so the test runs the product's sequence instead of a copy of it:

      1. ``scalars(step_losses)``   right behind the FORWARD: SUM all-reduce of an 8-float message whose first five words are
                                    [total, recon, kld, non-finite flag, invalid-features flag]; returns the reduced five.
                                    Words 3 and 4 of the result are the optimiser's guard: a flag raised on ANY rank is
                                    non-zero on EVERY rank, so all of them skip the same update and stay replicated.
      2. ``gradient(flat_grads)``   behind the BACKWARD: SUM all-reduce of the flat gradient (P floats), in place.
      3. ``decide(five, status)``   on the host, after the optimiser kernels are queued: raises what the reference raises
                                    (pace.py:97-98; features.py invariants) from the REDUCED flags — every rank raises alike.

    Both collectives go to the same process group; on the GPU the first runs on a side stream, the second on the compute
    stream.  They are issued in this order on every rank (a rank cannot reach 2 before 1: ``loss_and_grad`` issues 1 between
    its forward and its backward), which is what a communicator needs to pair them."""

    def __init__(self, group=None):
        self.group = group
        self._buf = None

    def scalars(self, step_losses: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        if self._buf is None or self._buf.device != step_losses.device:
            self._buf = torch.zeros(8, dtype=torch.float32, device=step_losses.device)
        self._buf[:5].copy_(step_losses[:5])
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        return self._buf[:5]

    @property
    def guard(self) -> torch.Tensor:
        """[non-finite flag, invalid-features flag] of the last ``scalars`` call, summed over ranks (a device view)."""
        return self._buf[3:5]

    def gradient(self, flat_grads: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grads

    @staticmethod
    def decide(five, status: int = 0):
        """``five``: the five scalars as host floats (global when they went through ``scalars``); ``status``: this rank's own
        validation bits.  Raises ValueError like the reference; returns None when the step stands."""
        if status != 0 or five[4] != 0.0:
            raise ValueError(f"batch violates the feature invariants (status bits {status:#x}"
                             f"{'' if status else ', raised on another rank'})")
        if five[3] != 0.0:
            raise ValueError("NaN detected in the output of the PACE-VAE step")    # pace.py:97-98
