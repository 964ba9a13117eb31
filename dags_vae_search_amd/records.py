"""Compact DAG batches and the device-side feature front-end.

The reference turns every parquet row into an igraph object, wraps it, and builds ~7.7 KB of dense float/bool features
per DAG in Python (pace.py:1345-1478), then ``torch.cat``s B of those dicts per step (main.py:75-92).  Once the train
step takes 2.5 ms that host work would dominate by orders of magnitude.  Here a DAG is its row codec in 3 bytes per
vertex (11 for n > 13) — ``labels[v]`` (u8) and ``preds[v]`` (u16, or u64 when n > 13; bit u <=> edge u -> v, the
``e{v}`` string) — kept ON the GPU for
the whole dataset; a batch is an index gather, and ``dvs_build_records`` (HIP) does the PACE wrapping, FIFO-Kahn
positions and ancestor closure per DAG.  ``PaceVaeV3.loss_direct`` / ``encode_direct`` / ``train_batch`` accept a
``CompactBatch`` wherever they accept the reference's feature dict.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from .features import LabeledGraph, _as_labels_edges


@dataclass
class CompactBatch:
    labels: torch.Tensor      # [B, n] uint8
    preds: torch.Tensor       # [B, n] int16 (bit pattern of the u16 predecessor mask); int64 when n > 13 (wide path)

    def __len__(self) -> int:
        return self.labels.shape[0]

    def to(self, device) -> "CompactBatch":
        return CompactBatch(self.labels.to(device), self.preds.to(device))

    def __getitem__(self, idx) -> "CompactBatch":
        return CompactBatch(self.labels[idx], self.preds[idx])


def encode_graphs(graphs: Sequence, n: int, label_key: str = "type") -> CompactBatch:
    """Row codec -> compact arrays.  ``graphs``: LabeledGraph / (labels, edges) / igraph-like / parquet-row dicts
    (``l{v}``, ``e{v}``).  Edges must go from lower to higher vertex id (the codec's order, labeled.py:132-154)."""
    B = len(graphs)
    labels = np.zeros((B, n), np.uint8)
    wide = n > 13                      # n + 3 tokens > one 16-token tile: 64-bit predecessor rows
    udt, sdt = (np.uint64, np.int64) if wide else (np.uint16, np.int16)
    preds = np.zeros((B, n), udt)
    for b, g in enumerate(graphs):
        if isinstance(g, dict):
            for v in range(n):
                labels[b, v] = int(g[f"l{v}"])
                conn = g[f"e{v}"]
                if len(conn) != v:
                    raise ValueError(f"{v} elements expected to be in 'e{v}'")
                m = 0
                for u in range(v):
                    if int(conn[u]) == 1:
                        m |= 1 << u
                preds[b, v] = m
            continue
        lab, edges = _as_labels_edges(g, label_key)
        assert len(lab) == n, f"Expected {n}, got instead {len(lab)}"
        labels[b] = lab
        for u, v in edges:
            if not (0 <= u < v < n):
                raise ValueError("compact encoding needs edges u -> v with u < v (topological vertex order)")
            preds[b, v] |= udt(1 << u)
    return CompactBatch(torch.from_numpy(labels), torch.from_numpy(preds.view(sdt)))


def decode_graphs(batch: CompactBatch):
    lab = batch.labels.cpu().numpy()
    pr = batch.preds.cpu().numpy()
    pr = pr.view(np.uint64 if pr.dtype == np.int64 else np.uint16)
    out = []
    for b in range(lab.shape[0]):
        edges = [(u, v) for v in range(lab.shape[1]) for u in range(v) if (int(pr[b, v]) >> u) & 1]
        out.append(LabeledGraph([int(x) for x in lab[b]], edges))
    return out


class CompactDagDataset:
    """A whole dataset as two small device tensors; ``batches()`` yields shuffled CompactBatch index gathers
    (the analogue of DataLoader(shuffle=True, collate_fn=pace_collate_fn), main.py:149-156)."""

    def __init__(self, graphs: Sequence, n: int, device="cuda"):
        self.n = n
        self.data = encode_graphs(graphs, n).to(device)

    def __len__(self) -> int:
        return len(self.data)

    def batches(self, batch_size: int, shuffle: bool = True, generator: Optional[torch.Generator] = None,
                drop_last: bool = False) -> Iterable[CompactBatch]:
        N = len(self)
        dev = self.data.labels.device
        perm = torch.randperm(N, generator=generator).to(dev) if shuffle else torch.arange(N, device=dev)
        for s in range(0, N, batch_size):
            idx = perm[s:s + batch_size]
            if drop_last and idx.numel() < batch_size:
                break
            yield self.data[idx]
