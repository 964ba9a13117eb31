"""Batched BIC scorer for discrete Bayesian networks on the GPU.

Mirror of the reference's ``BNLearnWrapper`` (src/problem/bn/bnlearn.py:10-61): ``score(labeled_graph)`` returns
``bnlearn::score(net, data, type="bic")`` of the DAG whose vertex v stands for data-set variable ``labels[v]``.  The
reference starts one ``Rscript`` per graph; here ``score_batch`` scores thousands of structures in one HIP launch
(csrc/k_bic.hip: LDS contingency counts + fp64 log-likelihood).  The data set is passed in (the reference pulls it from
R's ``data(asia)``; its CSV copies are data/bn_asia/target.csv, data/bn_sachs/target.csv).
"""
from __future__ import annotations

import csv
import ctypes
from typing import List, Sequence

import numpy as np
import torch

from . import _lib as dl
from .features import LABEL_KEY, _as_labels_edges


def load_discrete_csv(path: str):
    """(names, level-coded uint8 array [S, n]); levels coded by sorted name (BIC does not depend on the coding)."""
    rows = list(csv.reader(open(path)))
    names, rows = rows[0], rows[1:]
    cols = []
    for c in range(len(names)):
        levels = sorted(set(r[c] for r in rows))
        index = {v: i for i, v in enumerate(levels)}
        cols.append(np.fromiter((index[r[c]] for r in rows), np.uint8, len(rows)))
    return names, np.stack(cols, 1)


class BNLearnWrapper:
    def __init__(self, dataset_name: str, metric_name: str = "bic", data=None, device="cuda"):
        """``data``: path of a CSV with a header row of variable names, or a level-coded integer array [S, n]."""
        if metric_name != "bic":
            raise NotImplementedError(f"only the 'bic' score is built (got {metric_name!r})")
        if data is None:
            raise ValueError("pass data=<csv path or level-coded array>: the reference's R data sets are not bundled")
        self.dataset_name = dataset_name
        self.metric_name = metric_name
        arr = load_discrete_csv(data)[1] if isinstance(data, str) else np.asarray(data)
        if arr.ndim != 2 or arr.shape[1] > dl.MAX_TOKENS or arr.min() < 0 or arr.max() > 15:
            raise ValueError("data must be [samples, n_vars <= 48] with level codes 0..15")
        self.n_samples, self.n_vars = arr.shape
        words = (self.n_vars + 15) // 16
        packed = np.zeros((self.n_samples, words), np.uint64)
        for i in range(self.n_vars):
            packed[:, i // 16] |= arr[:, i].astype(np.uint64) << np.uint64(4 * (i % 16))
        self.lib = dl.load()
        self.device = torch.device(device)
        self._data = torch.from_numpy(packed.view(np.int64)).to(self.device)
        self._card = torch.from_numpy((arr.max(0) + 1).astype(np.uint8)).to(self.device)

    def _parent_masks(self, graphs: Sequence, label_key: str) -> np.ndarray:
        n = self.n_vars
        masks = np.zeros((len(graphs), n), np.uint64)
        for b, g in enumerate(graphs):
            labels, edges = _as_labels_edges(g, label_key)
            assert n == len(labels), f"Expected {n} vertices, but got {len(labels)}"                      # bnlearn.py:34
            assert sorted(labels) == list(range(n)), f"Expected graph labels from 0 to {n - 1}, but got {labels}"   # :35
            for u, v in edges:
                masks[b, labels[v]] |= np.uint64(1) << np.uint64(labels[u])
        return masks

    def score_batch(self, labeled_graphs: Sequence, label_key: str = LABEL_KEY) -> List[float]:
        masks = self._parent_masks(labeled_graphs, label_key)
        return self.score_masks(torch.from_numpy(masks.view(np.int64))).cpu().tolist()

    def score_masks(self, parents: torch.Tensor) -> torch.Tensor:
        """parents: int64 [B, n_vars] bit rows in data-set variable indices (bit u of [b, v] <=> u -> v); -> f64 [B]."""
        parents = parents.to(self.device).contiguous()
        B = parents.shape[0]
        scratch = torch.empty(B, self.n_vars, dtype=torch.float64, device=self.device)
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        dl.check(self.lib, self.lib.dvs_bic_scores(B, self.n_vars, self.n_samples, p(self._data), p(self._card), p(parents),
                                                   p(scratch), p(out), p(status),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                 "dvs_bic_scores")
        if int(status.item()) & 16:
            raise ValueError("a variable's parent set is too large for the on-chip counting paths (dense table: 36 864 "
                             "cells; sorted samples: 16 384 samples, 63 key bits)")
        return out

    def score_compact(self, batch) -> torch.Tensor:
        """BIC of a ``CompactBatch`` (records.py) that lives on the device -> float64 [B] on the device: the relabelling
        (dvs_bic_parent_masks) and the scoring (dvs_bic_scores) are both HIP launches, nothing touches the host."""
        labels = batch.labels.to(self.device).contiguous()
        preds = batch.preds.to(self.device).contiguous()
        B, n = labels.shape
        assert n == self.n_vars, f"Expected {self.n_vars} vertices, but got {n}"                             # bnlearn.py:34
        parents = torch.empty(B, n, dtype=torch.int64, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        dl.check(self.lib, self.lib.dvs_bic_parent_masks(B, n, 1 if preds.dtype == torch.int64 else 0, p(labels), p(preds),
                                                         p(parents), p(status),
                                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                 "dvs_bic_parent_masks")
        out = self.score_masks(parents)
        if int(status.item()) & 32:
            raise AssertionError(f"Expected graph labels from 0 to {n - 1}")                                  # bnlearn.py:35
        return out

    def score(self, labeled_graph, label_key: str = LABEL_KEY) -> float:
        return self.score_batch([labeled_graph], label_key)[0]
