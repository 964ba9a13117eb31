"""Synthetic labelled DAGs with the distribution of the reference's encoder datasets.

Imitates src/encoders/utils.py:18-57,96-202 + src/toolkit/labeled.py:281-333 without igraph: the edge count is drawn
from the curriculum ``linspace(n-1, floor(0.4*n(n-1)/2), 20)`` weighted ``(k+1)^2`` (data/synthetic_v12_c2/
encoder_dataset.py:18-25), ``m`` of the n(n-1)/2 upper-triangular slots are sampled uniformly (rejected unless weakly
connected), labels are a random sample of range(card) when card >= n (``label_random_method='sample'``), else zeros.
Vertices come out already in topological (row-codec) order.
"""
from __future__ import annotations

import numpy as np

from .features import LabeledGraph


def _weakly_connected(n, edges) -> bool:
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for u, v in edges:
        parent[find(u)] = find(v)
    return len({find(v) for v in range(n)}) == 1


def synthetic_dags(n: int, card: int, count: int, seed: int = 42, density_limit: float = 0.4, steps_limit: int = 20):
    rng = np.random.default_rng(seed)
    slots = [(u, v) for v in range(n) for u in range(v)]
    lo, hi = n - 1, max(n - 1, int(density_limit * n * (n - 1) / 2))
    counts = np.unique(np.linspace(lo, hi, steps_limit).astype(int))
    w = (np.arange(len(counts)) + 1.0) ** 2
    w /= w.sum()
    graphs = []
    while len(graphs) < count:
        m = int(rng.choice(counts, p=w))
        for _ in range(100):
            idx = rng.choice(len(slots), size=m, replace=False)
            edges = sorted(slots[i] for i in idx)
            if _weakly_connected(n, edges):
                break
        else:
            continue
        labels = list(rng.permutation(card)[:n]) if card >= n else [0] * n
        graphs.append(LabeledGraph([int(x) for x in labels], edges))
    return graphs
