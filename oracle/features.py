"""ORACLE (test infrastructure, never shipped): CPU restatement of the feature front-end.

Restates, in plain numpy / Python, what the reference does between a parquet row and
the dense feature dict consumed by ``PaceVaeV3.loss_direct``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Reference being restated (paths relative to /root/reference):
  * row codec            src/toolkit/labeled.py:132-154   (``l{i}`` label, ``e{i}`` '0/1' string)
  * PACE wrapping        src/encoders/pace.py:1250-1288   (start/input/output vertices, label+3)
  * topological order    src/encoders/pace.py:1245-1248   (igraph ``topological_sorting``)
  * ancestor mask        src/encoders/pace.py:1307-1343
  * dense features       src/encoders/pace.py:1345-1478
  * collate              experiments/03_synthetic_12/main.py:75-92

Pinned by: tests/models/test_pace_utils.py:18-61 (one wrapping fixture incl. positions),
tests/toolkit/test_labeled.py:49-64 (row codec) and the 1 408-row (graph -> mu) known answer
(experiments/01_bn_asia: data/test parquet + model_checkpoint_110 + predictor_dataset), see
tests/golden/gen_golden.py.
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Sequence, Tuple

import numpy as np

LABEL_INPUT = 0   # pace.py:1153
LABEL_OUTPUT = 1  # pace.py:1154
LABEL_START = 2   # pace.py:1155
NUM_HEADS = 8


def row_to_labeled(row: Dict, n: int) -> Tuple[List[int], List[Tuple[int, int]]]:
    """labeled.py:132-154 — vertex v has label ``l{v}``; char u of ``e{v}`` == '1' means edge u -> v (u < v)."""
    labels, edges = [], []
    for v in range(n):
        labels.append(int(row[f"l{v}"]))
        conn = row[f"e{v}"]
        if len(conn) != v:
            raise ValueError(f"{v} elements expected to be in 'e{v}'")  # labeled.py:108-112
        for u in range(v):
            if int(conn[u]) == 1:
                edges.append((u, v))
    return labels, edges


def labeled_to_row(labels: Sequence[int], edges: Sequence[Tuple[int, int]]) -> Dict:
    """Inverse of :func:`row_to_labeled` for graphs already in topological vertex order (u < v)."""
    n = len(labels)
    es = set(edges)
    row = {f"l{v}": int(labels[v]) for v in range(n)}
    for v in range(n):
        row[f"e{v}"] = "".join("1" if (u, v) in es else "0" for u in range(v))
    return row


def topological_order_fifo(num_vertices: int, edges: Sequence[Tuple[int, int]]) -> List[int]:
    """igraph ``Graph.topological_sorting(mode='out')`` semantics (pace.py:1247): Kahn's algorithm with a
    FIFO queue, seeded with the zero-in-degree vertices in id order, out-neighbours relaxed in ascending id."""
    out = [[] for _ in range(num_vertices)]
    indeg = [0] * num_vertices
    for u, v in edges:
        out[u].append(v)
        indeg[v] += 1
    for lst in out:
        lst.sort()
    q = deque(v for v in range(num_vertices) if indeg[v] == 0)
    order = []
    while q:
        u = q.popleft()
        order.append(u)
        for v in out[u]:
            indeg[v] -= 1
            if indeg[v] == 0:
                q.append(v)
    if len(order) != num_vertices:
        raise ValueError("graph is not a dag")
    return order


def pace_wrap(labels: Sequence[int], edges: Sequence[Tuple[int, int]]):
    """pace.py:1250-1288.  Returns (pace_labels[N], pace_edges, positions[N]) with N = n + 3.

    Quirk kept on purpose (pace.py:1286): ``positions[v] = order[v]`` — the topological *order list* is
    assigned by vertex index, it is not the inverse permutation."""
    n = len(labels)
    N = n + 3
    out_id = N - 1
    pl = [0] * N
    pl[0] = LABEL_START
    pl[1] = LABEL_INPUT
    pl[out_id] = LABEL_OUTPUT
    pe: List[Tuple[int, int]] = [(0, 1)]
    preds = [[] for _ in range(n)]
    for u, v in edges:
        preds[v].append(u)
    for v in range(n):
        pl[v + 2] = int(labels[v]) + 3
        if not preds[v]:
            pe.append((1, v + 2))
        else:
            pe.extend((u + 2, v + 2) for u in sorted(preds[v]))
    outdeg = [0] * N
    for u, _ in pe:
        outdeg[u] += 1
    for v in range(N):
        if outdeg[v] == 0 and v != out_id:
            pe.append((v, out_id))
    positions = topological_order_fifo(N, pe)
    return pl, pe, positions


def reachability(adj: np.ndarray) -> np.ndarray:
    """pace.py:1307-1338 — reach[a][b] = there is a path a -> b of length >= 1, or a == b."""
    N = adj.shape[0]
    reach = adj.astype(bool).copy()
    cur = adj.astype(np.int64)
    a = adj.astype(np.int64)
    for _ in range(1, N - 1):
        new = (cur @ a) > 0
        new &= ~reach
        if not new.any():
            break
        reach |= new
        cur = new.astype(np.int64)
    reach |= np.eye(N, dtype=bool)
    return reach


def dense_features_one(labels: Sequence[int], edges: Sequence[Tuple[int, int]], card: int,
                       num_heads: int = NUM_HEADS) -> Dict:
    """pace.py:1345-1478 for ONE graph (batch of 1), numpy arrays with the reference's shapes/dtypes."""
    n = len(labels)
    N, C = n + 3, card + 3
    pl, pe, pos = pace_wrap(labels, edges)
    lab1h = np.zeros((1, N, C), np.float32)
    pos1h = np.zeros((1, N, N), np.float32)
    lab1h[0, np.arange(N), pl] = 1.0
    pos1h[0, np.arange(N), pos] = 1.0
    adj = np.zeros((N, N), np.float32)
    for u, v in pe:
        adj[u, v] = 1.0
    non_reach = ~reachability(adj)
    source = np.repeat(non_reach[None, 1:, 1:], num_heads, 0)            # pace.py:1429-1433 (not transposed)
    target = np.repeat(non_reach.T[None], num_heads, 0)                   # pace.py:1435-1438 + .transpose(1,2) at 1474
    memory = np.zeros((num_heads, N, N - 1), bool)                        # pace.py:1446-1453 (all graphs full size)
    return {
        "vertex_label_features": lab1h,
        "vertex_position_features": pos1h,
        "adjacency_matrices": adj[None],
        "source_masks": source.copy(),
        "target_masks": target.copy(),
        "memory_masks": memory,
        "num_vertices": [N],
        "vertex_labels": [list(pl[1:])],                                  # pace.py:1464
    }


def collate(items: Sequence[Dict]) -> Dict:
    """experiments/03_synthetic_12/main.py:75-92 (``pace_collate_fn``), numpy flavour."""
    keys = ["vertex_label_features", "vertex_position_features", "adjacency_matrices",
            "source_masks", "target_masks", "memory_masks"]
    out = {k: np.concatenate([it[k] for it in items], 0) for k in keys}
    out["num_vertices"] = [it["num_vertices"][0] for it in items]
    out["vertex_labels"] = [it["vertex_labels"][0] for it in items]
    return out


def dense_features(graphs: Sequence[Tuple[Sequence[int], Sequence[Tuple[int, int]]]], card: int) -> Dict:
    return collate([dense_features_one(l, e, card) for (l, e) in graphs])


def to_torch(features: Dict):
    import torch
    out = {}
    for k, v in features.items():
        out[k] = torch.from_numpy(v) if isinstance(v, np.ndarray) else v
    return out


# ---------------------------------------------------------------------------------------------------
# Synthetic DAGs imitating src/encoders/utils.py:18-57,96-202 + src/toolkit/labeled.py:281-333
# (edge-count curriculum weighted (k+1)^2, uniform edge slots in topological vertex order, weakly
# connected, labels = random permutation when card >= n else zeros).  SURVEY.md §8d.
# ---------------------------------------------------------------------------------------------------

def _weakly_connected(n: int, edges) -> bool:
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for u, v in edges:
        parent[find(u)] = find(v)
    return len({find(v) for v in range(n)}) == 1


def synthetic_dags(n: int, card: int, count: int, seed: int = 42, density_limit: float = 0.4,
                   steps_limit: int = 20):
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
        graphs.append(([int(x) for x in labels], edges))
    return graphs
