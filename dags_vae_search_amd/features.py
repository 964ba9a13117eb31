"""Host-side feature front-end: labelled DAG -> PACE DAG -> the reference's dense feature dict.

Mirrors (same names, argument meaning and error behaviour):
  * ``LabeledDag.from_dict_to_graph``            src/toolkit/labeled.py:132-154   (parquet row codec)
  * ``PaceVaeV3.from_labeled_graph_to_pace_graph`` src/encoders/pace.py:1250-1288
  * ``PaceVaeV3.generate_mask``                   src/encoders/pace.py:1307-1343
  * ``PaceVaeV3.prepare_features``                src/encoders/pace.py:1345-1478
  * ``pace_collate_fn``                           experiments/03_synthetic_12/main.py:75-92

The reference builds these with igraph (absent here); this module needs only numpy.  A graph may be given as
  * a :class:`LabeledGraph` (labels + edge list, vertices already in the row codec's order),
  * a parquet-row ``dict`` (``l{i}`` / ``e{i}`` columns), or
  * any igraph-like object exposing ``vcount()``, ``get_edgelist()`` and ``vs[label_key]``.
Transitive closure uses one N-bit integer per vertex (Warshall on bit rows) instead of repeated dense matmuls.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

LABEL_KEY = "type"           # src/toolkit/labeled.py:9
POSITION_KEY = "position"    # src/encoders/pace.py:14


@dataclass
class LabeledGraph:
    """Minimal stand-in for the igraph.Graph the reference passes around (labelled DAG on n vertices)."""
    labels: List[int]
    edges: List[Tuple[int, int]] = field(default_factory=list)

    def vcount(self) -> int:
        return len(self.labels)

    def get_edgelist(self) -> List[Tuple[int, int]]:
        return list(self.edges)

    def copy(self) -> "LabeledGraph":
        return LabeledGraph(list(self.labels), list(self.edges))


class LabeledDag:
    """Row codec of the reference's toolkit (src/toolkit/labeled.py:13-154), igraph-free."""

    def __init__(self, num_vertices: int, label_cardinality: int, dict_label_prefix: str = "l",
                 dict_edges_prefix: str = "e", validation: bool = True):
        if num_vertices <= 0:
            raise ValueError("`num_vertices` must be greater than 0, got {}".format(num_vertices))
        if label_cardinality <= 0:
            raise ValueError("`label_cardinality` must be greater than 0, got {}".format(label_cardinality))
        self.num_vertices = num_vertices
        self.label_cardinality = label_cardinality
        self.dict_label_prefix = dict_label_prefix
        self.dict_edges_prefix = dict_edges_prefix
        self.validation = validation

    def is_valid_dict(self, pydict: Dict, quiet: bool = True) -> bool:
        try:
            if not isinstance(pydict, dict):
                raise ValueError("pydict must be a dict")
            n = self.num_vertices
            nl = len([k for k in pydict if k.startswith(self.dict_label_prefix)])
            ne = len([k for k in pydict if k.startswith(self.dict_edges_prefix)])
            if nl != n:
                raise AssertionError(f"Expected {n} label fields, got instead {nl}")
            if ne != n:
                raise AssertionError(f"Expected {n} edges fields, got instead {ne}")
            for v in range(n):
                lk, ek = f"{self.dict_label_prefix}{v}", f"{self.dict_edges_prefix}{v}"
                if lk not in pydict:
                    raise ValueError(f"{lk} expected to be in pydict")
                if not (0 <= pydict[lk] < self.label_cardinality):
                    raise AssertionError(f"Label of vertex '{v}' expected to be in range 0 ... "
                                         f"{self.label_cardinality - 1}, got instead {pydict[lk]}")
                if ek not in pydict:
                    raise ValueError(f"'{ek}' expected to be in pydict")
                if len(pydict[ek]) != v:
                    raise ValueError(f"{v} elements expected to be in '{ek}'")
        except (ValueError, AssertionError):
            if quiet:
                return False
            raise
        return True

    def from_dict_to_graph(self, pydict: Dict, validation: Optional[bool] = None) -> LabeledGraph:
        if validation or (validation is None and self.validation):
            self.is_valid_dict(pydict, quiet=False)
        labels, edges = [], []
        for v in range(self.num_vertices):
            labels.append(int(pydict[f"{self.dict_label_prefix}{v}"]))
            conn = pydict[f"{self.dict_edges_prefix}{v}"]
            edges.extend((u, v) for u in range(v) if int(conn[u]) == 1)
        return LabeledGraph(labels, edges)

    def is_valid_graph(self, graph, quiet: bool = True) -> bool:
        """src/toolkit/labeled.py:186-218: a DAG on num_vertices vertices with labels in range(label_cardinality)."""
        def fail(msg):
            if quiet:
                return False
            raise AssertionError(msg)
        if graph is None:
            return fail("graph is missing")
        labels, edges = _as_labels_edges(graph)
        n = len(labels)
        indeg = [0] * n
        out = [[] for _ in range(n)]
        for u, v in edges:
            if not (0 <= u < n and 0 <= v < n):
                return fail("graph is not a dag")
            out[u].append(v)
            indeg[v] += 1
        stack = [v for v in range(n) if indeg[v] == 0]
        seen = 0
        while stack:
            u = stack.pop()
            seen += 1
            for v in out[u]:
                indeg[v] -= 1
                if indeg[v] == 0:
                    stack.append(v)
        if seen != n:
            return fail("graph is not a dag")
        if n != self.num_vertices:
            return fail(f"Graph expected to has {self.num_vertices} got instead {n}")
        for i, label in enumerate(labels):
            if not (0 <= label < self.label_cardinality):
                return fail(f"Label of vertex '{i}' expected to be in range 0 ... {self.label_cardinality - 1}, "
                            f"got instead {label}")
        return True

    def graph_equals(self, graph1, graph2, attributes_match: bool = True) -> bool:
        """src/toolkit/labeled.py:238-260: label-preserving isomorphism.  Distinct labels (asia, sachs, card >= n data
        sets) make the vertex correspondence unique, so this is an edge-set comparison after relabelling; graphs with
        repeated labels go through networkx's VF2 like the reference (networkx is optional here)."""
        if graph1 is None or graph2 is None:
            return False
        l1, e1 = _as_labels_edges(graph1)
        l2, e2 = _as_labels_edges(graph2)
        if len(l1) != len(l2) or len(e1) != len(e2):
            return False
        if attributes_match and sorted(l1) != sorted(l2):
            return False
        if attributes_match and len(set(l1)) == len(l1):
            at = {lab: v for v, lab in enumerate(l2)}
            return sorted((at[l1[u]], at[l1[v]]) for u, v in e1) == sorted(e2)
        import networkx as nx
        g1, g2 = nx.DiGraph(), nx.DiGraph()
        for g, labels, edges in ((g1, l1, e1), (g2, l2, e2)):
            g.add_nodes_from((v, {LABEL_KEY: lab}) for v, lab in enumerate(labels))
            g.add_edges_from(edges)
        match = (lambda a, b: a[LABEL_KEY] == b[LABEL_KEY]) if attributes_match else None
        return nx.is_isomorphic(g1, g2, node_match=match)

    def from_graph_to_dict_writable(self, graph: LabeledGraph) -> Dict:
        es = set(graph.get_edgelist())
        out = {f"{self.dict_label_prefix}{v}": int(graph.labels[v]) for v in range(self.num_vertices)}
        for v in range(self.num_vertices):
            out[f"{self.dict_edges_prefix}{v}"] = "".join("1" if (u, v) in es else "0" for u in range(v))
        return out


def _as_labels_edges(graph, label_key: str = LABEL_KEY) -> Tuple[List[int], List[Tuple[int, int]]]:
    if isinstance(graph, LabeledGraph):
        return graph.labels, graph.edges
    if isinstance(graph, tuple) and len(graph) == 2:
        return list(graph[0]), list(graph[1])
    if hasattr(graph, "vcount") and hasattr(graph, "get_edgelist"):      # igraph.Graph
        return [int(x) for x in graph.vs[label_key]], [tuple(e) for e in graph.get_edgelist()]
    raise TypeError(f"unsupported graph object of type {type(graph).__name__}")


def pace_arrays(labels: Sequence[int], edges: Iterable[Tuple[int, int]], n_tokens: int,
                label_input: int = 0, label_output: int = 1, label_start: int = 2):
    """PACE wrapping (pace.py:1250-1288) as arrays: (labels[N], child bit-rows[N], positions[N]).

    v0 = start, v1 = input, v_{N-1} = output, user vertex k -> k+2 with label+3; sources hang off the input vertex,
    sinks feed the output vertex.  positions[v] = order[v] where ``order`` is the FIFO-Kahn topological order with
    ascending-id tie-breaks — the list is assigned BY VERTEX INDEX exactly as pace.py:1286 does."""
    n = len(labels)
    N = n_tokens
    assert N - 3 == n, f"Expected {N - 3}, got instead {n}"          # pace.py:1251
    out_id = N - 1
    pl = np.empty(N, np.int64)
    pl[0], pl[1], pl[out_id] = label_start, label_input, label_output
    pl[2:2 + n] = np.asarray(labels, np.int64) + 3
    child = [0] * N                     # child[u] bit v  <=>  edge u -> v
    child[0] = 1 << 1
    has_pred = [False] * n
    for u, v in edges:
        child[u + 2] |= 1 << (v + 2)
        has_pred[v] = True
    for v in range(n):
        if not has_pred[v]:
            child[1] |= 1 << (v + 2)
    for v in range(N - 1):
        if child[v] == 0:
            child[v] = 1 << out_id
    # FIFO Kahn, ascending ids (igraph Graph.topological_sorting semantics)
    indeg = [0] * N
    for u in range(N):
        c = child[u]
        while c:
            low = c & -c
            indeg[low.bit_length() - 1] += 1
            c ^= low
    queue = [v for v in range(N) if indeg[v] == 0]
    head = 0
    while head < len(queue):
        u = queue[head]
        head += 1
        c = child[u]
        while c:
            low = c & -c
            v = low.bit_length() - 1
            indeg[v] -= 1
            if indeg[v] == 0:
                queue.append(v)
            c ^= low
    if len(queue) != N:
        raise ValueError("graph is not a dag")
    return pl, child, np.asarray(queue, np.int64)


def ancestor_closure(child: List[int]) -> List[int]:
    """reach[a] bit b <=> path a -> ... -> b of length >= 1 or a == b (pace.py:1307-1338), Warshall on bit rows."""
    N = len(child)
    reach = [child[a] | (1 << a) for a in range(N)]
    for k in range(N):
        bk = 1 << k
        rk = reach[k]
        for a in range(N):
            if reach[a] & bk:
                reach[a] |= rk
    return reach


def prepare_features(graphs, n_tokens: int, n_classes: int, num_heads: int = 8, label_key: str = LABEL_KEY,
                     label_input: int = 0, label_output: int = 1, label_start: int = 2,
                     fixed_memory_len: Optional[int] = None, device=None) -> Dict:
    """pace.py:1345-1478: the dense feature dict with the reference's keys, shapes and dtypes."""
    B, N, C = len(graphs), n_tokens, n_classes
    lab = np.zeros((B, N, C), np.float32)
    pos = np.zeros((B, N, N), np.float32)
    adj = np.zeros((B, N, N), np.float32)
    reach = np.zeros((B, N, N), bool)
    vertex_labels = []
    ar = np.arange(N)
    bits = (1 << ar).astype(object)
    for b, g in enumerate(graphs):
        labels, edges = _as_labels_edges(g, label_key)
        pl, child, order = pace_arrays(labels, edges, N, label_input, label_output, label_start)
        if pl.max() >= C:
            raise IndexError(f"vertex label {int(pl.max()) - 3} out of range for cardinality {C - 3}")
        lab[b, ar, pl] = 1.0
        pos[b, ar, order] = 1.0
        rc = ancestor_closure(child)
        for a in range(N):
            adj[b, a] = [(child[a] >> j) & 1 for j in range(N)]
            reach[b, a] = [(rc[a] >> j) & 1 for j in range(N)]
        vertex_labels.append([int(x) for x in pl[1:]])
    non_reach = ~reach
    source = np.repeat(non_reach[:, None, 1:, 1:], num_heads, 1).reshape(B * num_heads, N - 1, N - 1)
    target = np.repeat(non_reach.transpose(0, 2, 1)[:, None], num_heads, 1).reshape(B * num_heads, N, N)
    memory = np.zeros((B * num_heads, N, N - 1), bool)
    if fixed_memory_len is not None and fixed_memory_len < N - 1:
        memory[:, :, fixed_memory_len:] = True                                  # pace.py:1446-1453
    out = {
        "vertex_label_features": torch.from_numpy(lab),
        "vertex_position_features": torch.from_numpy(pos),
        "adjacency_matrices": torch.from_numpy(adj),
        "source_masks": torch.from_numpy(np.ascontiguousarray(source)),
        "target_masks": torch.from_numpy(np.ascontiguousarray(target)),
        "memory_masks": torch.from_numpy(memory),
        "num_vertices": [N] * B,
        "vertex_labels": vertex_labels,
    }
    if device is not None:
        for k, v in out.items():
            if torch.is_tensor(v):
                out[k] = v.to(device)
    return out


def pace_collate_fn(data: Sequence[Dict]) -> Dict:
    """experiments/03_synthetic_12/main.py:75-92 (identical behaviour)."""
    keys_to_cat = ["vertex_label_features", "vertex_position_features", "adjacency_matrices", "source_masks",
                   "target_masks", "memory_masks"]
    batch = {key: torch.cat([el[key] for el in data], dim=0) for key in keys_to_cat}
    batch["num_vertices"] = [el["num_vertices"][0] for el in data]
    batch["vertex_labels"] = [el["vertex_labels"][0] for el in data]
    return batch


def collate_graph_batch(data):
    """src/train_utils.py:39-40."""
    return [g.copy() for g in data]
