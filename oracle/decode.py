"""ORACLE (test infrastructure, never shipped): CPU restatement of ``PaceVaeV3.decode`` (reference
src/encoders/pace.py:1666-1749) with its helpers ``prepare_features_v2`` (1480-1611), ``generate_mask`` (1307-1343),
``compute_graph_positions`` (1245-1248) and ``from_pace_graph_to_labeled_graph`` (1290-1305), igraph-free.

PARITY UNPINNED against a *run* of the reference: its decode needs a real igraph (absent in the build container) and
draws from numpy's / torch's global generators (``np.random.choice`` 1712, ``torch.rand_like`` 1728), which no other
implementation can reproduce.  What pins this file: the decoder stack it calls is the pinned one of pace_oracle.py;
the feature building for the grown graph follows the same mask/position functions as oracle/features.py (pinned by
the 254-row known answer); and the reference's published reconstruction accuracy of the shipped asia checkpoint
(experiments/01_bn_asia/main.py:560: valid 1.000, exact 0.935) is reproduced statistically in tests.  Randomness is
INJECTED: ``uniforms[b, idx, 0]`` replaces the uniform behind ``np.random.choice`` at step ``idx`` (inverse CDF,
``cdf.searchsorted(u, side='right')`` — numpy's own algorithm), ``uniforms[b, idx, 1 + vi]`` replaces
``random_score[b]`` for edge candidate ``vi``.

Quirks kept on purpose (each is what the reference does):
  * the grown graph never gets the start->input edge 0->1 (decode adds edges only from vertices vi+1 >= 1);
  * at the last step the vertex is labelled ``output`` but is hooked to the loose ends only if the SAMPLED type was
    ``output`` (1738-1743); otherwise its in-edges are sampled like any other vertex;
  * a graph that samples ``output`` early stops growing; the reference then fails in
    from_pace_graph_to_labeled_graph (IndexError on the missing vertices) — here such graphs are returned as ``None``
    by ``to_labeled`` and the caller decides;
  * padding tokens of a short graph: label ``output``, position max+1, attend each other only (1540-1583).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import features as ofeat
from . import pace_oracle as po

LABEL_INPUT, LABEL_OUTPUT, LABEL_START = 0, 1, 2


class GrownGraph:
    """A PACE graph under construction: labels per vertex and directed edges (u -> v)."""

    def __init__(self):
        self.labels: List[int] = [LABEL_START, LABEL_INPUT]
        self.edges: List[Tuple[int, int]] = []
        self.finished = False

    @property
    def nv(self) -> int:
        return len(self.labels)

    def positions(self) -> List[int]:
        return ofeat.topological_order_fifo(self.nv, self.edges)     # positions[v] = order[v] (pace.py:1286 quirk)

    def out_degree_zero(self) -> List[int]:
        has_out = {u for u, _ in self.edges}
        return [v for v in range(self.nv) if v not in has_out]


def features_v2(graphs: Sequence[GrownGraph], N: int, C: int, heads: int = 8):
    """prepare_features_v2 (pace.py:1480-1611) for partially grown graphs, as torch tensors."""
    B = len(graphs)
    lab = np.zeros((B, N, C), np.float32)
    pos = np.zeros((B, N, N), np.float32)
    adj = np.zeros((B, N, N), np.float32)
    tmask = np.ones((B, N, N), bool)
    for b, g in enumerate(graphs):
        nv = g.nv
        p = g.positions()
        labels = list(g.labels) + [LABEL_OUTPUT] * (N - nv)
        poss = list(p) + [max(p) + 1] * (N - nv)
        lab[b, np.arange(N), labels] = 1.0
        pos[b, np.arange(N), poss] = 1.0
        a = np.zeros((nv, nv), np.float32)
        for u, v in g.edges:
            a[u, v] = 1.0
        adj[b, :nv, :nv] = a
        reach = ofeat.reachability(a)                    # reach[a][b]: a reaches b or a == b
        tmask[b, :nv, :nv] = ~reach
        tmask[b, nv:, nv:] = False
    tmask = np.transpose(tmask, (0, 2, 1))               # .transpose(1, 2) at pace.py:1606
    tm = np.repeat(tmask[:, None], heads, axis=1).reshape(B * heads, N, N)
    return torch.from_numpy(lab), torch.from_numpy(pos), torch.from_numpy(adj), torch.from_numpy(tm)


def decode(P, cfg: po.PaceConfig, z: torch.Tensor, uniforms: np.ndarray) -> List[GrownGraph]:
    """pace.py:1666-1749 with injected uniforms [B, N, N]; returns the grown PACE graphs."""
    N, C = cfg.N, cfg.C
    B = z.shape[0]
    with torch.no_grad():
        memory = F.linear(z, P["fc3.weight"], P["fc3.bias"]).reshape(-1, N, cfg.d_model).transpose(0, 1)
        graphs = [GrownGraph() for _ in range(B)]
        for idx in range(2, N):
            lab, pos, adj, tm = features_v2(graphs, N, C, cfg.heads)
            x = po._embed(P, cfg, lab, pos, adj, False)
            out = po._decoder(P, cfg, x.transpose(0, 1), memory, tm, False).transpose(0, 1)
            hid = out[:, idx - 1, :]
            t1 = torch.relu(F.linear(hid, P["add_node.0.weight"], P["add_node.0.bias"]))
            probs = torch.softmax(F.linear(t1, P["add_node.2.weight"], P["add_node.2.bias"]), 1).numpy()
            new_types = []
            for b in range(B):
                cdf = probs[b].astype(np.float64).cumsum()
                cdf /= cdf[-1]
                new_types.append(int(min(cdf.searchsorted(float(uniforms[b, idx, 0]), side="right"), C - 1)))
            pair = torch.cat([torch.stack([hid] * (idx - 1), 1), out[:, :idx - 1, :]], -1)
            e = torch.relu(F.linear(pair, P["add_edge.0.weight"], P["add_edge.0.bias"]))
            score = torch.sigmoid(F.linear(e, P["add_edge.2.weight"], P["add_edge.2.bias"])).numpy()[:, :, 0]
            for b, g in enumerate(graphs):
                if not g.finished:
                    g.labels.append(new_types[b] if idx < N - 1 else LABEL_OUTPUT)
            for vi in range(idx - 2, -1, -1):
                for b, g in enumerate(graphs):
                    if g.finished:
                        continue
                    last = g.nv - 1
                    if new_types[b] == LABEL_OUTPUT:
                        for v in g.out_degree_zero():
                            if v != last:
                                g.edges.append((v, last))
                        g.finished = True
                        continue
                    if float(uniforms[b, idx, 1 + vi]) < float(score[b, vi]):
                        g.edges.append((vi + 1, last))
    return graphs


def to_labeled(g: GrownGraph, N: int) -> Optional[Tuple[List[int], List[Tuple[int, int]]]]:
    """from_pace_graph_to_labeled_graph (pace.py:1290-1305): user vertex k = PACE vertex k + 2, label - 3; the edges
    INTO PACE vertex 2 are skipped (``vertex_id == graph_label_start`` compares a vertex id with a label, 1298).
    None where the reference raises (graph shorter than N vertices)."""
    if g.nv < N:
        return None
    labels = [g.labels[v] - 3 for v in range(2, N - 1)]
    es = set(g.edges)
    edges = []
    for v in range(2, N - 1):
        if v == LABEL_START:
            continue
        for u in range(2, v + 2):
            if (u, v) in es:
                edges.append((u - 2, v - 2))
    return labels, edges
