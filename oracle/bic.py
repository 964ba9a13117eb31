"""ORACLE (test infrastructure, never shipped): numpy restatement of the reference's BIC scorer.

The reference scores a labelled DAG by shelling out to R: BNLearnWrapper.score (src/problem/bn/bnlearn.py:27-61)
relabels vertex v -> dataset variable labels[v], prints the adjacency matrix and runs
``bnlearn::score(net, dataset, type = "bic")`` (src/problem/bn/bnlearn_scripts/bnlearn_score.R:25-39).  R and bnlearn
are absent in the build container; bnlearn's discrete BIC is the published decomposable score
    BIC = sum_v [ sum_{j,k} N_vjk log(N_vjk / N_vj)  -  (log S / 2) (r_v - 1) q_v ]
(N_vjk: samples with variable v in state k and its parents in configuration j; r_v levels; q_v = product of the
parents' level counts, unobserved configurations included; S samples).  PINNED by the reference's own fixtures:
tests/problem/bn/test_bnlearn.py:22-55 (asia DAG -> -13331.093616667435, reproduced to the last digit) and the `target`
column of experiments/01_bn_asia/predictor_dataset (254 rows kept in tests/golden/asia_known_answer.npz, max abs error
4e-12).  Data: data/bn_asia/target.csv, data/bn_sachs/target.csv (level-coded copies in tests/golden/bn_*_data.npz).
"""
from typing import Sequence, Tuple

import numpy as np


def local_score(data: np.ndarray, card: np.ndarray, v: int, parents: Sequence[int]) -> float:
    S = data.shape[0]
    key = np.zeros(S, np.int64)
    q = 1
    for p in sorted(parents):
        key = key * int(card[p]) + data[:, p]
        q *= int(card[p])
    r = int(card[v])
    njk = np.bincount(key * r + data[:, v], minlength=q * r).reshape(q, r).astype(np.float64)
    nj = njk.sum(1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        ll = np.where(njk > 0, njk * np.log(njk / nj), 0.0).sum()
    return float(ll - 0.5 * np.log(S) * (r - 1) * q)


def parent_sets(labels: Sequence[int], edges: Sequence[Tuple[int, int]], n: int):
    """bnlearn.py:40-45: vertex v stands for dataset variable labels[v]."""
    assert len(labels) == n and sorted(labels) == list(range(n)), f"Expected graph labels from 0 to {n - 1}, but got {labels}"
    par = [[] for _ in range(n)]
    for u, v in edges:
        par[labels[v]].append(labels[u])
    return par


def bic(data: np.ndarray, card: np.ndarray, labels: Sequence[int], edges: Sequence[Tuple[int, int]]) -> float:
    n = data.shape[1]
    par = parent_sets(labels, edges, n)
    return sum(local_score(data, card, v, par[v]) for v in range(n))
