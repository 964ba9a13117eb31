"""GPU: batched BIC scorer (dvs_bic_scores) through the BNLearnWrapper mirror vs the reference's known answers."""
import time

import numpy as np
import pytest
import torch

from oracle import bic as obic
from tests.helpers import graphs_from, load_npz

pytestmark = pytest.mark.gpu


def test_bic_known_answer_and_targets_on_gpu():
    from dags_vae_search_amd import BNLearnWrapper, LabeledGraph
    data = load_npz("bn_asia_data.npz")["data"]
    ev = BNLearnWrapper("asia", "bic", data=data)
    e = {1: [1], 2: [0, 0], 3: [0, 0, 0], 4: [0, 1, 0, 0], 5: [1, 1, 0, 0, 0], 6: [0, 1, 0, 0, 1, 0], 7: [0, 0, 0, 1, 1, 1, 0]}
    edges = [(u, v) for v, bits in e.items() for u, b in enumerate(bits) if b]
    assert ev.score(LabeledGraph(list(range(8)), edges)) == pytest.approx(-13331.093616667435, abs=1e-5)   # test_bnlearn.py:55
    z = load_npz("asia_known_answer.npz")
    graphs = [LabeledGraph(list(l), list(e2)) for l, e2 in graphs_from(z, 8)]
    got = np.asarray(ev.score_batch(graphs))
    assert np.abs(got - z["bic"]).max() < 1e-6                      # 254 targets written by the reference through Rscript
    with pytest.raises(AssertionError):
        ev.score(LabeledGraph([0, 1, 2, 3, 4, 5, 6, 6], []))


@pytest.mark.parametrize("name,n", [("asia", 8), ("sachs", 11)])
def test_bic_batch_4096_matches_oracle_sample(name, n):
    from dags_vae_search_amd import BNLearnWrapper
    from dags_vae_search_amd.synthetic import synthetic_dags
    data = load_npz(f"bn_{name}_data.npz")["data"]
    card = (data.max(0) + 1).astype(np.uint8)
    ev = BNLearnWrapper(name, "bic", data=data)
    graphs = synthetic_dags(n, n, 4096, seed=9)
    masks = torch.from_numpy(ev._parent_masks(graphs, "type").view(np.int64)).cuda()
    out = ev.score_masks(masks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = ev.score_masks(masks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"\nBIC {name}: {4096 / dt:.0f} structures/s ({dt * 1e3:.2f} ms per 4096)")
    got = out.cpu().numpy()
    for b in range(0, 4096, 64):
        g = graphs[b]
        assert got[b] == pytest.approx(obic.bic(data, card, g.labels, g.edges), abs=1e-7)
    again = ev.score_masks(masks).cpu().numpy()
    assert np.array_equal(got, again)                               # integer counts + fixed-order fp64 sums


def test_bic_sort_path_large_parent_sets():
    """sachs variables with 9 and 10 ternary parents (3^10, 3^11 cells > the LDS table): the sorted-samples counting path
    against the oracle, plus graphs that mix both paths."""
    from dags_vae_search_amd import BNLearnWrapper, LabeledGraph
    data = load_npz("bn_sachs_data.npz")["data"]
    card = (data.max(0) + 1).astype(np.uint8)
    ev = BNLearnWrapper("sachs", "bic", data=data)
    graphs = [LabeledGraph(list(range(11)), [(u, 10) for u in range(10)]),
              LabeledGraph(list(range(11))[::-1], [(u, 10) for u in range(1, 10)] + [(0, 1)]),
              LabeledGraph([3, 1, 4, 0, 5, 9, 2, 6, 8, 7, 10], [(u, v) for v in range(11) for u in range(v)])]   # complete DAG
    got = ev.score_batch(graphs)
    for g, val in zip(graphs, got):
        assert val == pytest.approx(obic.bic(data, card, g.labels, g.edges), abs=1e-7)
