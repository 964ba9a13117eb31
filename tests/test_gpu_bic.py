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


def test_predictor_data_pipeline_reproduces_the_references_1408_rows(tmp_path):
    """the mirror of prepare_predictor_data (experiments/01_bn_asia/main.py:268-303; utils.py:15-59) —
    encode -> BIC -> (vector, target) rows, all on the device — against BOTH columns of the data set the reference itself
    wrote with checkpoint 110 and Rscript (experiments/01_bn_asia/predictor_dataset, 1 408 rows = 22 batches of 64; graphs
    re-associated by tests/golden/gen_predictor_graphs.py)."""
    import pyarrow.parquet as pq
    from dags_vae_search_amd import BNLearnWrapper, LabeledGraph, PaceVaeV3, prepare_predictor_data
    fix = load_npz("asia_predictor.npz")
    ck = load_npz("asia_ckpt110.npz")
    graphs = [LabeledGraph(list(l), list(e)) for l, e in graphs_from(load_npz("asia_predictor_graphs.npz"), 8)]
    assert len(graphs) == 1408
    model = PaceVaeV3(8, 8, 32, 8, 3, 64, 32, 32, 0.15)
    model.load_state_dict({k: torch.from_numpy(ck[k]) for k in ck.files})
    model = model.to("cuda:0")
    ev = BNLearnWrapper("asia", "bic", data=load_npz("bn_asia_data.npz")["data"])
    vec, tgt = prepare_predictor_data(model, graphs, ev, batch_size=64, output_dir=str(tmp_path / "predictor_dataset"))
    assert vec.is_cuda and tgt.is_cuda and vec.shape == (1408, 32) and tgt.dtype == torch.float64
    assert np.abs(vec.cpu().numpy() - fix["x"]).max() < 1e-5           # reference-written vectors (fp32)
    assert np.abs(tgt.cpu().numpy() - fix["y"]).max() < 1e-6           # reference-written targets (Rscript bnlearn)
    parts = list((tmp_path / "predictor_dataset").glob("part-*.parquet"))
    assert len(parts) == 22
    t = pq.read_table(tmp_path / "predictor_dataset" / "part-3.parquet").to_pylist()
    assert len(t) == 64 and set(t[0]) == {"vector", "target"}
    assert np.allclose(np.asarray(t[5]["vector"], np.float32), fix["x"][3 * 64 + 5], atol=1e-5)
    assert t[5]["target"] == pytest.approx(float(fix["y"][3 * 64 + 5]), abs=1e-6)
    # a plain callable evaluator (the reference's signature) gives the same targets
    _, tgt2 = prepare_predictor_data(model, graphs[:64], ev.score, batch_size=64)
    assert np.abs(tgt2.cpu().numpy() - fix["y"][:64]).max() < 1e-6
    with pytest.raises(AssertionError):
        from dags_vae_search_amd import encode_graphs
        bad = encode_graphs(graphs[:4], 8)
        bad.labels[0, 0] = bad.labels[0, 1]
        ev.score_compact(bad.to("cuda:0"))
