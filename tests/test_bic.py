"""BIC scorer: oracle vs the reference's known answers; HIP kernel (host emulator) vs the oracle."""
import ctypes

import numpy as np
import pytest

from oracle import bic as obic
from oracle import features as ofeat
from tests.helpers import graphs_from, load_npz


def _data(name):
    z = load_npz(f"bn_{name}_data.npz")
    d = z["data"]
    return d, (d.max(0) + 1).astype(np.uint8)


def test_oracle_bic_known_answer_and_predictor_targets():
    data, card = _data("asia")
    # tests/problem/bn/test_bnlearn.py:22-55
    e = {1: [1], 2: [0, 0], 3: [0, 0, 0], 4: [0, 1, 0, 0], 5: [1, 1, 0, 0, 0], 6: [0, 1, 0, 0, 1, 0], 7: [0, 0, 0, 1, 1, 1, 0]}
    edges = [(u, v) for v, bits in e.items() for u, b in enumerate(bits) if b]
    assert obic.bic(data, card, list(range(8)), edges) == pytest.approx(-13331.093616667435, abs=1e-5)
    # `target` column of experiments/01_bn_asia/predictor_dataset (written by prepare_predictor_data through Rscript)
    z = load_npz("asia_known_answer.npz")
    for (lab, edges), t in zip(graphs_from(z, 8), z["bic"]):
        assert obic.bic(data, card, lab, edges) == pytest.approx(float(t), abs=1e-8)
    with pytest.raises(AssertionError):
        obic.bic(data, card, [0, 1, 2, 3, 4, 5, 6, 6], [])


def _pack(data):
    S, n = data.shape
    packed = np.zeros((S, (n + 15) // 16), np.uint64)
    for i in range(n):
        packed[:, i // 16] |= data[:, i].astype(np.uint64) << np.uint64(4 * (i % 16))
    return packed


@pytest.mark.parametrize("name,n,count", [("asia", 8, 24), ("sachs", 11, 12)])
def test_emu_bic_kernel_matches_oracle(name, n, count):
    from tests.emu.harness import emu, ptr
    data, card = _data(name)
    graphs = ofeat.synthetic_dags(n, n, count, seed=5)
    if name == "sachs":      # a 10-parent sink: 3^11 cells, the sort path
        graphs.append((list(range(11)), [(u, 10) for u in range(10)]))
        graphs.append((list(range(11))[::-1], [(u, 10) for u in range(1, 10)] + [(0, 1)]))
        count = len(graphs)
    masks = np.zeros((count, n), np.uint64)
    for b, (lab, edges) in enumerate(graphs):
        for u, v in edges:
            masks[b, lab[v]] |= np.uint64(1) << np.uint64(lab[u])
    packed = _pack(data)
    scratch = np.zeros((count, n), np.float64)
    out = np.zeros(count, np.float64)
    status = np.zeros(1, np.int32)
    lib = emu()
    assert lib.dvs_bic_scores(count, n, data.shape[0], ptr(packed), ptr(card), ptr(masks), ptr(scratch), ptr(out),
                              ptr(status), None) == 0
    assert status[0] == 0
    ref = np.array([obic.bic(data, card, lab, edges) for lab, edges in graphs])
    assert np.abs(out - ref).max() < 1e-8
    assert lib.dvs_bic_scores(count, 49, data.shape[0], ptr(packed), ptr(card), ptr(masks), ptr(scratch), ptr(out),
                              ptr(status), None) != 0


@pytest.mark.parametrize("n,count", [(8, 40), (11, 20), (37, 6)])
def test_emu_parent_masks_from_the_row_codec(n, count):
    """dvs_bic_parent_masks (relabelling of bnlearn.py:34-45 on the row codec) == the host construction used above; a DAG
    whose labels are not a permutation raises status bit 5 and gets zero masks."""
    from dags_vae_search_amd.records import encode_graphs
    from tests.emu.harness import emu, ptr
    graphs = ofeat.synthetic_dags(n, n, count, seed=3, density_limit=0.4 if n <= 13 else 0.2)
    cb = encode_graphs(graphs, n)
    lab = np.ascontiguousarray(cb.labels.numpy())
    pr = np.ascontiguousarray(cb.preds.numpy())
    want = np.zeros((count, n), np.uint64)
    for b, (l, edges) in enumerate(graphs):
        for u, v in edges:
            want[b, l[v]] |= np.uint64(1) << np.uint64(l[u])
    got = np.full((count, n), 0xFFFF, np.uint64)
    status = np.zeros(1, np.int32)
    lib = emu()
    assert lib.dvs_bic_parent_masks(count, n, 1 if n > 13 else 0, ptr(lab), ptr(pr), ptr(got), ptr(status), None) == 0
    assert status[0] == 0 and np.array_equal(got, want)
    lab[1, 0] = lab[1, 1]                       # duplicate label: not a permutation
    assert lib.dvs_bic_parent_masks(count, n, 1 if n > 13 else 0, ptr(lab), ptr(pr), ptr(got), ptr(status), None) == 0
    assert status[0] == 32 and not got[1].any() and np.array_equal(got[2:], want[2:])
    assert lib.dvs_bic_parent_masks(count, 20, 0, ptr(lab), ptr(pr), ptr(got), ptr(status), None) != 0     # u16 rows, n > 16
