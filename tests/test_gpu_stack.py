"""Chained launches (k_fwd_stack / k_bwd_stack) against one launch per sublayer (DVS_SPLIT_STACK=1): same kernels' phase
functions, same workgroup -> DAG mapping, so losses, gradients and the parameters after train steps must agree BIT FOR BIT.
The switch is read once per process, hence the two child processes."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys
import torch
sys.path.insert(0, %(repo)r)
from oracle import features as ofeat
from oracle import pace_oracle as po
from dags_vae_search_amd import PaceVaeV3, LabeledGraph, optim as dopt
from dags_vae_search_amd.train import train_batch
NV, NB = %(n)d, %(b)d
cfg = po.PaceConfig(n=NV, card=NV)
params = po.init_params(cfg, seed=5)
m = PaceVaeV3(NV, NV, 32, 8, 3, 64, 32, 32, 0.15)
m.load_state_dict(params)
m = m.to("cuda:0").train()
graphs = ofeat.synthetic_dags(NV, NV, NB, seed=7, density_limit=0.2 if NV > 20 else 0.4)   # n = 12: ragged, 1000 DAGs = 125 workgroups of 8
f = m.prepare_features([LabeledGraph(l, e) for l, e in graphs])
m.seed(11)
losses = m.loss_and_grad(f).clone()
h = hashlib.sha256()
h.update(losses.cpu().numpy().tobytes())
h.update(m.flat_grads.cpu().numpy().tobytes())
g0 = m.flat_grads.double().clone()
opt = dopt.Adam(m.parameters(), lr=1e-3).attach(m)
for _ in range(3):
    loss_value, recon, kld = train_batch(f, m, opt)
h.update(m.flat_params.cpu().numpy().tobytes())
h.update(repr(loss_value).encode())
m.eval()
mu, logvar = m.encode_direct(f)
h.update(mu.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), loss_value)
print("VALUES", float(losses[0]), float(g0.norm()), float(g0.abs().max()), float(loss_value))
"""


def run_child(split: bool, latent_kernels: bool = False, nw: int = 8, n: int = 12, batch: int = 1000):
    env = dict(os.environ)
    env["DVS_SPLIT_STACK"] = "1" if split else "0"
    env["DVS_LATENT_KERNELS"] = "1" if latent_kernels else "0"
    env["DVS_WAVES_PER_WG"] = str(nw)
    out = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO, "n": n, "b": batch}], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][-1]
    vals = [float(x) for x in [ln for ln in out.stdout.splitlines() if ln.startswith("VALUES")][-1].split()[1:]]
    return line, vals


def test_chained_and_split_launches_agree_bitwise():
    """also: the latent block as phases of the encoder chains (default) against its own kernels inside otherwise chained
    launches (DVS_LATENT_KERNELS=1) — the 1000-DAG batch leaves the chains' last MFMA group half empty.  Both workgroup
    widths of the stack kernels (dvs_api.hip: waves_per_wg — 8 waves, and the narrow 4-wave mapping this batch size would
    pick by itself): chained == split bit for bit within a width; across widths the summation order of the weight gradients
    differs, so loss, gradient norm / maximum and the third train step's loss agree to rounding."""
    (chained, v8), (split, _), (own_latent, _) = run_child(False), run_child(True), run_child(False, latent_kernels=True)
    assert chained == split
    assert chained == own_latent
    (chained4, v4), (split4, _) = run_child(False, nw=4), run_child(True, nw=4)
    assert chained4 == split4
    assert abs(v4[0] - v8[0]) <= 1e-6 * abs(v8[0])                  # loss: per-DAG values are identical, the sum order too
    assert abs(v4[1] - v8[1]) <= 1e-5 * v8[1] and abs(v4[2] - v8[2]) <= 1e-4 * v8[2]
    assert abs(v4[3] - v8[3]) <= 1e-4 * abs(v8[3])                  # the loss of the third train step


def test_wide_path_token_local_chains_and_split_launches_agree_bitwise():
    """The tiled path (n = 37: three tiles per DAG) launches its attention sublayers on their own, but the token-local backward
    phases between two of them (a layer's q / k / v projections and the FFN of the layer below) travel as one chained launch
    since round 3: against one launch per phase (DVS_SPLIT_STACK=1) losses, gradients, three train steps and the encoder means
    must agree bit for bit.  600 DAGs: ragged against the 256 persistent workgroups (two DAGs in flight per workgroup in the
    attention backward: first pass fills, last pass has nothing to fill)."""
    (chained, _), (split, _) = run_child(False, n=37, batch=600), run_child(True, n=37, batch=600)
    assert chained == split
