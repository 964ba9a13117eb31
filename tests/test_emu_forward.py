"""Kernel indexing checks on the host emulator (tests/emu): forward pass vs the oracle, CPU only."""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from tests.emu.harness import EmuModel
from tests.helpers import load_golden, rel


@pytest.mark.parametrize("name,B", [("n12c12", 6), ("asia_rand", 5), ("n12c1", 4)])
def test_emu_forward_eval_matches_oracle(name, B):
    cfg, params, graphs, z = load_golden(name)
    graphs = graphs[:B]
    f_np = ofeat.dense_features(graphs, cfg.card)
    m = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=False)
    assert m.pack(f_np) == 0
    losses, mu, lv = m.forward()
    with torch.no_grad():
        total, recon, kld, aux = po.loss_direct(params, cfg, ofeat.to_torch(f_np), training=False, return_aux=True)
    assert np.abs(mu - aux["mu"].numpy()).max() < 2e-5
    assert np.abs(lv - aux["logvar"].numpy()).max() < 2e-5
    assert rel(losses[2], kld) < 1e-5
    # contract (BASELINE.json): ELBO within 1e-4 relative; asia_rand (trained ckpt, off-distribution graphs) is the
    # ill-conditioned case (~4e-5 from fp32 summation order), the others sit at ~1e-6
    assert rel(losses[1], recon) < 1e-4
    assert rel(losses[0], total) < 1e-4
    assert losses[3] == 0.0
