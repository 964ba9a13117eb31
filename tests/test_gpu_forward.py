"""GPU parity (forward): HIP path through the C ABI vs the oracle and the committed reference goldens."""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from tests.helpers import CONFIGS, load_golden, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CONFIGS))
def test_forward_eval_matches_reference_golden(name):
    from tests.gpu_common import gpu_forward
    cfg, params, graphs, z = load_golden(name)
    eng, flat, shape, losses, mu, lv, f_np = gpu_forward(cfg, params, graphs)
    assert np.abs(mu - z["eval/mu"]).max() < 5e-5
    assert np.abs(lv - z["eval/logvar"]).max() < 5e-5
    assert rel(losses[2], z["eval/kld"]) < 1e-4          # BASELINE.json: ELBO match < 1e-4 relative
    assert abs(float(losses[1]) - float(z["eval/recon"])) < 1e-4 * max(1.0, abs(float(z["eval/recon"])))
    assert rel(losses[0], z["eval/total"]) < 1e-4
    assert losses[3] == 0.0
    pre = eng.activation(len(graphs), 16)[:, :cfg.N].cpu()
    dec = po._ln(params, "decoder.layers.2.norm3", pre).numpy()
    assert np.abs(dec - z["eval/decoder_output"]).max() < 2e-4
