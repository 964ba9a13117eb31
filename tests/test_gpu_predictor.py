"""GPU: GP predictor mirror (dags_vae_search_amd.predictor.GPRegressionModel -> dvs_gp_predict) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import gp as ogp
from tests.helpers import load_npz

pytestmark = pytest.mark.gpu


def test_gp_predictor_matches_oracle_and_shipped_behaviour():
    from dags_vae_search_amd.predictor import GPRegressionModel
    fix = load_npz("asia_predictor.npz")
    x, y = torch.from_numpy(fix["x"]), torch.from_numpy(fix["y"])
    ntr = int(np.floor(0.8 * len(x)))
    model = GPRegressionModel(x[:ntr], y[:ntr])
    model.load_state_dict({"likelihood.noise_covar.raw_noise": torch.from_numpy(fix["raw_noise"]),
                           "mean_module.raw_constant": torch.from_numpy(fix["raw_constant"]),
                           "base_covar_module.raw_outputscale": torch.from_numpy(fix["raw_outputscale"]),
                           "base_covar_module.base_kernel.raw_lengthscale": torch.from_numpy(fix["raw_lengthscale"]),
                           "covar_module.inducing_points": torch.from_numpy(fix["inducing_points"])}).eval()
    pred = model.predict(x[ntr:]).cpu().numpy()
    h = ogp.hyper(fix)
    alpha = ogp.fit_alpha(fix["x"][:ntr], fix["y"][:ntr], fix["inducing_points"], h)
    ref = ogp.predict_mean(fix["x"][ntr:], fix["inducing_points"], alpha, h)
    assert np.abs(pred - ref).max() < 1e-4 * np.abs(ref).max()          # two fp64 solves of an ill-conditioned system
    assert float(np.abs(pred - fix["y"][ntr:]).mean()) == pytest.approx(686.1, abs=2.0)
    big = model.predict(torch.randn(4096, 32))
    assert big.shape == (4096,) and torch.isfinite(big).all()
    with pytest.raises(AssertionError):
        model.predict(torch.randn(4, 16))
