#!/usr/bin/env python3
"""Test MAE / MAPE of the SGPR predictor against training iterations, from ONE run of the reference's driver shape
(dags_vae_search_amd.predictor.train_predictor = experiments/01_bn_asia/main.py:315-393) on the reference's own 1 408-row
(vector, target) data set (tests/golden/asia_predictor.npz), next to the figures the reference's source carries as comments
(src/predictors/gp.py:95-106).  Parity against gpytorch is unpinned: this is a statistical comparison only.
    gpurun -- 'python tools/gp_curve.py > gpurun_out/r03_gp_curve.json'"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dags_vae_search_amd.predictor import train_predictor  # noqa: E402

REFERENCE_COMMENTS = {100: (3117.620849609375, 0.22864490747451782), 1000: (528.623046875, 0.03854367509484291),
                      2000: (338.1571350097656, 0.02457481063902378), 5000: (204.15782165527344, 0.014772910624742508)}


def main():
    fix = np.load(os.path.join(REPO, "tests", "golden", "asia_predictor.npz"))
    X, y = torch.from_numpy(fix["x"]), torch.from_numpy(fix["y"])
    t0 = time.perf_counter()
    model, rows = train_predictor(X, y, iterations=10000, log=lambda s: print(s, file=sys.stderr))
    dt = time.perf_counter() - t0
    const_mae = float((y[int(0.8 * len(y)):] - y[:int(0.8 * len(y))].mean()).abs().mean())
    for r in rows:
        ref = REFERENCE_COMMENTS.get(r["iterations"])
        r["reference_comment_mae_mape"] = list(ref) if ref else None
    print(json.dumps({"what": "SGPR predictor, test MAE / MAPE vs training iterations (one 10 000-iteration run, 80/20 split of the "
                              "reference's 1 408 rows, 500 inducing points, Adam lr 0.01)",
                      "rows": rows, "seconds": dt, "ms_per_iteration": dt / 10000 * 1e3,
                      "constant_predictor_mae": const_mae,
                      "note": "the reference's comment figures (gp.py:95-106) come from its own runs on a data set and a gpytorch "
                              "version this repository cannot reproduce (gpytorch absent): parity unpinned, shape of the curve only"}))


if __name__ == "__main__":
    main()
