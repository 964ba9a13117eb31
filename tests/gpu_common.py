"""Helpers shared by the -m gpu tests and __graft_entry__.smoke(): run the HIP path on cuda:0 and compare with the oracle."""
import numpy as np
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from tests.helpers import load_golden, rel


def gpu_forward(cfg, params, graphs, training=False, dropout=0.15, eps=None, seed=0):
    from dags_vae_search_amd.engine import PaceEngine
    dev = torch.device("cuda:0")
    eng = PaceEngine(cfg.N, cfg.C)
    f_np = ofeat.dense_features(graphs, cfg.card)
    feats = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in f_np.items()}
    eng.pack(feats)
    flat = eng.flatten(params, dev)
    B = len(graphs)
    shape = eng.shape(B, training=training, dropout=dropout, seed=seed)
    losses = torch.zeros(4, device=dev)
    mu = torch.zeros(B, 32, device=dev)
    lv = torch.zeros(B, 32, device=dev)
    e = None if eps is None else torch.as_tensor(eps, dtype=torch.float32, device=dev).contiguous()
    eng.loss_forward(shape, flat, e, losses, mu, lv)
    torch.cuda.synchronize()
    return eng, flat, shape, losses.cpu().numpy(), mu.cpu().numpy(), lv.cpu().numpy(), f_np


def smoke_check():
    cfg, params, graphs, z = load_golden("n12c12")
    graphs = graphs[:8]
    eng, flat, shape, losses, mu, lv, f_np = gpu_forward(cfg, params, graphs)
    with torch.no_grad():
        total, recon, kld = po.loss_direct(params, cfg, ofeat.to_torch(f_np), training=False)
    assert rel(losses[0], total) < 1e-4, (losses, float(total))
    assert rel(losses[2], kld) < 1e-4
