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
    losses = torch.zeros(5, device=dev)
    mu = torch.zeros(B, 32, device=dev)
    lv = torch.zeros(B, 32, device=dev)
    e = None if eps is None else torch.as_tensor(eps, dtype=torch.float32, device=dev).contiguous()
    eng.loss_forward(shape, flat, e, losses, mu, lv)
    torch.cuda.synchronize()
    return eng, flat, shape, losses.cpu().numpy(), mu.cpu().numpy(), lv.cpu().numpy(), f_np


def smoke_check():
    """One tiny train-mode forward+backward of the hot path (8 DAGs, n=12) on cuda:0, checked against the oracle."""
    from dags_vae_search_amd import LabeledGraph, PaceVaeV3
    cfg, params, graphs, z = load_golden("n12c12")
    graphs = graphs[:8]
    model = PaceVaeV3(12, 12, 32, 8, 3, 64, 32, 32, 0.0)
    model.load_state_dict(params)
    model = model.to("cuda:0").train()
    eps = torch.from_numpy(z["train0/eps"][:8])
    total, recon, kld = model.loss_direct(model.prepare_features([LabeledGraph(l, e) for l, e in graphs]), eps=eps)
    total.backward()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    cfg0 = po.PaceConfig(n=12, card=12, dropout=0.0)
    t, r, k = po.loss_direct(P, cfg0, ofeat.to_torch(ofeat.dense_features(graphs, 12)), training=True, eps=eps)
    t.backward()
    assert rel(total.item(), t.detach()) < 1e-4, (total.item(), float(t))
    for name, p in model.named_parameters():
        ref = P[name].grad
        assert (p.grad.cpu() - ref).abs().max() <= 2e-4 * max(ref.abs().max().item(), 1e-3), name      # the parity tests' bound
