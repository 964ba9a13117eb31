"""Diagnostic (run on the GPU box from the repo root): per-parameter error of one fused clip + Adam step against the
reference's golden step, split by the size of the clipped gradient — shows which entries move on rounding noise (the
key biases of the attentions, whose true gradient is exactly zero) and the relative gradient error per tensor."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load_golden
from tests.test_gpu_module import build_model, feats_for, DEV
from dags_vae_search_amd import optim as dopt
for name in ["n12c12", "asia_rand"]:
    cfg, params, graphs, z = load_golden(name)
    eps = torch.from_numpy(z["train0/eps"])
    gn = np.sqrt(sum(float((z[k].astype(np.float64) ** 2).sum()) for k in z.files if k.startswith("train0/grad/")))
    coef = min(1.0, 1.0 / (gn + 1e-6))
    m = build_model(cfg, params, dropout=0.0).train()
    f = feats_for(m, graphs)
    fo = dopt.Adam(m.parameters(), lr=1e-4).attach(m)
    m.loss_and_grad(f, eps=eps.to(DEV))
    g = {k: p.grad.clone().cpu().numpy() for k, p in m.named_parameters()}
    fo.step(max_grad_norm=1.0)
    rows = []
    for k, p in m.state_dict().items():
        err = np.abs(p.cpu().numpy() - z["step/param/" + k])
        gg = np.abs(z["train0/grad/" + k]) * coef
        gerr = np.abs(g[k] - z["train0/grad/" + k]).max() / max(np.abs(z["train0/grad/" + k]).max(), 1e-30)
        b7 = err[gg > 1e-7].max() if (gg > 1e-7).any() else 0
        b6 = err[gg > 1e-6].max() if (gg > 1e-6).any() else 0
        rows.append((err.max(), b7, b6, gerr, np.abs(z["train0/grad/" + k]).max(), k))
    rows.sort(reverse=True)
    print(name, "gn", gn)
    for r in rows[:14]:
        print("  maxerr %.2e  err(|g|>1e-7) %.2e  err(|g|>1e-6) %.2e  relgraderr %.1e  max|g| %.1e  %s" % r)
