"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the PACE-VAE train step.

A functional PyTorch-CPU restatement of what the reference computes in
``PaceVaeV3.loss_direct`` -> backward -> clip_grad_norm_ -> Adam.  Written from the reference's
behaviour (not copied); parameters are addressed by the reference's state-dict names so shipped
checkpoints load directly.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this; the product path never does.

Pinned (tests/golden/gen_golden.py) against the reference's own code run in the build container:
eval-mode and dropout-0/injected-eps train-mode (total, recon, kld, mu, logvar, decoder output, all
108 gradients) for asia (ckpt 110), n=12 card=1 (ckpt 78) and a fresh-seed n=12 card=12 model, plus
one train_batch golden (params/Adam state after one step).

Reference being restated (paths relative to /root/reference/src/encoders/pace.py unless noted):
  GnnPositionalEncoding.forward           186-221
  vertex_label_embed                      1181-1184
  TransformerEncoderLayer.forward         45-67    (nn.MultiheadAttention, need_weights path)
  TransformerDecoderLayer.forward         135-154  (cross-attention masked by tgt_mask, line 148)
  encode_direct                           1613-1641
  reparameterize                          1649-1664
  loss_log_likelihood_full_vectorized     1880-1972
  loss_direct                             1974-2035
  train_batch                             experiments/03_synthetic_12/main.py:95-118
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


@dataclass
class PaceConfig:
    n: int                      # user vertices
    card: int                   # label cardinality
    emb: int = 32               # vertices_embedding_size
    heads: int = 8
    layers: int = 3
    d_model: int = 64           # ff_hidden_size (also the FFN width, pace.py:1185)
    latent: int = 32
    fc_hidden: int = 32
    dropout: float = 0.15

    @property
    def N(self) -> int:
        return self.n + 3

    @property
    def C(self) -> int:
        return self.card + 3


def param_shapes(cfg: PaceConfig) -> Dict[str, tuple]:
    """The 108 state-dict entries, in the reference's registration order (pace.py:1176-1207)."""
    N, C, d, e = cfg.N, cfg.C, cfg.d_model, cfg.emb
    s: Dict[str, tuple] = {}
    s["vertex_position_embed.W1"] = (2 * N, 2 * e)
    s["vertex_position_embed.W2"] = (2 * e, e)
    s["vertex_label_embed.0.weight"] = (e, C)
    s["vertex_label_embed.0.bias"] = (e,)

    def attn(prefix):
        s[prefix + ".in_proj_weight"] = (3 * d, d)
        s[prefix + ".in_proj_bias"] = (3 * d,)
        s[prefix + ".out_proj.weight"] = (d, d)
        s[prefix + ".out_proj.bias"] = (d,)

    def ffn_norms(prefix, nn):
        s[prefix + ".linear1.weight"] = (d, d)
        s[prefix + ".linear1.bias"] = (d,)
        s[prefix + ".linear2.weight"] = (d, d)
        s[prefix + ".linear2.bias"] = (d,)
        for k in range(1, nn + 1):
            s[prefix + f".norm{k}.weight"] = (d,)
            s[prefix + f".norm{k}.bias"] = (d,)

    for l in range(cfg.layers):
        attn(f"encoder.layers.{l}.self_attn")
        ffn_norms(f"encoder.layers.{l}", 2)
    s["fc1.weight"] = (cfg.latent, N * d)
    s["fc1.bias"] = (cfg.latent,)
    s["fc2.weight"] = (cfg.latent, N * d)
    s["fc2.bias"] = (cfg.latent,)
    for l in range(cfg.layers):
        attn(f"decoder.layers.{l}.self_attn")
        attn(f"decoder.layers.{l}.multihead_attn")
        ffn_norms(f"decoder.layers.{l}", 3)
    s["add_node.0.weight"] = (cfg.fc_hidden, d)
    s["add_node.0.bias"] = (cfg.fc_hidden,)
    s["add_node.2.weight"] = (C, cfg.fc_hidden)
    s["add_node.2.bias"] = (C,)
    s["add_edge.0.weight"] = (d, 2 * d)
    s["add_edge.0.bias"] = (d,)
    s["add_edge.2.weight"] = (1, d)
    s["add_edge.2.bias"] = (1,)
    s["fc3.weight"] = (N * d, cfg.latent)
    s["fc3.bias"] = (N * d,)
    return s


def init_params(cfg: PaceConfig, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Same distributions as the reference's constructors (xavier-uniform gain 1.414 for W1/W2,
    pace.py:196-197; torch defaults for Linear / MultiheadAttention / LayerNorm).  NOT draw-for-draw
    identical to the reference's init order; goldens carry their own parameters."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("W1") or name.endswith("W2"):
            bound = 1.414 * math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif ".norm" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith("in_proj_weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            t = torch.zeros(shape)
        elif name.endswith("weight"):
            bound = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:  # Linear bias: U(-1/sqrt(fan_in), ..); fan_in from the matching weight
            w = p[name[:-4] + "weight"]
            bound = 1.0 / math.sqrt(w.shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        p[name] = t.float()
    return p


def _drop(x, p, training, mask=None):
    """nn.Dropout.  `mask` (already scaled by 1/keep, same shape as x) replaces torch's own draw when given:
    the parity tests inject the device's counter-based masks (oracle/rng.py)."""
    if not (training and p > 0.0):
        return x
    if mask is not None:
        return x * mask
    return F.dropout(x, p, training)


def _tile_mask(masks, site, x_bnf):
    """mask for a [B, N, width] tensor at dropout site `site` (None when torch's RNG is used)."""
    if masks is None:
        return None
    B, N, W = x_bnf.shape
    return torch.from_numpy(masks.tile(site, B, N, W))


# dropout sites (dags_vae_search_amd/csrc/dvs_api.hip site_enc/site_dec): 0,1 encoder-side embedding; 2,3 decoder-side
# embedding; 4+4l+{0 attn weights, 1 post-attn, 2 ffn hidden, 3 post-ffn}; 16+6l+{0,1 self; 2,3 cross; 4,5 ffn}
def site_enc(layer, k):
    return 4 + 4 * layer + k


def site_dec(layer, k):
    return 16 + 6 * layer + k


def _embed(P, cfg, lab1h, pos1h, adj, training, masks=None, site=0):
    """pace.py:201-221 + 1181-1184 + cat (1624-1630): -> [B, N, 64]."""
    pe = torch.cat((pos1h, torch.matmul(adj.transpose(1, 2), pos1h)), 2)
    pe = torch.relu(torch.matmul(pe, P["vertex_position_embed.W1"]))
    if cfg.dropout > 0.0001:
        pe = _drop(pe, cfg.dropout, training, _tile_mask(masks, site, pe))
    pe = torch.matmul(pe, P["vertex_position_embed.W2"])
    if cfg.dropout > 0.0001:
        pe = _drop(pe, cfg.dropout, training, _tile_mask(masks, site + 1, pe))
    le = torch.relu(F.linear(lab1h, P["vertex_label_embed.0.weight"], P["vertex_label_embed.0.bias"]))
    return torch.cat([le, pe], 2)


def _mha(P, prefix, cfg, q_in, kv_in, mask_bool, training, masks=None, site=0):
    """nn.MultiheadAttention(64, 8, dropout) explicit path (need_weights=True default, pace.py:52-56):
    packed in-proj, q scaled by 1/sqrt(dh), additive -inf mask, softmax, dropout on weights, out-proj.
    q_in/kv_in: [L, B, d]; mask_bool: [B*H, L, S] (True = masked)."""
    L, B, d = q_in.shape
    S = kv_in.shape[0]
    H = cfg.heads
    dh = d // H
    W, bias = P[prefix + ".in_proj_weight"], P[prefix + ".in_proj_bias"]
    q = F.linear(q_in, W[:d], bias[:d])
    k = F.linear(kv_in, W[d:2 * d], bias[d:2 * d])
    v = F.linear(kv_in, W[2 * d:], bias[2 * d:])
    q = q.reshape(L, B * H, dh).transpose(0, 1) * (1.0 / math.sqrt(dh))
    k = k.reshape(S, B * H, dh).transpose(0, 1)
    v = v.reshape(S, B * H, dh).transpose(0, 1)
    add_mask = torch.zeros(mask_bool.shape, dtype=q.dtype).masked_fill_(mask_bool, float("-inf"))
    w = torch.baddbmm(add_mask, q, k.transpose(1, 2))
    w = torch.softmax(w, dim=-1)
    w = _drop(w, cfg.dropout, training, None if masks is None else torch.from_numpy(masks.attn(site, B, L, H)))
    o = torch.bmm(w, v).transpose(0, 1).reshape(L, B, d)
    return F.linear(o, P[prefix + ".out_proj.weight"], P[prefix + ".out_proj.bias"])


def _ln(P, name, x):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-5)


def _seq_mask(masks, site, x_lbd):
    """mask for a sequence-first [L, B, d] activation."""
    if masks is None:
        return None
    L, B, d = x_lbd.shape
    return torch.from_numpy(masks.tile(site, B, L, d)).transpose(0, 1)


def _relu(hook, name, pre):
    """ReLU of a hidden layer.  `hook` (tests only, tests/relu_trace.py) sees the pre-activation under the layer's
    state-dict prefix and may evaluate the network on a prescribed linear piece (pre * mask) instead: the parity tests
    use it to count sign disagreements with the device at near-zero pre-activations and to compare gradients on the
    SAME piece of this piecewise-linear function."""
    return torch.relu(pre) if hook is None else hook(name, pre)


def _ffn(P, prefix, cfg, x, training, masks=None, site=0, relu=None):
    h = _relu(relu, prefix + ".linear1", F.linear(x, P[prefix + ".linear1.weight"], P[prefix + ".linear1.bias"]))
    h = _drop(h, cfg.dropout, training, _seq_mask(masks, site, h))
    return F.linear(h, P[prefix + ".linear2.weight"], P[prefix + ".linear2.bias"])


def _encoder(P, cfg, x, mask, training, masks=None, relu=None):
    for l in range(cfg.layers):
        pre = f"encoder.layers.{l}"
        a = _mha(P, pre + ".self_attn", cfg, x, x, mask, training, masks, site_enc(l, 0))
        x = _ln(P, pre + ".norm1", x + _drop(a, cfg.dropout, training, _seq_mask(masks, site_enc(l, 1), a)))
        f = _ffn(P, pre, cfg, x, training, masks, site_enc(l, 2), relu)
        x = _ln(P, pre + ".norm2", x + _drop(f, cfg.dropout, training, _seq_mask(masks, site_enc(l, 3), f)))
        if torch.isnan(x).any():  # pace.py:97-98
            raise ValueError(f"NaN detected in the output of encoder layer {l}")
    return x


def _decoder(P, cfg, t, memory, mask, training, masks=None, relu=None):
    for l in range(cfg.layers):
        pre = f"decoder.layers.{l}"
        a = _mha(P, pre + ".self_attn", cfg, t, t, mask, training, masks, site_dec(l, 0))
        t = _ln(P, pre + ".norm1", t + _drop(a, cfg.dropout, training, _seq_mask(masks, site_dec(l, 1), a)))
        a = _mha(P, pre + ".multihead_attn", cfg, t, memory, mask, training, masks, site_dec(l, 2))   # tgt_mask, pace.py:148
        t = _ln(P, pre + ".norm2", t + _drop(a, cfg.dropout, training, _seq_mask(masks, site_dec(l, 3), a)))
        f = _ffn(P, pre, cfg, t, training, masks, site_dec(l, 4), relu)
        t = _ln(P, pre + ".norm3", t + _drop(f, cfg.dropout, training, _seq_mask(masks, site_dec(l, 5), f)))
    return t


def encode_direct(P, cfg: PaceConfig, features: Dict, training: bool = False, masks=None, relu=None):
    """pace.py:1613-1641 -> (mu, logvar) [B, latent]."""
    x = _embed(P, cfg, features["vertex_label_features"], features["vertex_position_features"],
               features["adjacency_matrices"], training, masks, 0)
    mem = _encoder(P, cfg, x.transpose(0, 1), features["target_masks"], training, masks, relu)
    flat = mem.transpose(0, 1).reshape(-1, cfg.N * cfg.d_model)
    return (F.linear(flat, P["fc1.weight"], P["fc1.bias"]),
            F.linear(flat, P["fc2.weight"], P["fc2.bias"]))


def log_likelihood(P, cfg: PaceConfig, features: Dict, dec_out: torch.Tensor, relu=None) -> torch.Tensor:
    """pace.py:1880-1972: node log-softmax gather (target = label of the NEXT vertex, positions
    0..N-2) + edge BCE over all pairs j < i <= N-2 with truth adj[b, j+1, i+1]."""
    B, N = dec_out.shape[0], cfg.N
    adj = features["adjacency_matrices"]
    h = _relu(relu, "add_node.0", F.linear(dec_out, P["add_node.0.weight"], P["add_node.0.bias"]))
    logp = torch.log_softmax(F.linear(h, P["add_node.2.weight"], P["add_node.2.bias"]), dim=2)
    tgt = torch.zeros(B, N, dtype=torch.long)
    vl = torch.tensor([list(v)[:N] for v in features["vertex_labels"]], dtype=torch.long)
    tgt[:, :vl.shape[1]] = vl
    sizes = torch.tensor(features["num_vertices"])
    valid = (torch.arange(N).expand(B, N) < (sizes - 1).unsqueeze(1)).to(logp.dtype)
    ll = (torch.gather(logp, 2, tgt.unsqueeze(2)).squeeze(2) * valid).sum()

    M = int(sizes.max().item()) - 1
    ii, jj = torch.meshgrid(torch.arange(M), torch.arange(M), indexing="ij")
    keep = (ii > jj).unsqueeze(0) & (ii.unsqueeze(0) < (sizes - 1)[:, None, None]) \
        & (jj.unsqueeze(0) < (sizes - 1)[:, None, None])
    b_idx = torch.arange(B).view(-1, 1, 1).expand(-1, M, M)[keep]
    i_idx = ii.unsqueeze(0).expand(B, -1, -1)[keep]
    j_idx = jj.unsqueeze(0).expand(B, -1, -1)[keep]
    pair = torch.cat([dec_out[b_idx, i_idx], dec_out[b_idx, j_idx]], dim=1)
    if relu is not None and hasattr(relu, "aux"):
        relu.aux["pairs"] = (b_idx, i_idx, j_idx)
    e = _relu(relu, "add_edge.0", F.linear(pair, P["add_edge.0.weight"], P["add_edge.0.bias"]))
    logit = F.linear(e, P["add_edge.2.weight"], P["add_edge.2.bias"])
    truth = adj[b_idx, j_idx + 1, i_idx + 1].view(-1, 1)
    return ll - F.binary_cross_entropy_with_logits(logit, truth, reduction="sum")


def loss_direct(P, cfg: PaceConfig, features: Dict, beta: float = 0.005, training: bool = False,
                eps: Optional[torch.Tensor] = None, return_aux: bool = False, masks=None, relu=None):
    """pace.py:1974-2035 -> (total, recon, kld).  ``eps`` (already scaled by epsilon_scale=0.01)
    replaces the reference's ``randn_like(std) * 0.01`` draw when given (train mode only).  ``masks``
    (oracle.rng.DeviceMasks) injects the device's dropout masks instead of torch's own draws."""
    mu, logvar = encode_direct(P, cfg, features, training, masks, relu)
    if training:
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std) * 0.01
        z = mu + eps * std
    else:
        z = mu
    mem = F.linear(z, P["fc3.weight"], P["fc3.bias"]).reshape(-1, cfg.N, cfg.d_model).transpose(0, 1)
    x = _embed(P, cfg, features["vertex_label_features"], features["vertex_position_features"],
               features["adjacency_matrices"], training, masks, 2)
    dec = _decoder(P, cfg, x.transpose(0, 1), mem, features["target_masks"], training, masks, relu).transpose(0, 1)
    ll = log_likelihood(P, cfg, features, dec, relu)
    kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    total = -ll + beta * kld
    if return_aux:
        return total, -ll, kld, {"mu": mu, "logvar": logvar, "decoder_output": dec}
    return total, -ll, kld


class OracleTrainer:
    """train_batch (experiments/03_synthetic_12/main.py:95-118) over leaf parameter tensors:
    zero_grad, loss_direct(train), backward, clip_grad_norm_(1.0), Adam(lr 1e-4) step."""

    def __init__(self, cfg: PaceConfig, params: Dict[str, torch.Tensor], lr: float = 1e-4,
                 max_grad_norm: float = 1.0):
        self.cfg = cfg
        self.P = {k: v.clone().float().requires_grad_(True) for k, v in params.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)
        self.max_grad_norm = max_grad_norm

    def step(self, features: Dict, training: bool = True, eps: Optional[torch.Tensor] = None, masks=None):
        self.opt.zero_grad()
        total, recon, kld = loss_direct(self.P, self.cfg, features, training=training, eps=eps, masks=masks)
        value = float(total.item())
        total.backward()
        torch.nn.utils.clip_grad_norm_(list(self.P.values()), self.max_grad_norm)
        self.opt.step()
        return value, float(recon.detach()), float(kld.detach())
