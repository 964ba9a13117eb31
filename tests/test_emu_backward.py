"""Kernel checks on the host emulator (tests/emu): gradients, dropout-on train mode, clip+Adam — vs the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from oracle.rng import DeviceMasks
from tests.emu.harness import EmuModel, ptr
from tests.helpers import load_golden, rel


@pytest.fixture(params=[4, 8])
def waves_per_wg(request, monkeypatch):
    """Both workgroup widths of the one-tile stack kernels (dvs_api.hip: waves_per_wg): 8 waves = two cooperative groups,
    4 = the narrow mapping of small batches.  The emulator has 2 'CUs', so without the switch every small test batch would
    take the narrow one only."""
    monkeypatch.setenv("DVS_WAVES_PER_WG", str(request.param))
    return request.param


def _oracle_grads(cfg, params, f_np, **kw):
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    total, recon, kld = po.loss_direct(P, cfg, ofeat.to_torch(f_np), **kw)
    total.backward()
    return float(total.detach()), float(recon.detach()), float(kld.detach()), {k: v.grad.numpy() for k, v in P.items()}


def _check_grads(got, ref, tol):
    scale = max(float(np.abs(v).max()) for v in ref.values())
    for k, r in ref.items():
        err = float(np.abs(got[k] - r).max()) / max(float(np.abs(r).max()), 1e-4 * scale)
        assert err < tol, (k, err)


@pytest.mark.parametrize("name,B", [("n12c12", 5), ("asia_rand", 6), ("n12c1", 3), ("n37c37", -4),
                                    ("n13c5", -3), ("n14c14", -3), ("n29c7", -3), ("n45c45", -3)])
def test_emu_gradients_eval(name, B, waves_per_wg):
    """Eval-mode gradients against the oracle on a slice of the fixture's graphs (the reference's own gradients of the
    full fixtures are checked on the GPU and by tests/test_oracle.py); negative B = the LAST graphs of the fixture (the
    path / star DAGs of the alarm-size and edge-shape fixtures)."""
    cfg, params, graphs, z = load_golden(name)
    graphs = graphs[:B] if B > 0 else graphs[B:]
    B = len(graphs)
    f_np = ofeat.dense_features(graphs, cfg.card)
    m = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=False)
    assert m.pack(f_np) == 0
    m.forward()
    grads, flat = m.backward(1.0, 0.005)
    assert not np.isnan(flat).any()
    _, _, _, ref = _oracle_grads(cfg, params, f_np, training=False)
    _check_grads(grads, ref, 1e-3)


@pytest.mark.parametrize("name,B,seed", [("n12c12", 6, 1234), ("asia_rand", 4, 99), ("n37c37", 3, 4321)])
def test_emu_train_mode_with_dropout_and_hashed_eps(name, B, seed, waves_per_wg):
    """dropout 0.15 + counter-based eps: the oracle runs with the device's masks (oracle/rng.py)."""
    cfg, params, graphs, z = load_golden(name)
    f_np = ofeat.dense_features(graphs[:B], cfg.card)
    m = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=True, dropout=0.15, seed=seed, dag_offset=7)
    assert m.pack(f_np) == 0
    losses, mu, lv = m.forward()
    grads, _ = m.backward(1.0, 0.005)
    masks = DeviceMasks(seed, 0.15, dag_offset=7)
    total, recon, kld, ref = _oracle_grads(cfg, params, f_np, training=True, eps=torch.from_numpy(masks.eps(B)),
                                           masks=masks)
    assert rel(losses[0], total) < 1e-4 and rel(losses[2], kld) < 1e-4
    _check_grads(grads, ref, 2e-3)
    # dropout really is on: the eval-mode latent differs (a fresh-init model's LOSS is nearly mask-independent)
    m2 = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=False)
    m2.pack(f_np)
    _, mu_eval, _ = m2.forward()
    assert np.abs(mu_eval - mu).max() > 1e-3


def test_emu_train0_golden_and_adam_step():
    """train mode, dropout 0, injected eps -> reference golden losses; then clip(1.0)+Adam vs the golden step."""
    cfg, params, graphs, z = load_golden("n12c12")
    B = 8
    f_np = ofeat.dense_features(graphs[:B], cfg.card)
    eps = z["train0/eps"][:B]
    m = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=True, dropout=0.0)
    m.pack(f_np)
    losses, _, _ = m.forward(eps)
    grads, flat = m.backward(1.0, 0.005)
    cfg0 = po.PaceConfig(n=cfg.n, card=cfg.card, dropout=0.0)
    tr = po.OracleTrainer(cfg0, params)
    value, recon, kld = tr.step(ofeat.to_torch(f_np), training=True, eps=torch.from_numpy(eps))
    assert rel(losses[0], value) < 1e-5
    # clip + Adam on the emulator (dvs_clip_adam scales the gradient buffer in place, like clip_grad_norm_)
    grads = {k: v.copy() for k, v in grads.items()}
    gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    exp_avg = np.zeros_like(m.flat)
    exp_avg_sq = np.zeros_like(m.flat)
    scratch = np.zeros(4096, np.float32)
    p = m.flat.copy()
    # a raised guard (non-finite loss / invalid batch) must leave parameters, moments and gradients untouched
    for guard in ([1.0, 0.0], [0.0, 1.0]):
        p0, f0 = p.copy(), flat.copy()
        rc = m.lib.dvs_clip_adam(m.P, ptr(p), ptr(flat), ptr(exp_avg), ptr(exp_avg_sq), 1e-4, 0.9, 0.999, 1e-8, 1, 1.0,
                                 ptr(scratch), ptr(np.asarray(guard, np.float32)), None)
        assert rc == 0 and np.array_equal(p, p0) and np.array_equal(flat, f0)
        assert not exp_avg.any() and not exp_avg_sq.any()
    rc = m.lib.dvs_clip_adam(m.P, ptr(p), ptr(flat), ptr(exp_avg), ptr(exp_avg_sq), 1e-4, 0.9, 0.999, 1e-8, 1, 1.0,
                             ptr(scratch), ptr(np.zeros(2, np.float32)), None)
    assert rc == 0
    assert abs(np.sqrt(scratch[0]) - gn) / gn < 1e-5
    # the single-process step's variant (ABI 202): the slab reduction leaves the partial sums of squares behind
    # (dvs_loss_backward_sq) and the optimiser starts from them (dvs_clip_adam_from_partials): same norm, same update
    scratch2 = np.zeros(4096, np.float32)
    _, flat2 = m.backward(1.0, 0.005, clip_scratch=scratch2)
    nparts = ((m.P + 3) // 4 + 63) // 64
    assert np.array_equal(flat2, np.concatenate([v.reshape(-1) for v in [flat2]]))       # (gradient itself unchanged by the variant)
    assert abs(float(scratch2[2:2 + nparts].astype(np.float64).sum()) - gn ** 2) / gn ** 2 < 1e-5 and not scratch2[2 + nparts:].any()
    p2, ea2, eas2 = m.flat.copy(), np.zeros_like(m.flat), np.zeros_like(m.flat)
    rc = m.lib.dvs_clip_adam_from_partials(m.P, ptr(p2), ptr(flat2), ptr(ea2), ptr(eas2), 1e-4, 0.9, 0.999, 1e-8, 1, 1.0,
                                           ptr(scratch2), ptr(np.zeros(2, np.float32)), None)
    assert rc == 0 and abs(scratch2[0] - scratch[0]) <= 1e-5 * scratch[0]
    assert np.abs(flat2 - flat).max() <= 1e-5 * np.abs(flat).max()              # both clipped in place by ~the same coefficient
    worst_big = 0.0
    for name, off, shp in m.table:
        n = int(np.prod(shp))
        new = p[off:off + n].reshape(shp)
        ref = tr.P[name].detach().numpy()
        g = grads[name] * min(1.0, 1.0 / (gn + 1e-6))
        big = np.abs(g) > 1e-7          # below that the direction of Adam's first move is rounding noise (see the GPU twin)
        assert np.abs(new - ref).max() < 2.01e-4
        if big.any():
            worst_big = max(worst_big, float(np.abs(new - ref)[big].max()))
    assert worst_big < 1e-6


def test_emu_batch_sharding_is_additive(waves_per_wg):
    """Data-parallel property: loss and gradients of a batch equal the SUM over shards that keep the
    global DAG index (dag_offset) — also with dropout on."""
    cfg, params, graphs, z = load_golden("n12c12")
    B = 6
    f_all = ofeat.dense_features(graphs[:B], cfg.card)
    pn = {k: v.numpy() for k, v in params.items()}
    m = EmuModel(cfg, pn, B, training=True, dropout=0.15, seed=5)
    m.pack(f_all)
    la, _, _ = m.forward()
    ga, fa = m.backward()
    tot = np.zeros_like(fa)
    lsum = 0.0
    for lo, hi in ((0, 2), (2, 6)):
        f = ofeat.dense_features(graphs[lo:hi], cfg.card)
        ms = EmuModel(cfg, pn, hi - lo, training=True, dropout=0.15, seed=5, dag_offset=lo)
        ms.pack(f)
        l, _, _ = ms.forward()
        _, fl = ms.backward()
        tot += fl
        lsum += float(l[0])
    assert rel(lsum, la[0]) < 1e-6
    assert np.abs(tot - fa).max() < 1e-5 * max(1.0, np.abs(fa).max())


@pytest.mark.parametrize("n,card,B,train", [(1, 1, 3, False), (2, 5, 2, True), (13, 13, 3, True), (14, 14, 2, True),
                                            (13, 14, 2, False), (20, 3, 2, True), (29, 7, 2, True), (45, 45, 2, False),
                                            (45, 20, 2, True)])
def test_emu_edge_sizes_forward_and_gradients(n, card, B, train):
    """Size edges of the path: the smallest graph (N = 4 tokens), the one-tile limit (N = 16), the smallest tiled shape
    (N = 17), a shape that is wide only by its class count (C = 17), two tiles with few classes, two FULL tiles (N = 32),
    and the maximum (N = C = 48; N = 48 with dropout on) — loss and all 108 gradients against the oracle (fresh-seed parameters), dropout on where marked."""
    cfg = po.PaceConfig(n=n, card=card)
    params = po.init_params(cfg, seed=5)
    graphs = ofeat.synthetic_dags(n, card, B, seed=3, density_limit=0.2 if n > 20 else 0.4)
    f_np = ofeat.dense_features(graphs, card)
    m = EmuModel(cfg, {k: v.numpy() for k, v in params.items()}, B, training=train, dropout=0.15, seed=77, dag_offset=5)
    assert m.pack(f_np) == 0
    losses, _, _ = m.forward()
    grads, flat = m.backward(1.0, 0.005)
    assert not np.isnan(flat).any()
    kw = {}
    if train:
        masks = DeviceMasks(77, 0.15, dag_offset=5)
        kw = dict(training=True, eps=torch.from_numpy(masks.eps(B)), masks=masks)
    total, recon, kld, ref = _oracle_grads(cfg, params, f_np, **kw)
    assert rel(losses[0], total) < 1e-4 and rel(losses[2], kld) < 1e-4
    _check_grads(grads, ref, 1e-3)
