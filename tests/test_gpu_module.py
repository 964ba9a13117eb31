"""GPU parity through the drop-in Python surface (PaceVaeV3 / train_batch) -> C ABI -> HIP kernels."""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from oracle.rng import DeviceMasks
from tests.helpers import CONFIGS, grad_err, graphs_from, load_golden, load_npz, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_model(cfg, params, dropout=0.15):
    from dags_vae_search_amd import PaceVaeV3
    m = PaceVaeV3(max_num_vertices=cfg.n, vertex_label_cardinality=cfg.card, vertices_embedding_size=32, num_heads=8,
                  num_layers=3, ff_hidden_size=64, latent_layer_size=32, fc_hidden=32, dropout=dropout)
    m.load_state_dict(params)
    return m.to(DEV)


def feats_for(model, graphs):
    from dags_vae_search_amd import LabeledGraph
    return model.prepare_features([LabeledGraph(l, e) for l, e in graphs])


@pytest.mark.parametrize("name", list(CONFIGS))
def test_loss_direct_and_autograd_match_reference_golden(name):
    cfg, params, graphs, z = load_golden(name)
    model = build_model(cfg, params).eval()
    total, recon, kld = model.loss_direct(feats_for(model, graphs))
    assert rel(total.item(), z["eval/total"]) < 1e-4            # BASELINE.json: ELBO < 1e-4 relative
    assert rel(kld.item(), z["eval/kld"]) < 1e-4
    assert abs(recon.item() - float(z["eval/recon"])) < 1e-4 * max(1.0, abs(float(z["eval/recon"])))
    total.backward()
    err, worst = grad_err({k: p.grad for k, p in model.named_parameters()}, z, "eval/grad/")
    assert err < 2e-3, (worst, err)


@pytest.mark.parametrize("name", ["n12c12", "asia_rand"])
def test_train_mode_dropout0_injected_eps_and_one_step(name):
    """train mode, dropout 0, reference's captured eps -> golden loss/gradients, then golden clip+Adam step, through
    BOTH optimiser paths: stock torch.optim.Adam over the autograd-wrapped kernels and the fused flat-buffer Adam."""
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.train import train_batch
    cfg, params, graphs, z = load_golden(name)
    eps = torch.from_numpy(z["train0/eps"])
    gn = np.sqrt(sum(float((z[k].astype(np.float64) ** 2).sum()) for k in z.files if k.startswith("train0/grad/")))
    coef = min(1.0, 1.0 / (gn + 1e-6))

    def check_step(model):
        # Adam's first step moves every parameter by lr * g / (|g| + 1e-8): entries whose clipped gradient is above 1e-7
        # must land on the reference's value (measured: 1.4e-7); below that the step direction is rounding noise in ANY
        # implementation — e.g. the key bias of every attention, whose true gradient is exactly zero (softmax shift
        # invariance) — and only the size of the move is bounded (at most 2 lr apart).
        worst = 0.0
        for k, p in model.state_dict().items():
            err = np.abs(p.cpu().numpy() - z["step/param/" + k])
            sure = np.abs(z["train0/grad/" + k]) * coef > 1e-7
            assert err.max() < 2.01e-4, k
            if sure.any():
                worst = max(worst, float(err[sure].max()))
        assert worst < 1e-6

    # (a) reference sequence with a stock optimiser
    model = build_model(cfg, params, dropout=0.0).train()
    f = feats_for(model, graphs)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.zero_grad()
    total, recon, kld = model.loss_direct(f, eps=eps)
    assert rel(total.item(), z["train0/total"]) < 1e-4
    total.backward()
    err, worst = grad_err({k: p.grad for k, p in model.named_parameters()}, z, "train0/grad/")
    assert err < 2e-3, (worst, err)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    check_step(model)
    # (b) fused path
    model2 = build_model(cfg, params, dropout=0.0).train()
    fo = dopt.Adam(model2.parameters(), lr=1e-4).attach(model2)
    losses = model2.loss_and_grad(f, eps=eps.to(DEV))
    fo.step(max_grad_norm=1.0)
    assert rel(losses[0].item(), z["train0/total"]) < 1e-4
    assert abs(fo.grad_norm.item() - gn) / gn < 1e-4
    check_step(model2)


@pytest.mark.parametrize("name,B", [("n12c12", 48), ("asia_rand", 32)])
def test_train_mode_dropout_on_matches_oracle_with_device_masks(name, B):
    cfg, params, graphs, z = load_golden(name)
    graphs = graphs[:B]
    model = build_model(cfg, params, dropout=0.15).train()
    model.seed(77)
    model.dag_offset = 5
    f = feats_for(model, graphs)
    total, recon, kld = model.loss_direct(f)
    total.backward()
    seed = (77 << 32) | 1                                  # PaceVaeV3._next_seed: (seed << 32) | step
    masks = DeviceMasks(seed, 0.15, dag_offset=5)
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    f_np = ofeat.dense_features(graphs, cfg.card)
    t, r, k = po.loss_direct(P, cfg, ofeat.to_torch(f_np), training=True, eps=torch.from_numpy(masks.eps(B)), masks=masks)
    t.backward()
    assert rel(total.item(), t.detach()) < 1e-4 and rel(kld.item(), k.detach()) < 1e-4
    scale = max(float(p.grad.abs().max()) for p in P.values())
    for kname, p in model.named_parameters():
        ref = P[kname].grad.numpy()
        e = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-4 * scale)
        assert e < 3e-3, (kname, e)


def test_encode_direct_known_answer():
    """254 (graph -> mu) rows written by the reference's own pipeline (experiments/01_bn_asia predictor dataset)."""
    z = load_npz("asia_known_answer.npz")
    ck = load_npz("asia_ckpt110.npz")
    cfg = po.PaceConfig(n=8, card=8)
    model = build_model(cfg, {k: torch.from_numpy(ck[k]) for k in ck.files}).eval()
    mu, logvar = model.encode_direct(feats_for(model, graphs_from(z, 8)))
    assert np.abs(mu.cpu().numpy() - z["mu"]).max() < 1e-5


def test_errors_are_loud():
    from dags_vae_search_amd import LabeledGraph, PaceVaeV3
    cfg, params, graphs, z = load_golden("asia")
    model = build_model(cfg, params).eval()
    f = feats_for(model, graphs[:4])
    bad = dict(f)
    bad["vertex_label_features"] = f["vertex_label_features"] * 0.5
    with pytest.raises(ValueError):
        model.loss_direct(bad)
    with pytest.raises(AssertionError):
        model.prepare_features([LabeledGraph([0, 1, 2], [(0, 1)])])
    cpu_model = PaceVaeV3(8, 8, 32, 8, 3, 64, 32, 32, 0.15)
    with pytest.raises(RuntimeError):
        cpu_model.loss_direct({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in f.items()})
    with pytest.raises(NotImplementedError):
        PaceVaeV3(46, 46, 32, 8, 3, 64, 32, 32, 0.15)      # > 48 tokens
    with pytest.raises(NotImplementedError):
        PaceVaeV3(8, 8)                      # reference defaults (256-wide, 6 layers) are not this build


def test_full_size_properties_n12_b4096():
    """BASELINE metric shape (n=12, card=12, B=4096): size-independent properties instead of an oracle run:
    shard additivity of loss and gradient (the data-parallel contract), bitwise run-to-run determinism, and
    training steps that reduce the loss."""
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.dist import shard_features
    from dags_vae_search_amd.train import train_batch
    cfg = po.PaceConfig(n=12, card=12)
    params = po.init_params(cfg, seed=3)
    graphs = ofeat.synthetic_dags(12, 12, 4096, seed=42)
    model = build_model(cfg, params).train()
    f = feats_for(model, graphs)
    model.seed(1)
    l1 = model.loss_and_grad(f).clone()
    g1 = model.flat_grads.clone()
    model.seed(1)
    l2 = model.loss_and_grad(f).clone()
    assert torch.equal(l1, l2) and torch.equal(g1, model.flat_grads)          # deterministic, no float atomics
    tot = torch.zeros_like(g1)
    lsum = 0.0
    for rank in range(2):
        shard, off = shard_features(f, rank, 2)
        model.seed(1)
        model.dag_offset = off
        ls = model.loss_and_grad(shard)
        tot += model.flat_grads
        lsum += ls[0].item()
    model.dag_offset = 0
    assert rel(lsum, l1[0].item()) < 1e-5
    assert (tot - g1).abs().max().item() < 2e-4 * g1.abs().max().item()
    opt = dopt.Adam(model.parameters(), lr=1e-3).attach(model)
    first = None
    for step in range(12):
        loss_value, recon, kld = train_batch(f, model, opt)
        first = loss_value if first is None else first
    assert np.isfinite(loss_value) and loss_value < 0.97 * first


def test_asia_b4096_elbo_matches_cpu_oracle():
    """BASELINE config 2 (asia n=8, batch 4096 on one MI355X): ELBO vs the CPU path on the identical batch < 1e-4
    relative, in eval mode and in train mode with the device's dropout masks (fp32 kernels; bf16 is not used)."""
    from dags_vae_search_amd.synthetic import synthetic_dags
    ck = load_npz("asia_ckpt110.npz")
    params = {k: torch.from_numpy(ck[k]) for k in ck.files}
    cfg = po.PaceConfig(n=8, card=8)
    graphs = synthetic_dags(8, 8, 4096, seed=21)
    model = build_model(cfg, params)
    f = model.prepare_features(graphs)
    f_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in f.items()}
    model.eval()
    total, recon, kld = model.loss_direct(f)
    with torch.no_grad():
        t, r, k = po.loss_direct(params, cfg, f_cpu, training=False)
    assert rel(total.item(), t) < 1e-4 and rel(recon.item(), r) < 1e-4 and rel(kld.item(), k) < 1e-4
    model.train()
    model.seed(3)
    total, recon, kld = model.loss_direct(f)
    masks = DeviceMasks((3 << 32) | 1, 0.15)
    with torch.no_grad():
        t, r, k = po.loss_direct(params, cfg, f_cpu, training=True, eps=torch.from_numpy(masks.eps(4096)), masks=masks)
    assert rel(total.item(), t) < 1e-4 and rel(kld.item(), k) < 1e-4


def test_sachs_shape_n11():
    """BASELINE config 4's model shape (sachs: n=11, card=11, N=14 tokens) against the oracle, fwd + gradients."""
    from dags_vae_search_amd.synthetic import synthetic_dags
    cfg = po.PaceConfig(n=11, card=11)
    params = po.init_params(cfg, seed=11)
    graphs = synthetic_dags(11, 11, 64, seed=4)
    model = build_model(cfg, params).eval()
    f = model.prepare_features(graphs)
    total, recon, kld = model.loss_direct(f)
    total.backward()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    t, r, k = po.loss_direct(P, cfg, {k2: (v.cpu() if torch.is_tensor(v) else v) for k2, v in f.items()})
    t.backward()
    assert rel(total.item(), t.detach()) < 1e-4
    scale = max(float(p.grad.abs().max()) for p in P.values())
    for name, p in model.named_parameters():
        ref = P[name].grad.numpy()
        assert float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-4 * scale) < 2e-3, name


def test_compact_batch_front_end_equals_dense_features_path():
    """row codec -> records on the device (dvs_build_records) gives bit-identical losses/gradients to
    the dense prepare_features + dvs_pack_features path, and drives train_batch from a device-resident dataset."""
    from dags_vae_search_amd import CompactDagDataset, encode_graphs, optim as dopt
    from dags_vae_search_amd.synthetic import synthetic_dags
    from dags_vae_search_amd.train import train_batch
    cfg = po.PaceConfig(n=12, card=12)
    params = po.init_params(cfg, seed=5)
    graphs = synthetic_dags(12, 12, 512, seed=77)
    model = build_model(cfg, params).eval()
    dense = model.prepare_features(graphs)
    cb = encode_graphs(graphs, 12).to(DEV)
    la = [t.item() for t in model.loss_direct(dense)]
    lb = [t.item() for t in model.loss_direct(cb)]
    assert la == lb
    mu_a, _ = model.encode_direct(dense)
    mu_b, _ = model.encode_direct(cb)
    assert torch.equal(mu_a, mu_b)
    with pytest.raises(ValueError):
        bad = encode_graphs(graphs[:4], 12)
        bad.labels[0, 0] = 200
        model.loss_direct(bad.to(DEV))
    ds = CompactDagDataset(graphs, 12, device=DEV)
    model.train()
    opt = dopt.Adam(model.parameters(), lr=1e-3).attach(model)
    losses = []
    for epoch in range(3):
        for batch in ds.batches(256, generator=torch.Generator().manual_seed(epoch)):
            losses.append(train_batch(batch, model, opt)[0] / len(batch))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def _alarm_graphs(count, seed):
    """BASELINE config 5 inputs: n = 37 synthetic DAGs at density <= 0.2 plus deep chains (irregular, 37-level orders)."""
    from dags_vae_search_amd import LabeledGraph
    from dags_vae_search_amd.synthetic import synthetic_dags
    n = 37
    graphs = synthetic_dags(n, n, count - 8, seed=seed, density_limit=0.2)
    rng = np.random.default_rng(seed + 1)
    chain = [(v, v + 1) for v in range(n - 1)]
    for k in range(8):
        extra = [(v, v + 2 + k) for v in range(0, n - 2 - k, 3)]
        graphs.append(LabeledGraph([int(x) for x in rng.permutation(n)], sorted(set(chain + (extra if k else [])))))
    return graphs


def test_alarm_n37_b2048_wide_path():
    """BASELINE config 5 (alarm-size n = 37, batch 2048, one MI355X), tiled wide path: ELBO vs the CPU oracle on the
    identical batch < 1e-4 relative (eval), dropout-on gradients vs the oracle run with the device's masks on a
    sub-batch, bitwise determinism, shard additivity, compact front-end == dense features, and loss-reducing steps."""
    from dags_vae_search_amd import encode_graphs, optim as dopt
    from dags_vae_search_amd.dist import shard_features
    from dags_vae_search_amd.train import train_batch
    cfg = po.PaceConfig(n=37, card=37)
    params = po.init_params(cfg, seed=8)
    graphs = _alarm_graphs(2048, seed=31)
    model = build_model(cfg, params)
    f = model.prepare_features(graphs)
    f_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in f.items()}
    model.eval()
    total, recon, kld = model.loss_direct(f)
    with torch.no_grad():
        t, r, k = po.loss_direct(params, cfg, f_cpu, training=False)
    assert rel(total.item(), t) < 1e-4 and rel(recon.item(), r) < 1e-4 and rel(kld.item(), k) < 1e-4
    # device-side front-end (64-bit rows) gives the same records
    cb = encode_graphs(graphs, 37).to(DEV)
    assert [x.item() for x in model.loss_direct(cb)] == [total.item(), recon.item(), kld.item()]
    # dropout-on gradients on the last 24 DAGs (they include the chains)
    sub = graphs[-24:]
    fs = model.prepare_features(sub)
    model.train()
    model.seed(9)
    model.dag_offset = 3
    tt, rr, kk = model.loss_direct(fs)
    tt.backward()
    masks = DeviceMasks((9 << 32) | 1, 0.15, dag_offset=3)
    P = {k2: v.clone().requires_grad_(True) for k2, v in params.items()}
    fs_cpu = {k2: (v.cpu() if torch.is_tensor(v) else v) for k2, v in fs.items()}
    to, ro, ko = po.loss_direct(P, cfg, fs_cpu, training=True, eps=torch.from_numpy(masks.eps(24)), masks=masks)
    to.backward()
    assert rel(tt.item(), to.detach()) < 1e-4 and rel(kk.item(), ko.detach()) < 1e-4
    scale = max(float(p.grad.abs().max()) for p in P.values())
    for name, p in model.named_parameters():
        ref = P[name].grad.numpy()
        e = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-4 * scale)
        assert e < 3e-3, (name, e)
    model.dag_offset = 0
    model.zero_grad(set_to_none=True)
    # full-size properties
    model.seed(1)
    l1 = model.loss_and_grad(f).clone()
    g1 = model.flat_grads.clone()
    model.seed(1)
    l2 = model.loss_and_grad(f).clone()
    assert torch.equal(l1, l2) and torch.equal(g1, model.flat_grads)
    tot = torch.zeros_like(g1)
    lsum = 0.0
    for rank in range(2):
        shard, off = shard_features(f, rank, 2)
        model.seed(1)
        model.dag_offset = off
        ls = model.loss_and_grad(shard)
        tot += model.flat_grads
        lsum += ls[0].item()
    model.dag_offset = 0
    assert rel(lsum, l1[0].item()) < 1e-5
    assert (tot - g1).abs().max().item() < 2e-4 * g1.abs().max().item()
    opt = dopt.Adam(model.parameters(), lr=1e-3).attach(model)
    first = None
    for step in range(8):
        loss_value, _, _ = train_batch(f, model, opt)
        first = loss_value if first is None else first
    assert np.isfinite(loss_value) and loss_value < 0.97 * first


@pytest.mark.parametrize("n,B", [(12, 8192), (11, 8192)])
def test_batch_8192_per_gpu_shapes(n, B):
    """BASELINE configs 3 (n = 12, batch 8192 on one GPU) and 4 (sachs n = 11, 65 536 over 8 GPUs = 8192 per GPU):
    ELBO of the full batch vs the CPU oracle on a 512-DAG slice summed with the rest via shard additivity, and
    run-to-run determinism at the full size."""
    from dags_vae_search_amd.dist import shard_features
    from dags_vae_search_amd.synthetic import synthetic_dags
    cfg = po.PaceConfig(n=n, card=n)
    params = po.init_params(cfg, seed=n)
    graphs = synthetic_dags(n, n, B, seed=100 + n)
    model = build_model(cfg, params).eval()
    f = model.prepare_features(graphs)
    full = [x.item() for x in model.loss_direct(f)]
    parts = [0.0, 0.0, 0.0]
    for rank in range(16):
        shard, off = shard_features(f, rank, 16)
        model.dag_offset = off
        got = [x.item() for x in model.loss_direct(shard)]
        if rank == 5:       # one 512-DAG shard against the CPU path
            with torch.no_grad():
                t, r, k = po.loss_direct(params, cfg, {k2: (v.cpu() if torch.is_tensor(v) else v) for k2, v in shard.items()})
            assert rel(got[0], t) < 1e-4 and rel(got[2], k) < 1e-4
        parts = [a + b for a, b in zip(parts, got)]
    model.dag_offset = 0
    assert all(rel(a, b) < 1e-5 for a, b in zip(parts, full))
    model.train()
    model.seed(4)
    l1 = model.loss_and_grad(f).clone()
    g1 = model.flat_grads.clone()
    model.seed(4)
    l2 = model.loss_and_grad(f)
    assert torch.equal(l1, l2) and torch.equal(g1, model.flat_grads)


def test_train_model_epoch_loop_and_checkpoint_reload(tmp_path):
    """Row A13: the epoch loop of experiments/03_synthetic_12/main.py:121-198 (shuffled DataLoader over per-graph feature
    dicts, Adam, ReduceLROnPlateau on the last batch's loss, state_dict per epoch) + load_model_state
    (src/train_utils.py:11-36) on the saved checkpoint: a reloaded model encodes identically."""
    from dags_vae_search_amd import LabeledDag, PaceVaeV3, load_model_state, train_model
    from dags_vae_search_amd.datasets import LabeledDagDatasetInMemory
    from dags_vae_search_amd.synthetic import synthetic_dags
    tk = LabeledDag(num_vertices=8, label_cardinality=8)
    graphs = synthetic_dags(8, 8, 192, seed=6)
    torch.manual_seed(42)
    model = PaceVaeV3(8, 8, 32, 8, 3, 64, 32, 32, 0.15).to(DEV)
    ds = LabeledDagDatasetInMemory(None, tk, model, rows=[tk.from_graph_to_dict_writable(g) for g in graphs])
    hist = train_model(model, ds, epochs=4, batch_size=32, lr=1e-3, checkpoint_dir=str(tmp_path), log=lambda *_: None)
    assert len(hist) == 4 and np.isfinite(hist).all() and hist[-1] < hist[0]
    ck = tmp_path / "model_checkpoint_4.pth"
    sd = torch.load(ck, weights_only=True)
    assert len(sd) == 108 and sd["fc3.weight"].shape == (11 * 64, 32)
    fresh = PaceVaeV3(8, 8, 32, 8, 3, 64, 32, 32, 0.15)
    load_model_state(fresh, str(ck))
    fresh = fresh.to(DEV).eval()
    model.eval()
    f = model.prepare_features(graphs[:16])
    assert torch.equal(model.encode_direct(f)[0], fresh.encode_direct(f)[0])
