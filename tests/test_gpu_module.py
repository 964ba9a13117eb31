"""GPU parity through the drop-in Python surface (PaceVaeV3 / train_batch) -> C ABI -> HIP kernels."""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from oracle import pace_oracle as po
from oracle.rng import DeviceMasks
from tests.helpers import CONFIGS, EDGE_SHAPES, grad_err, graphs_from, load_golden, load_npz, rel
from tests.relu_trace import grad_errors, oracle_on_device_piece, record

# Gradient bound of the deterministic parity tests, as a fraction of the tensor maximum (the old 2e-3 was 50x
# what is measured; measured worst is printed by every test and kept in gpurun_out/parity_report.json).
GRAD_BOUND = 2e-4

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
    return model.prepare_features([g if isinstance(g, LabeledGraph) else LabeledGraph(*g) for g in graphs])


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
    record(f"golden_eval[{name}]", elbo_rel=rel(total.item(), z["eval/total"]), grad_err=err, worst=worst)
    assert err < GRAD_BOUND, (worst, err)


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

    def check_step(model, tag, clipped):
        """(1) EVERY entry: the update equals Adam's first step applied to the path's own clipped gradient,
        p - lr * g / (|g| + 1e-8) — pins the optimiser kernel on all 108 tensors, also where the direction is noise.
        (2) Entries whose clipped REFERENCE gradient is above 1e-7 land on the reference's parameters (measured 1.4e-7):
        below that the step direction is rounding noise in ANY implementation — e.g. the key bias of every attention,
        whose true gradient is exactly zero (softmax shift invariance) — and only the size of the move is bounded
        (2 lr).  (3) The share of entries outside the pinned set is asserted (and printed): the clip coefficient of
        these cases is 3e-4 / 3e-5, which pushes 2.8 % / 6.1 % of the clipped gradient entries below 1e-7."""
        worst, worst_formula = 0.0, 0.0
        unsure_total, n_total = 0, 0
        for k, p in model.state_dict().items():
            new = p.cpu().numpy()
            g = clipped[k].cpu().numpy().astype(np.float64)
            formula = params[k].numpy().astype(np.float64) - 1e-4 * g / (np.abs(g) + 1e-8)
            # fp32 rounding of the stored parameter (half an ulp of |p|) + of the 1e-4 step itself
            worst_formula = max(worst_formula, float((np.abs(new - formula) / (1.2e-7 * np.maximum(1.0, np.abs(new)) + 1e-10)).max()))
            err = np.abs(new - z["step/param/" + k])
            sure = np.abs(z["train0/grad/" + k]) * coef > 1e-7
            assert err.max() < 2.01e-4, k
            n_total += sure.size
            unsure_total += int((~sure).sum())
            if sure.any():
                worst = max(worst, float(err[sure].max()))
        record(f"adam_step[{name},{tag}]", pinned_worst=worst, own_gradient_formula_worst=worst_formula,
               unpinned_fraction=unsure_total / n_total)
        assert worst < 1e-6
        assert worst_formula < 1.0               # in units of the fp32 rounding allowance above
        assert unsure_total / n_total < 0.07

    # (a) reference sequence with a stock optimiser
    model = build_model(cfg, params, dropout=0.0).train()
    f = feats_for(model, graphs)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.zero_grad()
    total, recon, kld = model.loss_direct(f, eps=eps)
    assert rel(total.item(), z["train0/total"]) < 1e-4
    total.backward()
    err, worst = grad_err({k: p.grad for k, p in model.named_parameters()}, z, "train0/grad/")
    record(f"golden_train0[{name}]", elbo_rel=rel(total.item(), z["train0/total"]), grad_err=err, worst=worst)
    assert err < GRAD_BOUND, (worst, err)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    check_step(model, "stock", {k: p.grad for k, p in model.named_parameters()})
    # (b) fused path
    model2 = build_model(cfg, params, dropout=0.0).train()
    fo = dopt.Adam(model2.parameters(), lr=1e-4).attach(model2)
    losses = model2.loss_and_grad(f, eps=eps.to(DEV))
    fo.step(max_grad_norm=1.0)
    assert rel(losses[0].item(), z["train0/total"]) < 1e-4
    assert abs(fo.grad_norm.item() - gn) / gn < 1e-4
    check_step(model2, "fused", {k: p.grad for k, p in model2.named_parameters()})      # k_adam clips .grad in place


def dropout_on_parity(tag, model, params, cfg, graphs, seed, dag_offset, bound=GRAD_BOUND):
    """Train mode, dropout ON, the device's own masks injected into the oracle: ELBO < 1e-4, gradients within `bound` of
    the tensor maximum on the SAME linear piece of the network (tests/relu_trace.py: ReLU sign disagreements at
    near-zero pre-activations are counted, must be genuine ties, and the oracle is re-evaluated on the device's piece)."""
    B = len(graphs)
    model.train()
    model.zero_grad(set_to_none=True)        # .grad may alias flat_grads of an earlier fused step: autograd accumulates
    model.seed(seed)
    model.dag_offset = dag_offset
    f = feats_for(model, graphs)
    total, recon, kld = model.loss_direct(f)
    total.backward()
    masks = DeviceMasks((seed << 32) | 1, 0.15, dag_offset=dag_offset)          # PaceVaeV3._next_seed: (seed << 32) | step
    f_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in f.items()}
    t, k, g_piece, g_plain, info = oracle_on_device_piece(model, params, cfg, f_cpu, B, True,
                                                          eps=torch.from_numpy(masks.eps(B)), masks=masks)
    got = {n: p.grad for n, p in model.named_parameters()}
    err, worst, _ = grad_errors(got, g_piece)
    err_plain, worst_plain, _ = grad_errors(got, g_plain)
    record(tag, elbo_rel=rel(total.item(), t), grad_err=err, worst=worst, grad_err_plain_oracle=err_plain,
           worst_plain=worst_plain, **info)
    model.dag_offset = 0
    model.zero_grad(set_to_none=True)
    assert rel(total.item(), t) < 1e-4 and rel(kld.item(), k) < 1e-4
    assert info["relu_flips"] <= max(4, 2e-5 * info["relu_units"]), info
    assert err < bound, (worst, err, info)


@pytest.mark.parametrize("name,B,seed", [("n12c12", 48, 77), ("asia_rand", 32, 77), ("n12c12", 48, 5), ("n12c12", 48, 6)])
def test_train_mode_dropout_on_matches_oracle_with_device_masks(name, B, seed):
    cfg, params, graphs, z = load_golden(name)
    model = build_model(cfg, params, dropout=0.15)
    dropout_on_parity(f"dropout_on[{name},B={B},seed={seed}]", model, params, cfg, graphs[:B], seed, 5)


@pytest.mark.parametrize("name", EDGE_SHAPES)
def test_edge_of_range_shapes_dropout_on_and_compact_front_end(name):
    """Edges of the token range include/dvs.h promises (N = 16: full tile; 17: one valid row in the second tile; 32: two
    full tiles; 48: maximum, with 48 classes), two of them with card != n (pace.py:1159-1160,1188-1191,1921-1970).  The
    eval-mode ELBO and all 108 gradients of these fixtures against the REFERENCE's outputs run through the CONFIGS
    parametrisation above; here: train mode with dropout on against the oracle under the device's masks, the device-side
    front-end against the dense one, bitwise determinism and shard additivity at a batch that is not a multiple of the
    workgroup's DAG count."""
    from dags_vae_search_amd import encode_graphs
    from dags_vae_search_amd.dist import shard_features
    cfg, params, graphs, z = load_golden(name)
    model = build_model(cfg, params, dropout=0.15)
    dropout_on_parity(f"dropout_on[{name},B={len(graphs)},seed=13]", model, params, cfg, graphs, 13, 3)
    model.eval()
    f = feats_for(model, graphs)
    cb = encode_graphs(graphs, cfg.n).to(DEV)
    assert [x.item() for x in model.loss_direct(cb)] == [x.item() for x in model.loss_direct(f)]
    # 77 DAGs (odd, > one workgroup's share): determinism + shard additivity with dropout on
    from dags_vae_search_amd import LabeledGraph
    from dags_vae_search_amd.synthetic import synthetic_dags
    rng = np.random.default_rng(5)
    many = [LabeledGraph([int(x) for x in (g.labels if cfg.card >= cfg.n else rng.integers(0, cfg.card, cfg.n))], g.edges)
            for g in synthetic_dags(cfg.n, max(cfg.card, cfg.n), 77, seed=5, density_limit=0.2)]
    model.train()
    fm = model.prepare_features(many)
    model.seed(2)
    l1 = model.loss_and_grad(fm).clone()
    g1 = model.flat_grads.clone()
    model.seed(2)
    l2 = model.loss_and_grad(fm).clone()
    assert torch.equal(l1, l2) and torch.equal(g1, model.flat_grads)
    tot, lsum = torch.zeros_like(g1), 0.0
    for rank in range(2):
        shard, off = shard_features(fm, rank, 2)
        model.seed(2)
        model.dag_offset = off
        lsum += model.loss_and_grad(shard)[0].item()
        tot += model.flat_grads
    model.dag_offset = 0
    assert rel(lsum, l1[0].item()) < 1e-5
    assert (tot - g1).abs().max().item() < 2e-4 * g1.abs().max().item()


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
    # one 512-DAG slice of the SAME batch (global DAG indices 1536..2047, i.e. the masks the full batch drew) against the
    # CPU oracle: train mode, dropout on, ELBO + all 108 gradients (the headline shape had no oracle comparison)
    dropout_on_parity("full_size_n12_b4096_slice512", model, params, cfg, graphs[1536:2048], 1, 1536)
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
    worst = 0.0
    for name, p in model.named_parameters():
        ref = P[name].grad.numpy()
        worst = max(worst, float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-4 * scale))
    record("sachs_n11_eval", elbo_rel=rel(total.item(), t.detach()), grad_err=worst)
    assert worst < GRAD_BOUND


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
    # dropout-on gradients on the last 24 DAGs (they include the chains), several mask draws: round 1 saw 3.0e-3 on
    # encoder.layers.2.linear1.weight under one draw and < 1e-4 under others — isolated ReLU ties (tests/relu_trace.py)
    for seed in (9, 10, 11):
        dropout_on_parity(f"dropout_on[alarm n37,B=24,seed={seed}]", model, params, cfg, graphs[-24:], seed, 3)
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


def test_refused_launch_comes_back_as_an_error_code():
    """include/dvs.h code 20: a kernel launch the HIP runtime refuses (here: 200 KB of dynamic LDS on a 160 KB CU) must
    surface as a non-zero return code with the kernel's name and the HIP error string — not as success with stale results."""
    from dags_vae_search_amd import _lib as dl
    lib = dl.load()
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.dvs_debug_launch(1024, stream) == 0
    rc = lib.dvs_debug_launch(200 * 1024, stream)
    msg = lib.dvs_last_error().decode()
    assert rc == 20 and "k_debug_empty" in msg and "HIP error" in msg, (rc, msg)
    with pytest.raises(RuntimeError, match="k_debug_empty"):
        dl.check(lib, rc, "dvs_debug_launch")
    assert lib.dvs_debug_launch(1024, stream) == 0            # the failure is not sticky
    torch.cuda.synchronize()


def test_nonfinite_or_invalid_batch_leaves_weights_and_moments_untouched():
    """the reference raises inside loss_direct (pace.py:97-98) — before backward, clip and step
    (main.py:111-116) — so a NaN loss or an invalid batch leaves the model intact.  The fused step has its optimiser
    kernels enqueued before the host can know; they carry the two flags as a device-side guard (dvs_clip_adam)."""
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.train import train_batch
    cfg, params, graphs, z = load_golden("n12c12")
    model = build_model(cfg, params).train()
    f = feats_for(model, graphs)
    opt = dopt.Adam(model.parameters(), lr=1e-3).attach(model)
    for _ in range(2):
        train_batch(f, model, opt)

    def snapshot():
        st = opt.state[next(iter(opt.param_groups[0]["params"]))]
        return model.flat_params.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"]

    # (1) invalid batch: a label row that is not one-hot
    before = snapshot()
    bad = dict(f)
    bad["vertex_label_features"] = f["vertex_label_features"] * 0.5
    with pytest.raises(ValueError, match="feature invariants"):
        train_batch(bad, model, opt)
    torch.cuda.synchronize()
    after = snapshot()
    assert all(torch.equal(a, b) for a, b in zip(before[:3], after[:3])) and before[3] == after[3] == 2
    # (2) non-finite loss: exp(logvar) overflows
    with torch.no_grad():
        model.fc2.bias.fill_(1e30)
    before = snapshot()
    with pytest.raises(ValueError, match="NaN"):
        train_batch(f, model, opt)
    torch.cuda.synchronize()
    after = snapshot()
    assert all(torch.equal(a, b) for a, b in zip(before[:3], after[:3])) and before[3] == after[3] == 2
    # (3) the model is still usable: repair, step, and the step count moves on
    with torch.no_grad():
        model.fc2.bias.zero_()
    loss_value, _, _ = train_batch(f, model, opt)
    assert np.isfinite(loss_value) and opt._steps == 3 and not torch.equal(model.flat_params, after[0])


def test_fused_adam_state_dict_round_trip():
    """the fused optimiser's moments and step live in ordinary optimiser state: state_dict() -> a fresh
    optimiser on a fresh model -> load_state_dict() continues bit for bit like the uninterrupted run, and that run tracks
    torch.optim.Adam over the autograd path."""
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.train import train_batch
    cfg, params, graphs, z = load_golden("n12c12")

    def run(model, opt, steps, first_step):
        for i in range(steps):
            model._seed, model._step = 5, first_step + i            # same masks in every run
            train_batch(f, model, opt)
    a = build_model(cfg, params).train()
    f = feats_for(a, graphs)
    oa = dopt.Adam(a.parameters(), lr=1e-3).attach(a)
    run(a, oa, 3, 0)
    sd_model = {k: v.clone() for k, v in a.state_dict().items()}
    sd_opt = oa.state_dict()
    assert len(sd_opt["state"]) == 1 and sd_opt["state"][0]["step"] == 3
    assert sd_opt["state"][0]["exp_avg"].abs().sum() > 0
    sd_opt = {"state": {k: {kk: (vv.cpu().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                        for k, v in sd_opt["state"].items()}, "param_groups": sd_opt["param_groups"]}   # as torch.save/load
    run(a, oa, 2, 3)
    b = build_model(cfg, sd_model).train()
    ob = dopt.Adam(b.parameters(), lr=1e-3).attach(b)
    ob.load_state_dict(sd_opt)
    run(b, ob, 2, 3)
    assert ob._steps == 5 and torch.equal(a.flat_params, b.flat_params)
    # without the restored state the runs differ (the test would not notice a silent restart otherwise)
    c = build_model(cfg, sd_model).train()
    oc = dopt.Adam(c.parameters(), lr=1e-3).attach(c)
    run(c, oc, 2, 3)
    assert not torch.equal(a.flat_params, c.flat_params)
    # stock Adam over the autograd-wrapped kernels, same masks: same trajectory up to rounding.  Adam normalises every entry,
    # so the ~3 % of entries whose gradient is rounding noise (exact zeros of the key biases, ...) random-walk by lr per step
    # in ANY two implementations: bound those by the walk, pin the rest.
    d = build_model(cfg, params).train()
    od = torch.optim.Adam(d.parameters(), lr=1e-3)
    run(d, od, 5, 0)
    diff = (a.flat_params - d.flat_params).abs()
    assert diff.max().item() < 2 * 5 * 1e-3 and (diff < 2e-5).float().mean().item() > 0.95


def test_data_parallel_step_over_rccl_world1_equals_single_gpu_step(tmp_path):
    """train_batch(group=True) had never run through RCCL.  One rank, backend nccl (= RCCL): the two all-reduces
    (five scalars behind the forward, flat gradient behind the backward), clip-after-reduce and the guarded Adam give
    bit-identical parameters and losses to the single-GPU step taken through the same optimiser entry (norm from a pass over
    the gradient); the fused single-GPU train_batch — whose norm comes from the slab reduction's partial sums, a different
    summation order (ABI 202) — agrees to rounding."""
    import torch.distributed as dist
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.train import train_batch
    cfg, params, graphs, z = load_golden("n12c12")
    a = build_model(cfg, params).train()
    b = build_model(cfg, params).train()
    c = build_model(cfg, params).train()
    f = feats_for(a, graphs)
    oa = dopt.Adam(a.parameters(), lr=1e-3).attach(a)
    ob = dopt.Adam(b.parameters(), lr=1e-3).attach(b)
    oc = dopt.Adam(c.parameters(), lr=1e-3).attach(c)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1,
                                device_id=torch.device(DEV))
    try:
        for step in range(3):
            a.seed(step)
            b.seed(step)
            c.seed(step)
            la = train_batch(f, a, oa)
            lb = train_batch(f, b, ob, group=True)
            lc = c.loss_and_grad(f).clone()                    # forward + backward, then clip + Adam with the norm pass
            oc.step(max_grad_norm=1.0)
            assert lb[0] == lc[0].item() and lb[1].item() == lc[1].item() and lb[2].item() == lc[2].item()
            assert rel(la[0], lb[0]) < 1e-6 and rel(la[2].item(), lb[2].item()) < 1e-6
            assert lb[1].is_cuda and lb[1].dim() == 0
        assert torch.equal(c.flat_params, b.flat_params)
        assert (a.flat_params - b.flat_params).abs().max().item() < 1e-6
        bad = dict(f)
        bad["vertex_label_features"] = f["vertex_label_features"] * 0.5
        before = b.flat_params.clone()
        with pytest.raises(ValueError):
            train_batch(bad, b, ob, group=True)
        assert torch.equal(before, b.flat_params)
    finally:
        if created:
            dist.destroy_process_group()


def test_host_notification_carries_the_same_scalars_as_the_device_tail():
    """dvs_loss_forward_notify (the fused single-GPU step's early read): the pinned host words polled by read_step() equal the
    device-side tail bit for bit, the sequence word advances by one per step, the validation word is re-armed by the kernel,
    and an invalid batch still reaches the host (status bits) and the device guard (no update)."""
    from dags_vae_search_amd import optim as dopt
    from dags_vae_search_amd.train import train_batch
    cfg, params, graphs, z = load_golden("n12c12")
    model = build_model(cfg, params).train()
    opt = dopt.Adam(model.parameters(), lr=1e-4).attach(model)
    f = feats_for(model, graphs)
    model.seed(3)
    seqs = []
    for _ in range(3):
        loss_value, recon, kld = train_batch(f, model, opt)
        torch.cuda.synchronize()
        dev = model._notify_losses.cpu()        # this step's own device scalars (recon / kld handed to the caller are views of it)
        host = model._host_tail
        assert recon.is_cuda and kld.is_cuda and recon.dim() == 0      # main.py:111-118 returns 0-d device tensors
        assert torch.equal(host[:3], dev[:3]) and loss_value == float(dev[0])
        word = int(host.numpy().view("uint32")[3])
        assert (word & 0xFF) == 0 and float(dev[3]) == 0.0 and float(dev[4]) == 0.0     # no flag, no validation bit
        assert float(recon) == float(dev[1]) and float(kld) == float(dev[2])
        assert int(model._step_status.item()) == 0                      # re-armed on the device
        seqs.append(word >> 8)
    assert seqs == [seqs[0], seqs[0] + 1, seqs[0] + 2]
    before = model.flat_params.clone()
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in f.items()}
    key = next(k for k, v in bad.items() if torch.is_tensor(v) and v.dim() == 3 and v.dtype == torch.float32)
    bad[key][0, 0, :] = 0.5                                             # not one-hot any more
    with pytest.raises(ValueError):
        train_batch(bad, model, opt)
    torch.cuda.synchronize()
    assert torch.equal(before, model.flat_params)
    train_batch(f, model, opt)                                          # and the next valid step goes through
