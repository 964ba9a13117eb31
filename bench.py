#!/usr/bin/env python3
"""bench.py — DAGs/sec of the PACE-VAE train step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full train step on one batch of synthetic DAGs whose reference-layout dense features are already
resident in HBM: dvs_pack_features -> forward -> backward -> (RCCL SUM all-reduce of the flat gradient when N > 1)
-> fused clip_grad_norm_(1.0) + Adam(lr 1e-4), in train mode with dropout 0.15 — i.e. train_batch() of
experiments/03_synthetic_12/main.py:95-118 including its loss.item() host read.  Workload: BASELINE configs[1]/metric
shape: synthetic n=12, card=12, batch 4096 per GPU (weak scaling: global batch = 4096 * N, one gradient all-reduce).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (per-kernel durations measured with
HIP events on the launch stream inside libdvs_hip.so) and "cpu_baseline" (the oracle = CPU port of the reference step,
timed on this box's host cores on a bounded sample; rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

N_VERT, CARD = 12, 12            # --workload changes these (the driver's default run is the BASELINE metric shape)
WORKLOADS = {   # name: (n, card, default per-GPU batch, edge-density limit, description)
    "n12": (12, 12, 4096, 0.4, "synthetic_v12 DAGs (n=12, card=12, N=15 tokens)"),
    "asia": (8, 8, 4096, 0.4, "asia-shaped synthetic DAGs (n=8, card=8, N=11 tokens)"),
    "sachs": (11, 11, 8192, 0.4, "sachs-shaped synthetic DAGs (n=11, card=11, N=14 tokens)"),
    "alarm": (37, 37, 2048, 0.2, "alarm-size synthetic DAGs (n=37, card=37, N=40 tokens, tiled wide path)"),
}
DENSITY = 0.4
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix/vector peak
PEAK_HBM_GBS = 8000.0


def algo_flops_per_dag(N: int, C: int):
    """Algorithmic FLOPs per DAG of each kernel (multiply-add = 2), counted on the TRUE token count N (not the
    16-token tile) and on the reference's formulation (the edge head in its factored U_i+V_j form).
    proj = one N x 64 x 64 projection; core = QK^T + PV of all 8 heads."""
    proj = 2.0 * N * 64 * 64
    core = 2.0 * 2 * N * N * 64
    pairs = (N - 1) * (N - 2) / 2
    node = 2.0 * N * (64 * 32 + 32 * C)
    edge = 2 * proj + pairs * 64 * 3
    emb = 2.0 * N * 64 * 32 + N * 64 * 3
    latent_f = 2.0 * N * 64 * 64 + 2.0 * 32 * N * 64
    f = {
        "k_embed_fwd": emb, "k_attn_fwd": 4 * proj + core, "k_ffn_fwd": 2 * proj, "k_latent_fwd": latent_f,
        "k_loss_fwd": node + edge,
        # backward kernels: recompute (1x forward) + gradients (2x forward) of the ops they own
        "k_loss_bwd": 3 * (node + edge), "k_ffn_bwd": 3 * 2 * proj,
        "k_attn_bwd": (3 * proj + core) + 2 * (proj + core) + proj,     # recompute q,k,v,P ; dWo,dO + core bwd ; O
        "k_proj_bwd<3>": 2 * 3 * proj, "k_proj_bwd<2>": 2 * 2 * proj, "k_proj_bwd<1>": 2 * proj,
        "k_latent_bwd": 2.0 * 32 * N * 64 + 2.0 * N * 64 * 64, "k_fc_dw": latent_f, "k_embed_bwd": 2 * emb,
    }
    # chained launches (one-tile path): the sum over the sublayers one launch walks.  The decoder's 18 backward phases
    # go out as two launches of 9, so its per-launch figure is half the decoder total.
    f["k_fwd_stack<0>"] = 3 * (f["k_attn_fwd"] + f["k_ffn_fwd"])
    f["k_fwd_stack<1>"] = 3 * (2 * f["k_attn_fwd"] + f["k_ffn_fwd"])
    f["k_bwd_stack<0>"] = 3 * (f["k_ffn_bwd"] + 2 * f["k_attn_bwd"] + f["k_proj_bwd<1>"] + f["k_proj_bwd<2>"]
                               + f["k_proj_bwd<3>"]) / 2
    f["k_bwd_stack<1>"] = 3 * (f["k_ffn_bwd"] + f["k_attn_bwd"] + f["k_proj_bwd<3>"])
    return f


STACK_PHASES = {   # what one launch of a chained kernel walks (csrc/k_forward.hip, k_backward.hip)
    "k_fwd_stack<0>": "encoder forward: 3 x (attention, FFN)",
    "k_fwd_stack<1>": "decoder forward: 3 x (self-attention, cross-attention, FFN)",
    "k_bwd_stack<0>": "decoder backward, 9 of its 18 phases per launch: 3 x (FFN, cross-attention core, q projection, "
                      "k/v projections, self-attention core, q/k/v projections)",
    "k_bwd_stack<1>": "encoder backward: 3 x (FFN, attention core, q/k/v projections)",
}


def bytes_per_dag(N: int, C: int, P: int, B_local: int) -> float:
    """SURVEY.md §8d algorithmic bytes per DAG: consumed reference-layout features + amortised parameter traffic."""
    return N * C * 4 + 2 * N * N * 4 + 8 * N * N + 9.0 * P * 4 / B_local


def pmc_traffic_bytes(kernel: str, batch: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/r01_pmc_hbm.csv: FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), scaled to this batch; None if the
    profile is absent or does not list the kernel.  Counters cannot be collected from inside the timed process."""
    path = os.path.join(REPO, "profiles", "r01_pmc_hbm.csv")
    try:
        for line in open(path):
            parts = line.strip().split(",")
            if len(parts) == 4 and parts[0].replace("void ", "") == kernel:
                return float(parts[3]) * batch          # bytes per DAG x DAGs per launch
    except OSError:
        pass
    return None


def make_batch(batch: int, seed: int, device):
    from dags_vae_search_amd import prepare_features
    from dags_vae_search_amd.synthetic import synthetic_dags
    graphs = synthetic_dags(N_VERT, CARD, batch, seed=seed, density_limit=DENSITY)
    feats = prepare_features(graphs, N_VERT + 3, CARD + 3)
    dev_feats = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in feats.items()}
    return graphs, feats, dev_feats


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box gives a
    16-core share of a much larger host; os.cpu_count() there reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


def cpu_baseline(graphs, feats, steps: int = 3):
    """Reference CPU path = oracle port of train_batch (train mode, dropout 0.15, Adam), all usable host cores."""
    from oracle import pace_oracle as po
    cores = min(host_cores(), 32)
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads (os.cpu_count()={os.cpu_count()})")
    cfg = po.PaceConfig(n=N_VERT, card=CARD)
    torch.manual_seed(42)
    tr = po.OracleTrainer(cfg, po.init_params(cfg, seed=42))
    f = {k: v for k, v in feats.items()}
    tr.step(f)                       # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(f)
    dt = (time.perf_counter() - t0) / steps
    B = len(graphs)
    return {"value": B / dt, "unit": "DAGs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed train steps (+1 warm-up) of the oracle on the same n={N_VERT} card={CARD} "
                      f"B={B} batch, torch CPU fp32, {cores} threads", "ms_per_step": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="DAGs per GPU (default: the workload's)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="n12",
                    help="n12 = the BASELINE metric shape (default); the others are the remaining BASELINE configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the RCCL code path even with one rank (checks init / all-reduce plumbing on a 1-GPU box)")
    args = ap.parse_args()
    global N_VERT, CARD, DENSITY
    N_VERT, CARD, default_batch, DENSITY, workload_desc = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = default_batch

    # libraries (RCCL's version banner, ...) write to fd 1: keep stdout clean for the ONE JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from dags_vae_search_amd import PaceVaeV3, optim as dopt
    from dags_vae_search_amd import _lib as dl
    from dags_vae_search_amd.train import train_batch

    torch.set_num_threads(min(host_cores(), 16))
    torch.manual_seed(42)          # experiments/03_synthetic_12/main.py:122-124: same initial weights on every rank
    model = PaceVaeV3(max_num_vertices=N_VERT, vertex_label_cardinality=CARD, vertices_embedding_size=32, num_heads=8,
                      num_layers=3, ff_hidden_size=64, latent_layer_size=32, fc_hidden=32, dropout=0.15).to(device)
    model.seed(42)
    model.dag_offset = rank * args.batch
    opt = dopt.Adam(model.parameters(), lr=1e-4).attach(model)
    graphs, feats, dev_feats = make_batch(args.batch, seed=42 + rank, device=device)
    group = True if distributed else None

    def step():
        return train_batch(dev_feats, model, opt, max_grad_norm=1.0, group=group)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log("inputs ready; warm-up")
    for _ in range(args.warmup):
        step()
    sync()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_value, _, _ = step()
    sync()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    global_batch = args.batch * world
    value = global_batch * args.steps / dt

    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    # ---- per-kernel durations (HIP events on the launch stream), separate untimed steps --------------------------------
    lib = dl.load()
    lib.dvs_profile_enable(1)
    prof_steps = 3
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = dl.profile_collect(lib)
    lib.dvs_profile_enable(0)
    sync()

    if rank == 0:
        N, C = N_VERT + 3, CARD + 3
        P = sum(p.numel() for p in model.parameters())
        flops = algo_flops_per_dag(N, C)
        kern = {k: {"launches_per_step": c // prof_steps, "avg_us": 1e3 * ms / c, "ms_per_step": ms / prof_steps}
                for k, (c, ms) in prof.items()}
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        dom_key = dom[:-2] if dom.endswith("_w") else dom          # wide-path kernels: same algorithmic work
        ach = flops.get(dom_key, 0.0) * args.batch / (kern[dom]["avg_us"] * 1e-6) / 1e12
        step_flops = 3e6 * {8: 5.71, 11: 7.74, 12: 8.46, 37: 34.0}[N_VERT]      # 3 x forward MFLOP per DAG
        roofline = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic_bytes(dom, args.batch),
                    "avg_launch_us": kern[dom]["avg_us"], "launches_per_step": kern[dom]["launches_per_step"],
                    "algorithmic_flops_per_dag": flops.get(dom_key, 0.0),
                    **({"phases": STACK_PHASES[dom_key]} if dom_key in STACK_PHASES else {}),
                    "whole_step": {"tflops": value / world * step_flops / 1e12,
                                   "frac_f32_mfma_peak": value / world * step_flops / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                   "hbm_algorithmic_GBs": value / world * bytes_per_dag(N, C, P, args.batch) / 1e9,
                                   "frac_hbm_peak": value / world * bytes_per_dag(N, C, P, args.batch) / 1e9 / PEAK_HBM_GBS,
                                   "gpu_kernel_ms_per_step": sum(k["ms_per_step"] for k in kern.values())}}
        out = {
            "metric": "DAGs/sec VAE+predictor train step, n=12 batch 4096" if args.workload == "n12" else
                      f"DAGs/sec VAE+predictor train step, n={N_VERT} batch {args.batch}",
            "value": value, "unit": "DAGs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_desc + ", PACE-VAE train step: pack + fwd + bwd"
                                   " + clip_grad_norm_(1.0) + Adam(1e-4), train mode dropout 0.15",
                       "per_gpu_batch": args.batch, "global_batch": global_batch,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "last_loss_per_dag": loss_value / global_batch},
            "roofline": roofline,
            "kernels": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(graphs, feats)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
