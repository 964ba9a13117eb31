#!/usr/bin/env python3
"""Throughput of the batched device-side generation (PaceVaeV3.decode): graphs/s for one decode of a
batch of latent vectors, next to the oracle's CPU restatement of the reference loop on a bounded sample.
    python bench_decode.py [--n 12 --card 12 --batch 4096 --reps 5]
Prints one JSON line.  (The driver's metric is bench.py; this is the measurement of the decode row.)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--card", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()
    from dags_vae_search_amd import PaceVaeV3
    from oracle import decode as odec
    from oracle import pace_oracle as po
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(min(cores, 16))
    torch.manual_seed(42)
    model = PaceVaeV3(args.n, args.card, 32, 8, 3, 64, 32, 32, 0.15).to("cuda:0").eval()
    z = torch.randn(args.batch, 32, device="cuda:0")
    eng = model._eng()
    shape = eng.shape(args.batch, training=False, seed=1)
    eng.workspace(args.batch, z.device)
    for _ in range(2):
        eng.decode(shape, model.flat_params, z, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        eng.decode(shape, model.flat_params, z, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    t0 = time.perf_counter()
    model.decode(z, strict=False)
    dt_host = time.perf_counter() - t0
    cfg = po.PaceConfig(n=args.n, card=args.card)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    S = args.cpu_sample
    U = np.random.default_rng(0).random((S, cfg.N, cfg.N)).astype(np.float32)
    t0 = time.perf_counter()
    odec.decode(params, cfg, z[:S].cpu(), U)
    dc = time.perf_counter() - t0
    print(json.dumps({"metric": f"graphs/sec decode, n={args.n} batch {args.batch}", "value": args.batch / dt,
                      "unit": "graphs/s", "ms_per_decode": dt * 1e3,
                      "with_host_conversion_graphs_per_s": args.batch / dt_host,
                      "cpu_baseline": {"value": S / dc, "unit": "graphs/s", "kind": "port", "cores": torch.get_num_threads(),
                                       "sample": f"one oracle decode of {S} latents"}}))


if __name__ == "__main__":
    main()
