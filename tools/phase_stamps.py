#!/usr/bin/env python3
"""Per-phase time budget of the chained kernels from the diagnostic build (make -C dags_vae_search_amd/csrc stamps).

    gpurun -- 'python tools/phase_stamps.py [--batch 4096] > gpurun_out/phase_stamps.txt'

Runs a few n=12 train steps on libdvs_hip_stamps.so (NEVER loaded by the package itself), reads the s_memtime stamps lane 0
of every wave left at fixed points of every phase of the last step, and prints, per phase, the mean over workgroups of
  stage   entry -> DAG loop starts (first phase of a launch only: later phases are staged by the previous phase's tail)
  loop    the DAG loop, for the older wave group (waves 0-3) and the younger one (4-7)
  issue   issuing the next phase's prefetch loads
  waitA   waiting at the barrier behind the loop
  epi     epilogue (gradient slabs) up to the pre-commit barrier
  commit  barrier + registers -> LDS
  total   entry -> entry of the next phase
in shader cycles (s_memtime ticks)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    from dags_vae_search_amd import _lib as dl
    dl.LIB_NAME = os.environ.get("DVS_STAMPS_LIB", os.path.join(REPO, "tools", "_diag", "libdvs_hip_stamps.so"))      # variant builds of the diagnostic library
    from dags_vae_search_amd import PaceVaeV3, optim as dopt, prepare_features
    from dags_vae_search_amd.synthetic import synthetic_dags
    from dags_vae_search_amd.train import train_batch
    lib = dl.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model = PaceVaeV3(12, 12, 32, 8, 3, 64, 32, 32, 0.15).to(dev)
    opt = dopt.Adam(model.parameters(), lr=1e-4).attach(model)
    f = prepare_features(synthetic_dags(12, 12, args.batch, seed=42), 15, 15)
    f = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in f.items()}
    for _ in range(args.steps):
        train_batch(f, model, opt)
    torch.cuda.synchronize()
    WGS, WAVES, PH, IDS = 256, 8, 32, 8
    for name, fn in (("forward", lib.dvs_debug_read_stamps_fwd), ("backward", lib.dvs_debug_read_stamps_bwd)):
        buf = np.zeros(WGS * WAVES * PH * IDS, np.uint64)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert fn(buf.ctypes.data, buf.nbytes, 0) == 0
        t = buf.reshape(WGS, WAVES, PH, IDS).astype(np.int64)
        print(f"== {name}: cycles, mean over {WGS} workgroups (older group = wave 0, younger = wave 4) ==")
        print("phase   stage  loop_old loop_yng   issue   waitA_old waitA_yng     epi  commit   total")
        if name == "forward" and t[:, 0, 6, 3].any():        # latent block chained behind the encoder (ids 0..3, dvs_latent.h)
            lt = t[:, :, 6, :4]
            print("latent phase (wave 0 / wave 7): contraction %d / %d, partial sums + barrier %d / %d, epilogue %d / %d" % (
                np.mean(lt[:, 0, 1] - lt[:, 0, 0]), np.mean(lt[:, 7, 1] - lt[:, 7, 0]), np.mean(lt[:, 0, 2] - lt[:, 0, 1]),
                np.mean(lt[:, 7, 2] - lt[:, 7, 1]), np.mean(lt[:, 0, 3] - lt[:, 0, 2]), np.mean(lt[:, 7, 3] - lt[:, 7, 2])))
            t[:, :, 6, :] = 0
        for ph in range(PH):
            if not t[:, 0, ph, 1].any():
                continue
            e0, l0, l1 = t[:, :, ph, 0], t[:, :, ph, 1], t[:, :, ph, 2]
            i3, a4, e5, c6 = t[:, :, ph, 3], t[:, :, ph, 4], t[:, :, ph, 5], t[:, :, ph, 6]
            nxt = t[:, :, ph + 1, 0] if ph + 1 < PH and t[:, 0, ph + 1, 0].any() and (ph + 1) % 9 != 0 else None
            m = lambda a: float(np.mean(a))
            has_tail = i3[:, 0].any()
            stage = m(l0[:, 0] - e0[:, 0])
            loop_o, loop_y = m(l1[:, 0] - l0[:, 0]), m(l1[:, 4] - l0[:, 4])
            issue = m(i3[:, 0] - l1[:, 0]) if has_tail else 0.0
            wa_o = m(a4[:, 0] - i3[:, 0]) if has_tail else 0.0
            wa_y = m(a4[:, 4] - i3[:, 4]) if has_tail else 0.0
            epi = m(e5[:, 0] - a4[:, 0]) if e5[:, 0].any() else 0.0
            base = e5 if e5[:, 0].any() else a4
            commit = m(c6[:, 0] - base[:, 0]) if c6[:, 0].any() else 0.0
            total = m(nxt[:, 0] - e0[:, 0]) if nxt is not None else 0.0
            s7 = t[:, :, ph, 7]
            extra = ""
            if name == "backward" and s7[:, 0].any():       # id 7 = end of the FIRST DAG round (backward phases)
                extra = (f"   round 1 old/yng {m(s7[:, 0] - l0[:, 0]):6.0f} /{m(s7[:, 4] - l0[:, 4]):6.0f}"
                         f"   later rounds {m(l1[:, 0] - s7[:, 0]):6.0f} /{m(l1[:, 4] - s7[:, 4]):6.0f}")
            elif s7[:, 0].any():                              # DVS_STAMPS_ICACHE builds only (forward)
                extra = f"   commit repeated: {m(s7[:, 0] - c6[:, 0]):6.0f}"
            print(f"{ph:5d} {stage:7.0f} {loop_o:9.0f} {loop_y:8.0f} {issue:7.0f} {wa_o:11.0f} {wa_y:9.0f} {epi:7.0f} {commit:7.0f} {total:7.0f}" + extra)


if __name__ == "__main__":
    main()
