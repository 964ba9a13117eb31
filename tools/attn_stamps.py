#!/usr/bin/env python3
"""Inner time budget of the one-tile attention forward DAG loop (diagnostic build, make stamps): cycles per DAG, mean over
workgroups, for an older-group wave (0) and a younger-group wave (4).  gpurun -- 'python tools/attn_stamps.py'"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dags_vae_search_amd import _lib as dl
dl.LIB_NAME = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_diag", "libdvs_hip_stamps.so")   # absolute: outside the package
from dags_vae_search_amd import PaceVaeV3, optim as dopt, prepare_features
from dags_vae_search_amd.synthetic import synthetic_dags
from dags_vae_search_amd.train import train_batch
lib = dl.load()
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = PaceVaeV3(12, 12, 32, 8, 3, 64, 32, 32, 0.15).to(dev)
opt = dopt.Adam(model.parameters(), lr=1e-4).attach(model)
f = prepare_features(synthetic_dags(12, 12, 4096, seed=42), 15, 15)
f = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in f.items()}
for _ in range(3): train_batch(f, model, opt)
torch.cuda.synchronize()
fn = lib.dvs_debug_read_stamps_attn; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
buf = np.zeros(256*8*8, np.uint64); fn(buf.ctypes.data, buf.nbytes, 1)
train_batch(f, model, opt); torch.cuda.synchronize()
fn(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(256, 8, 8).astype(np.float64) / (9 * 2)     # 9 attention phases x 2 DAGs per wave
names = ["load+LN (vmcnt 0)", "split + q,k,v products", "scores + softmax", "prob dropout", "P V", "split + out-proj", "post dropout, residual, LN stats, store", "loop overhead"]
print("attention forward, cycles per DAG (mean over 256 workgroups)")
for k in range(8): print(f"  {names[k]:42s} old group (wave 0) {t[:,0,k].mean():7.0f}   young group (wave 4) {t[:,4,k].mean():7.0f}")
print(f"  {'sum':42s} {t[:,0,:].sum(1).mean():7.0f} {t[:,4,:].sum(1).mean():7.0f}")
