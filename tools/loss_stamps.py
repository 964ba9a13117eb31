#!/usr/bin/env python3
"""Inner time budget of k_loss_fwd (diagnostic build, make stamps): cycles per wave and launch, mean over workgroups.
gpurun -- 'python tools/loss_stamps.py'"""
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
fn = lib.dvs_debug_read_stamps_loss; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
buf = np.zeros(256*8*8, np.uint64); fn(buf.ctypes.data, buf.nbytes, 1)
train_batch(f, model, opt); torch.cuda.synchronize()
fn(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(256, 8, 8).astype(np.float64)
names = ["staging + barrier", "tile load + LayerNorm", "node head", "U, V products + park", "pair walk", "reduction + store"]
print("k_loss_fwd, cycles per wave and launch (2 DAGs per wave), mean over 256 workgroups")
for k in range(6): print(f"  {names[k]:28s} wave 0 {t[:,0,k].mean():7.0f}   wave 4 {t[:,4,k].mean():7.0f}")
print(f"  {'sum':28s} {t[:,0,:].sum(1).mean():7.0f} {t[:,4,:].sum(1).mean():7.0f}")

fb = lib.dvs_debug_read_stamps_lossb; fb.restype = ctypes.c_int; fb.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
bufb = np.zeros(256*4*12, np.uint64); fb(bufb.ctypes.data, bufb.nbytes, 1)
train_batch(f, model, opt); torch.cuda.synchronize()
fb(bufb.ctypes.data, bufb.nbytes, 0)
t = bufb.reshape(256, 4, 12).astype(np.float64)
names = ["staging + barrier + init", "tile load + LN + transpose", "node head fwd + bwd", "U, V recompute + park", "pass 1 (i walks j)",
         "pass 2 (j walks i)", "transposes + dWa, dWb", "d h products, LN backward, store", "closing barrier", "slab epilogue"]
print("k_loss_bwd, cycles per wave and launch (4 DAGs per wave), mean over 256 workgroups")
for k in range(10): print(f"  {names[k]:34s} wave 0 {t[:,0,k].mean():7.0f}   wave 3 {t[:,3,k].mean():7.0f}")
print(f"  {'sum':34s} {t[:,0,:].sum(1).mean():7.0f} {t[:,3,:].sum(1).mean():7.0f}")
