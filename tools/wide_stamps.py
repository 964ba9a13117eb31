import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from dags_vae_search_amd import _lib as dl
dl.LIB_NAME = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_diag", "libdvs_hip_stamps.so")   # absolute: outside the package
from dags_vae_search_amd import PaceVaeV3, optim as dopt, prepare_features
from dags_vae_search_amd.synthetic import synthetic_dags
from dags_vae_search_amd.train import train_batch
lib = dl.load()
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = PaceVaeV3(37, 37, 32, 8, 3, 64, 32, 32, 0.15).to(dev)
opt = dopt.Adam(model.parameters(), lr=1e-4).attach(model)
f = prepare_features(synthetic_dags(37, 37, 2048, seed=42, density_limit=0.2), 40, 40)
f = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in f.items()}
for _ in range(2): train_batch(f, model, opt)
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
fn = {"bwd": lib.dvs_debug_read_stamps_wb, "heads": lib.dvs_debug_read_stamps_wc}.get(which, lib.dvs_debug_read_stamps_w); fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
buf = np.zeros(256*8*8, np.uint64); fn(buf.ctypes.data, buf.nbytes, 1)
train_batch(f, model, opt); torch.cuda.synchronize()
fn(buf.ctypes.data, buf.nbytes, 0)
t = buf.reshape(256, 8, 8).astype(np.float64) / (9 * 8)     # 9 launches x 8 DAGs per workgroup
if which == "heads":
    t = buf.reshape(256, 8, 8).astype(np.float64) / 8      # one launch each, 8 DAGs per workgroup
    print("cycles per DAG (mean over WGs): k_embed_bwd_w 0 tile stage 1 barrier 2 scatters 3 barrier | k_loss_bwd_w 4 heads, U, V 5 barrier + pass 1 6 barrier + pass 2 7 barrier + edge matrices, d h, store + barrier")
    for w in range(4): print("wave", w, np.round(t[:, w, :8].mean(0)).astype(int).tolist(), "sums", int(t[:, w, 0:4].mean(0).sum()), int(t[:, w, 4:8].mean(0).sum()))
elif which == "bwd":
    print("k_attn_bwd_w, cycles per DAG (mean over WGs): 0 prologue(per launch/72) 1 fill: loads issued 2 fill: dO tile 3 core 4 (core end) 5 barrier 6 stores + parks | fill: q / k / v parked 7 barrier + dWo + barrier")
    for w in range(8): print("wave", w, np.round(t[:, w, :8].mean(0)).astype(int).tolist(), "sum", int(t[:, w, 1:8].mean(0).sum()))
else:
    print("k_attn_fwd_w, cycles per DAG (mean over WGs): k: 0 prologue(per launch/72) 1 stage1 2 barrier 3 stage2 4 barrier 5 stage3 6 barrier")
    for w in range(8): print("wave", w, np.round(t[:, w, :7].mean(0)).astype(int).tolist(), "sum", int(t[:, w, 1:7].mean(0).sum()))
