"""Histogram of the weight-gradient GEMMs of one C2 step: which are launched on the spot, which are deferred."""
import collections, sys
sys.path.insert(0, "/root/repo")
import torch
import d2r_amd
from d2r_amd import functional as F, modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, ParamStore
sys.argv = [sys.argv[0]]
import bench
d2r_amd.configure_runtime()
dev = torch.device("cuda:0")
torch.manual_seed(2023)
model = M.UnimoModelF(default_args(DR_step=3), VisionConfig(), TextConfig())
model.to(dev).set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
opt = FusedAdamW(store, lr=3e-5)
batch = bench.synthetic_batch(32, 128, 224, dev, seed=0)
def step():
    loss, _ = model(*batch); loss.backward(); opt.step(); opt.zero_grad()
step(); torch.cuda.synchronize()
direct, deferred = collections.Counter(), collections.Counter()
g0, d0 = F.gemm, F._defer_wgrad_raw
def gemm(layout, Mm, N, K, *a, **k):
    if layout == F.GEMM_TN:
        direct[(Mm, N, K, k.get("dbias") is not None)] += 1
    return g0(layout, Mm, N, K, *a, **k)
def defer(dt, N, K, Mm, *a, **k):
    deferred[(N, K, Mm)] += 1
    return d0(dt, N, K, Mm, *a, **k)
F.gemm, F._defer_wgrad_raw = gemm, defer
step(); torch.cuda.synchronize()
print("launched on the spot (out rows, out cols, reduction length, dbias):")
for k, v in sorted(direct.items(), key=lambda kv: -kv[1]): print("  ", k, v)
print("deferred (out rows, out cols, reduction length):")
for k, v in sorted(deferred.items(), key=lambda kv: -kv[1]): print("  ", k, v)
