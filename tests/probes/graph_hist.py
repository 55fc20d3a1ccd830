"""Histogram of autograd node types in one training step's backward graph (which torch-native plumbing ops remain?)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
tc = TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=2, image_size=224, patch_size=16)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc)
model.to(dev).set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
batch = bench.synthetic_batch(4, 128, 224, dev, 0)
loss, _ = model(*batch)
seen, hist, stack = set(), collections.Counter(), [loss.grad_fn]
multi = collections.Counter()
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    hist[type(fn).__name__] += 1
    for nxt, _ in fn.next_functions:
        if nxt is not None:
            multi[nxt] += 1
            stack.append(nxt)
for k, v in hist.most_common():
    print(f"{v:5d}  {k}")
fan = collections.Counter()
for fn, c in multi.items():
    if c > 1 and type(fn).__name__ != "AccumulateGrad":
        fan[(type(fn).__name__, c)] += 1
print("nodes whose output feeds several consumers (gradient accumulation adds):")
for (k, c), v in sorted(fan.items(), key=lambda x: -x[1] * x[0][1]):
    print(f"   {v:3d} x {k} with {c} consumers")
