"""Bisect which part of the step breaks hipGraph capture.  usage: graph_bisect.py <variant>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, ParamStore
from d2r_amd import functional as Fn

variant = sys.argv[1]
dev = torch.device("cuda:0")
tc = TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32)
torch.manual_seed(0)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc)
model.to(dev).set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
opt = FusedAdamW(store, lr=1e-3)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30000, (4, 16), generator=g); ids[:, 0] = 101
batch = tuple(t.to(dev) for t in (ids, torch.ones(4, 16, dtype=torch.long), torch.zeros(4, 16, dtype=torch.long),
                                  torch.randint(0, 3, (4,), generator=g), torch.randn(4, 3, 64, 64, generator=g)))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        loss, _ = model(*batch); loss.backward(); opt.step(); opt.zero_grad()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
x = torch.randn(256, 256, device=dev)
with torch.cuda.graph(graph):
    if variant == "torch":
        y = x @ x
    elif variant == "cast":
        y = Fn.cast(x, torch.bfloat16)
    elif variant == "adam":
        opt.step_captured()
    elif variant == "zero":
        opt.zero_grad()
    elif variant == "fwd_nograd":
        with torch.no_grad():
            loss, _ = model(*batch)
    elif variant == "fwd":
        loss, _ = model(*batch)
    elif variant == "fwdbwd":
        loss, _ = model(*batch); loss.backward()
    elif variant == "all":
        loss, _ = model(*batch); loss.backward(); opt.step_captured(); opt.zero_grad()
print(variant, "captured", flush=True)
opt.stage_hyper(); graph.replay(); torch.cuda.synchronize()
print(variant, "replayed OK", flush=True)
