"""Where do the device-to-device copies of one training step come from?  (aten::copy_ call sites by Python stack)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, ParamStore
import d2r_amd
d2r_amd.configure_runtime()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L = 32, 128
tc = TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc).to(dev)
model.set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
opt = FusedAdamW(store, lr=3e-5)
ids = torch.randint(1000, 30000, (B, L), device=dev); ids[:, 0] = 101
batch = (ids, torch.ones(B, L, dtype=torch.long, device=dev), torch.zeros(B, L, dtype=torch.long, device=dev),
         torch.randint(0, 3, (B,), device=dev), torch.randn(B, 3, 224, 224, device=dev))

def step():
    loss, _ = model(*batch)
    loss.backward()
    opt.step()
    opt.zero_grad()

for _ in range(2):
    step()
torch.cuda.synchronize()
counts = collections.Counter()
orig = torch.Tensor.copy_
import traceback

def site():
    st = traceback.extract_stack()[:-2]
    for fr in reversed(st):
        if "d2r_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "autograd/other"

with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if ("copy" in e.name.lower() or "Memcpy" in e.name or "clone" in e.name or "contiguous" in e.name or "cat" == e.name[-3:]
                                   or "fill" in e.name.lower() or "zero" in e.name.lower() or "add" in e.name.lower()) and e.device_type == torch.autograd.DeviceType.CPU]
c = collections.Counter()
for e in ev:
    stack = [s for s in (e.stack or []) if "d2r_amd" in s]
    c[(e.name, stack[0].strip()[-90:] if stack else "(no d2r_amd frame: autograd engine)")] += 1
for (name, where), n in c.most_common(40):
    print(f"{n:4d}  {name:28s} {where}")
