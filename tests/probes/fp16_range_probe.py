"""Which activations / activation gradients of the fp16 path live in fp16's subnormal range (|x| < 6.1e-5: precision lost)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import d2r_oracle as O
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore

gpu = torch.device("cuda:0")
S = float(os.environ.get("LOSS_SCALE", "1"))
cfg = O.OracleConfig(text_layers=2, vision_layers=2, image_size=64, patch_size=32, DR_step=4, num_cells=4, num_classes=7)
sd = O.seeded_state_dict(cfg, seed=21, router_bias="normal")
model = M.UnimoModelF(default_args(DR_step=4, num_cells=4), VisionConfig(num_hidden_layers=2, image_size=64, patch_size=32),
                      TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), num_classes=7)
model.load_state_dict(sd, strict=True)
M.COMPOSITE_ROUTING = False
M.COMPOSITE_LAYERS = False
model.to(gpu).set_compute_dtype(torch.float16).train()
store = ParamStore(model, torch.float16)
batch = O.synthetic_batch(cfg, 3, 10, seed=4)
rows = []


def stats(t):
    a = t.detach().float().abs().flatten()
    nz = a[a > 0]
    if nz.numel() == 0:
        return None
    return float(nz.median()), float((nz < 6.1e-5).float().mean()), float(a.max()), float((a == 0).float().mean())


def fhook(name):
    def h(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for i, o in enumerate(outs):
            if isinstance(o, (tuple, list)):
                o = o[0] if len(o) and torch.is_tensor(o[0]) else None
            if torch.is_tensor(o) and o.dtype == torch.float16:
                st = stats(o)
                if st:
                    rows.append(("fwd", f"{name}[{i}]", *st))
                if o.requires_grad:
                    o.register_hook(lambda g, name=name, i=i: rows.append(("bwd", f"{name}[{i}]", *(stats(g) or (0, 0, 0, 1)))))
    return h


for n, m in model.named_modules():
    if n and n.count(".") <= 5:
        m.register_forward_hook(fhook(n))
loss, _ = model(*[t.to(gpu) for t in batch])
(loss * S).backward()
torch.cuda.synchronize()
print(f"loss scale {S}")
for kind, name, med, sub, mx, zero in rows:
    if sub > 0.02 or mx > 3e4:
        print(f"{kind} {name:70s} median|x| {med:.2e} subnormal fraction {sub:.3f} max {mx:.2e} zeros {zero:.3f}")
