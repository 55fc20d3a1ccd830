import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M, functional as F
from d2r_amd.config import TextConfig
from d2r_amd.params import ParamStore
gpu = torch.device("cuda:0")
torch.manual_seed(3)
layer = M.BertLayer(TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
class Wrap(M.D2RModule):
    def __init__(self, layer):
        super().__init__(); self.layer = layer
model = Wrap(layer).to(gpu); model.set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
B, L = 3, 37
x0 = torch.randn(B, L, 768, device=gpu).bfloat16(); gy = torch.randn(B, L, 768, device=gpu).bfloat16()
mask = torch.zeros(B, L, device=gpu)
res = {}
for rep in range(2):
    for comp in (False, True):
        for wg in (False, True):
            M.COMPOSITE_LAYERS, F.WGRAD_STREAMS = comp, wg
            store.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = layer(x, mask); y.backward(gy); torch.cuda.synchronize()
            res[(comp, wg, rep)] = store.flat_g.clone()
base = res[(False, False, 0)]
for k, g in res.items():
    worst = max(((float((g[o:o+n] - base[o:o+n]).norm() / (base[o:o+n].norm() + 1e-3 * base.norm())), nm) for nm, p, o, n, _ in store.entries))
    print(k, "worst rel vs per-op/no-wgrad:", worst)
