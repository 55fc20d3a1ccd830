import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d2r_amd import functional as F
dev = torch.device("cuda:0")
for (B, L) in ((32, 197), (32, 128)):
    nc, D = 6, 768
    gates = torch.rand(B, nc, nc, device=dev).requires_grad_(True)
    embs = [torch.randn(B, L, D, device=dev).bfloat16().requires_grad_(True) for _ in range(nc)]
    embs[1] = torch.randn(B, D, device=dev).bfloat16().requires_grad_(True)
    embs[5] = torch.randn(B, D, device=dev).bfloat16().requires_grad_(True)
    probs, outs = F.route_aggregate(gates, *embs)
    gs = [torch.randn_like(o) for o in outs]
    gp = torch.randn_like(probs)
    def bwd():
        torch.autograd.grad([probs] + outs, [gates] + embs, [gp] + gs, retain_graph=True)
    for _ in range(3): bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bwd()
    e1.record(); torch.cuda.synchronize()
    print(f"route_aggregate bwd P=6 B={B} L={L}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per backward (both launches + host)")
