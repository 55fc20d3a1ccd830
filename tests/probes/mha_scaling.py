"""Is the fused multi-head attention latency-bound or throughput-bound?  Kernel time of d2r_mha forward / backward against the number of
(sample, head) workgroups (batch 2 ... 64 at 12 heads), 128 and 197 tokens, fp16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d2r_amd._lib as _L
if os.environ.get("PROBE_LIB"):  # A/B of two builds of the library in one GPU call
    _L.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
from d2r_amd import functional as F
from d2r_amd._lib import KernelTimer
dev = torch.device("cuda:0")
E, H = 768, 12
for L in (128, 197):
    for B in (2, 8, 21, 32, 64):
        qkv = (0.5 * torch.randn(B, L, 3 * E, device=dev)).half().requires_grad_(True)
        fwd = lambda: F.attention_qkv(qkv, H, 64 ** -0.5)
        g = torch.randn_like(fwd())
        def fb():
            fwd().backward(g)
        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        with KernelTimer() as kt:
            for _ in range(20):
                fb()
        s = kt.summary()
        tf = sum(r["ms"] for k, r in s.items() if k.endswith("_fwd")) * 1e3 / 20
        tb = sum(r["ms"] for k, r in s.items() if k.endswith("_bwd")) * 1e3 / 20
        print("L %3d  B %2d  workgroups %4d (%.2f per CU): fwd %6.1f us  bwd %6.1f us" % (L, B, B * H, B * H / 256.0, tf, tb), flush=True)
