"""Isolated timing of the attention cores (fused MHA, cross-modal attention) at the bench shapes.
usage: python tests/probes/bench_attention.py   -> us per launch and algorithmic GB/s (q+k+v+o bytes, x2 for bwd)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d2r_amd._lib as _L
if os.environ.get("PROBE_LIB"):  # A/B of two builds of the library in one GPU call
    _L.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
from d2r_amd import functional as F

dev = torch.device("cuda:0")
B, E = 32, 768


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, Lq, Lk, H, packed in (("bert self-attn", 128, 128, 12, True), ("clip self-attn", 197, 197, 12, True),
                                ("imrc text 16h", 128, 128, 16, False), ("imrc image 16h", 197, 197, 16, False),
                                ("xattn text->image", 128, 197, 1, False), ("xattn image->text", 197, 128, 1, False)):
    scale = 100.0 / 768 ** 0.5 if H == 1 else (E // H) ** -0.5
    if packed:
        qkv = (0.5 * torch.randn(B, Lq, 3 * E, device=dev)).bfloat16().requires_grad_(True)
        fwd = lambda: F.attention_qkv(qkv, H, scale)
    else:
        q = (0.5 * torch.randn(B, Lq, E, device=dev)).bfloat16().requires_grad_(True)
        k = (0.5 * torch.randn(B, Lk, E, device=dev)).bfloat16().requires_grad_(True)
        v = torch.randn(B, Lk, E, device=dev).bfloat16().requires_grad_(True)
        fwd = lambda: F.attention(q, k, v, H, scale)
    out = fwd()
    g = torch.randn_like(out)
    t_f = timeit(lambda: fwd())

    def fb():
        o = fwd()
        o.backward(g)

    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    from d2r_amd._lib import KernelTimer
    with KernelTimer() as kt:  # per-launch HIP events: device time of the kernels only (the autograd host cost is excluded)
        for _ in range(20):
            fb()
    summ = kt.summary()
    t_f = sum(r["ms"] for k, r in summ.items() if k.endswith("_fwd")) * 1e3 / 20
    t_b = sum(r["ms"] for k, r in summ.items() if k.endswith("_bwd")) * 1e3 / 20
    nbytes = B * (2 * Lq + 2 * Lk) * E * 2
    print(f"{name:20s} fwd {t_f:7.1f} us ({nbytes / t_f / 1e3:7.1f} GB/s)   bwd {t_b:7.1f} us "
          f"({2 * nbytes / max(t_b, 1e-3) / 1e3:7.1f} GB/s)   [algorithmic bytes fwd {nbytes / 1e6:.1f} MB; "
          f"launches fwd {sum(r['calls'] for k, r in summ.items() if k.endswith('_fwd')) // 20} bwd {sum(r['calls'] for k, r in summ.items() if k.endswith('_bwd')) // 20}]", flush=True)
