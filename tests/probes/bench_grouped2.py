"""Grouped weight-gradient GEMM: 128x128 LDS-DMA kernel (tile code 101) vs the 64x64 generic kernel (100), same data, one
process; correctness of both against fp32 torch.  python tests/probes/bench_grouped2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib
from d2r_amd.functional import _parr, _stream
dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for T in (4096, 6304):
    for (N, K, n) in ((768, 768, 16), (1536, 768, 9), (3072, 768, 7), (768, 3072, 7), (2304, 768, 7)):
        gs = [torch.randn(T, N, device=dev).bfloat16() for _ in range(n)]
        xs = [torch.randn(T, K, device=dev).bfloat16() for _ in range(n)]
        sinks = [torch.zeros(N, K, device=dev) for _ in range(n)]
        bs = [torch.zeros(N, device=dev) for _ in range(n)]
        A, B, C_, D = _parr(gs), _parr(xs), _parr(sinks), _parr(bs)
        call = lambda: _lib.call("d2r_gemm_tn_grouped", _lib.BF16, N, K, T, N, K, K, A, B, C_, D, n, 1.0, _stream())
        out = []
        for code in (100, 101, 100, 101):
            lib.d2r_gemm_tuning(1, 1, code)
            for s_, b_ in zip(sinks, bs):
                s_.zero_(); b_.zero_()
            call(); torch.cuda.synchronize()
            ref = gs[n - 1].float().t() @ xs[n - 1].float()
            err = float((sinks[n - 1] - ref).abs().max() / ref.abs().max())
            errb = float((bs[n - 1] - gs[n - 1].float().sum(0)).abs().max() / gs[n - 1].float().sum(0).abs().max())
            t = timeit(call)
            out.append(f"[{code}] {2.0 * n * N * K * T / t / 1e12:.0f} TF (err {err:.1e}/{errb:.1e})")
        print(f"T={T} {N}x{K} x{n}: " + "  ".join(out), flush=True)
lib.d2r_gemm_tuning(1, 1, 101)
