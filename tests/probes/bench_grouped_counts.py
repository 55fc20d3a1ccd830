"""Grouped weight-gradient launches at the group sizes the C2 step really issues (kernel-only times)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from d2r_amd import _lib
from d2r_amd.functional import _parr, _stream
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for T in (4096, 6304):
    for (N, K, counts) in ((768, 768, (6, 14, 16)), (1536, 768, (6, 9)), (2304, 768, (1, 6, 7)), (3072, 768, (1, 6, 7)),
                           (768, 3072, (1, 6, 7)), (64, 768, (3,))):
        for n in counts:
            gs = [torch.randn(T, N, device=dev).bfloat16() for _ in range(n)]
            xs = [torch.randn(T, K, device=dev).bfloat16() for _ in range(n)]
            sinks = [torch.zeros(N, K, device=dev) for _ in range(n)]
            bs = [torch.zeros(N, device=dev) for _ in range(n)]
            A, B, C_, D = _parr(gs), _parr(xs), _parr(sinks), _parr(bs)
            t = timeit(lambda: _lib.call("d2r_gemm_tn_grouped", _lib.BF16, N, K, T, N, K, K, A, B, C_, D, n, 1.0, _stream()))
            print(f"T={T} {N}x{K} x{n}: {2.0 * n * N * K * T / t / 1e12:.0f} TFLOP/s ({t * 1e6:.1f} us per launch, {t * 1e6 / n:.1f} per GEMM)", flush=True)
