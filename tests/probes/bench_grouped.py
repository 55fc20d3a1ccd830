import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    for (N, K) in ((768, 768), (1536, 768), (2304, 768), (3072, 768), (768, 3072)):
        n = 16 if N * K < 1500000 else 13
        gs = [torch.randn(T, N, device=dev).bfloat16() for _ in range(n)]
        xs = [torch.randn(T, K, device=dev).bfloat16() for _ in range(n)]
        sinks = [torch.zeros(N, K, device=dev) for _ in range(n)]
        bs = [torch.zeros(N, device=dev) for _ in range(n)]
        A, B, C_, D = _parr(gs), _parr(xs), _parr(sinks), _parr(bs)
        for xcd in (0, 1):
            _lib.load().d2r_gemm_tuning(1 + (0 if xcd else 256), 1, -1)
            t = timeit(lambda: _lib.call("d2r_gemm_tn_grouped", _lib.BF16, N, K, T, N, K, K, A, B, C_, D, n, 1.0, _stream()))
            print(f"T={T} {N}x{K} x{n} grouped, xcd remap {xcd}: {2.0 * n * N * K * T / t / 1e12:.0f} TFLOP/s ({t * 1e6 / n:.1f} us per GEMM)")
_lib.load().d2r_gemm_tuning(1, 1, -1)
