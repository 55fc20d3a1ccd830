"""Would deferring the small weight-gradient GEMMs and launching them as one batched kernel (no split-K, no slabs, no
reduce launch) pay?  Batched TN via uniform strides emulates the grouped launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib, functional as F
from d2r_amd._lib import BF16, F32, GEMM_TN
dev = torch.device("cuda:0")
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for T in (4096, 6304):
    for nb in (1, 4, 16, 32):
        M = N = 768
        a = torch.randn(nb, T, M, device=dev).bfloat16()
        b = torch.randn(nb, T, N, device=dev).bfloat16()
        c = torch.zeros(nb, M, N, device=dev)
        db = torch.zeros(nb, M, device=dev)
        if nb == 1:
            t = timeit(lambda: F.gemm(GEMM_TN, M, N, T, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, dtype=BF16, c_dtype=F32,
                                      beta=1.0, splitk_ws=ws, dbias=db.data_ptr()))
            print(f"T={T} single + split-K + dbias: {2.0 * M * N * T / t / 1e12:.0f} TFLOP/s ({t * 1e6:.1f} us)")
        for tile in (1, 2, 3):
            _lib.load().d2r_gemm_tuning(1, 1, tile)
            t = timeit(lambda: F.gemm(GEMM_TN, M, N, T, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, dtype=BF16, c_dtype=F32,
                                      beta=1.0, nb=nb, sA=(T * M, 0), sB=(T * N, 0), sC=(M * N, 0)))
            print(f"T={T} batch {nb:2d} tile {tile} (no split, no dbias): {2.0 * nb * M * N * T / t / 1e12:.0f} TFLOP/s ({t * 1e6 / nb:.1f} us per GEMM)")
        _lib.load().d2r_gemm_tuning(1, 1, -1)
