"""Is the time of an LDS-DMA GEMM launch a step function of workgroups per CU?  One shape family (M, K fixed), N swept so that the
grid goes from one workgroup per CU (256 tiles) through one and a half to two and three, per forced tile variant.
    python tests/probes/tile_balance.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib
from d2r_amd import functional as F
from d2r_amd._lib import BF16, GEMM_NN, GEMM_NT
dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for layout, name in ((GEMM_NT, "NT"), (GEMM_NN, "NN")):
    for M in (4096, 6304):
        for K in (768, 3072):
            for v, bn in ((5, 64), (6, 128)):
                lib.d2r_gemm_tuning(1, 1, v)
                out = []
                for N in ((512, 640, 768, 1024, 1280, 1536) if bn == 64 else (1024, 1536, 2048, 2304, 3072)):
                    a = torch.randn(M, K, device=dev).bfloat16()
                    b = torch.randn((N, K) if layout == GEMM_NT else (K, N), device=dev).bfloat16()
                    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                    run = lambda: F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), b.shape[1], c.data_ptr(), N, dtype=BF16, c_dtype=BF16)
                    us = min(timeit(run), timeit(run))
                    tiles = ((M + 127) // 128) * (N // bn)
                    out.append(f"N={N}: {tiles} tiles ({tiles / 256:.2f}/CU) {us:.1f} us {2.0 * M * N * K / us * 1e-6:.0f} TF")
                print(f"{name} M={M} K={K} tile 128x{bn}: " + " | ".join(out), flush=True)
lib.d2r_gemm_tuning(1, 1, -1)
