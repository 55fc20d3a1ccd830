"""Ablation of the LDS-DMA GEMM kernel (D2R_GEMM_DBG: 1 = no MFMA, 2 = no DMA issue, 3 = no epilogue stores) on the workload's shapes.
NOTE: the switches exist only in a measurement build of the library: D2R_GEMM_PROBES=1 python -m d2r_amd.build (then rebuild without it)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from d2r_amd import _lib
from d2r_amd import functional as F
from d2r_amd._lib import BF16, GEMM_NN, GEMM_NT
dev = torch.device("cuda:0")
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for layout, M, N, K in ((GEMM_NN, 4096, 768, 768), (GEMM_NN, 6304, 768, 768), (GEMM_NT, 4096, 768, 768), (GEMM_NN, 6304, 768, 3072)):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) if layout == GEMM_NT else torch.randn(K, N, device=dev)).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), b.shape[1], c.data_ptr(), N, dtype=BF16, c_dtype=BF16)
    print(f"layout {layout} {M}x{N}x{K} dbg={os.environ.get('D2R_GEMM_DBG','0')}: {timeit(run):.2f} us (back-to-back train)", flush=True)
