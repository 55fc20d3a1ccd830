"""Where does workgroup (0,0) of the LDS-DMA GEMM kernel spend its cycles?  (s_memtime stamps, d2r_gemm_debug_stamps)
    python tests/probes/gemm_stamps.py
NOTE: the stamps exist only in a measurement build of the library: D2R_GEMM_PROBES=1 python -m d2r_amd.build (then rebuild without it)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib, functional as F
from d2r_amd._lib import BF16, GEMM_NN, GEMM_NT

dev = torch.device("cuda:0")
lib = _lib.load()
lib.d2r_gemm_debug_stamps.argtypes = [C.c_void_p]
lib.d2r_gemm_debug_stamps.restype = None
buf = torch.zeros(8 * 8, dtype=torch.int64, device=dev)  # [wave][0..3 stamps, 4..7 per-phase cycle sums]
for lay, M, N, K in (("NN", 4096, 768, 768), ("NT", 4096, 768, 768), ("NN", 6304, 768, 3072), ("NT", 6304, 3072, 768), ("NT", 4096, 768, 3072),
                     ("NN", 6304, 3072, 768)):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) if lay == "NT" else torch.randn(K, N, device=dev)).mul_(0.03).bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ldb = K if lay == "NT" else N
    layout = GEMM_NT if lay == "NT" else GEMM_NN
    run = lambda: F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), ldb, c.data_ptr(), N, dtype=BF16, c_dtype=BF16)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    buf.zero_()
    lib.d2r_gemm_debug_stamps(C.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.d2r_gemm_debug_stamps(None)
    t = buf.cpu().view(8, 8)
    r = t[0]
    nk = K // 64
    print(f"{lay} {M}x{N}x{K}: {us:6.1f} us back to back ({2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s) | wave 0 of WG 0: to first tile {int(r[1] - r[0])}, "
          f"K loop {int(r[2] - r[1])} = {int(r[2] - r[1]) / nk:.0f} per 64-deep step ({nk} steps), epilogue + drain {int(r[3] - r[2])}, total {int(r[3] - r[0])} cycles; "
          f"per step (non-pipelined kernels): DMA issue {int(r[4]) / nk:.0f}, wait for the tile + barrier {int(r[5]) / nk:.0f}, fragment reads + MFMAs {int(r[6]) / nk:.0f}, closing barrier {int(r[7]) / nk:.0f}", flush=True)
