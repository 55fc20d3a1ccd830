"""A/B of the LDS-DMA GEMM variants (forced tile codes of d2r_gemm_tuning) on the workload's NT / NN / TN shapes, interleaved
in one process, with a correctness check against torch.matmul on the same data.  Run on the GPU box:
    python tests/probes/gemm_variants.py [NT|NN|TN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib
from d2r_amd import functional as F
from d2r_amd._lib import BF16, F32, GEMM_NN, GEMM_NT, GEMM_TN
dev = torch.device("cuda:0")
lib = _lib.load()
NAMES = {GEMM_NT: "NT", GEMM_NN: "NN", GEMM_TN: "TN"}
SHAPES = [(GEMM_NT, 4096, 768, 768), (GEMM_NT, 6304, 768, 768), (GEMM_NT, 4096, 2304, 768), (GEMM_NT, 6304, 3072, 768), (GEMM_NT, 6304, 768, 3072),
          (GEMM_NT, 6304, 1536, 768), (GEMM_NN, 4096, 768, 768), (GEMM_NN, 6304, 768, 3072), (GEMM_NN, 6304, 3072, 768), (GEMM_NN, 4096, 768, 2304),
          (GEMM_TN, 768, 768, 4096), (GEMM_TN, 3072, 768, 6304), (GEMM_TN, 768, 3072, 4096), (GEMM_TN, 2304, 768, 4096)]
want = [a for a in sys.argv[1:] if a in ("NT", "NN", "TN")]
if want:
    SHAPES = [s for s in SHAPES if NAMES[s[0]] in want]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "-1,4,5,6,7,8,9").split(",")]

def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

for layout, M, N, K in SHAPES:
    a = torch.randn((M, K) if layout != GEMM_TN else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if layout == GEMM_NT else (K, N), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = (a.float() @ b.float().t()) if layout == GEMM_NT else ((a.float() @ b.float()) if layout == GEMM_NN else (a.float().t() @ b.float()))
    run = lambda: F.gemm(layout, M, N, K, a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], c.data_ptr(), N, dtype=BF16, c_dtype=BF16)
    res = {}
    for rnd in range(2):
        for v in VARIANTS:
            lib.d2r_gemm_tuning(1, 1, v)
            if rnd == 0:
                c.zero_()
                run()
                err = float((c.float() - ref).abs().max() / ref.abs().max())
                if err > 2e-2:
                    print(f"  !! variant {v} WRONG on {NAMES[layout]} {M}x{N}x{K}: rel err {err:.3f}")
            res.setdefault(v, []).append(2.0 * M * N * K / timeit(run) / 1e12)
    tt = timeit(lambda: torch.matmul(a, b.t()) if layout == GEMM_NT else (torch.matmul(a, b) if layout == GEMM_NN else torch.matmul(a.t(), b)))
    print(f"{NAMES[layout]} M={M} N={N} K={K}: " + " ".join(f"[{v}]:{max(r):.0f}" for v, r in res.items()) + f" | hipBLASLt {2.0 * M * N * K / tt / 1e12:.0f}", flush=True)
lib.d2r_gemm_tuning(1, 1, -1)
