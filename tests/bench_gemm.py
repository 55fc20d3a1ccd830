"""A/B micro-benchmark of the MFMA GEMM family on the workload's shapes (not a pytest; run on the GPU box):
    python tests/bench_gemm.py
Interleaves the tuning variants in ONE process (cdna_hip_programming.md rule 24) and prints TFLOP/s per variant."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from d2r_amd import _lib
from d2r_amd import functional as F
from d2r_amd._lib import BF16, F32, GEMM_NN, GEMM_NT, GEMM_TN

dev = torch.device("cuda:0")
NAMES_EARLY = {GEMM_NT: "NT", GEMM_NN: "NN", GEMM_TN: "TN"}
lib = _lib.load()
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
dbias = torch.zeros(4096, dtype=torch.float32, device=dev)

# (layout, M, N, K): forward linears, dX, dW of the C2 workload (B=32, L=128 / 197 tokens)
SHAPES = [
    (GEMM_NT, 4096, 768, 768), (GEMM_NT, 6304, 768, 768), (GEMM_NT, 6304, 3072, 768), (GEMM_NT, 6304, 768, 3072),
    (GEMM_NN, 4096, 768, 768), (GEMM_NN, 6304, 768, 3072), (GEMM_NN, 6304, 3072, 768),
    (GEMM_TN, 768, 768, 4096), (GEMM_TN, 768, 768, 6304), (GEMM_TN, 2304, 768, 4096), (GEMM_TN, 2304, 768, 6304),
    (GEMM_TN, 3072, 768, 6304), (GEMM_TN, 768, 3072, 6304), (GEMM_TN, 3072, 768, 4096), (GEMM_TN, 768, 3072, 4096),
]
if os.environ.get("BENCH_GEMM_ONLY"):
    SHAPES = [s for s in SHAPES if NAMES_EARLY[s[0]] == os.environ["BENCH_GEMM_ONLY"]]
NAMES = {GEMM_NT: "NT", GEMM_NN: "NN", GEMM_TN: "TN"}


def operands(layout, M, N, K, dtype):
    a = torch.randn((M, K) if layout != GEMM_TN else (K, M), device=dev).to(dtype)
    b = torch.randn((N, K) if layout == GEMM_NT else (K, N), device=dev).to(dtype)
    c = torch.empty(M, N, device=dev, dtype=torch.float32 if layout == GEMM_TN else dtype)
    return a, b, c


def run(layout, M, N, K, a, b, c, use_ws):
    F.gemm(layout, M, N, K, a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], c.data_ptr(), N,
           dtype=BF16 if a.dtype == torch.bfloat16 else F32, c_dtype=F32 if c.dtype == torch.float32 else BF16,
           beta=1.0 if layout == GEMM_TN else 0.0, splitk_ws=ws if use_ws else None,
           dbias=dbias.data_ptr() if layout == GEMM_TN else None)  # weight-gradient GEMMs carry the bias gradient


def timeit(fn, iters=20):
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


# first field: LDS buffers, +256 = XCD-aware tile order OFF; tile 4/5: LDS-DMA pipelined 128x128 / 128x64
variants = [(1, 1, -1), (257, 1, -1), (1, 1, 1), (257, 1, 1), (1, 1, 3), (257, 1, 3), (1, 1, 5), (257, 1, 5)]
dtype = torch.bfloat16
print("variant = (lds_buffers, vector_epilogue, tile[-1 auto,1:64x64,2:128x64,3:128x128]); TFLOP/s")
for layout, M, N, K in SHAPES:
    a, b, c = operands(layout, M, N, K, dtype)
    res = {}
    for rnd in range(2):
        for v in variants:
            lib.d2r_gemm_tuning(*v)
            t = timeit(lambda: run(layout, M, N, K, a, b, c, layout == GEMM_TN))
            res.setdefault(v, []).append(2.0 * M * N * K / t / 1e12)
    # correctness of the LDS-DMA variants against hipBLASLt on the same data
    for tile in (4, 5):
        lib.d2r_gemm_tuning(1, 1, tile)
        c.zero_()
        run(layout, M, N, K, a, b, c, False)
        ref = (a.float() @ b.float().t()) if layout == GEMM_NT else ((a.float() @ b.float()) if layout == GEMM_NN else (a.float().t() @ b.float()))
        err = float((c.float() - ref).abs().max() / ref.abs().max())
        if err >= 2e-2:
            bad = ((c.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
            print(f"  !! glds tile {tile} WRONG on {NAMES[layout]} {M}x{N}x{K}: rel err {err:.3f}; {len(bad)} bad elements, "
                  f"rows {int(bad[:, 0].min())}..{int(bad[:, 0].max())} cols {int(bad[:, 1].min())}..{int(bad[:, 1].max())}")
    best = max(res, key=lambda v: min(res[v]))
    line = " ".join(f"{v}:{min(r):.0f}" for v, r in res.items())
    # hipBLASLt through torch as a yard-stick (same shapes, same data)
    if layout == GEMM_NT:
        tt = timeit(lambda: torch.matmul(a, b.t()))
    elif layout == GEMM_NN:
        tt = timeit(lambda: torch.matmul(a, b))
    else:
        tt = timeit(lambda: torch.matmul(a.t(), b))
    print(f"{NAMES[layout]} M={M} N={N} K={K}: best {best} {min(res[best]):.0f} TF/s | hipBLASLt {2.0 * M * N * K / tt / 1e12:.0f} TF/s\n    {line}", flush=True)
lib.d2r_gemm_tuning(1, 1, -1)
