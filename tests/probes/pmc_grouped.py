"""One cold launch per (shape, group size, XCD remap on/off) of the grouped weight-gradient kernel, to be run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o f -- python3 tests/probes/pmc_grouped.py
The dispatch order printed here matches the order of the gemm_kernel<...,true> rows in f_counter_collection.csv;
FETCH_SIZE (KiB, x2 on gfx950) against the operand bytes tells how often the operand strips are re-fetched past L2."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from d2r_amd import _lib
from d2r_amd.functional import _parr, _stream
dev = torch.device("cuda:0")
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)  # > MALL (256 MB) + L2: evicts the operands between launches
T = 4096
for (N, K, n) in ((768, 768, 16), (768, 768, 8), (3072, 768, 6), (3072, 768, 8), (768, 3072, 6), (2304, 768, 7)):
    gs = [torch.randn(T, N, device=dev).bfloat16() for _ in range(n)]
    xs = [torch.randn(T, K, device=dev).bfloat16() for _ in range(n)]
    sinks = [torch.zeros(N, K, device=dev) for _ in range(n)]
    A, B, C_ = _parr(gs), _parr(xs), _parr(sinks)
    for xcd in (1, 0):
        _lib.load().d2r_gemm_tuning(1 + (0 if xcd else 256), 1, -1)
        flush.fill_(1)
        torch.cuda.synchronize()
        _lib.call("d2r_gemm_tn_grouped", _lib.BF16, N, K, T, N, K, K, A, B, C_, None, n, 1.0, _stream())
        torch.cuda.synchronize()
        operands = n * T * (N + K) * 2
        print(f"dispatch: {N}x{K} x{n} T={T} xcd_remap={xcd}: operands {operands / 1e6:.1f} MB, C read+write {2 * n * N * K * 4 / 1e6:.1f} MB", flush=True)
_lib.load().d2r_gemm_tuning(1, 1, -1)
