"""In-launch split-K of the 128-wide kernel against the unsplit launch, alone on the GPU (back-to-back launches, events around 20)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from d2r_amd import _lib
from d2r_amd import functional as F
dev = torch.device("cuda", 0)
lowp = torch.float16
dt = _lib.F16
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
shapes = [(4096, 768, 3072), (6304, 768, 3072), (4096, 768, 2304), (6304, 768, 2304), (4096, 768, 13824), (6304, 768, 13824), (4096, 768, 768), (4096, 3072, 768), (4096, 1536, 3072)]
for layout, name in ((0, "NT"), (1, "NN")):
    for M, N, K in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).to(lowp)
        b = (torch.randn((N, K) if layout == 0 else (K, N), device=dev) * 0.5).to(lowp)
        c = torch.empty(M, N, device=dev, dtype=lowp)
        out = []
        for w in (None, ws):
            def go():
                F.gemm(layout, M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), dtype=dt, c_dtype=dt, splitk_ws=w)
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                go()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 20 * 1e3)
        fl = 2.0 * M * N * K
        print("%s %5d x %5d x %5d: unsplit %6.1f us %5.0f TF | with workspace %6.1f us %5.0f TF  (%+.0f %%)" % (name, M, N, K, out[0], fl / out[0] / 1e6, out[1], fl / out[1] / 1e6, (out[1] / out[0] - 1) * 100))
