"""Which hipBLASLt kernels (macro tile, wave layout: readable from the kernel names) serve the path's NT / NN shapes, and how long they take.
Run under rocprofv3 --kernel-trace --stats; a marker kernel (fill of N elements, N = shape index) separates the shapes in the trace."""
import torch
shapes = [(4096, 3072, 768), (4096, 768, 3072), (4096, 2304, 768), (4096, 768, 768), (6304, 3072, 768), (6304, 768, 3072), (6304, 2304, 768), (6304, 768, 768)]
dev = torch.device("cuda", 0)
for lay in ("NT", "NN"):
    for i, (M, N, K) in enumerate(shapes):
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        b = torch.randn(N, K, device=dev, dtype=torch.float16) if lay == "NT" else torch.randn(K, N, device=dev, dtype=torch.float16)
        for _ in range(6):
            c = a @ (b.t() if lay == "NT" else b)
        torch.cuda.synchronize()
        print(lay, M, N, K, flush=True)
