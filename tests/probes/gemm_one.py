"""One GEMM shape, a few launches (for rocprofv3 --pmc): python tests/probes/gemm_one.py NT 6304 768 768 [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib, functional as F
from d2r_amd._lib import BF16, F32, GEMM_NN, GEMM_NT, GEMM_TN
lay = {"NT": GEMM_NT, "NN": GEMM_NN, "TN": GEMM_TN}[sys.argv[1]]
M, N, K = (int(v) for v in sys.argv[2:5])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
dev = torch.device("cuda:0")
a = torch.randn((M, K) if lay != GEMM_TN else (K, M), device=dev).bfloat16()
b = torch.randn((N, K) if lay == GEMM_NT else (K, N), device=dev).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == GEMM_TN else torch.bfloat16)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
_lib.load().d2r_gemm_tuning(1, 1, tile)
for _ in range(5):
    F.gemm(lay, M, N, K, a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], c.data_ptr(), N, dtype=BF16,
           c_dtype=F32 if lay == GEMM_TN else BF16, beta=1.0 if lay == GEMM_TN else 0.0, splitk_ws=ws if lay == GEMM_TN else None)
torch.cuda.synchronize()
