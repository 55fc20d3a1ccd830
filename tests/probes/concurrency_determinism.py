"""Does a single op give bit-identical results when other kernels run concurrently on another stream?  (An op that reads
uninitialised LDS / registers is deterministic alone and varies under concurrency.)  Ops of the middle-layer router path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import functional as F
from d2r_amd._lib import F32, GEMM_NN, GEMM_NT, GEMM_TN

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, D, nc = 2, 16, 768, 6
noise_stream = torch.cuda.Stream()
big_a = torch.randn(2048, 2048, device=dev).bfloat16()
big_b = torch.randn(2048, 2048, device=dev).bfloat16()
big_c = torch.empty(2048, 2048, device=dev, dtype=torch.bfloat16)
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def noise(n=6):
    with torch.cuda.stream(noise_stream):
        for i in range(n):
            F.gemm(GEMM_NT, 2048, 2048, 2048, big_a.data_ptr(), 2048, big_b.data_ptr(), 2048, big_c.data_ptr(), 2048, dtype=1, c_dtype=1)
            junk.fill_(i)
            x = torch.randn(64, 197, 768, device=dev).bfloat16()
            F.layer_norm(x, torch.ones(768, device=dev), torch.zeros(768, device=dev), 1e-5) if hasattr(F, "layer_norm") else None


NOISE_OPS = []  # extra ops (same kernels as the op under test, other data) run on the noise stream


def agg_case():
    gates = torch.rand(B, nc, nc, device=dev)
    embs = [torch.randn(B, L, D, device=dev).bfloat16().requires_grad_(True) for _ in range(nc)]
    embs[1] = torch.randn(B, D, device=dev).bfloat16().requires_grad_(True)
    embs[5] = torch.randn(B, D, device=dev).bfloat16().requires_grad_(True)
    gates.requires_grad_(True)
    w = [torch.randn(B, L, D, device=dev) for _ in range(nc)]
    wp = torch.randn(B, nc, nc, device=dev)

    def run():
        for t in embs + [gates]:
            t.grad = None
        probs, outs = F.route_aggregate(gates, *embs)
        loss = (probs * wp).sum() + sum((o.float() * wi).sum() for o, wi in zip(outs, w))
        loss.backward()
        return [gates.grad.clone()] + [e.grad.clone() for e in embs]
    return run


def gemm_case(layout, M, N, K, nb, lda, ldb, ldc, sA, sB, sC, beta):
    # operand buffers sized for the LARGEST case below (6 batches of [128, 768] outputs, [2, 768] / [128, 768] inputs): every (pointer +
    # batch stride * 5 + rows * ld) of every case stays inside
    a = torch.randn(1 << 20, device=dev)
    b = torch.randn(1 << 20, device=dev)
    c0 = torch.randn(1 << 20, device=dev)
    need_c = (nb - 1) * sC + (M - 1) * ldc + N
    need_a = (nb - 1) * sA + ((K - 1) * lda + M if layout == GEMM_TN else (M - 1) * lda + K)
    need_b = (nb - 1) * sB + ((N - 1) * ldb + K if layout == GEMM_NT else (K - 1) * ldb + N)
    assert need_c <= c0.numel() and need_a <= a.numel() and need_b <= b.numel(), (need_a, need_b, need_c)

    def run():
        c = c0.clone()
        F.gemm(layout, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc, dtype=F32, c_dtype=F32, nb=nb, sA=(sA, 0), sB=(sB, 0),
               sC=(sC, 0), beta=beta)
        return [c]
    return run


hid, P = 128, 6
cases = {
    "route_aggregate fwd+bwd (P=6)": agg_case(),
    "router gx: batched NN [B,hid] = dG[B,P] W2[P,hid] (K=6)": gemm_case(GEMM_NN, B, hid, P, nc, nc * P, hid, nc * hid, P, P * hid, hid, 0.0),
    "router gw: batched TN [P,hid] += dG^T h (K=B)": gemm_case(GEMM_TN, P, hid, B, nc, nc * P, nc * hid, hid, P, hid, P * hid, 1.0),
    "router g0: batched NN [B,E] = dhp[B,hid] W0[hid,E]": gemm_case(GEMM_NN, B, D, hid, nc, nc * hid, D, D, hid, hid * D, B * D, 0.0),
    "router gw0: batched TN [hid,E] += dhp^T pooled (K=B)": gemm_case(GEMM_TN, hid, D, B, nc, nc * hid, D, D, hid, B * D, hid * D, 1.0),
}
if os.environ.get("SAME_KERNEL_NOISE"):
    torch.manual_seed(123)
    other_agg = agg_case()
    _noise0 = noise

    def noise(n=3):
        _noise0(1)
        with torch.cuda.stream(noise_stream):
            for _ in range(n):
                other_agg()

for name, run in cases.items():
    ref = run()
    torch.cuda.synchronize()
    bad_quiet = bad_noisy = 0
    for rep in range(60):
        out = run()
        torch.cuda.synchronize()
        bad_quiet += any(not torch.equal(a, b) for a, b in zip(ref, out))
    for rep in range(200):
        noise(3)
        out = run()
        torch.cuda.synchronize()
        bad_noisy += any(not torch.equal(a, b) for a, b in zip(ref, out))
    print(f"{name}: differing repetitions alone {bad_quiet}/60, with concurrent kernels on another stream {bad_noisy}/200", flush=True)
