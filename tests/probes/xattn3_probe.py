"""Timing probe of the cross-attention kernels at the C2 shapes (three problems per launch, as the routing layers issue them).
    D2R_X3_DBG=<mode> python tests/probes/xattn3_probe.py        (modes: see xattn3.hip; 0 = the real kernel)
NOTE: D2R_X3_DBG / D2R_X3_STAMPS act only on a measurement build of the library: D2R_X3_PROBES=1 python -m d2r_amd.build (then rebuild without it)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib
if os.environ.get("PROBE_LIB"):  # A/B of two builds of the library in one GPU call
    _lib.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
from d2r_amd import functional as F

dev = torch.device("cuda:0")
E, B, nc = 768, 32, 3
st = torch.cuda.current_stream().cuda_stream
arr = lambda ts: F._iparr([t if isinstance(t, int) else t.data_ptr() for t in ts])
for Lq, Lk in ((128, 197), (197, 128), (128, 128), (197, 197)):
    ncore = nc if Lq != Lk else 1
    q = [torch.randn(B, Lq, E, device=dev).mul_(0.15).bfloat16() for _ in range(ncore)]
    kv = [torch.randn(B, Lk, 2 * E, device=dev).mul_(0.15).bfloat16() for _ in range(ncore)]
    o = [torch.empty(B, Lq, E, dtype=torch.bfloat16, device=dev) for _ in range(ncore)]
    lse = [torch.empty(B, Lq, dtype=torch.float32, device=dev) for _ in range(ncore)]
    dO = [torch.randn(B, Lq, E, device=dev).bfloat16() for _ in range(ncore)]
    dq = [torch.empty_like(x) for x in q]
    dkv = [torch.empty_like(x) for x in kv]
    lkp = (Lk + 7) // 8 * 8
    P = [torch.empty(B, Lq, lkp, dtype=torch.bfloat16, device=dev) for _ in range(ncore)]
    dS = [torch.empty_like(x) for x in P]
    scale = 100.0 / math.sqrt(768)

    def fwd():
        _lib.call("d2r_xattn_fwd_multi", 1, ncore, arr(q), E, Lq * E, arr(kv), 2 * E, Lk * 2 * E, arr([t.data_ptr() + E * 2 for t in kv]), 2 * E,
                  Lk * 2 * E, arr(o), E, Lq * E, None, E, Lq * E, None, arr(lse), B, Lq, Lk, E, scale, st)

    def bwd():
        _lib.call("d2r_xattn_bwd_multi", 1, ncore, arr(q), E, Lq * E, arr(kv), 2 * E, Lk * 2 * E, arr([t.data_ptr() + E * 2 for t in kv]), 2 * E,
                  Lk * 2 * E, arr(dO), E, Lq * E, arr(o), E, Lq * E, None, E, Lq * E, None, arr(lse), arr(dq), E, Lq * E, arr(dkv), 2 * E,
                  Lk * 2 * E, arr([t.data_ptr() + E * 2 for t in dkv]), 2 * E, Lk * 2 * E, arr(P), arr(dS), lkp, B, Lq, Lk, E, scale, st)

    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        mb = ncore * B * (2 * Lq + 2 * Lk) * E * 2 / 1e6 * (2 if name == "bwd" else 1)
        print(f"dbg={os.environ.get('D2R_X3_DBG', '0')} Lq={Lq} Lk={Lk} ncore={ncore} {name}: {us:7.1f} us  ({mb / us * 1e-3 * 1e3:.0f} GB/s algorithmic)", flush=True)
    if os.environ.get("D2R_X3_STAMPS"):  # where does block 0 of the forward kernel spend its cycles?
        import ctypes as C
        lib = _lib.load()
        buf = torch.zeros(4 * 64, dtype=torch.int64, device=dev)
        lib.d2r_xattn3_debug_stamps.argtypes = [C.c_void_p]
        lib.d2r_xattn3_debug_stamps.restype = None
        for rep in range(3):
            lib.d2r_xattn3_debug_stamps(C.c_void_p(buf.data_ptr()))
            fwd()
            torch.cuda.synchronize()
            lib.d2r_xattn3_debug_stamps(None)
        t = buf.cpu().view(4, 64)
        nkc = (Lk + 15) // 16
        for w in range(4):
            r = t[w]
            k = [int(r[2 + i] - (r[1] if i == 0 else r[1 + i])) for i in range(nkc)]
            nv = (nkc + 1) // 2  # value tiles go two per step
            v = [int(r[19 + i] - (r[18] if i == 0 else r[18 + i])) for i in range(nv)]
            print(f"  stamps Lq={Lq} Lk={Lk} wave {w}: prologue {int(r[1] - r[0])} | score tiles {k} | softmax {int(r[18] - r[1 + nkc])} | value tile pairs {v} | "
                  f"ring release {int(r[35] - r[18 + nv])} | epilogue {int(r[36] - r[35])} | total {int(r[36] - r[0])} cycles", flush=True)
