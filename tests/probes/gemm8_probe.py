"""256 x 256 deep-pipelined GEMM (csrc/gemm8.hip): correctness against fp32 torch on the workload's shapes (ragged rows, ragged
reduction length, epilogue operands, both 16-bit types), then A/B timing against the 128-wide LDS-DMA kernels and hipBLASLt on the
same data, interleaved in one process.      python tests/probes/gemm8_probe.py [check] [time] [NT|NN|TN]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib
from d2r_amd import functional as F
from d2r_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F16, GEMM_NN, GEMM_NT

dev = torch.device("cuda:0")
if "stamps" in sys.argv[1:]:  # the measurement build (D2R_G8_STAMPS=1 python -m d2r_amd.build, copied beside this file)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libd2r_hip_stamps.so")
lib = _lib.load()
VARS = [0]  # (a second loop variant - DMA issue inside the MFMA cluster - was measured slower and removed: profiles/gemm8_probe_r04.log "v1")
args = sys.argv[1:]
do_check = "check" in args or not any(a in args for a in ("check", "time", "stamps"))
do_time = "time" in args or not any(a in args for a in ("check", "time", "stamps"))
lay = [a for a in args if a in ("NT", "NN", "TN")] or ["NT", "NN", "TN"]
TDT = {BF16: torch.bfloat16, F16: torch.float16}


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


def fwd_case(layout, M, N, K, dt, bias=False, act=ACT_NONE, res=False, beta=0.0, ldc_pad=0):
    td = TDT[dt]
    a = (torch.randn(M, K, device=dev) * 0.5).to(td)
    b = (torch.randn((N, K) if layout == GEMM_NT else (K, N), device=dev) * 0.5).to(td)
    ldc = N + ldc_pad
    c = (torch.randn(M, ldc, device=dev)).to(td)
    c0 = c.clone()
    bi = torch.randn(N, device=dev) if bias else None
    r = torch.randn(M, N, device=dev).to(td) if res else None
    ref = a.float() @ (b.float().t() if layout == GEMM_NT else b.float())
    if bias:
        ref = ref + bi
    if act == ACT_RELU:
        ref = torch.relu(ref)
    elif act == ACT_GELU:
        ref = torch.nn.functional.gelu(ref.to(td).float())  # (the kernel rounds the pre-activation to 16 bits first)
    if res:
        ref = ref + r.float()
    if beta:
        ref = ref + beta * c0[:, :N].float()
    run = lambda: F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), b.shape[1], c.data_ptr(), ldc, dtype=dt, c_dtype=dt,
                         bias=bi.data_ptr() if bias else None, act=act, residual=r.data_ptr() if res else None, ldr=N, beta=beta)
    return run, c, c0, ref, a, b


def check_fwd():
    bad = 0
    cases = []
    for layout in (GEMM_NT, GEMM_NN):
        if ("NT" if layout == GEMM_NT else "NN") not in lay:
            continue
        for dt in (BF16, F16):
            cases += [(layout, 4096, 3072, 768, dt, dict()), (layout, 6304, 768, 3072, dt, dict(bias=True, act=ACT_RELU)),
                      (layout, 6304, 2304, 768, dt, dict(bias=True, res=True)), (layout, 300, 520, 128, dt, dict(bias=True, act=ACT_GELU, ldc_pad=8)),
                      (layout, 4096, 768, 768, dt, dict(beta=1.0, bias=True)), (layout, 257, 264, 192, dt, dict())]
    for layout, M, N, K, dt, kw in cases:
        lib.d2r_gemm_tuning(1, 1, 11)
        run, c, c0, ref, a, b = fwd_case(layout, M, N, K, dt, **kw)
        run()
        torch.cuda.synchronize()
        err = float((c[:, :N].float() - ref).abs().max() / ref.abs().max())
        pad_ok = bool(torch.equal(c[:, N:], c0[:, N:]))
        tol = 1.2e-2 if dt == BF16 else 2.5e-3
        ok = err < tol and pad_ok
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} fwd {'NT' if layout == GEMM_NT else 'NN'} {M}x{N}x{K} {'bf16' if dt == BF16 else 'fp16'} {kw}: rel err {err:.2e} pad untouched {pad_ok}", flush=True)
        if not kw.get("beta"):
            # bit-identical to itself across launches (no race in the pipeline): 5 repeats
            first = c.clone()
            for _ in range(5):
                c.copy_(c0)
                run()
                torch.cuda.synchronize()
                if not torch.equal(c, first):
                    bad += 1
                    print("   BAD: repeat differs", float((c.float() - first.float()).abs().max()), flush=True)
                    break
    lib.d2r_gemm_tuning(1, 1, -1)
    return bad


def iarr(t, vals):
    arr = (t * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def tn_group(shapes, dt, beta, with_bias=True, seed=0):
    """shapes: [(Nf, Kf, T)] -> dW[Nf,Kf] = dY[T,Nf]^T X[T,Kf]"""
    td = TDT[dt]
    g = torch.Generator(device=dev).manual_seed(seed)
    dys = [(torch.randn(T, Nf, device=dev, generator=g) * 0.5).to(td) for Nf, Kf, T in shapes]
    xs = [(torch.randn(T, Kf, device=dev, generator=g) * 0.5).to(td) for Nf, Kf, T in shapes]
    sinks = [torch.randn(Nf, Kf, device=dev, generator=g) for Nf, Kf, T in shapes]
    bs = [torch.randn(Nf, device=dev, generator=g) for Nf, Kf, T in shapes]
    n = len(shapes)
    call = lambda: _lib.call("d2r_gemm_tn_grouped_v", dt, n, iarr(C.c_int, [s[0] for s in shapes]), iarr(C.c_int, [s[1] for s in shapes]),
                             iarr(C.c_int, [s[2] for s in shapes]), iarr(C.c_int64, [s[0] for s in shapes]), iarr(C.c_int64, [s[1] for s in shapes]),
                             iarr(C.c_int64, [s[1] for s in shapes]), iarr(C.c_void_p, [t.data_ptr() for t in dys]),
                             iarr(C.c_void_p, [t.data_ptr() for t in xs]), iarr(C.c_void_p, [t.data_ptr() for t in sinks]),
                             iarr(C.c_void_p, [t.data_ptr() for t in bs]) if with_bias else None, beta, F._stream())
    return call, dys, xs, sinks, bs


def check_tn():
    bad = 0
    for dt in (BF16, F16):
        for beta in (1.0, 0.0):
            shapes = [(768, 768, 4096), (3072, 768, 6304), (768, 3072, 4096), (2304, 768, 6304), (768, 768, 6304), (1536, 768, 200), (264, 520, 4096),
                      (768, 768, 4096), (768, 1536, 6304)]
            call, dys, xs, sinks, bs = tn_group(shapes, dt, beta)
            s0 = [s.clone() for s in sinks]
            b0 = [b.clone() for b in bs]
            lib.d2r_gemm_tuning(1, 1, 103)
            call()
            torch.cuda.synchronize()
            for i, (Nf, Kf, T) in enumerate(shapes):
                ref = dys[i].float().t() @ xs[i].float() + beta * s0[i]
                refb = dys[i].float().sum(0) + b0[i]
                err = float((sinks[i] - ref).abs().max() / ref.abs().max())
                errb = float((bs[i] - refb).abs().max() / refb.abs().max())
                ok = err < 2e-5 * (T ** 0.5) and errb < 1e-4
                bad += not ok
                print(f"{'ok ' if ok else 'BAD'} TN {Nf}x{Kf} T={T} {'bf16' if dt == BF16 else 'fp16'} beta={beta}: rel err {err:.2e} bias {errb:.2e}", flush=True)
            if beta == 0.0:
                first = [s.clone() for s in sinks]
                for _ in range(5):
                    call()
                    torch.cuda.synchronize()
                    if any(not torch.equal(a_, b_) for a_, b_ in zip(sinks, first)):
                        bad += 1
                        print("   BAD: repeat differs", flush=True)
                        break
    return bad


def time_fwd():
    for layout, name in ((GEMM_NT, "NT"), (GEMM_NN, "NN")):
        if name not in lay:
            continue
        for M in (4096, 6304):
            for N, K in ((3072, 768), (768, 3072), (2304, 768), (13824, 768), (4096, 4096), (8192, 8192)) if M == 4096 else ((3072, 768), (2304, 768), (13824, 768)):
                run, c, c0, ref, a, b = fwd_case(layout, M, N, K, F16 if os.environ.get("DT") == "f16" else BF16)
                res = {}
                for rnd in range(3):
                    for v in [-1] + VARS:
                        lib.d2r_gemm_tuning(1, 1, 110 if v < 0 else 111)  # -1: the 128-wide kernels' own choice, else the wide tiles forced
                        lib.d2r_gemm_tuning(1, 1, -1 if v < 0 else 11)
                        res.setdefault(v, []).append(timeit(run))
                tt = timeit((lambda: torch.matmul(a, b.t())) if layout == GEMM_NT else (lambda: torch.matmul(a, b)))
                fl = 2.0 * M * N * K
                t256 = ((M + 255) // 256) * ((N + 255) // 256)
                print(f"{name} M={M} N={N} K={K} ({t256} wide tiles): 128-wide {min(res[-1]) * 1e6:6.1f} us {fl / min(res[-1]) / 1e12:5.0f} TF | " +
                      " | ".join(f"gemm8 v{v} {min(res[v]) * 1e6:6.1f} us {fl / min(res[v]) / 1e12:5.0f} TF" for v in VARS) +
                      f" | hipBLASLt {tt * 1e6:6.1f} us {fl / tt / 1e12:5.0f} TF", flush=True)
    lib.d2r_gemm_tuning(1, 1, 111)
    lib.d2r_gemm_tuning(1, 1, -1)


def time_tn():
    dt = F16 if os.environ.get("DT") == "f16" else BF16
    groups = {
        "768x768 x16 T=4096": [(768, 768, 4096)] * 16, "768x768 x16 T=6304": [(768, 768, 6304)] * 16,
        "3072x768 x13 T=6304": [(3072, 768, 6304)] * 13, "768x3072 x13 T=4096": [(768, 3072, 4096)] * 13, "2304x768 x13 T=4096": [(2304, 768, 4096)] * 13,
        "encoder 7 layers T=4096": [(2304, 768, 4096), (768, 768, 4096), (3072, 768, 4096), (768, 3072, 4096)] * 7,
        "encoder 7 layers T=6304": [(2304, 768, 6304), (768, 768, 6304), (3072, 768, 6304), (768, 3072, 6304)] * 7,
        "routing module T=4096": [(768, 768, 4096)] * 30 + [(2304, 768, 4096)] * 3 + [(13824, 768, 6304)],
    }
    for name, shapes in groups.items():
        fl = sum(2.0 * a_ * b_ * c_ for a_, b_, c_ in shapes)
        out = []
        for beta in (1.0, 0.0):
            call, dys, xs, sinks, bs = tn_group(shapes, dt, beta)
            res = {}
            for rnd in range(2):
                for v in [-1] + VARS:
                    lib.d2r_gemm_tuning(1, 1, 102 if v < 0 else 103)
                    res.setdefault(v, []).append(timeit(call, 10))
            out.append(f"beta={beta:.0f}: 128-wide {min(res[-1]) * 1e6:7.1f} us {fl / min(res[-1]) / 1e12:5.0f} TF | " +
                       " | ".join(f"gemm8 v{v} {min(res[v]) * 1e6:7.1f} us {fl / min(res[v]) / 1e12:5.0f} TF" for v in VARS))
            del dys, xs, sinks, bs
        print(f"TN {name}: " + "  ||  ".join(out), flush=True)
    lib.d2r_gemm_tuning(1, 1, 103)


def stamps():
    """Cycle stamps of workgroup 0 (waves 0 and 4): per phase of the first five K-tiles, cycles from the first barrier to the end of
    the MFMA cluster ("mfma"), to the second barrier ("bar2") and through the load half to the next first barrier ("load")."""
    import numpy as np
    buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    lib.d2r_gemm8_debug_stamps(C.c_void_p(buf.data_ptr()))
    td = torch.bfloat16
    for v in VARS:
        lib.d2r_gemm_tuning(1, 1, 11)
        for name in ("NT 4096x4096x4096", "NT 4096x3072x768", "TN encoder 7 layers T=4096"):
            if name.startswith("NT"):
                M, N, K = (int(x) for x in name.split()[1].split("x"))
                run, c, c0, ref, a, b = fwd_case(GEMM_NT, M, N, K, BF16)
            else:
                run, *_ = tn_group([(2304, 768, 4096), (768, 768, 4096), (3072, 768, 4096), (768, 3072, 4096)] * 7, BF16, 1.0)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            buf.zero_()
            run()
            torch.cuda.synchronize()
            st = buf.cpu().numpy().reshape(8, 64)
            for w in (0, 4):
                t0 = st[w][0]
                line = [f"prologue {st[w][1] - t0}"]
                for t in range(5):
                    sb = 2 + t * 12
                    if st[w][sb + 11] == 0:
                        break
                    ph = []
                    for p_ in range(4):
                        a0, m1, b2 = st[w][sb + 3 * p_: sb + 3 * p_ + 3]
                        nxt = st[w][sb + 3 * p_ + 3] if (3 * p_ + 3 < 12 or t < 4) and sb + 3 * p_ + 3 < 62 else b2
                        ph.append(f"[mfma {m1 - a0} bar2 {b2 - m1} load {nxt - b2}]")
                    line.append(f"tile{t} " + " ".join(ph))
                print(f"v{v} {name} wave {w}: total {st[w][63] - t0} | " + " | ".join(line), flush=True)
    lib.d2r_gemm_tuning(1, 1, -1)


bad = 0
if "stamps" in args:
    stamps()
    sys.exit(0)
if do_check:
    for v in VARS:
        print(f"--- kernel variant {v}", flush=True)
        bad += check_fwd()
        if "TN" in lay:
            bad += check_tn()
    print("CHECK", "FAILED" if bad else "passed", bad, flush=True)
if do_time and not bad:
    time_fwd()
    if "TN" in lay:
        time_tn()
sys.exit(1 if bad else 0)
