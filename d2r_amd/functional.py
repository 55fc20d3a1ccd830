"""Differentiable ops of the D2R hot path: thin ``torch.autograd.Function`` wrappers whose forward AND backward
are launches of the hand-written gfx950 kernels in libd2r_hip.so (C ABI: include/d2r_hip.h).

PyTorch supplies device memory, the current HIP stream and the autograd tape — no arithmetic.  Nothing here
falls back to ATen: a missing library or a failing launch raises ``d2r_amd._lib.D2RError``.

dtype policy: activations and GEMM weights are ``T`` in {float32, bfloat16, float16}; accumulation, softmax logits and
statistics, router gates, biases, LayerNorm parameters, losses and every reduction are fp32.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_TANH_RELU, BF16, F16, F32,
                   GEMM_NN, GEMM_NT, GEMM_TN, GemmDesc)

_ACT_FROM_OUTPUT = (ACT_RELU, ACT_TANH, ACT_TANH_RELU, ACT_SIGMOID)
_ACT_FROM_PREACT = (ACT_GELU, ACT_QUICK_GELU)


_raw_stream = torch._C._cuda_getCurrentRawStream  # one C call (torch.cuda.current_stream() costs ~9 us of Python)
_current_device = torch._C._cuda_getDevice


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device: every d2r kernel is launched on it."""
    return _raw_stream(_current_device())


LOWP = (torch.bfloat16, torch.float16)  # the 16-bit compute dtypes: same kernels, the MFMA operand type differs
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_DT_NAME = {F32: "f32", BF16: "bf16", F16: "f16"}


def _dt_of(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"d2r_amd supports float32 / bfloat16 / float16 tensors, got {dtype}") from None


def _dt(t: torch.Tensor) -> int:
    return _dt_of(t.dtype)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.D2RError("d2r_amd ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _parr(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


_WS = {}

# Weight-gradient GEMMs (dW = dY^T X) feed nothing but the optimiser: they are issued on a side stream per compute
# stream so that they fill the CUs the (small, latency-bound) dX chain leaves idle.  Whoever consumes the flat
# gradient buffer (optimiser step, all-reduce, zero_grad, a test reading .grad) calls wgrad_join() first.
WGRAD_STREAMS = False  # measured slower on MI355X (35.4 vs 33.7 ms/step): off
_WGRAD = {}


def wgrad_stream(*tensors) -> Optional[int]:
    """Side stream of the current stream (raw handle), already waiting on the work enqueued so far; `tensors` are the
    operands the side stream will read (kept alive for it).  None when the feature is off."""
    if not WGRAD_STREAMS:
        return None
    cur = torch.cuda.current_stream()
    side = _WGRAD.get(cur.cuda_stream)
    if side is None:
        side = _WGRAD[cur.cuda_stream] = torch.cuda.Stream()
    side.wait_stream(cur)
    for t in tensors:
        t.record_stream(side)
    return side.cuda_stream


def wgrad_side_stream_handle() -> Optional[int]:
    """Raw handle of the current stream's side stream WITHOUT synchronising (the C composite orders itself)."""
    if not WGRAD_STREAMS:
        return None
    cur = torch.cuda.current_stream()
    side = _WGRAD.get(cur.cuda_stream)
    if side is None:
        side = _WGRAD[cur.cuda_stream] = torch.cuda.Stream()
    return side


def wgrad_streams():
    return list(_WGRAD.values())


# ------------------------------------------------------------------------------------------------------
# deferred weight gradients: nothing in the backward pass reads dW, so the small ones (768x768 class: 144 output
# tiles each — alone they need split-K slabs and a reduce launch) are queued per stream and shape and launched
# sixteen at a time by d2r_gemm_tn_grouped (2x faster per GEMM, one launch instead of thirty-two).
# ------------------------------------------------------------------------------------------------------
DEFER_WGRAD = True
DEFER_SHORT_WGRAD = True  # also the rank-B updates of the pooled-vector linears
_WGRAD_Q = {}  # stream handle -> {"stream": torch stream, "jobs": {shape key: [job, ...]}}
_WGRAD_FLUSH_AT = 32
D2R_LAYER_GROUP = 7  # encoder layers per grouped launch of their (large) weight gradients
# (measured at C2: 1 -> 32.6 ms/step, 2 -> 31.7, 4 / 6 / 13 -> 31.3-31.5; each stream runs 13 composite layers per step, so seven
# gives groups of 7 + 6 and no single-problem launch; at most ~0.8 GB of scratch kept alive)


def _defer_wgrad(g, x, lda_x, sink, bsink, N, K, M, w_master, bias):
    """Queue dW[N,K] += g[M,N]^T x[M,K] (and db[N] += colsum g) on the current stream."""
    _defer_wgrad_raw(_dt(x), N, K, M, lda_x, g.data_ptr(), x.data_ptr(), sink.data_ptr(),
                     None if bsink is None else bsink.data_ptr(), (w_master, bias), (g, x))


def _defer_wgrad_raw(dt, N, K, M, lda_x, g_ptr, x_ptr, sink_ptr, bsink_ptr, params, keepalive, flush_at=None):
    cur = torch.cuda.current_stream()
    q = _WGRAD_Q.get(cur.cuda_stream)
    if q is None:
        q = _WGRAD_Q[cur.cuda_stream] = {"stream": cur, "jobs": {}}
    key = (dt, N, K, M, lda_x, bsink_ptr is not None)
    jobs = q["jobs"].setdefault(key, [])
    # a parameter used at two call sites: its two products must not share a launch (the problems of a grouped launch run
    # in parallel and accumulate non-atomically) — launch what is queued first
    if any(j[2] == sink_ptr or (bsink_ptr is not None and j[3] == bsink_ptr) for j in jobs):
        _flush_wgrad_group(key, q["jobs"].pop(key))
        jobs = q["jobs"].setdefault(key, [])
    jobs.append((g_ptr, x_ptr, sink_ptr, bsink_ptr, params, keepalive))  # keepalive: tensors the launch will read
    if len(jobs) >= (flush_at or _WGRAD_FLUSH_AT):
        _flush_wgrad_group(key, q["jobs"].pop(key))


def _iparr(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, v in enumerate(ptrs):
        arr[i] = v
    return arr


DEFER_LN = True  # second stage of the encoder layers' LayerNorm backward: one launch per group of layers


def _defer_ln_sum(rows, D, ws_ptr, g_sink, b_sink, params, keepalive, flush_at=None):
    """Queue the summation of one LayerNorm's per-block partial gradients (left in scratch by d2r_encoder_layer_bwd) into its
    gamma / beta sinks; launched with the grouped weight gradients of the same layers (d2r_layernorm_bwd_sum_grouped)."""
    cur = torch.cuda.current_stream()
    q = _WGRAD_Q.get(cur.cuda_stream)
    if q is None:
        q = _WGRAD_Q[cur.cuda_stream] = {"stream": cur, "jobs": {}}
    key = ("ln", int(rows), int(D))
    jobs = q["jobs"].setdefault(key, [])
    if any(j[1] == g_sink for j in jobs):  # one LayerNorm used at two call sites: its sums must not share a launch
        _flush_wgrad_group(key, q["jobs"].pop(key))
        jobs = q["jobs"].setdefault(key, [])
    jobs.append((ws_ptr, g_sink, b_sink, None, params, keepalive))
    if len(jobs) >= (flush_at or _WGRAD_FLUSH_AT):
        _flush_wgrad_group(key, q["jobs"].pop(key))


def _flush_wgrad_group(key, jobs):
    _flush_wgrad_groups([(key, jobs)])


def _flush_wgrad_groups(groups):
    """Launches queued weight-gradient groups [(shape key, jobs)] of ONE stream: the LayerNorm sums per key, every GEMM job of
    every shape in ONE d2r_gemm_tn_grouped_v call (the 256-wide deep-pipelined kernel takes different shapes in one launch: the
    four weight gradients of seven encoder layers are 756 tiles instead of four launches of 63-252)."""
    gemm_jobs = []
    for key, jobs in groups:
        if key[0] == "ln":
            _, rows, D = key
            _lib.call("d2r_layernorm_bwd_sum_grouped", _iparr([j[0] for j in jobs]), _iparr([j[1] for j in jobs]), _iparr([j[2] for j in jobs]),
                      len(jobs), rows, D, 1, _stream(), meta=dict(group="d2r_layernorm_bwd_sum"))
        else:
            gemm_jobs.extend((key, j) for j in jobs)
    # one dtype and one "has bias sink" flavour per call
    flavours = {}
    for key, j in gemm_jobs:
        flavours.setdefault((key[0], key[5]), []).append((key, j))
    for (dt, has_b), kj in flavours.items():
        n = len(kj)
        Ms, Ns, Ks = [k[1] for k, _ in kj], [k[2] for k, _ in kj], [k[3] for k, _ in kj]  # dW [N_out, K_in] over M tokens
        lda, ldb, ldc = Ms, [k[4] for k, _ in kj], Ns
        meta = None
        if _lib._timer is not None:
            es = 4 if dt == F32 else 2
            meta = dict(group=f"gemm_{_DT_NAME[dt]}_TN_grouped", flops=sum(2.0 * a * b * c for a, b, c in zip(Ms, Ns, Ks)),
                        bytes=float(sum((c * a + c * b) * es + 2 * a * b * 4 for a, b, c in zip(Ms, Ns, Ks))))
        _lib.call("d2r_gemm_tn_grouped_v", dt, n, _iarr32(Ms), _iarr32(Ns), _iarr32(Ks), _iarr64(lda), _iarr64(ldb), _iarr64(ldc),
                  _iparr([j[0] for _, j in kj]), _iparr([j[1] for _, j in kj]), _iparr([j[2] for _, j in kj]),
                  _iparr([j[3] for _, j in kj]) if has_b else None, 1.0, _stream(), meta=meta)
    for key, jobs in groups:  # data-parallel bucket readiness (d2r_amd.dp)
        for j in jobs:
            for p in j[4]:
                cb = getattr(p, "_d2r_ready_cb", None) if p is not None else None
                if cb is not None:
                    cb(p)


def _iarr32(vals):
    arr = (C.c_int * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def _iarr64(vals):
    arr = (C.c_int64 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def _flush_stream_queue(q):
    """Every queued group of one stream's queue, together."""
    groups = [(key, q["jobs"].pop(key)) for key in list(q["jobs"])]
    if groups:
        _flush_wgrad_groups(groups)


def flush_wgrads_if(total: int):
    """Launches the current stream's queued weight gradients once at least `total` GEMM jobs wait (the encoder layers call this
    after queueing their four products: groups of D2R_LAYER_GROUP whole layers leave as one launch)."""
    q = _WGRAD_Q.get(torch.cuda.current_stream().cuda_stream)
    if q is not None and sum(len(v) for k, v in q["jobs"].items() if k[0] != "ln") >= total:
        _flush_stream_queue(q)


EARLY_FLUSH = False  # set by d2r_amd.dp when the gradient all-reduce overlaps with backward
_early_flushed = set()


def _flush_before_encoders():
    """Overlapped data parallelism: the routing modules' queued weight gradients are launched when the backward pass of a
    stream reaches its first whole encoder layer (everything downstream of the encoders is done by then), so that their
    buckets can be reduced during the encoders' backward instead of after it."""
    h = torch.cuda.current_stream().cuda_stream
    if h in _early_flushed:
        return
    _early_flushed.add(h)
    q = _WGRAD_Q.get(h)
    if q is not None:
        _flush_stream_queue(q)


def flush_wgrads():
    """Launches every queued weight-gradient group on the stream it was queued on.  Runs at the end of each backward
    pass (before the streams are joined) and may be called by anything that needs the gradients earlier."""
    for q in _WGRAD_Q.values():
        if q["jobs"]:
            with torch.cuda.stream(q["stream"]):
                _flush_stream_queue(q)


_COMPUTE_STREAMS = []  # extra streams the forward forks onto (d2r_amd.modules registers its text / vision streams)
_join_queued = False


def register_compute_stream(s):
    if s not in _COMPUTE_STREAMS:
        _COMPUTE_STREAMS.append(s)


def _backward_join_cb():
    """Final callback of a backward pass; autograd runs it on the stream that was current around .backward(): that
    stream now waits for every compute and weight-gradient stream, so whatever follows (optimiser step, all-reduce,
    a host read of .grad) is ordered after ALL gradient writes — independent of which autograd leaves happened to run."""
    global _join_queued
    _join_queued = False
    _early_flushed.clear()
    flush_wgrads()
    cur = torch.cuda.current_stream()
    for st in _COMPUTE_STREAMS:
        cur.wait_stream(st)
    for st in _WGRAD.values():
        cur.wait_stream(st)


def _ensure_backward_join():
    """Called from backward functions: queues _backward_join_cb once per backward pass."""
    global _join_queued
    if not _join_queued:
        _join_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(_backward_join_cb)


def wgrad_join():
    """Makes the current stream wait for every weight-gradient side stream."""
    if _WGRAD:
        cur = torch.cuda.current_stream()
        for side in _WGRAD.values():
            cur.wait_stream(side)


def _workspace(nbytes: int, device, stream: Optional[int] = None) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): kernels that share it run stream-ordered."""
    key = (device.index, _stream() if stream is None else stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        ws[-4096:].zero_()  # tile counters of the in-kernel split-K GEMM (include/d2r_hip.h, d2r_gemm_desc.workspace): zero at first use, kept zero
        _WS[key] = ws
    return ws


def _rows2d(x: torch.Tensor):
    """Describes x as [M, K] rows with a uniform row stride (last-dim stride 1). Returns (M, K, ld) or None."""
    if x.stride(-1) != 1 and x.shape[-1] != 1:
        return None
    K = x.shape[-1]
    if x.dim() == 1:
        return 1, K, K
    if x.dim() == 2:
        ld = x.stride(0) if x.shape[0] > 1 else max(K, 1)
        return x.shape[0], K, max(ld, K)
    if x.is_contiguous():
        return x.numel() // K, K, K
    return None


def _as_rows(x: torch.Tensor):
    r = _rows2d(x)
    if r is None:
        x = x.contiguous()
        r = _rows2d(x)
    return x, r


# ------------------------------------------------------------------------------------------------------
# raw GEMM launcher
# ------------------------------------------------------------------------------------------------------
def gemm(layout, M, N, K, A, lda, B, ldb, Cc, ldc, *, dtype, c_dtype, nb=1, nh=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
         alpha=1.0, beta=0.0, bias=None, act=ACT_NONE, residual=None, ldr=0, sR=(0, 0), preact=None, tag=None,
         splitk_ws=None, s_bias=0, dbias=None, stream=None):
    d = GemmDesc(dtype=dtype, c_dtype=c_dtype, layout=layout, act=act, M=M, N=N, K=K, nb=nb, nh=nh, alpha=alpha,
                 beta=beta, A=A, lda=lda, sAb=sA[0], sAh=sA[1], B=B, ldb=ldb, sBb=sB[0], sBh=sB[1], C=Cc, ldc=ldc,
                 sCb=sC[0], sCh=sC[1], bias=bias, residual=residual, ldr=ldr, sRb=sR[0], sRh=sR[1], preact=preact)
    d.s_bias_b = s_bias
    d.dbias = dbias
    if splitk_ws is not None:  # deterministic split-K scratch (dW GEMMs): fp32 partial slabs
        d.workspace, d.workspace_bytes = splitk_ws.data_ptr(), splitk_ws.numel()
    meta = None
    if _lib._timer is not None:  # algorithmic flops / bytes of this launch for bench.py's roofline
        z, es, cs = nb * nh, (4 if dtype == F32 else 2), (4 if c_dtype == F32 else 2)
        meta = dict(group=tag or f"gemm_{_DT_NAME[dtype]}_{('NT', 'NN', 'TN')[layout]}",
                    flops=2.0 * M * N * K * z,
                    bytes=float(z) * ((M * K + N * K) * es + M * N * cs * (1 + (beta != 0.0) + bool(residual) + bool(preact))))
    _lib.call("d2r_gemm", C.byref(d), _stream() if stream is None else stream, meta=meta)


def colsum(g: torch.Tensor, M: int, N: int, ld: int) -> torch.Tensor:
    out = torch.empty(N, dtype=torch.float32, device=g.device)
    nbytes = _lib.load().d2r_colsum_workspace(M, N)
    ws = _workspace(nbytes, g.device)
    _lib.call("d2r_colsum", _dt(g), g.data_ptr(), ld, M, N, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty_like(x, dtype=dtype)
    _lib.call("d2r_cast", _dt(x), x.data_ptr(), _dt(out), out.data_ptr(), x.numel(), _stream())
    return out


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast(g.contiguous(), ctx.src), None


class _StreamJoin(torch.autograd.Function):
    """Identity on a pair of tensors, issued on the launching stream at a fork/join point of the two-stream
    forward.  Its only job is to give the BACKWARD the same join-then-fork shape: the gradients coming from the two
    side streams are accumulated at this node (on the launching stream) before they flow on to the two encoders, so
    autograd never makes the side streams wait on each other directly (that pattern breaks hipGraph capture)."""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a), b.view_as(b)

    @staticmethod
    def backward(ctx, ga, gb):
        return ga, gb


def stream_join(a, b):
    return _StreamJoin.apply(a, b)


class _SplitColumns(torch.autograd.Function):
    """[.., n*w] -> n contiguous [.., w] column blocks (one row-copy launch each way per block); the backward assembles the
    blocks' gradients into ONE tensor (autograd would otherwise add n zero-padded full-size tensors)."""

    @staticmethod
    def forward(ctx, x, n):
        x = x.contiguous()
        W = x.shape[-1]
        w = W // n
        rows = x.numel() // W
        es = x.element_size()
        outs = []
        for i in range(n):
            o = torch.empty(*x.shape[:-1], w, dtype=x.dtype, device=x.device)
            _lib.call("d2r_copy_rows", o.data_ptr(), w * es, x.data_ptr() + i * w * es, W * es, w * es, rows, _stream())
            outs.append(o)
        ctx.meta = (n, w, W, rows, es, x.shape, x.dtype)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, w, W, rows, es, shape, dtype = ctx.meta
        g = torch.empty(shape, dtype=dtype, device=[t for t in gs if t is not None][0].device)
        for i, gi in enumerate(gs):
            if gi is None:
                g[..., i * w:(i + 1) * w].zero_()
                continue
            gi = gi.contiguous()
            _lib.call("d2r_copy_rows", g.data_ptr() + i * w * es, W * es, gi.data_ptr(), w * es, w * es, rows, _stream())
        return g, None


def split_columns(x, n):
    return list(_SplitColumns.apply(x, n))


def cast_ad(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


# ------------------------------------------------------------------------------------------------------
# K11 linear: y = act(x W^T + b) (+ residual)
# ------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_master, bias, w_compute, act, residual, out_dtype):
        _require_cuda(x, w_compute)
        x, (M, K, lda) = _as_rows(x)
        N = w_compute.shape[0]
        assert w_compute.shape[1] == K and w_compute.is_contiguous(), "weight must be contiguous [N,K]"
        assert x.dtype == w_compute.dtype, f"activation {x.dtype} vs weight {w_compute.dtype}"
        assert not (act != ACT_NONE and residual is not None), "activation + residual in one epilogue is unused on this path"
        odt = out_dtype or x.dtype
        y = torch.empty(*x.shape[:-1], N, dtype=odt, device=x.device)
        pre = torch.empty_like(y) if act in _ACT_FROM_PREACT else None
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == y.shape and residual.dtype == odt
        gemm(GEMM_NT, M, N, K, x.data_ptr(), lda, w_compute.data_ptr(), K, y.data_ptr(), N, dtype=_dt(x),
             c_dtype=_dt(y), bias=_ptr(bias), act=act, residual=_ptr(residual), ldr=N, preact=_ptr(pre),
             splitk_ws=_workspace(64 << 20, x.device) if M <= 64 else None)
        ctx.act, ctx.dims, ctx.has_bias, ctx.has_res = act, (M, N, K, lda), bias is not None, residual is not None
        ctx.save_for_backward(x, w_compute, pre if pre is not None else (y if act in _ACT_FROM_OUTPUT else None))
        ctx.w_needs = w_master.requires_grad
        ctx.w_master = w_master
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, g):
        _ensure_backward_join()
        x, w, ref = ctx.saved_tensors
        M, N, K, lda = ctx.dims
        g = g.contiguous()
        gres = g if ctx.has_res else None
        if ctx.act != ACT_NONE:
            gg = torch.empty_like(g)
            _lib.call("d2r_act_bwd", _dt(g), ctx.act, g.data_ptr(), ref.data_ptr(), gg.data_ptr(), g.numel(), _stream())
            g = gg
        if g.dtype != x.dtype:  # fp32-output GEMMs (router logits, SAF scores): operands must share a dtype
            g = cast(g, x.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dxc = torch.empty(M, K, dtype=x.dtype, device=x.device)
            gemm(GEMM_NN, M, K, N, g.data_ptr(), N, w.data_ptr(), K, dxc.data_ptr(), K, dtype=_dt(x), c_dtype=_dt(x),
                 splitk_ws=_workspace(64 << 20, x.device) if M <= 64 else None)
            dx = dxc.view(*x.shape[:-1], K) if x.dim() != 2 else dxc
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        db_ptr = None
        if want_db and ctx.w_needs:  # bias gradient = column sums of g: a side product of the dW GEMM below
            bsink = getattr(ctx.bias, "_d2r_grad", None)
            if bsink is None:
                db = torch.zeros(N, dtype=torch.float32, device=x.device)
                db_ptr = db.data_ptr()
            else:
                db_ptr = bsink.data_ptr()
        if ctx.w_needs:
            sink = getattr(ctx.w_master, "_d2r_grad", None)  # flat fp32 gradient buffer (d2r_amd.params.ParamStore)
            if (sink is not None and DEFER_WGRAD and not WGRAD_STREAMS and db is None and 262144 <= N * K <= 1500000
                    and (M >= 1024 or DEFER_SHORT_WGRAD) and lda == K):
                # small-output weight gradient: queued, launched with up to fifteen others of its shape (grouped GEMM;
                # below 64 output tiles a group cannot fill the chip and split-K on the spot is faster).
                # The queue holds a reference to g, so autograd cannot accumulate into it in place meanwhile (it only
                # does that to tensors it owns exclusively) even when g is also handed on as the skip gradient.
                _defer_wgrad(g, x, lda, sink, bsink if want_db else None, N, K, M, ctx.w_master, ctx.bias if want_db else None)
            elif sink is not None:
                # dW accumulates straight into the zero-initialised flat buffer: no temp, no autograd add kernel;
                # nothing downstream in backward reads it, so it runs on the weight-gradient side stream
                # (not when g is also handed on as the skip-connection gradient: autograd may then accumulate into
                # that very tensor in place on the main stream while the side stream still reads it)
                side = wgrad_stream(g, x) if (db is None and not ctx.has_res) else None
                gemm(GEMM_TN, N, K, M, g.data_ptr(), N, x.data_ptr(), lda, sink.data_ptr(), K, dtype=_dt(x),
                     c_dtype=F32, beta=1.0, splitk_ws=_workspace(64 << 20, x.device, side), dbias=db_ptr, stream=side)
                cb = getattr(ctx.w_master, "_d2r_ready_cb", None)  # data-parallel bucket readiness (d2r_amd.dp)
                if cb is not None:
                    cb(ctx.w_master)
            else:
                dw = torch.empty(N, K, dtype=torch.float32, device=x.device)
                gemm(GEMM_TN, N, K, M, g.data_ptr(), N, x.data_ptr(), lda, dw.data_ptr(), K, dtype=_dt(x), c_dtype=F32,
                     splitk_ws=_workspace(64 << 20, x.device), dbias=db_ptr)
            if want_db and db is None:
                cb = getattr(ctx.bias, "_d2r_ready_cb", None)
                if cb is not None:
                    cb(ctx.bias)
        elif want_db:
            db = colsum(g, M, N, N)
        return dx, dw, db, None, None, gres, None


def linear(x, w_master, bias, w_compute=None, act=ACT_NONE, residual=None, out_dtype=None):
    """F.linear with fused bias/activation/residual epilogue.  ``w_master`` is the fp32 parameter that receives
    the gradient; ``w_compute`` the (possibly bf16) copy multiplied with."""
    return _Linear.apply(x, w_master, bias, w_master if w_compute is None else w_compute, act, residual, out_dtype)


class _GroupedLinear(torch.autograd.Function):
    """G independent linears as ONE batched GEMM: y[:, g] = act(x_g W_g^T + b_g).
    x: [B, G*K] (columns grouped) or, with x_gm, group-major [G, B, K]; W: fused [G*N, K]; b: [G*N]; y: [B, G*N].
    Serves the 20+20 rank-15 merge projections of Block (models/XModules.py:541-546) and the six routers of a
    routing layer (models/Router.py:24)."""

    @staticmethod
    def forward(ctx, x, w_master, bias, w_compute, G, act, x_gm):
        x = x.contiguous()
        K = w_compute.shape[1]
        N = w_compute.shape[0] // G
        B = x.shape[1] if x_gm else x.shape[0]
        xs = (K, B * K) if x_gm else (G * K, K)  # (row stride, group stride) of x
        assert x.dtype == w_compute.dtype and act not in _ACT_FROM_PREACT
        y = torch.empty(B, G * N, dtype=x.dtype, device=x.device)
        gemm(GEMM_NT, B, N, K, x.data_ptr(), xs[0], w_compute.data_ptr(), K, y.data_ptr(), G * N, dtype=_dt(x),
             c_dtype=_dt(y), nb=G, sA=(xs[1], 0), sB=(N * K, 0), sC=(N, 0), bias=bias.data_ptr(), act=act, s_bias=N)
        ctx.save_for_backward(x, w_compute, y if act in _ACT_FROM_OUTPUT else None)
        ctx.cfg = (B, G, N, K, xs, act, x_gm)
        ctx.w_master = w_master
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, ref = ctx.saved_tensors
        B, G, N, K, xs, act, x_gm = ctx.cfg
        g = g.contiguous()
        if act != ACT_NONE:
            gg = torch.empty_like(g)
            _lib.call("d2r_act_bwd", _dt(g), act, g.data_ptr(), ref.data_ptr(), gg.data_ptr(), g.numel(), _stream())
            g = gg
        dx = torch.empty_like(x)
        gemm(GEMM_NN, B, K, N, g.data_ptr(), G * N, w.data_ptr(), K, dx.data_ptr(), xs[0], dtype=_dt(x), c_dtype=_dt(x),
             nb=G, sA=(N, 0), sB=(N * K, 0), sC=(xs[1], 0))
        sink = getattr(ctx.w_master, "_d2r_grad", None)
        dw = None
        if sink is not None:
            tgt, beta = sink, 1.0
        else:
            dw = torch.empty(G * N, K, dtype=torch.float32, device=x.device)
            tgt, beta = dw, 0.0
        gemm(GEMM_TN, N, K, B, g.data_ptr(), G * N, x.data_ptr(), xs[0], tgt.data_ptr(), K, dtype=_dt(x), c_dtype=F32,
             nb=G, sA=(N, 0), sB=(xs[1], 0), sC=(N * K, 0), beta=beta)
        if sink is not None:
            cb = getattr(ctx.w_master, "_d2r_ready_cb", None)
            if cb is not None:
                cb(ctx.w_master)
        db = colsum(g, B, G * N, G * N)
        return dx, dw, db, None, None, None, None


def grouped_linear(x, w_master, bias, w_compute, G, act=ACT_NONE, x_gm=False):
    return _GroupedLinear.apply(x, w_master, bias, w_master if w_compute is None else w_compute, G, act, x_gm)


class _MatmulNT(torch.autograd.Function):
    """C = A B^T for 2-D row-major A [M,K], B [N,K]; fp32 output (similarity matrices, [B,B])."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        M, K = a.shape
        N = b.shape[0]
        c = torch.empty(M, N, dtype=torch.float32, device=a.device)
        gemm(GEMM_NT, M, N, K, a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, dtype=_dt(a), c_dtype=F32)
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        M, K = a.shape
        N = b.shape[0]
        g = cast(g.contiguous(), a.dtype)
        da = torch.empty_like(a)
        db = torch.empty_like(b)
        gemm(GEMM_NN, M, K, N, g.data_ptr(), N, b.data_ptr(), K, da.data_ptr(), K, dtype=_dt(a), c_dtype=_dt(a))
        gemm(GEMM_TN, N, K, M, g.data_ptr(), N, a.data_ptr(), K, db.data_ptr(), K, dtype=_dt(a), c_dtype=_dt(a))
        return da, db


# ------------------------------------------------------------------------------------------------------
# data parallelism, global-batch-exact mode (d2r_amd.dp.DataParallel(global_batch_exact=True)): the three places where
# the reference couples the samples of a batch - the BatchNorm1d(1) of every GLAC cell (models/XModules.py:376,381), the [B,B]
# path / cls similarity matrices and the batch-softmax JS loss (models/InteractionModule.py:53, models/modeling_unimo.py:845-849,
# models/XModules.py:32-41) - see the WHOLE batch: BatchNorm sums are all-reduced (2 doubles, forward and backward), `paths` and the
# cls vectors are all-gathered.  Every rank then computes the same JS term of the global batch and adds it to its local cross
# entropy mean: the average of the ranks' losses is the reference's loss on the global batch, and the averaged gradients are its
# gradients.  None: local-batch statistics and a local [b,b] JS term per rank (what DDP of the reference would give).
# ------------------------------------------------------------------------------------------------------
DP_EXACT = None  # (process group, world size, rank)


class _GatherBatch(torch.autograd.Function):
    """x [b, ...] of every rank -> [world * b, ...] (rank order).  Every rank evaluates the SAME function of the gathered tensor, so
    the sum over ranks of the gradients w.r.t. this rank's rows is world x this rank's own copy: no communication in the backward."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist
        group, world, rank = DP_EXACT
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        ctx.meta = (world, rank, x.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        world, rank, b = ctx.meta
        return g[rank * b:(rank + 1) * b] * float(world)


def gather_batch(x):
    return x if DP_EXACT is None else _GatherBatch.apply(x)


def _allreduce_pair(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=DP_EXACT[0])


_BN_SYNC_BUFS = {}  # id -> fp64 device tensor handed to a whole-module C call as bn_sync_buf


def _bn_sync(user, pair, stream):
    """d2r_interaction_desc.bn_sync: sums the two doubles at `pair` over the ranks (called from inside d2r_interaction_fwd / _bwd)."""
    try:
        buf = _BN_SYNC_BUFS[int(user)]
        off = (int(pair) - buf.data_ptr()) // 8
        assert 0 <= off <= buf.numel() - 2 and torch.cuda.current_stream().cuda_stream == (stream or 0)
        _allreduce_pair(buf[off:off + 2])
        return 0
    except Exception:  # (an exception must not unwind through the C frames)
        import traceback
        traceback.print_exc()
        return 1


_BN_SYNC_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)(_bn_sync)


def _bn_sync_fields(d, device):
    """Fills the global-batch-exact BatchNorm fields of a d2r_interaction_desc (no-op outside that mode)."""
    if DP_EXACT is None:
        d.bn_sync, d.bn_sync_buf = None, None
        return None
    buf = torch.zeros(4 * d.nlayer, dtype=torch.float64, device=device)
    _BN_SYNC_BUFS[id(buf)] = buf
    d.bn_sync = C.cast(_BN_SYNC_CB, C.c_void_p)
    d.bn_sync_user, d.bn_sync_buf, d.bn_world = id(buf), buf.data_ptr(), DP_EXACT[1]
    return buf


def matmul_nt(a, b):
    return _MatmulNT.apply(a, b)


# ------------------------------------------------------------------------------------------------------
# attention: O = softmax(scale * Q K^T + mask) V (+ residual), H heads; logits kept fp32
# ------------------------------------------------------------------------------------------------------
DETERMINISTIC = os.environ.get("D2R_DETERMINISTIC", "0") == "1"  # serialise the two routing modules' backward passes (bit-reproducible steps)
FUSED_MHA = True  # 0: three-launch path (the only one for fp32)
FUSED_XATTN = True


def _attn_fwd(q, k, v, geo, H, scale, mask, residual, dtype, device, p_drop=0.0):
    """q, k, v: (data_ptr, row stride, batch stride) in elements of `dtype`; geo = (B, Lq, Lk, E).  Returns (o, P, seed).
    p_drop > 0 (training-time attention-probability dropout, models/modeling_unimo.py:204,388): the fused multi-head core
    applies the mask in registers (same counter-based generator and element indexing as the three-launch path, where P is
    saved BEFORE dropout); either way the backward regenerates the mask from the seed."""
    B, Lq, Lk, E = geo
    d = E // H
    Lkp = (Lk + 7) // 8 * 8
    dt = _dt_of(dtype)
    tag = "xattn_core_fwd" if H == 1 else "mha_core_fwd"
    if FUSED_MHA and H > 1 and _lib.load().d2r_mha_supported(dt, Lq, Lk, d):
        # one launch, scores/probabilities stay in registers; the saved state is the row log-sum-exp, not P
        o = torch.empty(B, Lq, E, dtype=dtype, device=device)
        lse = torch.empty(B, H, Lq, dtype=torch.float32, device=device)
        seed = _next_dropout_seed() if p_drop > 0.0 else 0
        _lib.call("d2r_mha_fwd", dt, q[0], q[1], q[2], k[0], k[1], k[2], v[0], v[1], v[2], o.data_ptr(), E, Lq * E,
                  _ptr(residual), E, Lq * E, _ptr(mask), lse.data_ptr(), B, H, Lq, Lk, d, scale, p_drop, seed, _stream(),
                  meta=dict(group=tag, algo_bytes=float(B * (2 * Lq + 2 * Lk) * E * 2)))
        return o, lse, seed
    if p_drop <= 0.0 and FUSED_XATTN and H == 1 and _lib.load().d2r_xattn_supported(dt, Lq, Lk, E):
        o = torch.empty(B, Lq, E, dtype=dtype, device=device)
        lse = torch.empty(B, 1, Lq, dtype=torch.float32, device=device)
        _lib.call("d2r_xattn_fwd", dt, q[0], q[1], q[2], k[0], k[1], k[2], v[0], v[1], v[2], o.data_ptr(), E, Lq * E,
                  _ptr(residual), E, Lq * E, _ptr(mask), lse.data_ptr(), B, Lq, Lk, E, scale, _stream(),
                  meta=dict(group=tag, algo_bytes=float(B * (2 * Lq + 2 * Lk) * E * 2)))
        return o, lse, 0
    S = torch.empty(B, H, Lq, Lkp, dtype=torch.float32, device=device)
    gemm(GEMM_NT, Lq, Lk, d, q[0], q[1], k[0], k[1], S.data_ptr(), Lkp, dtype=dt, c_dtype=F32, nb=B, nh=H,
         sA=(q[2], d), sB=(k[2], d), sC=(H * Lq * Lkp, Lq * Lkp), tag=tag)
    if _lib._timer is not None:  # algorithmic bytes of the fused core = B(2Lq+2Lk)D s
        _lib._timer.records[-1][1]["algo_bytes"] = float(B * (2 * Lq + 2 * Lk) * E * (4 if dt == F32 else 2))
    P = S if dtype == torch.float32 else torch.empty(B, H, Lq, Lkp, dtype=dtype, device=device)
    _lib.call("d2r_softmax_fwd", F32, dt, S.data_ptr(), P.data_ptr(), Lkp, B * H * Lq, Lk, scale, _ptr(mask), H * Lq,
              _stream(), meta=dict(group=tag))  # padding columns of P are never read (K = Lk below)
    seed, Pd = 0, P
    if p_drop > 0.0:  # dropout on the probabilities: P itself is what the softmax backward needs
        seed = _next_dropout_seed()
        Pd = torch.empty_like(P)
        _lib.call("d2r_dropout", _dt(P), P.data_ptr(), None, Pd.data_ptr(), P.numel(), p_drop, seed, _stream())
    o = torch.empty(B, Lq, E, dtype=dtype, device=device)
    gemm(GEMM_NN, Lq, d, Lk, Pd.data_ptr(), Lkp, v[0], v[1], o.data_ptr(), E, dtype=dt, c_dtype=dt, nb=B, nh=H,
         sA=(H * Lq * Lkp, Lq * Lkp), sB=(v[2], d), sC=(Lq * E, d), residual=_ptr(residual), ldr=E, sR=(Lq * E, d),
         tag=tag)
    return o, P, seed


def _attn_bwd(g, q, k, v, P, dq, dk, dv, geo, H, scale, dtype, device, mask=None, p_drop=0.0, seed=0, o=None, res=None):
    """g: contiguous [B,Lq,E]; q/k/v and dq/dk/dv: (ptr, row stride, batch stride).  P: probabilities [B,H,Lq,Lkp]
    of the unfused forward, or the fp32 log-sum-exp [B,H,Lq] of the fused one."""
    B, Lq, Lk, E = geo
    d = E // H
    dt = _dt_of(dtype)
    tag = "xattn_core_bwd" if H == 1 else "mha_core_bwd"
    if P.dim() == 3 and H == 1:  # fused single-head forward: dS, P and dQ in one launch, then dV / dK as ONE grouped batched launch
        Lkp = (Lk + 7) // 8 * 8
        Pb = torch.empty(B, Lq, Lkp, dtype=dtype, device=device)
        dS = torch.empty(B, Lq, Lkp, dtype=dtype, device=device)
        one = lambda x: _iparr([x])
        _lib.call("d2r_xattn_bwd_multi", dt, 1, one(q[0]), q[1], q[2], one(k[0]), k[1], k[2], one(v[0]), v[1], v[2], one(g.data_ptr()), E, Lq * E,
                  None if o is None else one(o.data_ptr()), E, Lq * E, None if res is None else one(res.data_ptr()), E, Lq * E,
                  _ptr(mask), one(P.data_ptr()), one(dq[0]), dq[1], dq[2], one(dk[0]), dk[1], dk[2], one(dv[0]), dv[1], dv[2],
                  one(Pb.data_ptr()), one(dS.data_ptr()), Lkp, B, Lq, Lk, E, scale, _stream(),
                  meta=dict(group=tag, algo_bytes=float(2 * B * (2 * Lq + 2 * Lk) * E * 2)))
        return
    if P.dim() == 3:  # fused forward ran: recompute P from q, k and the log-sum-exp
        dsum = torch.empty_like(P) if (Lq > 256 or Lk > 256) else None  # long sequences: D handed from the dQ to the dK/dV kernel
        _lib.call("d2r_mha_bwd", dt, q[0], q[1], q[2], k[0], k[1], k[2], v[0], v[1], v[2], g.data_ptr(), E, Lq * E,
                  _ptr(mask), P.data_ptr(), _ptr(dsum), dq[0], dq[1], dq[2], dk[0], dk[1], dk[2], dv[0], dv[1], dv[2], B, H, Lq, Lk,
                  d, scale, p_drop, seed, _stream(), meta=dict(group=tag, algo_bytes=float(2 * B * (2 * Lq + 2 * Lk) * E * 2)))
        return
    Lkp = P.shape[-1]
    sP, sG = (H * Lq * Lkp, Lq * Lkp), (Lq * E, d)
    Pd = P
    if p_drop > 0.0:  # regenerate the dropped probabilities from the seed (same element indexing as the forward)
        Pd = torch.empty_like(P)
        _lib.call("d2r_dropout", _dt(P), P.data_ptr(), None, Pd.data_ptr(), P.numel(), p_drop, seed, _stream())
    gemm(GEMM_TN, Lk, d, Lq, Pd.data_ptr(), Lkp, g.data_ptr(), E, dv[0], dv[1], dtype=dt, c_dtype=dt, nb=B, nh=H,
         sA=sP, sB=sG, sC=(dv[2], d), tag=tag)
    dP = torch.empty(B, H, Lq, Lkp, dtype=torch.float32, device=device)
    gemm(GEMM_NT, Lq, Lk, d, g.data_ptr(), E, v[0], v[1], dP.data_ptr(), Lkp, dtype=dt, c_dtype=F32, nb=B, nh=H,
         sA=sG, sB=(v[2], d), sC=sP, tag=tag)
    if p_drop > 0.0:  # gradient through the dropout: same mask, same 1/(1-p)
        _lib.call("d2r_dropout", F32, dP.data_ptr(), None, dP.data_ptr(), dP.numel(), p_drop, seed, _stream())
    dS = dP if dtype == torch.float32 else torch.empty(B, H, Lq, Lkp, dtype=dtype, device=device)
    _lib.call("d2r_softmax_bwd", dt, F32, P.data_ptr(), dP.data_ptr(), dS.data_ptr(), Lkp, B * H * Lq, Lk, scale,
              _stream(), meta=dict(group=tag))
    gemm(GEMM_NN, Lq, d, Lk, dS.data_ptr(), Lkp, k[0], k[1], dq[0], dq[1], dtype=dt, c_dtype=dt, nb=B, nh=H,
         sA=sP, sB=(k[2], d), sC=(dq[2], d), tag=tag)
    gemm(GEMM_TN, Lk, d, Lq, dS.data_ptr(), Lkp, q[0], q[1], dk[0], dk[1], dtype=dt, c_dtype=dt, nb=B, nh=H,
         sA=sP, sB=(q[2], d), sC=(dk[2], d), tag=tag)


def _desc(t, col0, ncols_total):
    """(ptr, row stride, batch stride) of the column block starting at col0 of a contiguous [B,L,ncols_total] tensor."""
    return (t.data_ptr() + col0 * t.element_size(), ncols_total, t.shape[1] * ncols_total)


class _Attention(torch.autograd.Function):
    """separate q [B,Lq,E], k, v [B,Lk,E]"""

    @staticmethod
    def forward(ctx, q, k, v, H, scale, mask, residual, p_drop=0.0):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Lq, E = q.shape
        geo = (B, Lq, k.shape[1], E)
        if residual is not None:
            residual = residual.contiguous()
        o, P, seed = _attn_fwd(_desc(q, 0, E), _desc(k, 0, E), _desc(v, 0, E), geo, H, scale, mask, residual, q.dtype,
                               q.device, p_drop)
        ctx.drop = (p_drop, seed)
        ctx.save_for_backward(q, k, v, P, o if H == 1 else None, residual if H == 1 else None)
        ctx.mask = mask
        ctx.cfg = (geo, H, scale, residual is not None)
        return o

    @staticmethod
    def backward(ctx, g):
        q, k, v, P, o, res = ctx.saved_tensors
        geo, H, scale, has_res = ctx.cfg
        E = geo[3]
        g = g.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _attn_bwd(g, _desc(q, 0, E), _desc(k, 0, E), _desc(v, 0, E), P, _desc(dq, 0, E), _desc(dk, 0, E), _desc(dv, 0, E),
                  geo, H, scale, q.dtype, q.device, ctx.mask, *ctx.drop, o=o, res=res)
        return dq, dk, dv, None, None, None, (g if has_res else None), None


class _AttentionQKV(torch.autograd.Function):
    """packed self-attention input qkv [B,L,3E] = one fused projection GEMM (q | k | v along the last dim); the
    backward writes dq/dk/dv straight into ONE [B,L,3E] gradient (no slice-backward copies, no adds)."""

    @staticmethod
    def forward(ctx, qkv, H, scale, mask, residual, p_drop=0.0):
        qkv = qkv.contiguous()
        B, L, E3 = qkv.shape
        E = E3 // 3
        geo = (B, L, L, E)
        if residual is not None:
            residual = residual.contiguous()
        o, P, seed = _attn_fwd(_desc(qkv, 0, E3), _desc(qkv, E, E3), _desc(qkv, 2 * E, E3), geo, H, scale, mask, residual,
                               qkv.dtype, qkv.device, p_drop)
        ctx.drop = (p_drop, seed)
        ctx.save_for_backward(qkv, P, o if H == 1 else None, residual if H == 1 else None)
        ctx.mask = mask
        ctx.cfg = (geo, H, scale, residual is not None)
        return o

    @staticmethod
    def backward(ctx, g):
        qkv, P, o, res = ctx.saved_tensors
        geo, H, scale, has_res = ctx.cfg
        E, E3 = geo[3], 3 * geo[3]
        g = g.contiguous()
        d = torch.empty_like(qkv)
        _attn_bwd(g, _desc(qkv, 0, E3), _desc(qkv, E, E3), _desc(qkv, 2 * E, E3), P, _desc(d, 0, E3), _desc(d, E, E3),
                  _desc(d, 2 * E, E3), geo, H, scale, qkv.dtype, qkv.device, ctx.mask, *ctx.drop, o=o, res=res)
        return d, None, None, None, (g if has_res else None), None


class _AttentionKV(torch.autograd.Function):
    """cross-attention: q [B,Lq,E] and packed kv [B,Lk,2E] (k | v from one fused projection of the other modality)."""

    @staticmethod
    def forward(ctx, q, kv, H, scale, mask, residual, p_drop=0.0):
        q, kv = q.contiguous(), kv.contiguous()
        B, Lq, E = q.shape
        geo = (B, Lq, kv.shape[1], E)
        if residual is not None:
            residual = residual.contiguous()
        o, P, seed = _attn_fwd(_desc(q, 0, E), _desc(kv, 0, 2 * E), _desc(kv, E, 2 * E), geo, H, scale, mask, residual,
                               q.dtype, q.device, p_drop)
        ctx.drop = (p_drop, seed)
        ctx.save_for_backward(q, kv, P, o if H == 1 else None, residual if H == 1 else None)
        ctx.mask = mask
        ctx.cfg = (geo, H, scale, residual is not None)
        return o

    @staticmethod
    def backward(ctx, g):
        q, kv, P, o, res = ctx.saved_tensors
        geo, H, scale, has_res = ctx.cfg
        E = geo[3]
        g = g.contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        _attn_bwd(g, _desc(q, 0, E), _desc(kv, 0, 2 * E), _desc(kv, E, 2 * E), P, _desc(dq, 0, E), _desc(dkv, 0, 2 * E),
                  _desc(dkv, E, 2 * E), geo, H, scale, q.dtype, q.device, ctx.mask, *ctx.drop, o=o, res=res)
        return dq, dkv, None, None, None, (g if has_res else None), None


def attention(q, k, v, num_heads, scale, mask=None, residual=None, p_drop=0.0):
    """q [B,Lq,E], k/v [B,Lk,E]; mask: fp32 additive [B,Lk] or None; residual [B,Lq,E] added to the output."""
    return _Attention.apply(q, k, v, num_heads, float(scale), mask, residual, float(p_drop))


def attention_qkv(qkv, num_heads, scale, mask=None, residual=None, p_drop=0.0):
    return _AttentionQKV.apply(qkv, num_heads, float(scale), mask, residual, float(p_drop))


def attention_kv(q, kv, num_heads, scale, mask=None, residual=None, p_drop=0.0):
    return _AttentionKV.apply(q, kv, num_heads, float(scale), mask, residual, float(p_drop))


# ------------------------------------------------------------------------------------------------------
# K15 whole encoder layer (16-bit compute dtypes): one C call forward, one backward
# ------------------------------------------------------------------------------------------------------
class LayerBundle:
    """Pointers of one encoder layer's parameters and fp32 gradient sinks (stable across steps: they live in the flat
    buffers of d2r_amd.params.ParamStore), pre-packed into a descriptor template."""

    def __init__(self, *, pre_ln, act, H, eps, qkv, o, fc1, fc2, ln1, ln2):
        """qkv/o/fc1/fc2: (weight leaf, bias leaf) with ._d2r_lp / ._d2r_grad; ln1/ln2: (gamma, beta) fp32 leaves."""
        self.params = [t for pair in (qkv, o, fc1, fc2, ln1, ln2) for t in pair]
        if any(getattr(t, "_d2r_grad", None) is None for t in self.params) or any(
                getattr(w, "_d2r_lp", None) is None for w in (qkv[0], o[0], fc1[0], fc2[0])):
            raise _lib.D2RError("LayerBundle needs a model prepared by ParamStore with a 16-bit weight shadow")
        E, Fi = o[0].shape[0], fc1[0].shape[0]
        assert qkv[0].shape == (3 * E, E) and fc2[0].shape == (E, Fi)
        self.E, self.F, self.H = E, Fi, H
        t = _lib.EncoderLayerDesc()
        self.tdtype = qkv[0]._d2r_lp.dtype  # the 16-bit compute dtype of the shadow: bf16 or fp16
        self.dt = _dt_of(self.tdtype)
        t.dtype, t.pre_ln, t.act, t.E, t.H, t.F, t.eps, t.scale = self.dt, int(pre_ln), act, E, H, Fi, eps, float((E // H) ** -0.5)
        for name, (w, b) in (("qkv", qkv), ("o", o), ("1", fc1), ("2", fc2)):
            setattr(t, "w_" + name, w._d2r_lp.data_ptr())
            setattr(t, "b_" + name, b.data_ptr())
            setattr(t, "gw_" + name, w._d2r_grad.data_ptr())
            setattr(t, "gb_" + name, b._d2r_grad.data_ptr())
        for name, (g, b) in (("ln1", ln1), ("ln2", ln2)):
            setattr(t, name + "_g", g.data_ptr())
            setattr(t, name + "_b", b.data_ptr())
            setattr(t, "g" + name + "_g", g._d2r_grad.data_ptr())
            setattr(t, "g" + name + "_b", b._d2r_grad.data_ptr())
        self.template = t
        self.key = (qkv[0]._d2r_lp.data_ptr(), qkv[0]._d2r_grad.data_ptr())

    def supports(self, x) -> bool:
        return (x.dtype == self.tdtype and x.is_cuda and x.dim() == 3 and x.shape[-1] == self.E
                and bool(_lib.load().d2r_mha_supported(self.dt, x.shape[1], x.shape[1], self.E // self.H)))


_SCRATCH = {}


def _layer_scratch(nbytes: int, device) -> torch.Tensor:
    key = (device.index, _stream())
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _SCRATCH[key] = ws
    return ws


class _EncoderLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, bundle, mask, p_attn, p_hidden):
        x = x.contiguous()
        B, L, E = x.shape
        T, Fi, H = B * L, bundle.F, bundle.H
        d = _lib.EncoderLayerDesc()
        C.memmove(C.byref(d), C.byref(bundle.template), C.sizeof(d))
        d.B, d.L = B, L
        d.mask = _ptr(mask)
        # training-time dropout: seeds drawn in the order the op-by-op path draws them (probabilities, attention output, FFN
        # output), so that both paths see the same masks under the same torch.manual_seed
        d.p_attn, d.p_hidden = p_attn, p_hidden
        if p_attn > 0.0:
            d.seed_attn = _next_dropout_seed()
        if p_hidden > 0.0:
            d.seed_hidden[0] = _next_dropout_seed()
            d.seed_hidden[1] = _next_dropout_seed()
        acts = torch.empty(T * (7 * E + 2 * Fi), dtype=x.dtype, device=x.device)  # qkv ctx h1 n1 h2 | f_pre f
        stats = torch.empty(B * H * L + 4 * T, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        a0, s0 = acts.data_ptr(), stats.data_ptr()
        d.x, d.y = x.data_ptr(), y.data_ptr()
        d.qkv, d.ctx, d.h1, d.n1, d.h2 = a0, a0 + 6 * T * E, a0 + 8 * T * E, a0 + 10 * T * E, a0 + 12 * T * E
        d.f_pre, d.f = a0 + 14 * T * E, a0 + 14 * T * E + 2 * T * Fi
        d.lse = s0
        s0 += 4 * B * H * L
        d.mean1, d.rstd1, d.mean2, d.rstd2 = s0, s0 + 4 * T, s0 + 8 * T, s0 + 12 * T
        _lib.call("d2r_encoder_layer_fwd", C.byref(d), _stream(), meta=dict(group="encoder_layer_fwd"))
        ctx.save_for_backward(x)
        ctx.d, ctx.keep, ctx.bundle = d, (acts, stats, mask), bundle
        return y

    @staticmethod
    def backward(ctx, g):
        _ensure_backward_join()
        if EARLY_FLUSH:
            _flush_before_encoders()
        (x,) = ctx.saved_tensors
        d, bundle = ctx.d, ctx.bundle
        g = g.contiguous()
        dx = torch.empty_like(g)
        need = _lib.load().d2r_encoder_layer_bwd_scratch(d.B, d.L, d.E, d.F)
        side = wgrad_side_stream_handle()
        defer = DEFER_WGRAD and side is None
        d.defer_wgrad = int(defer)
        d.defer_ln = int(defer and DEFER_LN)
        if defer:  # the four weight gradients are launched later, grouped with the other layers': scratch must survive
            scratch = torch.empty(need, dtype=torch.uint8, device=g.device)
            ws = _workspace(64 << 20, g.device)
            d.wgrad_stream = None
        elif side is None:
            scratch = _layer_scratch(need, g.device)
            ws = _workspace(64 << 20, g.device)
            d.wgrad_stream = None
        else:  # the weight-gradient GEMMs outlive this call on the side stream: fresh scratch, operands kept alive
            scratch = torch.empty(need, dtype=torch.uint8, device=g.device)
            for t in (scratch, g, x, ctx.keep[0]):
                t.record_stream(side)
            ws = _workspace(64 << 20, g.device, side.cuda_stream)
            d.wgrad_stream = side.cuda_stream
        d.dy, d.dx = g.data_ptr(), dx.data_ptr()
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
        d.splitk_ws, d.splitk_bytes = ws.data_ptr(), ws.numel()
        _lib.call("d2r_encoder_layer_bwd", C.byref(d), _stream(), meta=dict(group="encoder_layer_bwd"))
        if defer:
            T, E, Fi = d.B * d.L, d.E, d.F
            keep = (scratch, g, x, ctx.keep[0])
            P = bundle.params  # (w, b) of qkv, o, fc1, fc2, then the LayerNorm pairs
            attn_in = d.n1 if d.pre_ln else d.x
            ffn_in = d.h2 if d.pre_ln else d.n1
            for which, (N, K, xin, gw, gb) in enumerate(((3 * E, E, attn_in, d.gw_qkv, d.gb_qkv), (E, E, d.ctx, d.gw_o, d.gb_o),
                                                          (Fi, E, ffn_in, d.gw_1, d.gb_1), (E, Fi, d.f, d.gw_2, d.gb_2))):
                _defer_wgrad_raw(d.dtype, N, K, T, K, d.o_dy[which], xin, gw, gb, (P[2 * which], P[2 * which + 1]), keep,
                                 flush_at=1 << 30)
            if d.defer_ln:  # ... and the two LayerNorms' gamma / beta gradients: their partial sums sit in `scratch`
                for i, (gs, bs) in enumerate(((d.gln1_g, d.gln1_b), (d.gln2_g, d.gln2_b))):
                    _defer_ln_sum(T, E, d.o_lnws[i], gs, bs, (P[8 + 2 * i], P[9 + 2 * i]), keep, flush_at=1 << 30)
                ready = ()
            else:
                ready = bundle.params[8:]  # LayerNorm gradients were accumulated inside the call
            flush_wgrads_if(4 * D2R_LAYER_GROUP)  # whole layers: the four shapes of D2R_LAYER_GROUP layers leave as one launch
        else:
            ready = bundle.params
        ctx.keep = None
        for p in ready:  # data-parallel bucket readiness (d2r_amd.dp)
            cb = getattr(p, "_d2r_ready_cb", None)
            if cb is not None:
                cb(p)
        return dx, None, None, None, None, None


def encoder_layer(x, bundle: LayerBundle, mask=None, p_attn: float = 0.0, p_hidden: float = 0.0):
    """One whole BertLayer / CLIPEncoderLayer (16-bit compute dtype) as a single autograd node and a single C call each way;
    p_attn / p_hidden: training-time dropout on the attention probabilities / on the two dense outputs."""
    return _EncoderLayer.apply(x, bundle.params[0], bundle, mask, float(p_attn), float(p_hidden))


# ------------------------------------------------------------------------------------------------------
# K16 whole (Reversed_)InteractionModule (16-bit compute): one C call forward, one backward
# ------------------------------------------------------------------------------------------------------
class InteractionBundle:
    """Parameter / gradient-sink pointers of one interaction module (stable across steps: views of the flat buffers of
    d2r_amd.params.ParamStore), packed as the d2r_routing_layer_params table of include/d2r_hip.h.

    ``layers``: one dict per routing layer {RL name: (weight leaf, bias leaf)} plus 'bn': (weight, bias, running_mean,
    running_var); weight leaves carry ._d2r_grad (fp32 sink) and, for the compute-dtype linears, ._d2r_lp (16-bit shadow);
    the router linears (R0, R2) are multiplied as fp32 masters."""

    def __init__(self, layers, ncell, hid_router, heads_imrc, hid_imrc, kv_all=None):
        """kv_all: (weight leaf, bias leaf) of the module-wide fused k|v projection of `other` (every alignment cell of every
        layer, rows back to back in layer order: modules._InteractionBase._fusion_members)."""
        self.ncell, self.nlayer = ncell, len(layers)
        if kv_all is None or getattr(kv_all[0], "_d2r_lp", None) is None or getattr(kv_all[0], "_d2r_grad", None) is None:
            raise _lib.D2RError("InteractionBundle needs the fused k|v run of ParamStore (kv_all)")
        nkv = ((ncell > 1) + (ncell > 3) + (ncell > 4)) * len(layers)
        if tuple(kv_all[0].shape) != (nkv * 1536, 768):
            raise _lib.D2RError(f"kv_all has shape {tuple(kv_all[0].shape)}, expected ({nkv * 1536}, 768)")
        self.kv_all = kv_all
        self.tdtype = None  # the 16-bit compute dtype of the weight shadows (bf16 or fp16)
        self.hid_router, self.heads_imrc, self.hid_imrc = hid_router, heads_imrc, hid_imrc
        self.table = (_lib.RoutingLayerParams * self.nlayer)()
        self.params = []  # every leaf whose gradient sink the backward call writes (data-parallel readiness)
        key = []
        for l, spec in enumerate(layers):
            t = self.table[l]
            for name, leaves in spec.items():
                if name == "bn":
                    continue
                w, b = leaves
                for leaf in (w, b):
                    if getattr(leaf, "_d2r_grad", None) is None:
                        raise _lib.D2RError("InteractionBundle needs a model prepared by ParamStore")
                fp32 = name in ("R0", "R2")
                wc = w if fp32 else getattr(w, "_d2r_lp", None)
                if wc is None:
                    raise _lib.D2RError("InteractionBundle needs the 16-bit weight shadow of ParamStore")
                if not fp32:
                    self.tdtype = wc.dtype
                e = t.lin[_lib.RL[name]]
                e.w, e.b, e.gw, e.gb = wc.data_ptr(), b.data_ptr(), w._d2r_grad.data_ptr(), b._d2r_grad.data_ptr()
                if not name.endswith("_KV"):  # (views of the module-wide k|v run: kv_all reports for them)
                    self.params += [w, b]
                key += [e.w, e.gw]
            if "bn" in spec:
                bw, bb, rm, rv = spec["bn"]
                t.bn_weight, t.bn_bias, t.bn_running_mean, t.bn_running_var = bw.data_ptr(), bb.data_ptr(), rm.data_ptr(), rv.data_ptr()
                t.g_bn_weight, t.g_bn_bias = bw._d2r_grad.data_ptr(), bb._d2r_grad.data_ptr()
                self.params += [bw, bb]
                key += [t.bn_running_mean]
        self.params += [kv_all[0], kv_all[1]]
        key += [kv_all[0]._d2r_lp.data_ptr(), kv_all[0]._d2r_grad.data_ptr()]
        self.key = tuple(key)
        self.total_paths = ncell * ncell * (self.nlayer - 1) + ncell

    def supports(self, own, other) -> bool:
        return (own.dtype == self.tdtype and other.dtype == self.tdtype and own.is_cuda and own.dim() == 3 and own.shape[-1] == 768
                and other.shape[-1] == 768
                and bool(_lib.load().d2r_interaction_supported(_dt_of(self.tdtype), own.shape[1], other.shape[1], self.ncell, self.heads_imrc)))


class _Interaction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, own, other, anchor, bundle, train):
        own, other = own.contiguous(), other.contiguous()
        B, Lq, _ = own.shape
        Lk = other.shape[1]
        lib = _lib.load()
        d = _lib.InteractionDesc()
        d.dtype, d.B, d.Lq, d.Lk, d.ncell, d.nlayer = _dt(own), B, Lq, Lk, bundle.ncell, bundle.nlayer
        d.hid_router, d.heads_imrc, d.hid_imrc, d.train = bundle.hid_router, bundle.heads_imrc, bundle.hid_imrc, int(train)
        d.layers = bundle.table
        kw, kb = bundle.kv_all
        d.kv_all.w, d.kv_all.b, d.kv_all.gw, d.kv_all.gb = kw._d2r_lp.data_ptr(), kb.data_ptr(), kw._d2r_grad.data_ptr(), kb._d2r_grad.data_ptr()
        out = torch.empty_like(own)
        paths = torch.empty(B, bundle.total_paths, dtype=torch.float32, device=own.device)
        nbytes = lib.d2r_interaction_arena_bytes(B, Lq, Lk, bundle.ncell, bundle.nlayer, bundle.hid_router, bundle.hid_imrc)
        arena = torch.empty(nbytes, dtype=torch.uint8, device=own.device)
        ws = _workspace(64 << 20, own.device)
        d.own, d.other, d.out, d.paths = own.data_ptr(), other.data_ptr(), out.data_ptr(), paths.data_ptr()
        d.arena, d.arena_bytes, d.splitk_ws, d.splitk_bytes = arena.data_ptr(), arena.numel(), ws.data_ptr(), ws.numel()
        bnbuf = _bn_sync_fields(d, own.device)
        _lib.call("d2r_interaction_fwd", C.byref(d), _stream(), meta=dict(group="interaction_fwd"))
        ctx.save_for_backward(own, other, out)
        ctx.d, ctx.keep, ctx.bundle = d, (arena, bnbuf), bundle
        return out, paths

    @staticmethod
    def backward(ctx, d_out, d_paths):
        _ensure_backward_join()
        if DETERMINISTIC:
            # bit-reproducible steps with the two branch streams on: this module's backward first waits (on the GPU, no host
            # synchronisation) for what is queued on the other compute streams - the two whole-module backward passes then
            # no longer overlap (an open timing-dependent last-bits difference otherwise)
            cur = torch.cuda.current_stream()
            for st in _COMPUTE_STREAMS:
                if st != cur:
                    cur.wait_stream(st)
        own, other, out = ctx.saved_tensors
        d, bundle = ctx.d, ctx.bundle
        lib = _lib.load()
        d_own, d_other = torch.empty_like(own), torch.empty_like(other)
        if d_out is not None:
            d_out = d_out.contiguous()
        if d_paths is not None:
            d_paths = d_paths.contiguous()
        nbytes = lib.d2r_interaction_bwd_scratch(d.B, d.Lq, d.Lk, d.ncell, d.nlayer, d.hid_router, d.hid_imrc)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=own.device)
        ws = _workspace(64 << 20, own.device)
        d.splitk_ws, d.splitk_bytes = ws.data_ptr(), ws.numel()
        d.d_out, d.d_paths, d.d_own, d.d_other = _ptr(d_out), _ptr(d_paths), d_own.data_ptr(), d_other.data_ptr()
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
        _lib.call("d2r_interaction_bwd", C.byref(d), _stream(), meta=dict(group="interaction_bwd"))
        if ctx.keep[1] is not None:
            _BN_SYNC_BUFS.pop(id(ctx.keep[1]), None)
        ctx.keep = None
        for p in bundle.params:  # data-parallel bucket readiness (d2r_amd.dp): every sink of the module is written now
            cb = getattr(p, "_d2r_ready_cb", None)
            if cb is not None:
                cb(p)
        return d_own, d_other, None, None, None


class HeadBundle:
    """Parameter table of the classification head (Block fusion + fc) for the one-call path (d2r_head_fwd / d2r_head_bwd): fp32
    weights and their fp32 gradient sinks.  linears: (weight, bias) pairs in the order lin0, lin1, merge0, merge1, lin_out, fc
    (merge0 / merge1: the fused [chunks * rank * size, size] groups of Block's rank projections)."""

    def __init__(self, linears, mm: int, chunks: int, rank: int):
        self.params = []
        self.lp = []
        for w, b in linears:
            for t in (w, b):
                if t.dtype != torch.float32 or not t.is_cuda or getattr(t, "_d2r_grad", None) is None:
                    raise _lib.D2RError("HeadBundle: fp32 parameters prepared by ParamStore (gradient sinks) are required")
            self.lp.append(_lib.LinearParams(w.data_ptr(), b.data_ptr(), w._d2r_grad.data_ptr(), b._d2r_grad.data_ptr()))
            self.params += [w, b]
        (w0, _), (wm, _), (wo, _), (wf, _) = linears[0], linears[2], linears[4], linears[5]
        self.E, self.mm, self.chunks, self.rank, self.classes = w0.shape[1], int(mm), int(chunks), int(rank), wf.shape[0]
        size = self.mm // self.chunks
        if w0.shape != (self.mm, self.E) or wm.shape != (self.chunks * self.rank * size, size) or wo.shape != (self.E, self.mm):
            raise _lib.D2RError("HeadBundle: unexpected parameter shapes")
        self.key = (w0.data_ptr(), w0._d2r_grad.data_ptr())


class _Head(torch.autograd.Function):
    """(loss, logits, pooled) = head(text_pooled, vision_pooled, js_loss, labels): ONE C call each way.  Only `loss` carries a
    gradient (the end of the training graph); a gradient arriving at logits / pooled is refused."""

    @staticmethod
    def forward(ctx, x0, x1, js, labels, anchor, bundle):
        x0, x1, js = x0.contiguous(), x1.contiguous(), js.contiguous()
        labels = labels.contiguous().long()
        B = x0.shape[0]
        lib = _lib.load()
        d = _lib.HeadDesc()
        d.B, d.E, d.mm, d.chunks, d.rank, d.classes = B, bundle.E, bundle.mm, bundle.chunks, bundle.rank, bundle.classes
        d.lin0, d.lin1, d.merge0, d.merge1, d.lin_out, d.fc = bundle.lp
        loss = torch.empty((), dtype=torch.float32, device=x0.device)
        logits = torch.empty(B, bundle.classes, dtype=torch.float32, device=x0.device)
        pooled = torch.empty(B, bundle.E, dtype=torch.float32, device=x0.device)
        arena = torch.empty(lib.d2r_head_arena_bytes(B, bundle.E, bundle.mm, bundle.chunks, bundle.rank, bundle.classes), dtype=torch.uint8,
                            device=x0.device)
        ws = _workspace(64 << 20, x0.device)
        d.x0, d.x1, d.labels, d.js = x0.data_ptr(), x1.data_ptr(), labels.data_ptr(), js.data_ptr()
        d.loss, d.logits, d.pooled = loss.data_ptr(), logits.data_ptr(), pooled.data_ptr()
        d.arena, d.arena_bytes, d.splitk_ws, d.splitk_bytes = arena.data_ptr(), arena.numel(), ws.data_ptr(), ws.numel()
        _lib.call("d2r_head_fwd", C.byref(d), _stream(), meta=dict(group="head_fwd"))
        ctx.save_for_backward(x0, x1, labels, logits, pooled)
        ctx.d, ctx.keep, ctx.bundle = d, arena, bundle
        ctx.set_materialize_grads(False)
        return loss, logits, pooled

    @staticmethod
    def backward(ctx, g_loss, g_logits, g_pooled):
        _ensure_backward_join()
        x0, x1, labels, logits, pooled = ctx.saved_tensors
        d, bundle = ctx.d, ctx.bundle
        d_x0, d_x1 = torch.empty_like(x0), torch.empty_like(x1)
        d_js = torch.empty((), dtype=torch.float32, device=x0.device)
        if g_loss is None and g_logits is None and g_pooled is None:
            return d_x0.zero_(), d_x1.zero_(), d_js.zero_(), None, None, None
        # gradients arriving at the logits / Block's output (a second loss on them) are added to the cross-entropy path inside the call
        g_loss = torch.zeros((), dtype=torch.float32, device=x0.device) if g_loss is None else g_loss.contiguous().float()
        g_logits = None if g_logits is None else g_logits.contiguous().float()
        g_pooled = None if g_pooled is None else g_pooled.contiguous().float()
        d.d_logits, d.d_pooled = _ptr(g_logits), _ptr(g_pooled)
        lib = _lib.load()
        scratch = torch.empty(lib.d2r_head_bwd_scratch(d.B, d.E, d.mm, d.chunks, d.rank, d.classes), dtype=torch.uint8, device=x0.device)
        ws = _workspace(64 << 20, x0.device)
        d.splitk_ws, d.splitk_bytes = ws.data_ptr(), ws.numel()
        d.d_loss, d.d_x0, d.d_x1, d.d_js = g_loss.data_ptr(), d_x0.data_ptr(), d_x1.data_ptr(), d_js.data_ptr()
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
        _lib.call("d2r_head_bwd", C.byref(d), _stream(), meta=dict(group="head_bwd"))
        ctx.keep = None
        for p in bundle.params:  # data-parallel bucket readiness: every sink of the head is written now
            cb = getattr(p, "_d2r_ready_cb", None)
            if cb is not None:
                cb(p)
        return d_x0, d_x1, d_js, None, None, None


def head(x0, x1, js, labels, bundle: HeadBundle):
    """Block fusion + fc + cross entropy + (ce + js) as a single autograd node: -> (loss, logits [B, classes], pooled [B, 768])."""
    return _Head.apply(x0, x1, js, labels, bundle.params[0], bundle)


def interaction(own, other, bundle: InteractionBundle, train: bool):
    """One whole (Reversed_)InteractionModule as a single autograd node: -> (emb [B,Lq,768], paths fp32 [B,total_paths])."""
    return _Interaction.apply(own, other, bundle.params[0], bundle, bool(train))


# ------------------------------------------------------------------------------------------------------
# row kernels
# ------------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _lib.call("d2r_layernorm_fwd", _dt(x), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, rows, D,
                  y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        g = g.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        dx = torch.empty_like(x)
        dg = torch.empty(D, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        nbytes = _lib.load().d2r_layernorm_bwd_workspace(rows, D)
        ws = _workspace(nbytes, x.device)
        _lib.call("d2r_layernorm_bwd", _dt(x), g.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), rows, D, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                  _stream())
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        norm = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.call("d2r_l2norm_fwd", _dt(x), x.data_ptr(), y.data_ptr(), norm.data_ptr(), rows, D, _stream())
        ctx.save_for_backward(x, norm)
        return y

    @staticmethod
    def backward(ctx, g):
        x, norm = ctx.saved_tensors
        g = g.contiguous()
        D = x.shape[-1]
        dx = torch.empty_like(x)
        _lib.call("d2r_l2norm_bwd", _dt(x), g.data_ptr(), x.data_ptr(), norm.data_ptr(), dx.data_ptr(), x.numel() // D,
                  D, _stream())
        return dx


def l2norm(x):
    return _L2Norm.apply(x)


class _SoftmaxRows(torch.autograd.Function):
    """softmax over the last dim of a 2-D tensor (GESC feature gate, models/Cells.py:204)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        rows, cols = x.shape
        y = torch.empty_like(x)
        _lib.call("d2r_softmax_fwd", _dt(x), _dt(x), x.data_ptr(), y.data_ptr(), cols, rows, cols, 1.0, None, 1,
                  _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        rows, cols = y.shape
        dx = torch.empty_like(y)
        _lib.call("d2r_softmax_bwd", _dt(y), _dt(y), y.data_ptr(), g.data_ptr(), dx.data_ptr(), cols, rows, cols, 1.0,
                  _stream())
        return dx


def softmax_rows(x):
    return _SoftmaxRows.apply(x)


# ------------------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------------------
class _SqDiff(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _lib.call("d2r_sqdiff_fwd", _dt(a), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        _lib.call("d2r_sqdiff_bwd", _dt(a), a.data_ptr(), b.data_ptr(), g.data_ptr(), da.data_ptr(), db.data_ptr(),
                  a.numel(), _stream())
        return da, db


def sqdiff(a, b):
    return _SqDiff.apply(a, b)


class _MulAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, s, h):
        a, s, h = a.contiguous(), s.contiguous(), h.contiguous()
        out = torch.empty_like(a)
        _lib.call("d2r_muladd_fwd", _dt(a), a.data_ptr(), s.data_ptr(), h.data_ptr(), out.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(a, s)
        return out

    @staticmethod
    def backward(ctx, g):
        a, s = ctx.saved_tensors
        g = g.contiguous()
        da, ds = torch.empty_like(a), torch.empty_like(s)
        _lib.call("d2r_muladd_bwd", _dt(a), a.data_ptr(), s.data_ptr(), g.data_ptr(), da.data_ptr(), ds.data_ptr(),
                  a.numel(), _stream())
        return da, ds, g


def muladd(a, s, h):
    """a * s + h (FiLM modulation, models/Refinement.py:136)."""
    return _MulAdd.apply(a, s, h)


class _Lerp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_, a, b):
        g_, a, b = g_.contiguous(), a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _lib.call("d2r_lerp_fwd", _dt(a), g_.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(g_, a, b)
        return out

    @staticmethod
    def backward(ctx, d):
        g_, a, b = ctx.saved_tensors
        d = d.contiguous()
        dg, da, db = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        _lib.call("d2r_lerp_bwd", _dt(a), g_.data_ptr(), a.data_ptr(), b.data_ptr(), d.data_ptr(), dg.data_ptr(),
                  da.data_ptr(), db.data_ptr(), a.numel(), _stream())
        return dg, da, db


def lerp_gate(g, a, b):
    """g*a + (1-g)*b (models/Cells.py:205)."""
    return _Lerp.apply(g, a, b)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _lib.call("d2r_add", _dt(a), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


# nn.Dropout of the BERT path.  The mask is a pure function of (seed, element index): each call draws a fresh 63-bit
# seed from torch's CPU generator (so torch.manual_seed makes training runs reproducible; no device sync), and the
# backward pass regenerates the mask from it.
def _next_dropout_seed() -> int:
    return int(torch.empty((), dtype=torch.int64).random_().item())


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, p, seed):
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == x.shape and residual.dtype == x.dtype
        y = torch.empty_like(x)
        _lib.call("d2r_dropout", _dt(x), x.data_ptr(), _ptr(residual), y.data_ptr(), x.numel(), p, seed, _stream())
        ctx.cfg = (p, seed, residual is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed, has_res = ctx.cfg
        g = g.contiguous()
        dx = torch.empty_like(g)
        _lib.call("d2r_dropout", _dt(g), g.data_ptr(), None, dx.data_ptr(), g.numel(), p, seed, _stream())
        return dx, (g if has_res else None), None, None


def dropout(x, p: float, training: bool, residual=None):
    """nn.Dropout(p)(x) (+ residual).  Identity (plus the add) when not training or p == 0."""
    if not training or p <= 0.0:
        return x if residual is None else add(x, residual)
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout probability {p} outside [0, 1)")
    return _Dropout.apply(x, residual, float(p), _next_dropout_seed())


class _LinComb(torch.autograd.Function):
    """out = sum_k c_k x_k for fp32 device scalars (total loss)."""

    @staticmethod
    def forward(ctx, coefs, *xs):
        out = torch.empty((), dtype=torch.float32, device=xs[0].device)
        carr = (C.c_float * len(xs))(*coefs)
        _lib.call("d2r_lincomb", _parr(xs), carr, len(xs), out.data_ptr(), _stream())
        ctx.coefs = coefs
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs = []
        for c in ctx.coefs:
            d = torch.empty_like(g)
            _lib.call("d2r_axpby", F32, float(c), g.data_ptr(), 0.0, d.data_ptr(), 1, _stream())
            outs.append(d)
        return (None, *outs)


def lincomb(coefs, xs):
    return _LinComb.apply(tuple(float(c) for c in coefs), *xs)


# ------------------------------------------------------------------------------------------------------
# K1 router pooling and K8 route aggregation
# ------------------------------------------------------------------------------------------------------
class _MeanPool(torch.autograd.Function):
    """mean over tokens of n sources [B,L,D] -> fp32 [n,B,D] in one launch (models/Router.py:23)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [x.contiguous() for x in xs]
        B, L, D = xs[0].shape
        pooled = torch.empty(len(xs), B, D, dtype=torch.float32, device=xs[0].device)
        _lib.call("d2r_meanpool_fwd", _dt(xs[0]), _parr(xs), len(xs), B, L, D, pooled.data_ptr(), _stream(),
                  meta=dict(group="router_pool_fwd", bytes=float(len(xs) * B * L * D * xs[0].element_size() + len(xs) * B * D * 4)))
        ctx.meta = (B, L, D, xs[0].dtype, len(xs))
        return pooled

    @staticmethod
    def backward(ctx, g):
        B, L, D, dtype, n = ctx.meta
        g = g.contiguous()
        outs = []
        for i in range(n):
            dx = torch.empty(B, L, D, dtype=dtype, device=g.device)
            _lib.call("d2r_meanpool_bwd", _dt_of(dtype), g[i].data_ptr(), B, L, D,
                      dx.data_ptr(), 0, _stream())
            outs.append(dx)
        return tuple(outs)


def mean_pool(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    return _MeanPool.apply(*xs)


class _RouteAggregate(torch.autograd.Function):
    """K8: outs_i = sum_j p_hat[i,j] emb_j + gate_i relu(x0)  (P = ncell), or the final-layer rule (P = 1).
    ncell cells = the first ncell of [RIC, GLAC, IMRC, CMRC, CRCMC, GESC] (6 in the reference)."""

    @staticmethod
    def forward(ctx, gates, nc, *tensors):
        embs = [t.contiguous() for t in tensors[:nc]]
        refs = [r.contiguous() for r in tensors[nc:]]
        gates = gates.contiguous()
        B, L, D = embs[0].shape
        P = gates.shape[2]
        assert gates.shape[1] == nc and P in (nc, 1), "gates: fp32 [B, ncell, P]"
        final = P == 1
        if final:
            assert len(refs) == nc - 1, "the final layer takes ref_1..ref_{ncell-1} (ref_0 is x0)"
            all_refs = [embs[0]] + refs
        outs = [torch.empty(B, L, D, dtype=embs[0].dtype, device=embs[0].device) for _ in range(P)]
        probs = torch.empty(B, P, nc, dtype=torch.float32, device=gates.device)
        nfull = nc - 1 - (1 if nc > 5 else 0)  # cells with a [B,L,D] output (1 and 5 are per-sample broadcasts)
        _lib.call("d2r_route_aggregate_fwd", _dt(embs[0]), _parr(embs), _parr(all_refs) if final else None,
                  gates.data_ptr(), B, L, D, nc, P, _parr(outs), probs.data_ptr(), P * nc, _stream(),
                  meta=dict(group=f"route_aggregate_fwd_P{P}",
                            bytes=float(((nfull + P) * B * L * D + (nc - nfull) * B * D) * embs[0].element_size() + 4 * B * P * 2 * nc)))
        ctx.meta = (B, L, D, P, nc)
        ctx.save_for_backward(gates, *embs, *refs, *(outs if final else []))
        ctx.mark_non_differentiable()
        return (probs, *outs)

    @staticmethod
    def backward(ctx, dprobs, *douts):
        B, L, D, P, nc = ctx.meta
        final = P == 1
        saved = ctx.saved_tensors
        gates, embs = saved[0], list(saved[1:1 + nc])
        refs = list(saved[1 + nc:2 * nc]) if final else []
        out_saved = [saved[2 * nc]] if final else []
        dev, dtype = embs[0].device, embs[0].dtype
        douts = [(d if d is not None else torch.zeros(B, L, D, dtype=dtype, device=dev)).contiguous() for d in douts]
        dprobs = None if dprobs is None else dprobs.contiguous()
        dembs = [torch.empty_like(e) for e in embs]
        drefs = [torch.empty(B, L, D, dtype=dtype, device=dev) for _ in range(nc)] if final else None
        dgates = torch.empty_like(gates)
        nbytes = _lib.load().d2r_route_aggregate_bwd_workspace(B, L, D, P)
        ws = _workspace(nbytes, dev)
        nfull = nc - 1 - (1 if nc > 5 else 0)
        _lib.call("d2r_route_aggregate_bwd", _dt(embs[0]), _parr(embs), _parr([embs[0]] + refs) if final else None,
                  gates.data_ptr(), _parr(douts), _parr(out_saved) if final else None, _ptr(dprobs), P * nc, B, L, D, nc, P,
                  _parr(dembs), _parr(drefs) if final else None, dgates.data_ptr(), ws.data_ptr(), ws.numel(),
                  _stream(), meta=dict(group=f"route_aggregate_bwd_P{P}",
                                       bytes=float(((P + 2 * nfull + (nc if final else 0) + (1 if final else 0)) * B * L * D
                                                    + 2 * (nc - nfull) * B * D) * embs[0].element_size())))
        if final:
            # ref_0 is x0 itself: relu path + skip path
            dx0 = add(dembs[0], drefs[0])
            return (dgates, None, dx0, *dembs[1:], *drefs[1:])
        return (dgates, None, *dembs)


def route_aggregate(gates, *embs, refs: Optional[Sequence[torch.Tensor]] = None):
    """gates fp32 [B,ncell,P]; embs: the ncell cell outputs (embs[0] = the RIC input); returns (probs [B,P,ncell], [outs])."""
    res = _RouteAggregate.apply(gates, len(embs), *embs, *(refs or ()))
    return res[0], list(res[1:])


# ------------------------------------------------------------------------------------------------------
# K6 SAF gate, weighted row sum
# ------------------------------------------------------------------------------------------------------
class _SafGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, bn_w, bn_b, running_mean, running_var, train):
        a = a.contiguous()
        B, n = a.shape
        w = torch.empty_like(a)
        saved = torch.empty(2, dtype=torch.float32, device=a.device)
        ctx.exact = train and DP_EXACT is not None
        if ctx.exact:  # statistics over the samples of every rank
            sums = torch.empty(2, dtype=torch.float64, device=a.device)
            _lib.call("d2r_saf_gate_stats", a.data_ptr(), B, n, sums.data_ptr(), _stream())
            _allreduce_pair(sums)
            _lib.call("d2r_saf_gate_fwd_ex", a.data_ptr(), B, n, bn_w.data_ptr(), bn_b.data_ptr(), running_mean.data_ptr(),
                      running_var.data_ptr(), 1, w.data_ptr(), saved.data_ptr(), sums.data_ptr(), float(DP_EXACT[1] * B * n), None, 0, _stream())
        else:
            _lib.call("d2r_saf_gate_fwd", a.data_ptr(), B, n, bn_w.data_ptr(), bn_b.data_ptr(), running_mean.data_ptr(),
                      running_var.data_ptr(), 1 if train else 0, w.data_ptr(), saved.data_ptr(), _stream())
        ctx.save_for_backward(a, bn_w, bn_b, saved)
        ctx.train = train
        return w

    @staticmethod
    def backward(ctx, dw):
        a, bn_w, bn_b, saved = ctx.saved_tensors
        dw = dw.contiguous()
        B, n = a.shape
        da = torch.empty_like(a)
        dbw, dbb = torch.empty_like(bn_w), torch.empty_like(bn_b)
        if ctx.exact:
            gs = torch.empty(2, dtype=torch.float64, device=a.device)
            nt = float(DP_EXACT[1] * B * n)
            _lib.call("d2r_saf_gate_bwd_ex", a.data_ptr(), dw.data_ptr(), B, n, bn_w.data_ptr(), bn_b.data_ptr(), saved.data_ptr(), 1,
                      da.data_ptr(), dbw.data_ptr(), dbb.data_ptr(), 1, gs.data_ptr(), nt, None, 0, 0, _stream())
            _allreduce_pair(gs)
            _lib.call("d2r_saf_gate_bwd_ex", a.data_ptr(), None, B, n, bn_w.data_ptr(), bn_b.data_ptr(), saved.data_ptr(), 1,
                      da.data_ptr(), None, None, 2, gs.data_ptr(), nt, None, 0, 0, _stream())
            return da, dbw, dbb, None, None, None
        _lib.call("d2r_saf_gate_bwd", a.data_ptr(), dw.data_ptr(), B, n, bn_w.data_ptr(), bn_b.data_ptr(),
                  saved.data_ptr(), 1 if ctx.train else 0, da.data_ptr(), dbw.data_ptr(), dbb.data_ptr(), _stream())
        return da, dbw, dbb, None, None, None


def saf_gate(a, bn_w, bn_b, running_mean, running_var, train: bool):
    """l1norm(sigmoid(BatchNorm1d(1)(a))) over the last dim of fp32 a [B,n] (models/XModules.py:380-381)."""
    return _SafGate.apply(a, bn_w, bn_b, running_mean, running_var, bool(train))


class _WeightedRowSum(torch.autograd.Function):
    """out[b] = w[b] @ S[b]  (w [B,n] in S's dtype, S [B,n,D]) — torch.matmul(sim_attn, sim_emb), XModules.py:382."""

    @staticmethod
    def forward(ctx, w, S):
        w, S = w.contiguous(), S.contiguous()
        B, n, D = S.shape
        out = torch.empty(B, D, dtype=S.dtype, device=S.device)
        gemm(GEMM_NN, 1, D, n, w.data_ptr(), n, S.data_ptr(), D, out.data_ptr(), D, dtype=_dt(S), c_dtype=_dt(S), nb=B,
             sA=(n, 0), sB=(n * D, 0), sC=(D, 0))
        ctx.save_for_backward(w, S)
        return out

    @staticmethod
    def backward(ctx, g):
        w, S = ctx.saved_tensors
        g = g.contiguous()
        B, n, D = S.shape
        dw = torch.empty_like(w)
        gemm(GEMM_NT, 1, n, D, g.data_ptr(), D, S.data_ptr(), D, dw.data_ptr(), n, dtype=_dt(S), c_dtype=_dt(S), nb=B,
             sA=(D, 0), sB=(n * D, 0), sC=(n, 0))
        dS = torch.empty_like(S)
        gemm(GEMM_TN, n, D, 1, w.data_ptr(), n, g.data_ptr(), D, dS.data_ptr(), D, dtype=_dt(S), c_dtype=_dt(S), nb=B,
             sA=(n, 0), sB=(D, 0), sC=(n * D, 0))
        return dw, dS


def weighted_row_sum(w, S):
    return _WeightedRowSum.apply(w, S)


# ------------------------------------------------------------------------------------------------------
# losses and Block fusion core
# ------------------------------------------------------------------------------------------------------
class _JsDiv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, q):
        p, q = p.contiguous(), q.contiguous()
        out = torch.empty((), dtype=torch.float32, device=p.device)
        _lib.call("d2r_jsdiv_fwd", p.data_ptr(), q.data_ptr(), p.shape[0], out.data_ptr(), _stream())
        ctx.save_for_backward(p, q)
        return out

    @staticmethod
    def backward(ctx, g):
        p, q = ctx.saved_tensors
        g = g.contiguous()
        dp, dq = torch.empty_like(p), torch.empty_like(q)
        _lib.call("d2r_jsdiv_bwd", p.data_ptr(), q.data_ptr(), p.shape[0], g.data_ptr(), dp.data_ptr(), dq.data_ptr(),
                  _stream())
        return dp, dq


def js_div(p_logits, q_logits):
    assert p_logits.dtype == torch.float32 and q_logits.dtype == torch.float32
    return _JsDiv.apply(p_logits, q_logits)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous()
        labels = labels.contiguous().long()
        B, Cn = logits.shape
        out = torch.empty((), dtype=torch.float32, device=logits.device)
        _lib.call("d2r_ce_fwd", logits.data_ptr(), labels.data_ptr(), B, Cn, out.data_ptr(), _stream())
        ctx.save_for_backward(logits, labels)
        return out

    @staticmethod
    def backward(ctx, g):
        _ensure_backward_join()
        logits, labels = ctx.saved_tensors
        g = g.contiguous()
        B, Cn = logits.shape
        d = torch.empty_like(logits)
        _lib.call("d2r_ce_bwd", logits.data_ptr(), labels.data_ptr(), B, Cn, g.data_ptr(), d.data_ptr(), _stream())
        return d, None


def cross_entropy(logits, labels):
    assert logits.dtype == torch.float32
    return _CrossEntropy.apply(logits, labels)


class _BlockMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m0, m1, Cn, R, S):
        m0, m1 = m0.contiguous(), m1.contiguous()
        B = m0.shape[0]
        out = torch.empty(B, Cn * S, dtype=m0.dtype, device=m0.device)
        zraw = torch.empty(B, Cn * S, dtype=torch.float32, device=m0.device)
        _lib.call("d2r_block_merge_fwd", _dt(m0), m0.data_ptr(), m1.data_ptr(), B, Cn, R, S, out.data_ptr(),
                  zraw.data_ptr(), _stream())
        ctx.save_for_backward(m0, m1, zraw)
        ctx.meta = (B, Cn, R, S)
        return out

    @staticmethod
    def backward(ctx, g):
        m0, m1, zraw = ctx.saved_tensors
        B, Cn, R, S = ctx.meta
        g = g.contiguous()
        d0, d1 = torch.empty_like(m0), torch.empty_like(m1)
        _lib.call("d2r_block_merge_bwd", _dt(m0), m0.data_ptr(), m1.data_ptr(), zraw.data_ptr(), g.data_ptr(), B, Cn, R,
                  S, d0.data_ptr(), d1.data_ptr(), _stream())
        return d0, d1, None, None, None


def block_merge(m0, m1, chunks, rank, size):
    """m0, m1: [B, chunks, rank*size] -> [B, chunks*size] (models/XModules.py:541-549)."""
    return _BlockMerge.apply(m0, m1, chunks, rank, size)


# ------------------------------------------------------------------------------------------------------
# K12 embeddings
# ------------------------------------------------------------------------------------------------------
class _BertEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, tt, word, pos, typ, dtype):
        ids, tt = ids.contiguous().long(), tt.contiguous().long()
        B, L = ids.shape
        D = word.shape[1]
        out = torch.empty(B, L, D, dtype=dtype, device=word.device)
        _lib.call("d2r_bert_embed_fwd", _dt(out), ids.data_ptr(), tt.data_ptr(), word.data_ptr(), pos.data_ptr(),
                  typ.data_ptr(), B, L, D, word.shape[0], typ.shape[0], out.data_ptr(), _stream())
        ctx.save_for_backward(ids, tt)
        ctx.tables = (word, pos, typ)
        return out

    @staticmethod
    def backward(ctx, g):
        ids, tt = ctx.saved_tensors
        g = g.contiguous()
        B, L, D = g.shape
        tables = ctx.tables
        sinks = [getattr(t, "_d2r_grad", None) for t in tables]
        if all(s is not None for s in sinks):  # accumulate straight into the flat gradient buffer (no 94 MB temp)
            outs, ret = sinks, (None, None, None)
        else:
            outs = [torch.zeros(t.shape, dtype=torch.float32, device=g.device) for t in tables]
            ret = tuple(outs)
        _lib.call("d2r_bert_embed_bwd", _dt(g), g.data_ptr(), ids.data_ptr(), tt.data_ptr(), B, L, D, tables[2].shape[0],
                  0, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), _stream())
        if ret[0] is None:
            for t in tables:
                cb = getattr(t, "_d2r_ready_cb", None)
                if cb is not None:
                    cb(t)
        return (None, None) + ret + (None,)


def bert_embed(ids, tt, word, pos, typ, dtype):
    return _BertEmbed.apply(ids, tt, word, pos, typ, dtype)


class _ClipEmbed(torch.autograd.Function):
    """CLIPVisionEmbeddings (models/modeling_unimo.py:108-118): patch conv as a GEMM on unfolded patches."""

    @staticmethod
    def forward(ctx, pixels, w_master, w_compute, cls, pos, patch):
        pixels = pixels.contiguous().float()
        B, _, Hh, Ww = pixels.shape
        E = w_compute.shape[0]
        K = 3 * patch * patch
        npatch = (Hh // patch) * (Ww // patch)
        ntok = npatch + 1
        dtype = w_compute.dtype
        patches = torch.empty(B * npatch, K, dtype=dtype, device=pixels.device)
        _lib.call("d2r_patchify", _dt(patches), pixels.data_ptr(), B, Hh, Ww, patch, patches.data_ptr(), _stream())
        x = torch.empty(B, ntok, E, dtype=dtype, device=pixels.device)
        w2 = w_compute.view(E, K)
        es = x.element_size()
        gemm(GEMM_NT, npatch, E, K, patches.data_ptr(), K, w2.data_ptr(), K, x.data_ptr() + E * es, E, dtype=_dt(x),
             c_dtype=_dt(x), nb=B, sA=(npatch * K, 0), sC=(ntok * E, 0))
        _lib.call("d2r_clip_embed_finish", _dt(x), x.data_ptr(), cls.data_ptr(), pos.data_ptr(), B, ntok, E, _stream())
        ctx.save_for_backward(patches)
        ctx.meta = (B, npatch, ntok, E, K, tuple(w_master.shape))
        return x

    @staticmethod
    def backward(ctx, g):
        (patches,) = ctx.saved_tensors
        B, npatch, ntok, E, K, wshape = ctx.meta
        g = g.contiguous()
        dcls = torch.empty(E, dtype=torch.float32, device=g.device)
        dpos = torch.empty(ntok, E, dtype=torch.float32, device=g.device)
        _lib.call("d2r_clip_embed_bwd", _dt(g), g.data_ptr(), B, ntok, E, dcls.data_ptr(), dpos.data_ptr(), _stream())
        # dW[E,K] = dY_patch^T patches over all B*npatch rows: the class-token rows are dropped from dY by one strided copy,
        # then ONE split-K weight-gradient GEMM (was: one accumulating GEMM per sample)
        dw = torch.empty(E, K, dtype=torch.float32, device=g.device)
        gp = g[:, 1:, :].contiguous()
        gemm(GEMM_TN, E, K, B * npatch, gp.data_ptr(), E, patches.data_ptr(), K, dw.data_ptr(), K, dtype=_dt(g), c_dtype=F32,
             beta=0.0, splitk_ws=_workspace(64 << 20, g.device))
        return None, dw.view(wshape), None, dcls, dpos, None


def clip_embed(pixels, w_master, w_compute, cls, pos, patch):
    return _ClipEmbed.apply(pixels, w_master, w_compute, cls, pos, patch)
