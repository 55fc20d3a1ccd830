"""GPU parity of the 256 x 256 deep-pipelined GEMM (d2r_amd/csrc/gemm8.hip) through the C ABI: forward (NT) / dX (NN) products with
every epilogue operand, and the grouped weight gradients of DIFFERENT shapes (d2r_gemm_tn_grouped_v), against fp64 expressions of
the same products on the same 16-bit operands.  Shapes cover the edge cases of the pipeline: one K-tile, two, an odd count, a ragged
reduction length (6304 token rows = 98.5 tiles), ragged rows / columns, and the benchmark's own shapes.  Every case is launched
repeatedly and must be bit-identical to itself (the LDS hazards of the pipeline are ordered by barrier counts; a race would show as a
repeat that differs)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

LOWP = [torch.bfloat16, torch.float16]
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}  # half an ulp at 1.0 (one rounding of the stored result)


def _force_wide(on):
    from d2r_amd import _lib
    _lib.load().d2r_gemm_tuning(1, 1, 11 if on else -1)


@pytest.fixture()
def wide(gpu):
    _force_wide(True)
    yield
    _force_wide(False)


def _gemm(layout, a, b, c, *, bias=None, act=0, residual=None, beta=0.0, preact=None):
    from d2r_amd import functional as F
    from d2r_amd._lib import BF16, F16
    dt = BF16 if a.dtype == torch.bfloat16 else F16
    M, K = a.shape
    N = b.shape[0] if layout == 0 else b.shape[1]
    F.gemm(layout, M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), dtype=dt, c_dtype=dt,
           bias=None if bias is None else bias.data_ptr(), act=act, residual=None if residual is None else residual.data_ptr(),
           ldr=0 if residual is None else residual.stride(0), beta=beta, preact=None if preact is None else preact.data_ptr())


FWD_CASES = [
    # M, N, K, options
    (256, 256, 128, {}),                                   # one tile, two K-tiles
    (256, 256, 64 * 3, dict(bias=True)),                    # odd K-tile count
    (300, 520, 64 * 5, dict(bias=True, act=3, pad=8)),      # ragged rows and columns, padded output rows, gelu + saved pre-activation
    (1024, 768, 768, dict(bias=True, act=1, res=True)),     # relu + residual
    (4096, 3072, 768, dict(bias=True)),                     # benchmark shape (FFN up-projection of the text encoder)
    (6304, 768, 3072, dict(beta=1.0)),                      # ragged rows (24.6 tiles), deep reduction, accumulate into C
]


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("layout", [0, 1], ids=["NT", "NN"])
@pytest.mark.parametrize("case", FWD_CASES, ids=lambda c: "x".join(map(str, c[:3])))
def test_wide_forward_and_dx_products(case, layout, lowp, gpu, wide):
    M, N, K, kw = case
    g = torch.Generator(device=gpu).manual_seed(M + N + K + layout)
    a = (torch.randn(M, K, device=gpu, generator=g) * 0.5).to(lowp)
    b = (torch.randn((N, K) if layout == 0 else (K, N), device=gpu, generator=g) * 0.5).to(lowp)
    ldc = N + kw.get("pad", 0)
    c = torch.randn(M, ldc, device=gpu, generator=g).to(lowp)
    c0 = c.clone()
    bias = torch.randn(N, device=gpu, generator=g) if kw.get("bias") else None
    res = torch.randn(M, N, device=gpu, generator=g).to(lowp) if kw.get("res") else None
    act, beta = kw.get("act", 0), kw.get("beta", 0.0)
    pre = torch.zeros(M, ldc, device=gpu, dtype=lowp) if act == 3 else None
    _gemm(layout, a, b, c, bias=bias, act=act, residual=res, beta=beta, preact=pre)
    torch.cuda.synchronize()
    ref = a.double() @ (b.double().t() if layout == 0 else b.double())
    if bias is not None:
        ref = ref + bias.double()
    pre_ref = ref
    if act == 1:
        ref = torch.relu(ref)
    elif act == 3:
        ref = torch.nn.functional.gelu(ref.to(lowp).double())  # the kernel rounds the pre-activation to 16 bits before the activation
    if res is not None:
        ref = ref + res.double()
    if beta:
        ref = ref + beta * c0[:, :N].double()
    scale = float(ref.abs().max())
    err = float((c[:, :N].double() - ref).abs().max())
    # one rounding of the 16-bit result (+ one of the rounded pre-activation) on top of fp32 accumulation of exact products
    assert err <= 2.5 * ULP[lowp] * scale + 1e-6, f"max err {err:.3e} at scale {scale:.3e}"
    assert torch.equal(c[:, N:], c0[:, N:]), "the padding columns of the output rows were written"
    if pre is not None:
        assert float((pre[:, :N].double() - pre_ref).abs().max()) <= 1.5 * ULP[lowp] * float(pre_ref.abs().max()) + 1e-6
    if not beta:
        first = c.clone()
        for _ in range(4):
            c.copy_(c0)
            _gemm(layout, a, b, c, bias=bias, act=act, residual=res, beta=beta, preact=pre)
            torch.cuda.synchronize()
            assert torch.equal(c, first), "repeat differs: a race in the staging pipeline"


def _arr(t, vals):
    arr = (t * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def _grouped_v(shapes, dys, xs, sinks, bs, beta, lowp):
    from d2r_amd import _lib
    from d2r_amd import functional as F
    dt = _lib.BF16 if lowp == torch.bfloat16 else _lib.F16
    n = len(shapes)
    _lib.call("d2r_gemm_tn_grouped_v", dt, n, _arr(C.c_int, [s[0] for s in shapes]), _arr(C.c_int, [s[1] for s in shapes]),
              _arr(C.c_int, [s[2] for s in shapes]), _arr(C.c_int64, [t.stride(0) for t in dys]), _arr(C.c_int64, [t.stride(0) for t in xs]),
              _arr(C.c_int64, [t.stride(0) for t in sinks]), _arr(C.c_void_p, [t.data_ptr() for t in dys]), _arr(C.c_void_p, [t.data_ptr() for t in xs]),
              _arr(C.c_void_p, [t.data_ptr() for t in sinks]), None if bs is None else _arr(C.c_void_p, [t.data_ptr() for t in bs]), beta, F._stream())


# (out features, in features, token rows): 768-class cell linears, encoder shapes, a ragged reduction (6304 = 98.5 K-tiles), one and three
# K-tiles, ragged rows / columns, a per-sample product (32 rows: takes the rank-K kernel inside the same call), a narrow one (N = 64)
TN_SHAPES = [(768, 768, 4096), (3072, 768, 6304), (768, 3072, 4096), (2304, 768, 6304), (768, 768, 6304), (1536, 768, 200), (264, 520, 4096),
             (768, 1536, 128), (768, 768, 32), (768, 64, 4096), (256, 256, 192)]


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("with_bias", [True, False], ids=["dbias", "nobias"])
def test_grouped_weight_gradients_of_different_shapes(with_bias, beta, lowp, gpu):
    g = torch.Generator(device=gpu).manual_seed(7)
    dys = [(torch.randn(T, Nf, device=gpu, generator=g) * 0.5).to(lowp) for Nf, Kf, T in TN_SHAPES]
    xs = [(torch.randn(T, Kf, device=gpu, generator=g) * 0.5).to(lowp) for Nf, Kf, T in TN_SHAPES]
    sinks = [torch.randn(Nf, Kf, device=gpu, generator=g) for Nf, Kf, T in TN_SHAPES]
    bs = [torch.randn(Nf, device=gpu, generator=g) for Nf, Kf, T in TN_SHAPES] if with_bias else None
    s0 = [s.clone() for s in sinks]
    b0 = [b.clone() for b in bs] if with_bias else None
    _grouped_v(TN_SHAPES, dys, xs, sinks, bs, beta, lowp)
    torch.cuda.synchronize()
    for i, (Nf, Kf, T) in enumerate(TN_SHAPES):
        ref = dys[i].double().t() @ xs[i].double() + beta * s0[i].double()
        err = float((sinks[i].double() - ref).abs().max())
        # fp32 accumulation of exact 16-bit products over T rows
        assert err <= 3e-7 * (T ** 0.5) * float(ref.abs().max()) + 1e-6, f"problem {i} {Nf}x{Kf} T={T}: {err:.3e}"
        if with_bias:
            refb = dys[i].double().sum(0) + b0[i].double()
            assert float((bs[i].double() - refb).abs().max()) <= 3e-7 * (T ** 0.5) * float(refb.abs().max()) + 1e-6, f"bias gradient of problem {i}"
    if beta == 0.0 and not with_bias:
        first = [s.clone() for s in sinks]
        for _ in range(4):
            _grouped_v(TN_SHAPES, dys, xs, sinks, bs, beta, lowp)
            torch.cuda.synchronize()
            assert all(torch.equal(a_, b_) for a_, b_ in zip(sinks, first)), "repeat differs: a race in the staging pipeline"


def test_grouped_v_rejects_shared_outputs(gpu):
    from d2r_amd._lib import D2RError
    dy = torch.zeros(256, 256, device=gpu, dtype=torch.bfloat16)
    x = torch.zeros(256, 256, device=gpu, dtype=torch.bfloat16)
    sink = torch.zeros(256, 256, device=gpu)
    with pytest.raises(D2RError):
        _grouped_v([(256, 256, 256)] * 2, [dy, dy], [x, x], [sink, sink], None, 1.0, torch.bfloat16)


def test_same_shape_groups_take_the_wide_kernel_and_match_the_128_wide_one(gpu):
    """d2r_gemm_tn_grouped (one shape per call): the 256-wide kernel against the 128-wide LDS-DMA kernel on the same operands."""
    from d2r_amd import _lib
    from d2r_amd import functional as F
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(3)
    n, T, Nf, Kf = 13, 6304, 768, 768
    dys = [(torch.randn(T, Nf, device=gpu, generator=g) * 0.5).to(torch.float16) for _ in range(n)]
    xs = [(torch.randn(T, Kf, device=gpu, generator=g) * 0.5).to(torch.float16) for _ in range(n)]
    out = {}
    for code in (102, 103):
        lib.d2r_gemm_tuning(1, 1, code)
        sinks = [torch.zeros(Nf, Kf, device=gpu) for _ in range(n)]
        bs = [torch.zeros(Nf, device=gpu) for _ in range(n)]
        _lib.call("d2r_gemm_tn_grouped", _lib.F16, Nf, Kf, T, Nf, Kf, Kf, _arr(C.c_void_p, [t.data_ptr() for t in dys]),
                  _arr(C.c_void_p, [t.data_ptr() for t in xs]), _arr(C.c_void_p, [t.data_ptr() for t in sinks]),
                  _arr(C.c_void_p, [t.data_ptr() for t in bs]), n, 1.0, F._stream())
        torch.cuda.synchronize()
        out[code] = (sinks, bs)
    lib.d2r_gemm_tuning(1, 1, 103)
    for i in range(n):
        ref = dys[i].double().t() @ xs[i].double()
        for code in (102, 103):
            assert float((out[code][0][i].double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
            assert float((out[code][1][i].double() - dys[i].double().sum(0)).abs().max()) <= 3e-5 * float(dys[i].double().sum(0).abs().max())


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("layout", [0, 1], ids=["NT", "NN"])
def test_grouped_launch_of_independent_products_is_bit_identical_to_separate_launches(layout, lowp, gpu):
    """d2r_gemm_group (gemm_glds.hip: independent forward / dX products of the routing cells in one launch of the 128 x 128 LDS-DMA
    kernel; gemm.hip: up to four per-sample products of at most 32 rows in one launch of the skinny kernel) against one d2r_gemm per
    problem: every tile is computed exactly as in a launch of its own."""
    from d2r_amd import _lib
    from d2r_amd import functional as F
    from d2r_amd._lib import GemmDesc
    dt = _lib.BF16 if lowp == torch.bfloat16 else _lib.F16
    g = torch.Generator(device=gpu).manual_seed(5 + layout)
    # (M, N, K, bias, act, residual, beta, grad_ref act): the shapes of one routing layer (q|k|v, 768-wide linears), ragged rows,
    # a small one that stays out of the group (M < 128), a deep reduction
    specs = [(4096, 2304, 768, True, 0, False, 0.0, 0), (4096, 768, 768, True, 2, False, 0.0, 0), (4096, 768, 768, True, 1, True, 0.0, 0),
             (6304, 768, 768, False, 0, False, 1.0, 0), (6304, 768, 768, True, 0, True, 1.0, 0), (300, 136, 128, True, 3, False, 0.0, 0),
             (64, 768, 768, True, 0, False, 0.0, 0), (4096, 768, 3072, False, 0, False, 0.0, 1),
             # per-sample products (at most 32 rows: the skinny kernel): five of them = one group of four + one launch of its own;
             # tanh epilogue, an accumulating tail, the activation gradient of a reference, a 17-row one (two row tiles, ragged)
             (32, 768, 768, True, 2, False, 0.0, 0), (32, 768, 768, True, 0, False, 1.0, 0), (32, 768, 768, False, 0, False, 0.0, 2),
             (17, 768, 1536, True, 0, True, 0.0, 0), (32, 1536, 768, True, 2, False, 0.0, 0)]
    ops = []
    for M, N, K, hb, act, hr, beta, gact in specs:
        a = (torch.randn(M, K, device=gpu, generator=g) * 0.5).to(lowp)
        b = (torch.randn((N, K) if layout == 0 else (K, N), device=gpu, generator=g) * 0.5).to(lowp)
        c0 = torch.randn(M, N, device=gpu, generator=g).to(lowp)
        bias = torch.randn(N, device=gpu, generator=g) if hb else None
        res = torch.randn(M, N, device=gpu, generator=g).to(lowp) if hr else None
        gref = torch.randn(M, N, device=gpu, generator=g).to(lowp) if gact else None
        ops.append((a, b, c0, bias, res, gref, act, beta, gact))

    def descs(outs):
        arr = (GemmDesc * len(ops))()
        for d, (a, b, c0, bias, res, gref, act, beta, gact), c in zip(arr, ops, outs):
            M, K = a.shape
            N = c.shape[1]
            d.dtype, d.c_dtype, d.layout, d.act, d.M, d.N, d.K, d.nb, d.nh, d.alpha, d.beta = dt, dt, layout, act, M, N, K, 1, 1, 1.0, beta
            d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), K, b.data_ptr(), b.shape[1], c.data_ptr(), N
            d.bias = None if bias is None else bias.data_ptr()
            d.residual, d.ldr = (None, 0) if res is None else (res.data_ptr(), N)
            d.grad_ref, d.grad_act = (None, 0) if gref is None else (gref.data_ptr(), gact)
        return arr

    one = [o[2].clone() for o in ops]
    arr = descs(one)
    for i in range(len(ops)):
        _lib.call("d2r_gemm", C.byref(arr[i]), F._stream())
    grp = [o[2].clone() for o in ops]
    _lib.call("d2r_gemm_group", descs(grp), len(ops), F._stream())
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(one, grp)):
        assert torch.equal(x, y), f"problem {i} {specs[i]}: grouped launch differs, max {float((x.float() - y.float()).abs().max()):.3e}"
    # and both are right
    a, b, c0, bias, res, gref, act, beta, gact = ops[1]
    ref = torch.tanh(a.double() @ (b.double().t() if layout == 0 else b.double()) + bias.double())
    assert float((grp[1].double() - ref).abs().max()) <= 2.5 * ULP[lowp] * float(ref.abs().max()) + 1e-6


# ---- in-launch split-K of the 128-wide kernel (gemm_glds_splitk_kernel) --------------------------------------------------------------
SPLITK_CASES = [
    # M, N, K, options: N = 768-class products with a deep reduction (two, three or four workgroups per output tile; K >= 6144 splits)
    (4096, 768, 6144, dict(bias=True, res=True)),           # 192 tiles, 96 K-tiles
    (6304, 768, 13824, dict(beta=1.0)),                     # d_other of the text-branch routing module: 300 tiles, ragged rows, accumulated
    (4096, 768, 13824, dict()),                             # ... of the image-branch module: 216 K-tiles
    (300, 520, 64 * 101, dict(bias=True, act=3, pad=8)),    # ragged rows and columns, an odd K-tile count, gelu + saved pre-activation
    (4096, 768, 3072, dict(bias=True)),                     # below the threshold: the workspace must not change the launch
]


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("layout", [0, 1], ids=["NT", "NN"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=lambda c: "x".join(map(str, c[:3])))
def test_in_launch_split_k_products(case, layout, lowp, gpu):
    """Deep reductions over a partial round of 128 x 128 tiles run with 2-4 workgroups per tile whose partial sums meet inside the
    launch (d2r_gemm_desc.workspace given): same bound as the unsplit kernel against the fp64 product, bit-identical to itself over
    repeats on ONE workspace (the tile counters are back at zero after every launch), the workspace's slab region may hold anything,
    and without a workspace (or with the family switched off) the unsplit kernel answers within a rounding of the 16-bit result."""
    from d2r_amd import _lib
    from d2r_amd import functional as F
    M, N, K, kw = case
    g = torch.Generator(device=gpu).manual_seed(M + N + K + layout)
    a = (torch.randn(M, K, device=gpu, generator=g) * 0.5).to(lowp)
    b = (torch.randn((N, K) if layout == 0 else (K, N), device=gpu, generator=g) * 0.5).to(lowp)
    ldc = N + kw.get("pad", 0)
    c0 = torch.randn(M, ldc, device=gpu, generator=g).to(lowp)
    bias = torch.randn(N, device=gpu, generator=g) if kw.get("bias") else None
    res = torch.randn(M, N, device=gpu, generator=g).to(lowp) if kw.get("res") else None
    act, beta = kw.get("act", 0), kw.get("beta", 0.0)
    dt = _lib.BF16 if lowp == torch.bfloat16 else _lib.F16
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=gpu)
    ws.fill_(0xA5)  # garbage in the slab region ...
    ws[-4096:].zero_()  # ... the counters start at zero (the contract of d2r_gemm_desc.workspace)

    def run(workspace):
        c = c0.clone()
        pre = torch.zeros(M, ldc, device=gpu, dtype=lowp) if act == 3 else None
        F.gemm(layout, M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), dtype=dt, c_dtype=dt,
               bias=None if bias is None else bias.data_ptr(), act=act, residual=None if res is None else res.data_ptr(),
               ldr=0 if res is None else res.stride(0), beta=beta, preact=None if pre is None else pre.data_ptr(), splitk_ws=workspace)
        torch.cuda.synchronize()
        return c, pre

    c, pre = run(ws)
    assert int(ws[-4096:].view(torch.int32).abs().max()) == 0, "a tile counter was left behind"
    ref = a.double() @ (b.double().t() if layout == 0 else b.double())
    if bias is not None:
        ref = ref + bias.double()
    pre_ref = ref
    if act == 3:
        ref = torch.nn.functional.gelu(ref.to(lowp).double())
    if res is not None:
        ref = ref + res.double()
    if beta:
        ref = ref + beta * c0[:, :N].double()
    scale = float(ref.abs().max())
    err = float((c[:, :N].double() - ref).abs().max())
    assert err <= 2.5 * ULP[lowp] * scale + 1e-6, f"max err {err:.3e} at scale {scale:.3e}"
    assert torch.equal(c[:, N:], c0[:, N:]), "the padding columns of the output rows were written"
    if pre is not None:
        assert float((pre[:, :N].double() - pre_ref).abs().max()) <= 1.5 * ULP[lowp] * float(pre_ref.abs().max()) + 1e-6
    for _ in range(4):
        c2, _ = run(ws)
        assert torch.equal(c2, c), "repeat differs: the partial sums met in a different order, or a stale partial was read"
    plain, _ = run(None)
    # (two roundings between the fp32 sum and the stored value - the LDS-staged 16-bit result, then the residual / accumulate add - so
    #  two neighbouring values: 2 x 2^-10 x scale for fp16)
    assert float((plain[:, :N].double() - c[:, :N].double()).abs().max()) <= 4.0 * ULP[lowp] * scale + 1e-6
    if K < 6144:
        assert torch.equal(plain, c)
    else:  # (a split sums K in a different order: the fp32 sums differ in their last bits, some 16-bit results with them)
        assert not torch.equal(plain, c), "the product was not split: the test does not exercise the meeting"
    _lib.load().d2r_gemm_tuning(1, 1, 130)
    try:
        off, _ = run(ws)
    finally:
        _lib.load().d2r_gemm_tuning(1, 1, 131)
    assert torch.equal(off, plain), "with the family switched off a workspace must not change the launch"
