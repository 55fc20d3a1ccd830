"""GPU parity of every HIP op (forward AND backward, through the C ABI) against a plain PyTorch CPU fp64
expression of the same op.  fp32 mode is checked tightly; the 16-bit modes (bf16, fp16) within their rounding of the fp64 truth."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]
LOWP = (torch.bfloat16, torch.float16)


def tol(dtype, scale=1.0):
    # one rounding of a 16-bit type: 2^-9 (bf16) / 2^-12 (fp16) relative; the ops chain a handful of them
    return {torch.float32: 2e-5, torch.bfloat16: 2.5e-2, torch.float16: 3.2e-3}[dtype] * scale


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (scale * torch.randn(tuple(shape), generator=g)).to(dtype)


def check(name, got, ref, dtype, scale=None, loosen=1.0):
    got = got.detach().double().cpu()
    ref = ref.detach().double()
    s = max(float(ref.abs().max()), 1e-6) if scale is None else scale
    err = float((got - ref).abs().max())
    assert err <= loosen * tol(dtype) * s + 1e-7, f"{name}: max err {err:.3e} vs scale {s:.3e} ({dtype})"


def run_both(fn_gpu, fn_ref, inputs, dtype, gpu, wrt=None, name="op", out_scale=None, grad_loosen=1.0):
    """inputs: list of CPU tensors (fp32 master values).  Runs fwd+bwd on the GPU op and the fp64 reference."""
    wrt = range(len(inputs)) if wrt is None else wrt
    xs_g = []
    for i, x in enumerate(inputs):
        t = x.to(dtype) if x.is_floating_point() and getattr(x, "_keep32", False) is False else x
        t = t.to(gpu)
        if i in wrt:
            t.requires_grad_(True)
        xs_g.append(t)
    xs_r = []
    for i, x, xg in zip(range(len(inputs)), inputs, xs_g):
        t = xg.detach().double().cpu() if x.is_floating_point() else x.clone()
        if i in wrt:
            t.requires_grad_(True)
        xs_r.append(t)
    out_g = fn_gpu(*xs_g)
    out_r = fn_ref(*xs_r)
    if not isinstance(out_g, (tuple, list)):
        out_g, out_r = [out_g], [out_r]
    loss_g, loss_r = 0, 0
    for k, (og, orr) in enumerate(zip(out_g, out_r)):
        check(f"{name}.out{k}", og, orr, og.dtype if og.dtype in LOWP else dtype, out_scale)
        w = rnd(*orr.shape, seed=100 + k).double()
        loss_g = loss_g + (og.float() * w.float().to(gpu)).sum()
        loss_r = loss_r + (orr * w).sum()
    loss_g.backward()
    loss_r.backward()
    for i in wrt:
        assert xs_g[i].grad is not None, f"{name}: no grad for input {i}"
        check(f"{name}.grad{i}", xs_g[i].grad, xs_r[i].grad, dtype, loosen=grad_loosen)


def keep32(t):
    t._keep32 = True
    return t


def wt(*shape, dtype, seed=0, scale=1.0):
    """fp32 master weight whose values are exactly representable in `dtype` (so the bf16 shadow is exact)."""
    return keep32(rnd(*shape, seed=seed, scale=scale).to(dtype).float())


def lp(w, dtype):
    from d2r_amd import functional as F
    return F.cast(w.detach(), dtype)


ACTS = {"none": (0, lambda x: x), "relu": (1, torch.relu), "tanh": (2, torch.tanh),
        "gelu": (3, torch.nn.functional.gelu), "quick_gelu": (4, lambda x: x * torch.sigmoid(1.702 * x)),
        "tanh_relu": (5, lambda x: torch.relu(torch.tanh(x)))}


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act", list(ACTS))
def test_linear(gpu, dtype, act):
    from d2r_amd import functional as F
    code, ref = ACTS[act]
    x, w, b = rnd(3, 37, 96, dtype=torch.float32), wt(72, 96, dtype=dtype, scale=0.2), keep32(rnd(72, scale=0.5))

    def f(x, w, b):
        return F.linear(x, w, b, lp(w, dtype), act=code)

    run_both(f, lambda x, w, b: ref(x @ w.t() + b), [x, w, b], dtype, gpu, name=f"linear[{act}]")


@pytest.mark.parametrize("dtype", DT)
def test_linear_residual_and_strided(gpu, dtype):
    from d2r_amd import functional as F
    x, w, b, r = rnd(4, 9, 768), wt(768, 768, dtype=dtype, scale=0.05), keep32(rnd(768)), rnd(4, 9, 768)
    run_both(lambda x, w, b, r: F.linear(x, w, b, lp(w, dtype), residual=r), lambda x, w, b, r: x @ w.t() + b + r,
             [x, w, b, r], dtype, gpu, name="linear+res")
    # strided rows: x[:, 0] (BertPooler) without a copy
    run_both(lambda x, w, b: F.linear(x[:, 0], w, b, lp(w, dtype), act=2), lambda x, w, b: torch.tanh(x[:, 0] @ w.t() + b),
             [x, w, b], dtype, gpu, name="linear cls rows")


@pytest.mark.parametrize("dtype", DT)
def test_linear_fp32_out(gpu, dtype):
    from d2r_amd import functional as F
    x, w, b = rnd(5, 11, 768), wt(1, 768, dtype=dtype, scale=0.1), keep32(rnd(1))
    run_both(lambda x, w, b: F.linear(x, w, b, lp(w, dtype), out_dtype=torch.float32), lambda x, w, b: x @ w.t() + b,
             [x, w, b], dtype, gpu, name="linear N=1 fp32 out")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(2, 8, 8, 4, 0.3, False, False), (3, 16, 5, 1, 100 / math.sqrt(768), False, True),
                                 (2, 197, 197, 12, 0.125, True, False), (2, 40, 40, 16, 1 / math.sqrt(48), False, True),
                                 (2, 128, 128, 12, 0.125, True, False), (2, 70, 250, 12, 0.125, True, True),
                                 (2, 197, 197, 16, 1 / math.sqrt(48), False, True),
                                 (2, 128, 197, 1, 100 / math.sqrt(768), False, False), (2, 197, 128, 1, 100 / math.sqrt(768), False, False),
                                 (2, 197, 197, 1, 1.0, False, True), (2, 40, 250, 1, 0.3, True, False), (2, 33, 256, 1, 0.3, True, True),
                                 # sequences above 256 tokens (BASELINE configs[3]: 577 image tokens; configs[4]: 512 text tokens): the block loop
                                 (2, 577, 577, 12, 0.125, False, True), (1, 512, 512, 12, 0.125, True, False), (2, 300, 577, 16, 1 / math.sqrt(48), True, True),
                                 (1, 577, 130, 12, 0.125, True, False), (1, 577, 256, 1, 0.3, True, True), (1, 256, 577, 1, 0.3, True, False)])
def test_attention(gpu, dtype, cfg):
    from d2r_amd import functional as F
    B, Lq, Lk, H, scale, use_mask, use_res = cfg
    E = 768
    # single-head cases contract over all 768 features: keep the logits O(1) so that the comparison measures the
    # kernel and not the conditioning of a near-one-hot softmax
    qk = 0.15 if (H == 1 and Lq >= 33) else 0.5
    q, k, v = rnd(B, Lq, E, scale=qk), rnd(B, Lk, E, scale=qk, seed=1), rnd(B, Lk, E, seed=2)
    mask = torch.zeros(B, Lk)
    if use_mask:
        mask[0, Lk // 2:] = -10000.0
        if B > 1:
            mask[1, Lk - 3:] = -10000.0
    res = rnd(B, Lq, E, seed=3)

    def f(q, k, v, res):
        return F.attention(q, k, v, H, scale, mask=mask.to(gpu) if use_mask else None, residual=res if use_res else None)

    def r(q, k, v, res):
        d = E // H
        qh, kh, vh = (t.view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
        s = scale * qh @ kh.transpose(-1, -2)
        if use_mask:
            s = s + mask.double()[:, None, None, :]
        o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)
        return o + res if use_res else o + 0 * res

    run_both(f, r, [q, k, v, res], dtype, gpu, wrt=[0, 1, 2] + ([3] if use_res else []), name=f"attention{cfg}")


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 128, 128, 12), (2, 197, 197, 12), (2, 50, 37, 16), (1, 300, 577, 12)], ids=lambda c: "x".join(map(str, c)))
def test_fused_attention_dropout_matches_the_three_launch_path(gpu, lowp, cfg, monkeypatch):
    """Attention-probability dropout (models/modeling_unimo.py:388) inside the fused multi-head core - short sequences and the
    block loop above 256 tokens - against the unfused path (GEMM, softmax, d2r_dropout on the [B,H,Lq,Lkp] probabilities, GEMM)
    with the SAME seed: one mask, so outputs and all three gradients agree to 16-bit rounding; and the mask has the
    requested density."""
    from d2r_amd import functional as F
    B, Lq, Lk, H = cfg
    E, p = 768, 0.1
    scale = 1.0 / math.sqrt(E // H)
    q, k, v = (rnd(B, L_, E, seed=i, scale=0.5).to(lowp).to(gpu) for i, L_ in enumerate((Lq, Lk, Lk)))
    w = rnd(B, Lq, E, seed=9).to(gpu)
    mask = torch.zeros(B, Lk, device=gpu)
    mask[0, Lk - 5:] = -10000.0
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(F, "FUSED_MHA", fused)
        monkeypatch.setattr(F, "_next_dropout_seed", lambda: 424242)
        qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = F.attention(qg, kg, vg, H, scale, mask=mask, p_drop=p)
        assert (type(o.grad_fn).__name__ == "_AttentionBackward")
        (o.float() * w).sum().backward()
        torch.cuda.synchronize()
        out[fused] = (o.detach().float(), qg.grad.float(), kg.grad.float(), vg.grad.float())
    tol_ = {torch.bfloat16: 2e-2, torch.float16: 3e-3}[lowp]
    for name, a, b in zip(("o", "dq", "dk", "dv"), out[True], out[False]):
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        assert err <= tol_ * sc, f"{name}: fused vs three-launch with one seed differ by {err:.3e} (scale {sc:.3e})"
    # density of the mask: with v = 1 the output row is the sum of the kept, rescaled probabilities -> mean 1
    monkeypatch.setattr(F, "FUSED_MHA", True)
    ones = torch.ones_like(v)
    with torch.no_grad():
        o1 = F.attention(q, k, ones, H, scale, p_drop=p).float()
    assert abs(float(o1.mean()) - 1.0) < 2e-2 and float(o1.std()) > 1e-3, (float(o1.mean()), float(o1.std()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(2, 128, 12, True), (2, 197, 12, False), (2, 37, 16, False)])
def test_attention_packed_qkv_and_kv(gpu, dtype, cfg):
    """The packed layouts the fused projections produce: qkv [B,L,3E] (self-attention) and q + kv [B,Lk,2E]; gradients
    land in one packed tensor."""
    from d2r_amd import functional as F
    B, L, H, use_mask = cfg
    E, d = 768, 768 // H
    scale = 1 / math.sqrt(d)
    qkv = rnd(B, L, 3 * E, scale=0.5)
    mask = torch.zeros(B, L)
    if use_mask:
        mask[0, L // 3:] = -10000.0

    def ref_core(q, k, v):
        qh, kh, vh = (t.reshape(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
        s = scale * qh @ kh.transpose(-1, -2)
        if use_mask:
            s = s + mask.double()[:, None, None, :]
        return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, -1, E)

    run_both(lambda x: F.attention_qkv(x, H, scale, mask=mask.to(gpu) if use_mask else None),
             lambda x: ref_core(x[..., :E], x[..., E:2 * E], x[..., 2 * E:]), [qkv], dtype, gpu, name=f"attention_qkv{cfg}")
    q, kv = rnd(B, L + 5, E, scale=0.5, seed=4), rnd(B, L, 2 * E, scale=0.5, seed=5)
    run_both(lambda q, kv: F.attention_kv(q, kv, H, scale, mask=mask.to(gpu) if use_mask else None),
             lambda q, kv: ref_core(q, kv[..., :E], kv[..., E:]), [q, kv], dtype, gpu, name=f"attention_kv{cfg}")


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_l2norm_softmax(gpu, dtype):
    from d2r_amd import functional as F
    x, g, b = rnd(3, 7, 768, scale=2.0), keep32(1 + 0.1 * rnd(768)), keep32(0.1 * rnd(768, seed=1))
    run_both(lambda x, g, b: F.layer_norm(x, g, b, 1e-12), lambda x, g, b: torch.nn.functional.layer_norm(x, (768,), g, b, 1e-12),
             [x, g, b], dtype, gpu, name="layernorm")
    run_both(lambda x: F.l2norm(x), lambda x: x / (x.pow(2).sum(-1, keepdim=True).sqrt() + 1e-8), [x], dtype, gpu, name="l2norm")
    z = rnd(5, 768, scale=3.0)
    run_both(lambda z: F.softmax_rows(z), lambda z: torch.softmax(z, -1), [z], dtype, gpu, name="softmax_rows",
             out_scale=1.0 if dtype == torch.float32 else 0.05)


@pytest.mark.parametrize("dtype", DT)
def test_elementwise(gpu, dtype):
    from d2r_amd import functional as F
    a, b, c = rnd(2, 5, 768), rnd(2, 5, 768, seed=1), rnd(2, 5, 768, seed=2)
    run_both(F.sqdiff, lambda a, b: (a - b) ** 2, [a, b], dtype, gpu, name="sqdiff")
    run_both(F.muladd, lambda a, s, h: a * s + h, [a, b, c], dtype, gpu, name="muladd")
    g = torch.rand(2, 768)
    run_both(F.lerp_gate, lambda g, a, b: g * a + (1 - g) * b, [g, rnd(2, 768), rnd(2, 768, seed=5)], dtype, gpu, name="lerp")
    run_both(F.add, lambda a, b: a + b, [a, b], dtype, gpu, name="add")


@pytest.mark.parametrize("dtype", DT)
def test_meanpool_and_router(gpu, dtype):
    from d2r_amd import functional as F
    xs = [rnd(3, 13, 768, seed=i) for i in range(6)]
    run_both(lambda *x: F.mean_pool(list(x)), lambda *x: torch.stack([t.mean(1) for t in x]), xs, dtype, gpu, name="meanpool6")
    run_both(lambda x: F.mean_pool([x])[0], lambda x: x.mean(1), [xs[0]], dtype, gpu, name="meanpool1")


def _agg_ref(nc, gates, *tensors):
    embs_in, refs = list(tensors[:nc]), list(tensors[nc:])
    x0 = embs_in[0]
    B, L, D = x0.shape
    P = gates.shape[2]
    embs = [torch.relu(x0)] + [e[:, None].expand(B, L, D) if j in (1, 5) else e for j, e in enumerate(embs_in) if j > 0]
    G = gates.transpose(1, 2)  # [B,P,nc]
    thr, thr_f = float(torch.tensor(1e-4, dtype=torch.float32)), float(torch.tensor(1e-4 / nc, dtype=torch.float32))
    if P == 1:
        rr = [x0] + refs
        skip = (G < thr_f).double()
        num = sum(G[:, 0, j, None, None] * embs[j] + skip[:, 0, j, None, None] * rr[j] for j in range(nc))
        den = (skip.sum(-1) + G.sum(-1))[:, :, None]
        return (G, num / den)
    skip = (G.sum(-1) < thr).double()
    probs = G / (G.sum(-1, keepdim=True) + float(torch.tensor(1e-8, dtype=torch.float32)))
    outs = [sum(probs[:, i, j, None, None] * embs[j] for j in range(nc)) + skip[:, i, None, None] * embs[0] for i in range(P)]
    return (probs, *outs)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nc,final,regime", [(6, False, "open"), (6, False, "mixed"), (6, False, "closed"), (6, True, "open"),
                                             (6, True, "mixed"), (6, True, "closed"), (4, False, "mixed"), (4, True, "mixed"),
                                             (4, True, "closed"), (5, False, "open")])
def test_route_aggregate(gpu, dtype, nc, final, regime):
    """K8 for the reference's six cells and for declared subsets (first nc cells; BASELINE configs[4] uses 4): path
    normalisation / threshold gates over the existing cells, final-layer threshold 1e-4/nc."""
    from d2r_amd import functional as F
    B, L, D = 4, 19, 768
    P = 1 if final else nc
    g = torch.Generator().manual_seed(5)
    gates = torch.rand(B, nc, P, generator=g)
    if regime == "mixed":
        gates = gates * (torch.rand(B, nc, P, generator=g) > 0.5)
        gates[0] = 0.0  # one sample with every path closed -> skip connection
    elif regime == "closed":
        gates.zero_()
    gates = keep32(gates.float())
    embs = [rnd(B, D, seed=7 + j) if j in (1, 5) else rnd(B, L, D, seed=j) for j in range(nc)]
    refs = [rnd(B, L, D, seed=10 + i) for i in range(nc - 1)] if final else []
    inputs = [gates] + embs + refs

    def f(gates, *ts):
        probs, outs = F.route_aggregate(gates, *ts[:nc], refs=list(ts[nc:]) if final else None)
        return (probs, *outs)

    if final and regime == "closed":
        wrt = list(range(1, len(inputs)))  # d/dgate at g=0 with every path closed is well defined but huge
    else:
        wrt = None
    run_both(f, lambda gates, *ts: _agg_ref(nc, gates, *ts), inputs, dtype, gpu, wrt=wrt, name=f"aggregate nc={nc} final={final} {regime}")


@pytest.mark.parametrize("train", [True, False])
def test_saf_gate(gpu, train):
    from d2r_amd import functional as F
    B, n = 5, 33
    a, bw, bb = rnd(B, n, scale=2.0), torch.tensor([1.3]), torch.tensor([-0.2])
    rm, rv = torch.tensor([0.1]), torch.tensor([1.5])
    rm_g, rv_g = rm.clone().to(gpu), rv.clone().to(gpu)

    def ref(a, bw, bb):
        if train:
            mu, var = a.mean(), a.var(unbiased=False)
        else:
            mu, var = rm.double()[0], rv.double()[0]
        s = torch.sigmoid((a - mu) / torch.sqrt(var + 1e-5) * bw + bb)
        return s / (s.abs().sum(-1, keepdim=True) + 1e-8)

    run_both(lambda a, bw, bb: F.saf_gate(a, bw, bb, rm_g, rv_g, train), ref, [a, bw, bb], torch.float32, gpu, name="saf_gate")
    if train:
        N = B * n
        assert abs(float(rm_g[0]) - (0.9 * 0.1 + 0.1 * float(a.mean()))) < 1e-5
        assert abs(float(rv_g[0]) - (0.9 * 1.5 + 0.1 * float(a.var(unbiased=False)) * N / (N - 1))) < 1e-5


@pytest.mark.parametrize("dtype", DT)
def test_weighted_row_sum(gpu, dtype):
    from d2r_amd import functional as F
    w, S = torch.rand(3, 21), rnd(3, 21, 768)
    run_both(F.weighted_row_sum, lambda w, S: torch.bmm(w[:, None], S)[:, 0], [w, S], dtype, gpu, name="weighted_row_sum")


def test_losses(gpu):
    from d2r_amd import functional as F
    p, q = rnd(7, 7, scale=3.0), rnd(7, 7, scale=2.0, seed=3)

    def js(p, q):
        pp, qq = torch.softmax(p, -1), torch.softmax(q, -1)
        lm = ((pp + qq) / 2).log()
        kl = torch.nn.KLDivLoss(reduction="batchmean")
        return (kl(lm, pp) + kl(lm, qq)) / 2

    run_both(F.js_div, js, [p, q], torch.float32, gpu, name="js_div")
    logits, labels = rnd(9, 3, scale=2.0), torch.tensor([0, 2, 1, 1, 0, 2, 2, 0, 1])
    run_both(lambda l: F.cross_entropy(l, labels.to(gpu)), lambda l: torch.nn.functional.cross_entropy(l, labels),
             [logits], torch.float32, gpu, name="cross_entropy")
    a, b = rnd(), rnd(seed=4)
    run_both(lambda a, b: F.lincomb([1.0, -0.3], [a, b]), lambda a, b: a - 0.3 * b, [a, b], torch.float32, gpu, name="lincomb")
    x = rnd(6, 768, scale=0.2)
    run_both(lambda x: F.matmul_nt(x, x), lambda x: x @ x.t(), [x], torch.float32, gpu, name="matmul_nt")


@pytest.mark.parametrize("dtype", DT)
def test_block_merge(gpu, dtype):
    from d2r_amd import functional as F
    B, Cn, R, S = 3, 20, 15, 80
    m0, m1 = rnd(B, Cn, R * S, scale=0.5), rnd(B, Cn, R * S, scale=0.5, seed=1)

    def ref(m0, m1):
        z = (m0 * m1).view(B, Cn, R, S).sum(2)
        z = torch.sqrt(torch.relu(z)) - torch.sqrt(torch.relu(-z))
        return torch.nn.functional.normalize(z, p=2, dim=-1).reshape(B, Cn * S)

    # d/dz sqrt|z| is unbounded near z = 0: the largest gradients carry the fp32 rounding of 1/sqrt|z|
    run_both(lambda a, b: F.block_merge(a, b, Cn, R, S), ref, [m0, m1], dtype, gpu, name="block_merge", grad_loosen=5.0)


@pytest.mark.parametrize("dtype", DT)
def test_embeddings(gpu, dtype):
    from d2r_amd import functional as F
    B, L, D = 5, 37, 768  # 185 tokens over 50 ids: duplicates within and across the 64-token scan chunks
    ids = torch.randint(0, 50, (B, L))
    ids[0, 5:] = 0
    ids[:, 0] = 7
    tt = torch.randint(0, 2, (B, L))
    word, pos, typ = keep32(rnd(50, D)), keep32(rnd(40, D, seed=1)), keep32(rnd(2, D, seed=2))

    def ref(word, pos, typ):
        return torch.nn.functional.embedding(ids, word, padding_idx=0) + typ[tt] + pos[:L][None]

    run_both(lambda w, p, t: F.bert_embed(ids.to(gpu), tt.to(gpu), w, p, t, dtype), ref, [word, pos, typ], dtype, gpu,
             name="bert_embed")
    px = keep32(rnd(2, 3, 64, 64))
    w, cls, pe = wt(D, 3, 32, 32, dtype=dtype, scale=0.05), keep32(rnd(D, seed=3)), keep32(rnd(5, D, seed=4))

    def refc(px, w, cls, pe):
        pt = torch.nn.functional.conv2d(px, w, stride=32).flatten(2).transpose(1, 2)
        return torch.cat([cls.expand(2, 1, -1), pt], 1) + pe[None]

    run_both(lambda px, w, cls, pe: F.clip_embed(px, w, lp(w, dtype), cls, pe, 32), refc, [px, w, cls, pe], dtype, gpu,
             wrt=[1, 2, 3], name="clip_embed")


def test_adamw_matches_torch(gpu):
    from d2r_amd import _lib
    from d2r_amd.functional import _stream
    n = 1000 + 3
    w0, g = rnd(n), rnd(n, seed=1)
    for lp_dtype, lp_code in ((torch.bfloat16, _lib.BF16), (torch.float16, _lib.F16)):
        p = torch.nn.Parameter(w0.clone())
        opt = torch.optim.AdamW([p], lr=3e-3, weight_decay=1e-2)
        w = w0.clone().to(gpu)
        m, v = torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
        w16 = torch.zeros(n, dtype=lp_dtype, device=gpu)
        skip = torch.zeros(1, dtype=torch.int32, device=gpu)
        for step in range(1, 4):
            gs = g * step
            p.grad = gs.clone()
            opt.step()
            gg = (gs * 2.0).to(gpu)  # grad_scale 0.5 undoes the doubling (data-parallel SUM -> mean)
            _lib.call("d2r_adamw_step", w.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), w16.data_ptr(), lp_code, n, 3e-3, 0.9,
                      0.999, 1e-8, 1e-2, step, 0.5, skip.data_ptr(), _stream())
        assert float((w.cpu() - p.detach()).abs().max()) < 1e-6
        assert float((w16.float().cpu() - p.detach()).abs().max()) < (1e-2 if lp_dtype == torch.bfloat16 else 2e-3)
        # overflowed loss-scaled gradients: the check raises the flag and the flagged launch changes nothing
        before = (w.clone(), m.clone(), v.clone(), w16.clone())
        gg[17] = float("inf")
        _lib.call("d2r_grad_nonfinite", gg.data_ptr(), n, skip.data_ptr(), _stream())
        assert int(skip) == 1
        _lib.call("d2r_adamw_step", w.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), w16.data_ptr(), lp_code, n, 3e-3, 0.9,
                  0.999, 1e-8, 1e-2, 4, 0.5, skip.data_ptr(), _stream())
        for a, b in zip(before, (w, m, v, w16)):
            assert torch.equal(a, b)
        skip.zero_()
        gg[17] = float("nan")
        _lib.call("d2r_grad_nonfinite", gg.data_ptr(), n, skip.data_ptr(), _stream())
        assert int(skip) == 1
        skip.zero_()
        gg[17] = 3.0e38
        _lib.call("d2r_grad_nonfinite", gg.data_ptr(), n, skip.data_ptr(), _stream())
        assert int(skip) == 0


@pytest.mark.parametrize("cfg", [("NT", 32, 768, 768, 1), ("NT", 2, 768, 768, 1), ("NT", 17, 100, 128, 6), ("NT", 32, 1600, 768, 1), ("NT", 1, 1, 768, 1),
                                 ("NT", 32, 1200, 80, 20), ("NN", 32, 80, 1200, 20), ("NT", 5, 64, 1200, 2),
                                 ("NN", 32, 768, 768, 1), ("NN", 3, 768, 128, 6), ("NN", 32, 50, 1600, 1), ("NN", 32, 768, 4608, 1), ("NN", 17, 64, 80, 3),
                                 ("NN", 32, 1600, 768, 1)], ids=lambda c: "-".join(map(str, c)))
def test_skinny_fp32_gemm(gpu, cfg):
    """The fp32 kernel for products with at most 32 rows (router MLPs, poolers, Block head and chunk products; K a multiple of 16): plain, batched
    with per-batch bias rows, with activation + saved pre-activation, residual and accumulation, against fp64."""
    from d2r_amd import functional as F
    from d2r_amd._lib import F32, GEMM_NN, GEMM_NT, ACT_TANH_RELU, ACT_NONE
    lay, M, N, K, nb = cfg
    layout = GEMM_NT if lay == "NT" else GEMM_NN
    a = rnd(nb, M, K, seed=1).to(gpu)
    b = (rnd(nb, N, K, seed=2, scale=0.1) if lay == "NT" else rnd(nb, K, N, seed=2, scale=0.1)).to(gpu)
    bias = rnd(nb, N, seed=3).to(gpu)
    res = rnd(nb, M, N, seed=4).to(gpu)
    c0 = rnd(nb, M, N, seed=5).to(gpu)
    prod = a.double() @ (b.double().transpose(1, 2) if lay == "NT" else b.double())
    ldb = K if lay == "NT" else N
    # plain
    c = torch.empty(nb, M, N, device=gpu)
    F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), ldb, c.data_ptr(), N, dtype=F32, c_dtype=F32, nb=nb, sA=(M * K, 0), sB=(N * K, 0),
           sC=(M * N, 0))
    assert float((c.double() - prod).abs().max()) <= 2e-5 * float(prod.abs().max()) + 1e-6
    # bias (one row per batch) + relu(tanh) + saved pre-activation + residual + accumulate onto old C
    c, pre = c0.clone(), torch.empty(nb, M, N, device=gpu)
    F.gemm(layout, M, N, K, a.data_ptr(), K, b.data_ptr(), ldb, c.data_ptr(), N, dtype=F32, c_dtype=F32, nb=nb, sA=(M * K, 0), sB=(N * K, 0),
           sC=(M * N, 0), alpha=0.5, beta=1.0, bias=bias.data_ptr(), s_bias=N, act=ACT_TANH_RELU, residual=res.data_ptr(), ldr=N, sR=(M * N, 0),
           preact=pre.data_ptr())
    want_pre = 0.5 * prod + bias.double()[:, None, :]
    want = torch.relu(torch.tanh(want_pre)) + res.double() + c0.double()
    assert float((pre.double() - want_pre).abs().max()) <= 2e-5 * float(want_pre.abs().max()) + 1e-6
    assert float((c.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
def test_meanpool_backward_multi_matches_single_launches(gpu, lowp):
    """d2r_meanpool_bwd_multi (the pooled gradients of the six routers of a layer broadcast into six tensors in one launch, some
    accumulated, some overwritten) against one d2r_meanpool_bwd launch per tensor: bit-identical."""
    from d2r_amd import _lib
    from d2r_amd.functional import _parr, _stream
    code = _lib.BF16 if lowp == torch.bfloat16 else _lib.F16
    n, B, L, D = 6, 5, 37, 768
    dp = rnd(n, B, D, seed=1).to(gpu)
    base = [rnd(B, L, D, seed=10 + j).to(lowp).to(gpu) for j in range(n)]
    for mask in (1, 0x3f, 0b101010):
        one = [t.clone() for t in base]
        for j in range(n):
            _lib.call("d2r_meanpool_bwd", code, dp[j].data_ptr(), B, L, D, one[j].data_ptr(), (mask >> j) & 1, _stream())
        multi = [t.clone() for t in base]
        _lib.call("d2r_meanpool_bwd_multi", code, dp.data_ptr(), n, B, L, D, _parr(multi), mask, _stream())
        torch.cuda.synchronize()
        for j in range(n):
            assert torch.equal(one[j], multi[j]), (mask, j)


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
def test_matrix_vector_16bit_gemm(gpu, lowp):
    """N = 1 products of 16-bit operands (the SAF scores a = S w over B*(Lq+1) rows) take the wave-per-row kernel: fp32 and 16-bit
    output, bias, tanh, a row stride larger than K, against fp64 on the rounded operands."""
    from d2r_amd import functional as F
    from d2r_amd._lib import BF16, F16, F32, GEMM_NT, ACT_TANH
    code = BF16 if lowp == torch.bfloat16 else F16
    eps = 2.0 ** -8 if lowp == torch.bfloat16 else 2.0 ** -11
    for M, K, lda in ((4128, 768, 768), (70, 768, 1536), (6336, 96, 96)):
        a = rnd(M, lda, seed=1).to(lowp).to(gpu)
        b = rnd(1, K, seed=2, scale=0.1).to(lowp).to(gpu)
        bias = rnd(1, seed=3).to(gpu)
        prod = a[:, :K].double() @ b.double().t()
        c = torch.empty(M, 1, device=gpu)
        F.gemm(GEMM_NT, M, 1, K, a.data_ptr(), lda, b.data_ptr(), K, c.data_ptr(), 1, dtype=code, c_dtype=F32, bias=bias.data_ptr())
        want = prod + bias.double()
        assert float((c.double() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6, (M, K)
        c16 = torch.empty(M, 1, dtype=lowp, device=gpu)
        F.gemm(GEMM_NT, M, 1, K, a.data_ptr(), lda, b.data_ptr(), K, c16.data_ptr(), 1, dtype=code, c_dtype=code, alpha=0.5, act=ACT_TANH)
        want = torch.tanh(0.5 * prod)
        assert float((c16.double() - want).abs().max()) <= eps + 1e-6, (M, K)


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
def test_saf_rank_one_backward_products(gpu, lowp):
    """d2r_saf_dweights / d2r_saf_dscores (the rank-one products around the SAF gate in the backward pass of wsum[b] = w[b] @ S[b])
    against fp64 on the rounded operands: dw[b,i] = <dwsum[b], S[b,i]>, dS[b,i] = w[b,i] dwsum[b] + da[b,i] w_saf."""
    from d2r_amd import _lib
    from d2r_amd.functional import _stream
    code = _lib.BF16 if lowp == torch.bfloat16 else _lib.F16
    eps = 2.0 ** -8 if lowp == torch.bfloat16 else 2.0 ** -11
    for B, n in ((3, 17), (32, 129), (5, 198)):
        E = 768
        S = rnd(B, n, E, seed=1).to(lowp).to(gpu)
        dwsum = rnd(B, E, seed=2).to(lowp).to(gpu)
        w = rnd(B, n, seed=3).to(lowp).to(gpu)
        da = rnd(B, n, seed=4).to(gpu)
        w_saf = rnd(E, seed=5).to(lowp).to(gpu)
        dw = torch.empty(B, n, device=gpu)
        _lib.call("d2r_saf_dweights", code, dwsum.data_ptr(), S.data_ptr(), B, n, E, dw.data_ptr(), _stream())
        want = torch.einsum("be,bie->bi", dwsum.double(), S.double())
        assert float((dw.double() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6
        dS = torch.empty_like(S)
        _lib.call("d2r_saf_dscores", code, w.data_ptr(), dwsum.data_ptr(), da.data_ptr(), w_saf.data_ptr(), B, n, E, dS.data_ptr(), _stream())
        want = w.double()[:, :, None] * dwsum.double()[:, None, :] + da.double()[:, :, None] * w_saf.double()[None, None, :]
        assert float((dS.double() - want).abs().max()) <= eps * float(want.abs().max()) + 1e-6


@pytest.mark.parametrize("cfg", [(1600, 768, 32, 1), (768, 1600, 32, 1), (3, 768, 32, 1), (128, 768, 32, 6), (60, 80, 32, 20), (100, 70, 5, 1), (64, 64, 64, 2),
                                 (777, 130, 17, 1)], ids=lambda c: "x".join(map(str, c)))
def test_rank_k_fp32_tn_gemm(gpu, cfg):
    """The fp32 TN kernel for short reductions (K <= 64: weight gradients of the linears that see one row per sample - Block fusion head,
    routers, poolers): C = alpha A^T B (+ beta C) and dbias[m] += sum_k A[k,m], batched, ragged edges, against fp64."""
    from d2r_amd import functional as F
    from d2r_amd._lib import F32, GEMM_TN
    M, N, K, nb = cfg
    a = rnd(nb, K, M, seed=1).to(gpu)
    b = rnd(nb, K, N, seed=2).to(gpu)
    c0 = rnd(nb, M, N, seed=3).to(gpu)
    prod = a.double().transpose(1, 2) @ b.double()
    c = torch.empty(nb, M, N, device=gpu)
    F.gemm(GEMM_TN, M, N, K, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, dtype=F32, c_dtype=F32, nb=nb, sA=(K * M, 0), sB=(K * N, 0), sC=(M * N, 0))
    assert float((c.double() - prod).abs().max()) <= 2e-6 * float(prod.abs().max()) + 1e-6
    c = c0.clone()
    F.gemm(GEMM_TN, M, N, K, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, dtype=F32, c_dtype=F32, nb=nb, sA=(K * M, 0), sB=(K * N, 0), sC=(M * N, 0),
           alpha=0.5, beta=1.0)
    want = 0.5 * prod + c0.double()
    assert float((c.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6
    if nb == 1:  # bias gradient: accumulated into its sink
        db0 = rnd(M, seed=4).to(gpu)
        db, c = db0.clone(), c0.clone()
        F.gemm(GEMM_TN, M, N, K, a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, dtype=F32, c_dtype=F32, beta=1.0, dbias=db.data_ptr())
        want_b = db0.double() + a[0].double().sum(0)
        assert float((db.double() - want_b).abs().max()) <= 2e-6 * float(want_b.abs().max()) + 1e-6
        assert float((c.double() - (prod + c0.double())).abs().max()) <= 2e-6 * float((prod + c0.double()).abs().max()) + 1e-6


@pytest.mark.parametrize("lowp", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [("NT", 32, 768, 768), ("NT", 8, 768, 768), ("NT", 17, 96, 1600), ("NT", 1, 16, 64), ("NN", 32, 768, 768),
                                 ("NN", 3, 128, 768), ("NN", 32, 1600, 96)], ids=lambda c: "-".join(map(str, c)))
def test_skinny_16bit_gemm(gpu, cfg, lowp):
    """The 16-bit kernel for products with at most 32 rows (per-sample vectors of the routing cells; K a multiple of 32, NN: N a
    multiple of 16): plain with 16-bit and with fp32 output, strided A rows (token 0 of every sample), and the full epilogue
    (bias, tanh + saved pre-activation, residual, accumulation) against fp64 on the rounded operands."""
    from d2r_amd import functional as F
    from d2r_amd._lib import BF16, F16, F32, GEMM_NN, GEMM_NT, ACT_TANH, ACT_NONE
    lay, M, N, K = cfg
    code = BF16 if lowp == torch.bfloat16 else F16
    layout = GEMM_NT if lay == "NT" else GEMM_NN
    L = 5
    a_all = rnd(M, L, K, seed=1).to(lowp).to(gpu)  # rows of the product = token 0 of every sample: lda = L * K
    a = a_all[:, 0]
    b = (rnd(N, K, seed=2, scale=0.1) if lay == "NT" else rnd(K, N, seed=2, scale=0.1)).to(lowp).to(gpu)
    bias = rnd(N, seed=3).to(gpu)
    res = rnd(M, N, seed=4).to(lowp).to(gpu)
    c0 = rnd(M, N, seed=5).to(lowp).to(gpu)
    prod = a.double() @ (b.double().t() if lay == "NT" else b.double())
    ldb = K if lay == "NT" else N
    eps = 2.0 ** -8 if lowp == torch.bfloat16 else 2.0 ** -11
    for c_dt, c_code in ((lowp, code), (torch.float32, F32)):
        c = torch.empty(M, N, dtype=c_dt, device=gpu)
        F.gemm(layout, M, N, K, a_all.data_ptr(), L * K, b.data_ptr(), ldb, c.data_ptr(), N, dtype=code, c_dtype=c_code)
        tol = (eps if c_dt != torch.float32 else 1e-5) * float(prod.abs().max()) + 1e-6
        assert float((c.double() - prod).abs().max()) <= tol, (cfg, c_dt)
    c, pre = c0.clone(), torch.empty(M, N, dtype=lowp, device=gpu)
    F.gemm(layout, M, N, K, a_all.data_ptr(), L * K, b.data_ptr(), ldb, c.data_ptr(), N, dtype=code, c_dtype=code, alpha=0.5, beta=1.0,
           bias=bias.data_ptr(), act=ACT_TANH, residual=res.data_ptr(), ldr=N, preact=pre.data_ptr())
    want_pre = 0.5 * prod + bias.double()[None, :]
    want = torch.tanh(want_pre) + res.double() + c0.double()
    assert float((pre.double() - want_pre).abs().max()) <= eps * float(want_pre.abs().max()) + 1e-6
    assert float((c.double() - want).abs().max()) <= 2 * eps * float(want.abs().max()) + 1e-6


def test_copy_rows(gpu):
    """d2r_copy_rows: strided row gather / scatter in one launch (16-byte vector path and the byte path), checked against slicing."""
    from d2r_amd import _lib
    from d2r_amd.functional import _stream
    src = torch.arange(40 * 4096, dtype=torch.int32, device=gpu).view(40, 4096).to(torch.uint8)  # values mod 256
    for (rows, width, sp, dp, so, do) in ((32, 1536, 4096, 2048, 0, 0), (7, 1536 * 5, 1536 * 6, 1536 * 5, 1536, 0), (5, 37, 100, 64, 3, 1)):
        s_ = src.flatten()[: rows * sp + so + 64].clone()
        dst = torch.full((rows * dp + do + 64,), 255, dtype=torch.uint8, device=gpu)
        _lib.call("d2r_copy_rows", dst.data_ptr() + do, dp, s_.data_ptr() + so, sp, width, rows, _stream())
        torch.cuda.synchronize()
        want = dst.clone().fill_(255)
        for r in range(rows):
            want[do + r * dp: do + r * dp + width] = s_[so + r * sp: so + r * sp + width]
        assert torch.equal(dst, want), (rows, width, sp, dp, so, do)


def test_ops_refuse_cpu_tensors():
    from d2r_amd import functional as F
    from d2r_amd import D2RError
    with pytest.raises(D2RError):
        F.linear(torch.zeros(2, 8), torch.zeros(4, 8), None)


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("kind", ["bert", "clip", "bert-dropout"])
def test_encoder_layer_one_call_matches_op_by_op(gpu, kind, lowp, monkeypatch):
    """d2r_encoder_layer_fwd/bwd (one C call per layer and direction) against the op-by-op path built from the same
    kernels: the forward is bit-identical; the backward differs only where a skip-connection gradient is now added in
    fp32 inside a GEMM / LayerNorm epilogue instead of by a separate bf16 add.  "bert-dropout": train mode with the
    bert-base dropout of 0.1 on the attention probabilities and on both dense outputs - the same seeds give the same masks
    on both paths (the one-call path masks the probabilities inside the fused attention core)."""
    from d2r_amd import functional as F
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig
    from d2r_amd.params import ParamStore
    torch.manual_seed(3)
    if kind.startswith("bert"):
        pd = 0.1 if kind == "bert-dropout" else 0.0
        layer = M.BertLayer(TextConfig(num_hidden_layers=1, hidden_dropout_prob=pd, attention_probs_dropout_prob=pd))
        B, L = 3, 37
    else:
        layer = M.CLIPEncoderLayer(VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32))
        B, L = 2, 50

    class Wrap(M.D2RModule):
        def __init__(self, layer):
            super().__init__()
            self.layer = layer

    model = Wrap(layer).to(gpu)
    model.set_compute_dtype(lowp).train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "LayerNorm" in n or "layer_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    store = ParamStore(model, lowp)
    x0 = torch.randn(B, L, 768, device=gpu).to(lowp)
    gy = torch.randn(B, L, 768, device=gpu).to(lowp)
    mask = torch.zeros(B, L, device=gpu)
    mask[0, L // 2:] = -10000.0
    res = {}
    for composite in (False, True):
        M.COMPOSITE_LAYERS = composite
        seeds = iter(range(7000, 7100))  # both paths draw: probabilities, attention output, FFN output
        monkeypatch.setattr(F, "_next_dropout_seed", lambda: next(seeds))
        try:
            store.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = layer(x, mask) if kind.startswith("bert") else layer(x)
            assert (type(y.grad_fn).__name__ == "_EncoderLayerBackward") == composite
            y.backward(gy)
            torch.cuda.synchronize()
            res[composite] = (y.detach().clone(), x.grad.clone(), store.flat_g.clone())
        finally:
            M.COMPOSITE_LAYERS = True
    (y0, dx0, g0), (y1, dx1, g1) = res[False], res[True]
    assert torch.equal(y0, y1), "forward differs"
    if kind == "bert-dropout":  # the masks really were applied: the eval-mode output differs
        layer.eval()
        with torch.no_grad():
            assert not torch.equal(layer(x0, mask), y1)
        layer.train()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    assert rel(dx1, dx0) < 1e-2, rel(dx1, dx0)
    for n, p, o, k, _ in store.entries:
        r = float((g1[o:o + k] - g0[o:o + k]).norm() / (g0[o:o + k].norm() + 1e-3 * g0.norm()))  # floor: a key bias has a mathematically zero gradient
        assert r < 2e-2, (n, r)


@pytest.mark.parametrize("kind", ["bert", "clip"])
def test_encoder_layers_deferred_layernorm_sums_are_bit_identical(gpu, kind):
    """The second stage of the LayerNorm backward (per-block partial sums -> gamma / beta gradients) deferred out of
    d2r_encoder_layer_bwd and launched once per group of layers (d2r_layernorm_bwd_sum_grouped) against the sum inside the call:
    the same additions in the same order - every gradient of a three-layer stack is bit-identical."""
    from d2r_amd import functional as F
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig
    from d2r_amd.params import ParamStore
    torch.manual_seed(5)
    if kind == "bert":
        layers = [M.BertLayer(TextConfig(num_hidden_layers=3, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)) for _ in range(3)]
        B, L = 3, 37
    else:
        layers = [M.CLIPEncoderLayer(VisionConfig(num_hidden_layers=3, image_size=64, patch_size=32)) for _ in range(3)]
        B, L = 2, 50

    class Stack(M.D2RModule):
        def __init__(self, layers):
            super().__init__()
            self.layers = torch.nn.ModuleList(layers)

    model = Stack(layers).to(gpu)
    model.set_compute_dtype(torch.float16).train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "LayerNorm" in n or "layer_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    store = ParamStore(model, torch.float16)
    x0 = torch.randn(B, L, 768, device=gpu).half()
    gy = torch.randn(B, L, 768, device=gpu).half()
    mask = torch.zeros(B, L, device=gpu)
    res = {}
    saved = F.DEFER_LN
    try:
        for defer in (False, True):
            F.DEFER_LN = defer
            store.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = x
            for layer in model.layers:
                y = layer(y, mask) if kind == "bert" else layer(y)
            assert type(y.grad_fn).__name__ == "_EncoderLayerBackward"
            y.backward(gy)
            F.flush_wgrads()
            torch.cuda.synchronize()
            res[defer] = (x.grad.clone(), store.flat_g.clone())
    finally:
        F.DEFER_LN = saved
    assert torch.equal(res[False][0], res[True][0]), "input gradient differs"
    g0, g1 = res[False][1], res[True][1]
    ln = [(n, o, k) for n, p, o, k, _ in store.entries if "LayerNorm" in n or "layer_norm" in n]
    assert len(ln) == 12 and all(float(g0[o:o + k].abs().max()) > 0 for _, o, k in ln), "the LayerNorm gradients were not produced"
    bad = [n for n, p, o, k, _ in store.entries if not torch.equal(g0[o:o + k], g1[o:o + k])]
    assert not bad, f"gradients differ: {bad[:6]} ({len(bad)} tensors)"


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [("text", 6, 3, 3, 24, 10, True), ("image", 6, 4, 2, 10, 24, True), ("text", 4, 3, 2, 16, 7, True),
                                 ("image", 6, 3, 2, 12, 9, False), ("text", 6, 2, 2, 9, 5, True),
                                 # token rows >= 128: the cells' products leave as GROUPED launches (d2r_gemm_group) inside the call
                                 ("image", 6, 3, 4, 128, 197, False), ("text", 6, 3, 2, 128, 197, True), ("text", 4, 4, 2, 197, 128, True)],
                         ids=lambda c: f"{c[0]}-nc{c[1]}-dr{c[2]}-B{c[3]}x{c[4]}-{'train' if c[6] else 'eval'}")
def test_interaction_module_one_call_matches_op_by_op(gpu, cfg, lowp):
    """d2r_interaction_fwd/bwd (K16: one C call per module and direction) against the op-by-op path built from the same
    kernels, on identical weights and inputs with about half of the paths pruned: the forward (aggregated embedding, path
    similarities, BatchNorm running statistics) is bit-identical; the backward differs only in WHERE multi-consumer
    gradients are summed (GEMM epilogues instead of separate bf16 adds) — input gradients and every parameter gradient are
    compared per tensor."""
    from d2r_amd import modules as M
    from d2r_amd.config import default_args
    from d2r_amd.params import ParamStore
    branch, nc, dr, B, Lq, Lk, train = cfg
    torch.manual_seed(11)
    cls = M.InteractionModule if branch == "text" else M.Reversed_InteractionModule
    mod = cls(default_args(DR_step=dr), num_layer_routing=dr, num_cells=nc, path_hid=128).to(gpu)
    mod.set_compute_dtype(lowp).train(train)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("router.mlp.2.bias"):
                p.normal_()
    store = ParamStore(mod, lowp)
    own0 = torch.randn(B, Lq, 768, device=gpu).to(lowp)
    other0 = torch.randn(B, Lk, 768, device=gpu).to(lowp)
    r_emb = torch.randn(B, Lq, 768, device=gpu)
    r_sim = torch.randn(B, B, device=gpu)
    buffers0 = {k: v.clone() for k, v in mod.named_buffers()}
    res = {}
    for composite in (False, True):
        M.COMPOSITE_ROUTING = composite
        try:
            with torch.no_grad():
                for k, v in mod.named_buffers():
                    v.copy_(buffers0[k])
            store.zero_grad()
            own, other = own0.clone().requires_grad_(True), other0.clone().requires_grad_(True)
            text, image = (own, other) if branch == "text" else (other, own)
            (emb,), sim = mod(text, image)
            assert ("_InteractionBackward" in repr(emb.grad_fn)) == composite, emb.grad_fn
            ((emb.float() * r_emb).sum() + (sim * r_sim).sum()).backward()
            torch.cuda.synchronize()
            res[composite] = (emb.detach().clone(), sim.detach().clone(), own.grad.clone(), other.grad.clone(), store.flat_g.clone(),
                              {k: v.clone() for k, v in mod.named_buffers()})
        finally:
            M.COMPOSITE_ROUTING = True
    (e0, s0, do0, dt0, g0, b0), (e1, s1, do1, dt1, g1, b1) = res[False], res[True]
    assert torch.equal(e0, e1), f"forward differs: max {float((e0.float() - e1.float()).abs().max()):.3e}"
    assert torch.equal(s0, s1), "path similarities differ"
    for k in b0:
        assert torch.equal(b0[k], b1[k]), f"buffer {k} differs"
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    assert rel(do1, do0) < 2e-2, ("d own", rel(do1, do0))
    assert rel(dt1, dt0) < 2e-2, ("d other", rel(dt1, dt0))
    gn = float(g0.norm())
    worst = ("", 0.0)
    for n, p, o, k, _ in store.entries:
        a, b = g1[o:o + k], g0[o:o + k]
        zero_grad = n.endswith(("key.bias", "linears.1.bias", "crcmc.fc_2.bias"))  # (crcmc.fc_2 makes the keys of softmax(Q K^T), Cells.py:244)
        zero_grad |= train and n.endswith("attn_sim_w.bias")  # a constant shift of the SAF scores is removed by the batch-statistics BatchNorm
        if zero_grad:
            # a key bias shifts every logit of a softmax row equally: its gradient is mathematically zero, what both paths
            # hold is rounding noise of different summation orders
            assert float(a.norm()) < 1e-2 * gn and float(b.norm()) < 1e-2 * gn, (n, float(a.norm()), float(b.norm()), gn)
            continue
        r = float((a - b).norm() / (b.norm() + 1e-3 * gn / len(store.entries) ** 0.5))
        if r > worst[1]:
            worst = (n, r)
        assert r < 5e-2, (n, r, float(a.norm()), float(b.norm()))
    print(f"[{cfg}] one-call vs op-by-op: d_own {rel(do1, do0):.2e} d_other {rel(dt1, dt0):.2e} worst parameter gradient {worst[0]} {worst[1]:.2e}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "fp16"])
def test_head_one_call_matches_op_by_op(gpu, dtype):
    """d2r_head_fwd / d2r_head_bwd (K17: Block fusion + fc + cross entropy + loss as one C call each way) against the op-by-op path
    on a whole tiny model: the same launches with the same descriptors, so loss, logits and EVERY parameter gradient are
    bit-identical (the head runs in fp32 in every compute mode)."""
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    from d2r_amd.params import ParamStore
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    sd = O.seeded_state_dict(cfg, seed=3, router_bias="normal")
    batch = tuple(t.to(gpu) for t in O.synthetic_batch(cfg, 5, 12, seed=4))
    res = {}
    for composite in (False, True):
        M.COMPOSITE_HEAD = composite
        try:
            tc = TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
            vc = VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32)
            model = M.UnimoModelF(default_args(), vc, tc)
            model.load_state_dict(sd, strict=True)
            model.to(gpu).set_compute_dtype(dtype).train()
            model.model.use_streams = False
            store = ParamStore(model, dtype)
            loss, logits = model(*batch)
            assert ("_HeadBackward" in repr(loss.grad_fn)) == composite, loss.grad_fn
            (loss * 64.0).backward()
            torch.cuda.synchronize()
            res[composite] = (loss.detach().clone(), logits.detach().clone(), store.flat_g.clone(), [(n, o, k) for n, _, o, k, _ in store.entries])
        finally:
            M.COMPOSITE_HEAD = True
    (l0, g0, f0, ent), (l1, g1, f1, _) = res[False], res[True]
    assert torch.equal(l0, l1) and torch.equal(g0, g1), (float(l0), float(l1))
    bad = [n for n, o, k in ent if not torch.equal(f0[o:o + k], f1[o:o + k])]
    assert not bad, f"{len(bad)} parameter gradients differ, first {bad[:5]}"
    assert float(f1.abs().max()) > 0.0


def test_head_one_call_backpropagates_a_second_loss_on_the_logits(gpu):
    """A caller that adds its own loss on the returned logits (distillation, label smoothing outside the model) - the reference model
    allows it: the one-call head adds the incoming logit gradient to the cross-entropy path (d2r_head_desc.d_logits) and gives the same
    parameter gradients as the op-by-op head."""
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    from d2r_amd.params import ParamStore
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    sd = O.seeded_state_dict(cfg, seed=3, router_bias="normal")
    batch = tuple(t.to(gpu) for t in O.synthetic_batch(cfg, 5, 12, seed=4))
    w = torch.randn(5, 3, device=gpu)
    res = {}
    for composite in (False, True):
        M.COMPOSITE_HEAD = composite
        try:
            model = M.UnimoModelF(default_args(), VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32),
                                  TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
            model.load_state_dict(sd, strict=True)
            model.to(gpu).set_compute_dtype(torch.float32).train()
            model.model.use_streams = False
            store = ParamStore(model, torch.float32)
            loss, logits = model(*batch)
            (loss + 3.0 * (logits * w).sum()).backward()
            torch.cuda.synchronize()
            res[composite] = (store.flat_g.clone(), [(n, o, k) for n, _, o, k, _ in store.entries])
        finally:
            M.COMPOSITE_HEAD = True
    (f0, ent), (f1, _) = res[False], res[True]
    for n, o, k in ent:
        a, b = f1[o:o + k], f0[o:o + k]
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, n
    only_ce = float(f0.norm())
    assert only_ce > 0


@pytest.mark.parametrize("dtype", DT)
def test_dropout_kernel(gpu, dtype):
    """nn.Dropout semantics: keep fraction 1-p, survivors scaled by 1/(1-p), residual add fused, mask = pure function
    of (seed, index) so the backward regenerates it; new seed -> new mask; eval / p=0 -> identity."""
    from d2r_amd import functional as F
    torch.manual_seed(11)
    x = (1.0 + torch.rand(7, 333, 77)).to(dtype).to(gpu).requires_grad_(True)  # odd sizes: scalar tail, all > 0
    r = torch.randn(7, 333, 77).to(dtype).to(gpu).requires_grad_(True)
    p = 0.1
    y = F.dropout(x, p, True)
    keep = y != 0
    frac = float(keep.float().mean())
    assert abs(frac - (1 - p)) < 3e-3, frac
    tol = {torch.bfloat16: 1e-2, torch.float16: 1.5e-3, torch.float32: 1e-6}[dtype]
    atol = 1e-6 if dtype == torch.float16 else 1e-8  # fp16 values below 6e-5 are subnormal: absolute, not relative, rounding
    assert torch.allclose(y[keep].float(), x.detach()[keep].float() / (1 - p), rtol=tol, atol=atol)
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad != 0, keep & (g != 0))  # same mask in the backward
    assert torch.allclose(x.grad[keep].float(), g[keep].float() / (1 - p), rtol=tol, atol=atol)
    y2 = F.dropout(x, p, True)
    assert not torch.equal(y2 != 0, keep), "every call must draw a fresh mask"
    x.grad = None
    z = F.dropout(x, p, True, residual=r)
    z.backward(g)
    assert torch.equal(r.grad, g)
    assert float(((z - r).abs() > 1e-3).float().mean()) == pytest.approx(1 - p, abs=3e-3)
    assert F.dropout(x, p, False) is x and F.dropout(x, 0.0, True) is x
    torch.manual_seed(11)
    a = F.dropout(x, p, True)
    torch.manual_seed(11)
    b = F.dropout(x, p, True)
    assert torch.equal(a, b), "torch.manual_seed must restart the dropout stream"


def test_bert_layer_with_dropout_backward_is_consistent(gpu, monkeypatch):
    """BertLayer in train mode with hidden and attention-probability dropout 0.1 (the bert-base default): with the seed
    stream frozen the layer is a deterministic function, so the analytic gradient must match a central finite
    difference along a random direction (fp32 path)."""
    from d2r_amd import functional as F
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig
    torch.manual_seed(5)
    layer = M.BertLayer(TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)).to(gpu)
    layer.set_compute_dtype(torch.float32).train()
    B, L = 2, 19
    x0 = torch.randn(B, L, 768, device=gpu)
    w = torch.randn(B, L, 768, device=gpu)
    d = torch.randn(B, L, 768, device=gpu)
    mask = torch.zeros(B, L, device=gpu)
    mask[1, 12:] = -10000.0

    def f(x):
        seeds = iter(range(1000, 1100))
        monkeypatch.setattr(F, "_next_dropout_seed", lambda: next(seeds))
        return (layer(x, mask) * w).sum()

    x = x0.clone().requires_grad_(True)
    y = f(x)
    y.backward()
    analytic = float((x.grad * d).sum())
    eps = 1e-2
    with torch.no_grad():
        numeric = float((f(x0 + eps * d) - f(x0 - eps * d)) / (2 * eps))
    assert abs(analytic - numeric) <= 2e-2 * max(abs(analytic), abs(numeric), 1.0), (analytic, numeric)
    layer.eval()
    with torch.no_grad():
        assert torch.equal(layer(x0, mask), layer(x0, mask)), "eval mode must not drop anything"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(19, 300, 136, 72), (3, 200, 72, 1536), (3, 1576, 256, 384), (2, 200, 136, 264), (17, 192, 128, 128),
                                   (9, 32, 768, 768), (3, 5, 130, 77)],
                         ids=["19x136x72", "3x72x1536", "3x256x384-T1576", "2x136x264-T200", "17x128x128", "9x768x768-T32", "3x130x77-T5"])
def test_grouped_weight_gradient_gemm(gpu, dtype, shape):
    """d2r_gemm_tn_grouped: same-shape dW = dY^T X problems (19: two launches, 16 + 3; 24 tile columns: the launch order
    without the whole-problem-per-XCD remap) with bias-gradient side product and accumulation into pre-filled sinks,
    against torch.  Outputs >= 128 x 128 in bf16 run on the 128x128 LDS-DMA weight-gradient kernel: ragged output edges
    (136 x 264), token counts that are not a multiple of the 64-row K tile (1576 = 8 x 197, 200), 17 problems (16 + 1); at most 64
    token rows (one row per sample: the pooled-vector linears) run on the rank-K kernel."""
    from d2r_amd import _lib
    from d2r_amd.functional import _parr, _stream
    n, T, N, K = shape
    gs = [rnd(T, N, seed=i).to(dtype).to(gpu) for i in range(n)]
    xs = [rnd(T, K, seed=100 + i).to(dtype).to(gpu) for i in range(n)]
    sinks = [rnd(N, K, seed=200 + i).to(gpu) for i in range(n)]
    bsinks = [rnd(N, seed=300 + i).to(gpu) for i in range(n)]
    want_w = [s.double() + g.double().t() @ x.double() for s, g, x in zip(sinks, gs, xs)]
    want_b = [b.double() + g.double().sum(0) for b, g in zip(bsinks, gs)]
    dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float32: _lib.F32}[dtype]
    _lib.call("d2r_gemm_tn_grouped", dt, N, K, T, N, K, K, _parr(gs), _parr(xs), _parr(sinks), _parr(bsinks), n, 1.0, _stream())
    torch.cuda.synchronize()
    for i in range(n):
        scale = float(want_w[i].abs().max())
        assert float((sinks[i].double() - want_w[i]).abs().max()) <= 1e-5 * scale + 1e-6, i
        assert float((bsinks[i].double() - want_b[i]).abs().max()) <= 1e-5 * float(want_b[i].abs().max()) + 1e-6, i


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(5, 61, 192, 136), (2, 128, 768, 768), (3, 131, 256, 72)], ids=lambda s: "x".join(map(str, s)))
def test_linear_many_rows(gpu, dtype, shape):
    """>= 128 rows: in bf16 the forward (NT) and dX (NN) products run on the LDS-DMA 128x64 kernel (gemm_glds.hip), the
    workhorse of the real workload — ragged row / column edges, bias + GELU with saved pre-activation, bias + residual;
    dW / db go through the deferred grouped path only with a ParamStore, here through the generic TN kernel."""
    from d2r_amd import functional as F
    B, L, K, N = shape
    x, w, b, r = rnd(B, L, K), wt(N, K, dtype=dtype, scale=0.1), keep32(rnd(N, scale=0.5)), rnd(B, L, N, seed=7)
    gelu = lambda v: 0.5 * v * (1 + torch.erf(v / math.sqrt(2)))
    run_both(lambda x, w, b: F.linear(x, w, b, lp(w, dtype), act=3), lambda x, w, b: gelu(x @ w.t() + b), [x, w, b], dtype, gpu,
             name=f"linear gelu {shape}")
    run_both(lambda x, w, b, r: F.linear(x, w, b, lp(w, dtype), residual=r), lambda x, w, b, r: x @ w.t() + b + r,
             [x, w, b, r], dtype, gpu, name=f"linear+res {shape}")


@pytest.mark.parametrize("lowp", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(5, 128, 197, 3), (9, 197, 128, 3), (3, 40, 250, 2), (2, 130, 577, 3), (32, 128, 197, 3)], ids=lambda c: "x".join(map(str, c)))
def test_xattn_multi_launch_matches_single_launches(gpu, lowp, cfg):
    """d2r_xattn_fwd_multi / d2r_xattn_bwd_multi (several attention problems of one shape in one launch: the three alignment
    cores of a routing layer) give, per problem, exactly the bits of the single-problem entry points, and dk / dv (the grouped,
    batched key-side launch; packed k|v layout as in the routing cells) match an fp64 expression of the same products."""
    import ctypes as C
    from d2r_amd import _lib
    from d2r_amd import functional as F
    B, Lq, Lk, nc = cfg
    E, dt = 768, F._dt_of(lowp)
    scale = 100.0 / math.sqrt(768)
    lkp = (Lk + 7) // 8 * 8
    st = torch.cuda.current_stream().cuda_stream
    arr = lambda ts: F._iparr([t if isinstance(t, int) else t.data_ptr() for t in ts])
    q = [rnd(B, Lq, E, seed=10 + c, scale=0.15).to(lowp).to(gpu) for c in range(nc)]
    kv = [rnd(B, Lk, 2 * E, seed=20 + c, scale=0.15).to(lowp).to(gpu) for c in range(nc)]
    dO = [rnd(B, Lq, E, seed=30 + c).to(lowp).to(gpu) for c in range(nc)]
    es = 2

    def fwd(multi):
        o = [torch.empty(B, Lq, E, dtype=lowp, device=gpu) for _ in range(nc)]
        lse = [torch.empty(B, Lq, dtype=torch.float32, device=gpu) for _ in range(nc)]
        if multi:
            _lib.call("d2r_xattn_fwd_multi", dt, nc, arr(q), E, Lq * E, arr(kv), 2 * E, Lk * 2 * E, arr([t.data_ptr() + E * es for t in kv]), 2 * E,
                      Lk * 2 * E, arr(o), E, Lq * E, None, E, Lq * E, None, arr(lse), B, Lq, Lk, E, scale, st)
        else:
            for c in range(nc):
                _lib.call("d2r_xattn_fwd", dt, q[c].data_ptr(), E, Lq * E, kv[c].data_ptr(), 2 * E, Lk * 2 * E, kv[c].data_ptr() + E * es, 2 * E,
                          Lk * 2 * E, o[c].data_ptr(), E, Lq * E, None, E, Lq * E, None, lse[c].data_ptr(), B, Lq, Lk, E, scale, st)
        return o, lse

    o1, l1 = fwd(False)
    o3, l3 = fwd(True)
    for c in range(nc):
        assert torch.equal(o1[c], o3[c]) and torch.equal(l1[c], l3[c]), f"forward of problem {c} differs between the multi and the single launch"

    def bwd(ncore_per_call):
        dq = [torch.empty(B, Lq, E, dtype=lowp, device=gpu) for _ in range(nc)]
        dkv = [torch.empty(B, Lk, 2 * E, dtype=lowp, device=gpu) for _ in range(nc)]
        P = [torch.empty(B, Lq, lkp, dtype=lowp, device=gpu) for _ in range(nc)]
        dS = [torch.empty(B, Lq, lkp, dtype=lowp, device=gpu) for _ in range(nc)]
        for c0 in range(0, nc, ncore_per_call):
            sl = slice(c0, c0 + ncore_per_call)
            n = len(q[sl])
            _lib.call("d2r_xattn_bwd_multi", dt, n, arr(q[sl]), E, Lq * E, arr(kv[sl]), 2 * E, Lk * 2 * E, arr([t.data_ptr() + E * es for t in kv[sl]]),
                      2 * E, Lk * 2 * E, arr(dO[sl]), E, Lq * E, arr(o1[sl]), E, Lq * E, None, E, Lq * E, None, arr(l1[sl]), arr(dq[sl]), E, Lq * E, arr(dkv[sl]), 2 * E, Lk * 2 * E,
                      arr([t.data_ptr() + E * es for t in dkv[sl]]), 2 * E, Lk * 2 * E, arr(P[sl]), arr(dS[sl]), lkp, B, Lq, Lk, E, scale, st)
        return dq, dkv, P, dS

    a = bwd(1)
    b = bwd(nc)
    torch.cuda.synchronize()
    for c in range(nc):
        for name, x, y in zip(("dq", "dkv", "P", "dS"), [t[c] for t in a], [t[c] for t in b]):
            assert torch.equal(x[..., :Lk] if name in ("P", "dS") else x, y[..., :Lk] if name in ("P", "dS") else y), f"{name} of problem {c} differs"
    # dk / dv against fp64 from the kernel's own P / dS (isolates the grouped batched product)
    for c in range(nc):
        P64, dS64 = b[2][c][..., :Lk].double().cpu(), b[3][c][..., :Lk].double().cpu()
        dv_ref = P64.transpose(1, 2) @ dO[c].double().cpu()
        dk_ref = dS64.transpose(1, 2) @ q[c].double().cpu()
        check(f"dv[{c}]", b[1][c][..., E:], dv_ref, lowp)
        check(f"dk[{c}]", b[1][c][..., :E], dk_ref, lowp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "fp16"])
def test_dual_elementwise_launches_equal_two_single_launches(gpu, dtype):
    """d2r_add2 / d2r_act_bwd2 run two independent problems of one size in one launch (the text / image and a / b pairs of the routing
    cells' backward): element for element the arithmetic of two single calls, so the results are bit-identical - also for a size
    that is not a multiple of the pack width and for unaligned operands (the scalar tail / scalar path)."""
    from d2r_amd import _lib
    from d2r_amd import functional as F
    dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    st = F._stream()
    g = torch.Generator(device=gpu).manual_seed(3)
    for n, off in ((32 * 768, 0), (1003, 0), (4099, 1)):
        t = [torch.randn(n + 8, device=gpu, generator=g).to(dtype)[off:off + n] for _ in range(4)]
        for act in (_lib.ACT_TANH, _lib.ACT_RELU):
            o1, o2, p1, p2 = (torch.empty(n + 8, device=gpu, dtype=dtype)[off:off + n] for _ in range(4))
            _lib.call("d2r_act_bwd", dt, act, t[0].data_ptr(), t[1].data_ptr(), o1.data_ptr(), n, st)
            _lib.call("d2r_act_bwd", dt, act, t[2].data_ptr(), t[3].data_ptr(), o2.data_ptr(), n, st)
            _lib.call("d2r_act_bwd2", dt, act, t[0].data_ptr(), t[1].data_ptr(), p1.data_ptr(), t[2].data_ptr(), t[3].data_ptr(), p2.data_ptr(), n, st)
            torch.cuda.synchronize()
            assert torch.equal(o1, p1) and torch.equal(o2, p2), (n, off, "act_bwd2")
        o1, o2, p1, p2 = (torch.empty(n + 8, device=gpu, dtype=dtype)[off:off + n] for _ in range(4))
        _lib.call("d2r_add", dt, t[0].data_ptr(), t[1].data_ptr(), o1.data_ptr(), n, st)
        _lib.call("d2r_add", dt, t[2].data_ptr(), t[1].data_ptr(), o2.data_ptr(), n, st)
        _lib.call("d2r_add2", dt, t[0].data_ptr(), t[1].data_ptr(), p1.data_ptr(), t[2].data_ptr(), t[1].data_ptr(), p2.data_ptr(), n, st)
        torch.cuda.synchronize()
        assert torch.equal(o1, p1) and torch.equal(o2, p2), (n, off, "add2")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "fp16"])
def test_column_sums_small_and_large_row_counts_and_accumulation(gpu, dtype):
    """d2r_colsum over at most 32 rows writes its single slice of partial sums straight to the output (one launch); more rows go through the
    workspace and the fixed-order second stage; d2r_colsum_add adds the sums of at most 32 rows into an fp32 sink (the bias gradients of
    the routers) and refuses more rows.  Against fp64 column sums; vectorised and scalar paths (N a multiple of the pack width or not)."""
    from d2r_amd import _lib
    from d2r_amd import functional as F
    dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}[dtype]
    st = F._stream()
    g = torch.Generator(device=gpu).manual_seed(12)
    lib = _lib.load()
    for M, N, ld in ((32, 768, 768), (17, 1003, 1010), (1, 64, 64), (200, 768, 768), (4096, 136, 144)):
        x = torch.randn(M, ld, device=gpu, generator=g).to(dtype)
        ws = torch.empty(lib.d2r_colsum_workspace(M, N), dtype=torch.uint8, device=gpu)
        out = torch.full((N,), 7.0, device=gpu)
        _lib.call("d2r_colsum", dt, x.data_ptr(), ld, M, N, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        ref = x[:, :N].double().sum(0)
        tol = 1e-6 * (M ** 0.5) * float(x[:, :N].double().abs().max()) * M ** 0.5 + 1e-6
        assert float((out.double() - ref).abs().max()) <= tol, (M, N, "colsum")
        sink = torch.randn(N, device=gpu, generator=g)
        s0 = sink.clone()
        if M <= 32:
            _lib.call("d2r_colsum_add", dt, x.data_ptr(), ld, M, N, sink.data_ptr(), ws.data_ptr(), ws.numel(), st)
            torch.cuda.synchronize()
            assert torch.equal(sink, s0 + out), (M, N, "colsum_add is sink + colsum, one fp32 add per column")
        else:
            with pytest.raises(_lib.D2RError):
                _lib.call("d2r_colsum_add", dt, x.data_ptr(), ld, M, N, sink.data_ptr(), ws.data_ptr(), ws.numel(), st)
