"""Which precision policy meets |dlogits| <= 1e-3 and gradient cosine >= 0.99 at the benchmark shape?  CPU-only probe
(test infrastructure: imports the oracle).

Runs the pinned oracle at the C2 shape (12+12 layers, L=128, 197 image tokens, default init, seed 2023 — the model of
tests/test_gpu_bench_shapes.py) in fp64 (truth) and in fp32 under emulated precision policies:
  ops    : matmul-class operands rounded to a 16-bit type per region (what an MFMA consumes)
  store  : which op outputs are rounded as well: 'all' (every activation tensor is 16-bit: the round-2 HIP paths),
           'operand' (only tensors whose consumer is a matmul anyway, i.e. matmul / LayerNorm / activation outputs: the
           residual stream, sums and products stay fp32), 'none'
The rounding is straight-through for autograd, so the gradient shows what the FORWARD rounding alone does to the
backward pass (the dominant effect: Block's signed square root has an unbounded derivative at 0).

    python tests/probes/mixed_policy_c2.py [BATCH]            env: LAYERS SEQ IMG SEED POLICY
"""
import os
import sys
import time

import torch
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import d2r_oracle as O  # noqa: E402

REGION = ["other"]


def _wrap(name, region):
    fn = getattr(O, name)

    def inner(*a, **k):
        REGION.append(region)
        try:
            return fn(*a, **k)
        finally:
            REGION.pop()
    setattr(O, name, inner)


for n, r in (("bert_layer", "enc"), ("clip_layer", "enc"), ("vision_embed", "enc"), ("text_embed", "enc"),
             ("interaction_module", "routing"), ("block_fusion", "block"), ("router_gate", "router"),
             ("js_div", "loss"), ("_saf", "saf"), ("xalign", "xalign"), ("cls_pool", "pool")):
    _wrap(n, r)
for cname in list(O.CELLS):
    O.CELLS[cname] = getattr(O, "cell_" + cname)


class _STE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


def q(x, dt):
    if dt is None or not torch.is_tensor(x) or not x.is_floating_point():
        return x
    return _STE.apply(x, dt)


MM = {torch.nn.functional.linear, torch.bmm, torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__,
      torch.nn.functional.conv2d, torch.Tensor.bmm}
ACT = {torch.nn.functional.layer_norm, torch.nn.functional.relu, torch.tanh, torch.nn.functional.gelu, torch.sigmoid}
EW = ACT | {torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.mul, torch.Tensor.__mul__,
            torch.Tensor.__rmul__, torch.Tensor.sub, torch.Tensor.__sub__, torch.Tensor.pow, torch.Tensor.__truediv__,
            torch.Tensor.div, torch.cat, torch.nn.functional.embedding, torch.sqrt, torch.Tensor.sqrt}
BF, HF = torch.bfloat16, torch.float16


class Policy(TorchFunctionMode):
    def __init__(self, ops=None, store="none", store_dt=None):
        """ops: {region: dtype} ('*' default); store: 'none' | 'operand' | 'all'; store_dt: {region: dtype} (default: ops)."""
        super().__init__()
        self.ops, self.store, self.store_dt = ops or {}, store, store_dt

    def _dt(self, table, region):
        for r in reversed(REGION):
            if r in table:
                return table[r]
        return table.get("*")

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        region = REGION[-1]
        if region in ("router", "block", "loss") or not self.ops:
            return func(*args, **kwargs)
        if func in MM:
            dt = self._dt(self.ops, region)
            n = 3 if func is torch.nn.functional.conv2d else 2
            args = tuple(q(a, dt) if i < n else a for i, a in enumerate(args))
        out = func(*args, **kwargs)
        if self.store == "none" or not torch.is_tensor(out) or not out.is_floating_point():
            return out
        rounds = (func in MM or func in EW) if self.store == "all" else (func in MM or func in ACT)
        if not rounds:
            return out
        if func in MM and func is not torch.nn.functional.linear and out.shape[-1] in (128, 197, 50, 64):
            return out  # attention scores stay fp32 inside the fused cores
        if out.dim() >= 2 and out.shape[-1] == 1:
            return out
        return q(out, self._dt(self.store_dt or self.ops, region))


POLICIES = {
    "fp32": dict(),
    "bf16 ops, all storage bf16 (round-2 bf16 path)": dict(ops={"*": BF}, store="all"),
    "fp16 ops, all storage fp16 (round-2 fp16 path)": dict(ops={"*": HF}, store="all"),
    "bf16 ops, operand-only storage (fp32 residual streams)": dict(ops={"*": BF}, store="operand"),
    "fp16 ops, operand-only storage (fp32 residual streams)": dict(ops={"*": HF}, store="operand"),
    "bf16 ops only": dict(ops={"*": BF}),
    "fp16 ops only": dict(ops={"*": HF}),
    "enc bf16 / rest fp16, all storage": dict(ops={"*": HF, "enc": BF}, store="all"),
    "enc bf16 / rest fp16, operand-only storage": dict(ops={"*": HF, "enc": BF}, store="operand"),
    "enc fp16 / rest bf16, operand-only storage": dict(ops={"*": BF, "enc": HF}, store="operand"),
}


def main():
    torch.set_num_threads(8)
    env = lambda k, d: int(os.environ.get(k, d))
    layers, L, IMG, seed = env("LAYERS", "12"), env("SEQ", "128"), env("IMG", "224"), env("SEED", "2023")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    torch.manual_seed(seed)
    tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=layers, image_size=IMG, patch_size=16)
    model = M.UnimoModelF(default_args(), vc, tc)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    cfg = O.OracleConfig(text_layers=layers, vision_layers=layers, image_size=IMG, patch_size=16)
    batch = O.synthetic_batch(cfg, B, L, seed=9, ragged=False)

    def run(dtype, pol, round_params=None, cot=None):
        def prep(k, v):
            if not (v.is_floating_point() and "running_" not in k):
                return v.clone()
            v = v.detach().clone()
            if round_params is not None and v.dim() >= 2:  # weight matrices only (what an MFMA reads); biases / norms exact
                v = v.to(round_params)
            return v.to(dtype).requires_grad_(not O.is_dead_param(k))
        sd = {k: prep(k, v) for k, v in sd0.items()}
        ids, mask, tt, labels, images = batch
        t0 = time.time()
        with Policy(**pol):
            loss, logits, aux = O.forward(sd, cfg, ids, mask, tt, labels, images.to(dtype), train=True)
            if cot is not None:  # fixed cotangents at Block's two inputs instead of the loss (tests/test_gpu_bench_shapes.py)
                tp = O.cls_pool(sd, "model.text_pool", aux["emb_text"])
                vp = O.cls_pool(sd, "model.vision_pool", aux["emb_image"])
                loss = (tp * cot[0].to(dtype)).sum() + (vp * cot[1].to(dtype)).sum() + aux["js_loss"]
            loss.backward()
        grads = {k: v.grad.detach().double() for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
        return float(loss), logits.detach().double(), grads, time.time() - t0

    l64, lg64, g64, dt = run(torch.float64, {})
    print(f"fp64 truth: loss {l64:.6f} max|logit| {float(lg64.abs().max()):.3f} ({dt:.0f} s)", flush=True)
    def report(name, l, lg, g, dt, l64=l64, lg64=lg64, g64=g64):
        dot = sum(float((g[k] * g64[k]).sum()) for k in g64)
        gg = sum(float(g[k].pow(2).sum()) for k in g64)
        rr = sum(float(g64[k].pow(2).sum()) for k in g64)
        print(f"{name:58s} |dlogits| {float((lg - lg64).abs().max()):.2e} |dloss| {abs(l - l64):.2e}  grad cos {dot / (gg * rr) ** 0.5:.4f} "
              f"|g|/|ref| {(gg / rr) ** 0.5:.3f} ({dt:.0f} s)", flush=True)

    # Conditioning of the problem itself: fp64 ARITHMETIC throughout, only the weight matrices rounded once to a 16-bit type
    # (a relative perturbation of 2^-12 / 2^-9 of the data; no accumulated rounding, no kernel of ours).
    if not os.environ.get("POLICY"):
        for nm, rp in (("fp64 arithmetic, weight matrices rounded to fp16 once", HF), ("fp64 arithmetic, weight matrices rounded to bf16 once", BF)):
            report(nm, *run(torch.float64, {}, round_params=rp))
        gen = torch.Generator().manual_seed(71)
        cot = (torch.randn(B, 768, generator=gen, dtype=torch.float64) * 0.05, torch.randn(B, 768, generator=gen, dtype=torch.float64) * 0.05)
        c64 = run(torch.float64, {}, cot=cot)
        print("fixed cotangents at Block's inputs (objective <text_pooled,c1> + <vision_pooled,c2> + js):")
        for nm, rp in (("  fp64 arithmetic, weight matrices rounded to fp16 once", HF), ("  fp64 arithmetic, weight matrices rounded to bf16 once", BF)):
            report(nm, *run(torch.float64, {}, round_params=rp, cot=cot), l64=c64[0], lg64=c64[1], g64=c64[2])
        for nm in ("fp16 ops only", "fp16 ops, all storage fp16 (round-2 fp16 path)", "bf16 ops only"):
            report("  " + nm, *run(torch.float32, POLICIES[nm], cot=cot), l64=c64[0], lg64=c64[1], g64=c64[2])
    only = os.environ.get("POLICY")
    for name, pol in POLICIES.items():
        if only and only not in name:
            continue
        l, lg, g, dt = run(torch.float32, pol)
        dot = sum(float((g[k] * g64[k]).sum()) for k in g64)
        gg = sum(float(g[k].pow(2).sum()) for k in g64)
        rr = sum(float(g64[k].pow(2).sum()) for k in g64)
        enc = [k for k in g64 if ".encoder." in k]
        dote = sum(float((g[k] * g64[k]).sum()) for k in enc)
        gge = sum(float(g[k].pow(2).sum()) for k in enc)
        rre = sum(float(g64[k].pow(2).sum()) for k in enc)
        print(f"{name:58s} |dlogits| {float((lg - lg64).abs().max()):.2e} |dloss| {abs(l - l64):.2e}  grad cos {dot / (gg * rr) ** 0.5:.4f} "
              f"|g|/|ref| {(gg / rr) ** 0.5:.3f}  encoders cos {dote / (gge * rre) ** 0.5:.4f} ({dt:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
