"""Where does the bf16 logit error come from?  CPU-only probe (test infrastructure: imports the oracle).

Runs the pinned oracle under a torch-function mode that emulates precision policies of the HIP path:
  ops   : matmul-class operands (linear / bmm / matmul / conv2d inputs) are rounded to bf16 (MFMA operands)
  store : outputs of every op are rounded to bf16 (activation storage) inside the named regions
and prints |logits - fp64 oracle| per fixture and policy.  Used to decide what the "amp" mode keeps in fp32.

    python tests/probes/bf16_budget.py [case ...]
"""
import os
import sys

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import d2r_oracle as O  # noqa: E402
from oracle import golden_cases as GC  # noqa: E402

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
# routing_layer's aggregation runs inside interaction_module; the cells look these up through CELLS
for cname in list(O.CELLS):
    O.CELLS[cname] = getattr(O, "cell_" + cname)


LOWP = torch.float16 if os.environ.get("LOWP", "bf16") == "fp16" else torch.bfloat16


def q(x):
    return x.to(LOWP).to(x.dtype) if torch.is_tensor(x) and x.is_floating_point() else x


MM = {torch.nn.functional.linear, torch.bmm, torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__,
      torch.nn.functional.conv2d, torch.Tensor.bmm}
EW = {torch.nn.functional.layer_norm, torch.nn.functional.relu, torch.tanh, torch.nn.functional.gelu, torch.sigmoid,
      torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.mul, torch.Tensor.__mul__,
      torch.Tensor.__rmul__, torch.Tensor.sub, torch.Tensor.__sub__, torch.Tensor.pow, torch.Tensor.__truediv__,
      torch.Tensor.div, torch.cat, torch.nn.functional.embedding, torch.sqrt, torch.Tensor.sqrt}


class Policy(TorchFunctionMode):
    def __init__(self, ops=True, store=(), exact=()):
        super().__init__()
        self.ops, self.store, self.exact = ops, set(store), set(exact)

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        region = REGION[-1]
        if region in ("router", "block", "loss") or self.exact & set(REGION):  # kept fp32 end to end
            return func(*args, **kwargs)
        if func in MM and self.ops:
            n = 3 if func is torch.nn.functional.conv2d else 2
            args = tuple(q(a) if i < n else a for i, a in enumerate(args))
        out = func(*args, **kwargs)
        if region in self.store and (func in MM or func in EW) and torch.is_tensor(out) and out.is_floating_point():
            if func in MM and func is not torch.nn.functional.linear and out.shape[-1] < 40:
                return out  # attention scores stay fp32 inside the fused cores
            if out.dim() >= 2 and out.shape[-1] in (1,):
                return out
            return q(out)
        return out


POLICIES = {
    "fp32 (no rounding)": dict(ops=False, store=()),
    "ops only (fp32 storage everywhere)": dict(ops=True, store=()),
    "ops + bf16 storage in encoders": dict(ops=True, store=("enc",)),
    "ops + bf16 storage in routing": dict(ops=True, store=("routing", "saf")),
    "ops + bf16 storage everywhere (today)": dict(ops=True, store=("enc", "routing", "saf", "other")),
    "ops only, xalign exact": dict(ops=True, store=(), exact=("xalign",)),
    "ops only, encoders exact": dict(ops=True, store=(), exact=("enc",)),
    "ops only, routing exact": dict(ops=True, store=(), exact=("routing",)),
    "ops only, poolers exact": dict(ops=True, store=(), exact=("pool",)),
    "today, xalign exact": dict(ops=True, store=("enc", "routing", "saf", "other"), exact=("xalign",)),
    "today, poolers+xalign exact": dict(ops=True, store=("enc", "routing", "saf", "other"), exact=("xalign", "pool")),
    "today, routing exact": dict(ops=True, store=("enc", "routing", "saf", "other"), exact=("routing",)),
    "today, encoders exact": dict(ops=True, store=("enc", "routing", "saf", "other"), exact=("enc",)),
}


def run_case(case, pol):
    cfg = case.cfg()
    sd = O.seeded_state_dict(cfg, seed=case.seed, router_bias=case.router_bias)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", case.name + ".npz"), allow_pickle=False))
    batch = [torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "token_type_ids", "labels", "images")]
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad(), Policy(**pol):
        loss, logits, aux = O.forward(sd32, cfg, *batch[:4], batch[4].float(), train=case.train)
    ref = torch.from_numpy(g["logits"]).double()
    return float((logits.double() - ref).abs().max()), float(abs(float(loss) - float(g["loss"]))), float(ref.abs().max())


def run_default_init(pol):
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    torch.manual_seed(2023)
    layers, B, L = 4, 4, 32
    tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=layers, image_size=96, patch_size=32)
    model = M.UnimoModelF(default_args(), vc, tc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.OracleConfig(text_layers=layers, vision_layers=layers, image_size=96, patch_size=32)
    ids, mask, tt, labels, images = O.synthetic_batch(cfg, B, L, seed=5)
    with torch.no_grad():
        lo, logits_o, _ = O.forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg,
                                    ids, mask, tt, labels, images.double(), train=False)
        with Policy(**pol):
            l2, logits, _ = O.forward(sd, cfg, ids, mask, tt, labels, images, train=False)
    return float((logits.double() - logits_o).abs().max()), abs(float(l2) - float(lo)), float(logits_o.abs().max())


if __name__ == "__main__":
    torch.set_num_threads(8)
    want = sys.argv[1:]
    only = os.environ.get("POLICY")
    for name, pol in POLICIES.items():
        if only and only not in name:
            continue
        line = f"{name:45s}"
        e, el, s = run_default_init(pol)
        line += f" default-init {e:.2e}/{el:.2e}"
        for case in GC.MODEL_CASES:
            if want and case.name not in want:
                continue
            e, el, s = run_case(case, pol)
            line += f" | {case.name} {e:.2e}/{el:.2e} (scale {s:.2f})"
        print(line, flush=True)
