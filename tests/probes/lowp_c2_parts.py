"""C2-shape (12+12 layers, L=128, 197 image tokens, batch 8) default-init gradients of the 16-bit paths against the fp32 HIP path
(itself within 1e-6 of the oracle): cosine per part of the model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import d2r_oracle as O
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore

gpu = torch.device("cuda:0")
env = lambda k, d: int(os.environ.get(k, d))
layers, B, L, IMG = env("LAYERS", "12"), env("BATCH", "8"), env("SEQ", "128"), env("IMG", "224")
DR, CELLS, CLASSES = env("DR", "3"), env("CELLS", "6"), env("CLASSES", "3")
torch.manual_seed(env("SEED", "2023"))
tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=layers, image_size=IMG, patch_size=16)
mk = lambda: M.UnimoModelF(default_args(DR_step=DR, num_cells=CELLS), vc, tc, num_classes=CLASSES)
sd = {k: v.detach().clone() for k, v in mk().state_dict().items()}
cfg = O.OracleConfig(text_layers=layers, vision_layers=layers, image_size=IMG, patch_size=16, DR_step=DR, num_cells=CELLS, num_classes=CLASSES)
batch = [t.to(gpu) for t in O.synthetic_batch(cfg, B, L, seed=9, ragged=False)]
grads = {}
for dtype in (torch.float32, torch.float16, torch.bfloat16):
    model = mk()
    model.load_state_dict(sd, strict=True)
    model.to(gpu).set_compute_dtype(dtype).train()
    store = ParamStore(model, dtype)
    loss, logits = model(*batch)
    s = 1024.0 if dtype == torch.float16 else 1.0
    (loss * s).backward()
    torch.cuda.synchronize()
    grads[dtype] = {n: (p.grad.detach().double().cpu() / s) for n, p in model.named_parameters() if p.grad is not None}
    print(str(dtype), "loss", float(loss), "logits", logits[0].tolist(), flush=True)
    del model, store
ref = grads[torch.float32]
for dtype in (torch.float16, torch.bfloat16):
    parts, tot = {}, [0.0, 0.0, 0.0]
    for n, r in ref.items():
        g = grads[dtype][n]
        d, gg, rr = float((g * r).sum()), float(g.pow(2).sum()), float(r.pow(2).sum())
        key = ".".join(n.split(".")[:int(os.environ.get("DEPTH", "4"))]) if n.startswith("model.") else n.split(".")[0]
        a = parts.setdefault(key, [0.0, 0.0, 0.0]); a[0] += d; a[1] += gg; a[2] += rr
        tot[0] += d; tot[1] += gg; tot[2] += rr
    print(f"== {dtype}: global cos {tot[0] / (tot[1] * tot[2]) ** 0.5:.4f} |g|/|ref| {(tot[1] / tot[2]) ** 0.5:.3f}")
    for k, (d, gg, rr) in parts.items():
        print(f"   {k:60s} cos {d / max((gg * rr) ** 0.5, 1e-300):.4f} |g|/|ref| {(gg / max(rr, 1e-300)) ** 0.5:.3f} |ref| {rr ** 0.5:.2e}")

if os.environ.get('NO_CONTROL'):
    sys.exit(0)
# Control experiment: the fp32 path with ONLY the two pooled vectors entering Block rounded to fp16 / bf16 (everything else fp32)
for rd in (torch.float16, torch.bfloat16):
    model = mk()
    model.load_state_dict(sd, strict=True)
    model.to(gpu).set_compute_dtype(torch.float32).train()
    store = ParamStore(model, torch.float32)
    orig = M.Block.forward

    def rounded(self, x, _o=orig, _rd=rd):
        return _o(self, [t + (t.to(_rd).float() - t).detach() for t in x])  # value rounded, gradient passed straight through

    M.Block.forward = rounded
    try:
        loss, logits = model(*batch)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        M.Block.forward = orig
    tot = [0.0, 0.0, 0.0]
    bf = [0.0, 0.0, 0.0]
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g, r = p.grad.detach().double().cpu(), ref[n]
        d, gg, rr = float((g * r).sum()), float(g.pow(2).sum()), float(r.pow(2).sum())
        tot[0] += d; tot[1] += gg; tot[2] += rr
        if "block_fusion" in n:
            bf[0] += d; bf[1] += gg; bf[2] += rr
    print(f"== fp32 path, Block inputs rounded to {rd}: global cos {tot[0] / (tot[1] * tot[2]) ** 0.5:.4f}; block_fusion parameters cos "
          f"{bf[0] / (bf[1] * bf[2]) ** 0.5:.4f} |g|/|ref| {(bf[1] / bf[2]) ** 0.5:.3f}")
    del model, store
