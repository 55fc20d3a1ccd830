"""Gradient cosine of the fp16 path against the fp64 oracle at the reference's default init (the computation of
tests/test_gpu_model.py::test_default_init_gradients_vs_oracle) over several batch seeds and init seeds, for the library given as argv[1]
(default: the production build).  Question: is a difference between two builds of the epilogue arithmetic larger than the spread over seeds?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from d2r_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore
from oracle import d2r_oracle as O
gpu = torch.device("cuda", 0)
dtype = torch.float16
layers, B, L = 2, 4, 24
out = []
for init_seed, batch_seed in ((2023, 6), (2023, 7), (2023, 8), (2023, 9), (7, 6), (11, 6), (13, 7), (17, 8)):
    torch.manual_seed(init_seed)
    tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=layers, image_size=96, patch_size=32)
    model = M.UnimoModelF(default_args(), vc, tc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.OracleConfig(text_layers=layers, vision_layers=layers, image_size=96, patch_size=32)
    ids, mask, tt, labels, images = O.synthetic_batch(cfg, B, L, seed=batch_seed)
    osd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    lo, _, _ = O.forward(osd, cfg, ids, mask, tt, labels, images.double(), train=True)
    lo.backward()
    model.to(gpu).set_compute_dtype(dtype).train()
    ParamStore(model, dtype)
    loss, _ = model(ids.to(gpu), mask.to(gpu), tt.to(gpu), labels.to(gpu), images.to(gpu))
    (loss * 1024.0).backward()
    torch.cuda.synchronize()
    dot = ng = nr = 0.0
    side = [0.0, 0.0, 0.0]  # parameters that do not sit behind Block's signed square root (they see the JS loss only)
    for name, p in model.named_parameters():
        ref = osd[name].grad
        if ref is None or p.grad is None:
            continue
        got = p.grad.detach().double().cpu() / 1024.0
        d_, g_, r_ = float((got * ref).sum()), float(got.pow(2).sum()), float(ref.pow(2).sum())
        dot += d_; ng += g_; nr += r_
        if name.startswith(("model.self_text", "model.self_vision", "model.text_cls_pool", "model.vision_cls_pool")):
            side = [side[0] + d_, side[1] + g_, side[2] + r_]
    cos = dot / (ng * nr) ** 0.5
    out.append(cos)
    print("init %4d batch %2d: loss err %.2e  cos %.4f  |g|/|ref| %.3f  side cos %.5f" % (init_seed, batch_seed, abs(float(loss) - float(lo)), cos, (ng / nr) ** 0.5,
          side[0] / max((side[1] * side[2]) ** 0.5, 1e-300)), flush=True)
print("min %.4f median %.4f max %.4f" % (min(out), sorted(out)[len(out) // 2], max(out)))
