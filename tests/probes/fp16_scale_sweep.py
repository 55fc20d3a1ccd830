"""fp16 gradient direction vs the oracle as a function of the loss scale (4-cell / 7-class seeded case and the default-init case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import d2r_oracle as O
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore

gpu = torch.device("cuda:0")


def case(seeded):
    if seeded:
        cfg = O.OracleConfig(text_layers=2, vision_layers=2, image_size=64, patch_size=32, DR_step=4, num_cells=4, num_classes=7)
        sd = O.seeded_state_dict(cfg, seed=21, router_bias="normal")
        mk = lambda: M.UnimoModelF(default_args(DR_step=4, num_cells=4), VisionConfig(num_hidden_layers=2, image_size=64, patch_size=32),
                                   TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), num_classes=7)
        batch = O.synthetic_batch(cfg, 3, 10, seed=4)
    else:
        torch.manual_seed(2023)
        cfg = O.OracleConfig(text_layers=2, vision_layers=2, image_size=96, patch_size=32)
        mk = lambda: M.UnimoModelF(default_args(), VisionConfig(num_hidden_layers=2, image_size=96, patch_size=32),
                                   TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
        sd = {k: v.detach().clone() for k, v in mk().state_dict().items()}
        batch = O.synthetic_batch(cfg, 4, 24, seed=6)
    osd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    lo, _, _ = O.forward(osd, cfg, batch[0], batch[1], batch[2], batch[3], batch[4].double(), train=True)
    lo.backward()
    for dtype, scales in ((torch.bfloat16, [1.0]), (torch.float16, [1.0, 1024.0, 2.0 ** 14, 2.0 ** 16, 2.0 ** 20])):
        for s in scales:
            model = mk()
            model.load_state_dict(sd, strict=True)
            model.to(gpu).set_compute_dtype(dtype).train()
            store = ParamStore(model, dtype)
            loss, _ = model(*[t.to(gpu) for t in batch])
            (loss * s).backward()
            torch.cuda.synchronize()
            dot = ng = nr = 0.0
            parts = {}
            for name, p in model.named_parameters():
                ref = osd[name].grad
                if ref is None:
                    continue
                got = p.grad.detach().double().cpu() / s
                d, g, r = float((got * ref).sum()), float(got.pow(2).sum()), float(ref.pow(2).sum())
                dot, ng, nr = dot + d, ng + g, nr + r
                key = ".".join(name.split(".")[:3]) if name.startswith("model.") else name.split(".")[0]
                a = parts.setdefault(key, [0.0, 0.0, 0.0]); a[0] += d; a[1] += g; a[2] += r
            cos = dot / max((ng * nr) ** 0.5, 1e-300)
            worst = sorted((d_ / max((g_ * r_) ** 0.5, 1e-300), k) for k, (d_, g_, r_) in parts.items())[:4]
            print(f"{'seeded-4cell' if seeded else 'default-init'} {str(dtype)[6:]:9s} scale {s:9.0f} finite {bool(torch.isfinite(store.flat_g).all())} "
                  f"cos {cos:.4f} |g|/|ref| {(ng / nr) ** 0.5:.3f} worst parts {[(round(c, 3), k) for c, k in worst]}", flush=True)
            del model, store


case(True)
case(False)
