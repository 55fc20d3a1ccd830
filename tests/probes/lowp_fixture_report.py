"""Measures the 16-bit compute dtypes against the reference's golden fixtures (GPU): logits / loss error and the direction
of the stored full gradients, per fixture, for bf16 and fp16.  Prints one line per (fixture, dtype)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from conftest import load_golden
import test_gpu_model as T
from d2r_amd.params import ParamStore

gpu = torch.device("cuda:0")
SCALE = float(os.environ.get("LOSS_SCALE", "1024"))
for case in T._model_cases():
    g = load_golden(case.name)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        model, sd, cfg = T._build_model(case, dtype, gpu)
        store = ParamStore(model, dtype)
        batch = [T._t(g[k], gpu) for k in ("input_ids", "attention_mask", "token_type_ids", "labels", "images")]
        loss, logits = model(*batch)
        s = SCALE if dtype == torch.float16 else 1.0
        (loss * s).backward()
        torch.cuda.synchronize()
        store.flat_g.mul_(1.0 / s)
        e_logit = T._err(logits, torch.from_numpy(g["logits"]))
        e_loss = T._err(loss, torch.from_numpy(np.asarray(g["loss"])))
        e_js = T._err(model.last_aux["js_loss"], torch.from_numpy(np.asarray(g["js_loss"])))
        params = dict(model.named_parameters())
        dots, rels = [], []
        for key in [k for k in g if k.startswith("grad/")]:
            ref = torch.from_numpy(g[key]).double().flatten()
            if float(ref.abs().max()) == 0.0:
                continue
            got = params[key[5:]].grad.detach().double().cpu().flatten()
            dots.append((float(got @ ref), float(got.norm()) ** 2, float(ref.norm()) ** 2))
        d = np.asarray(dots)
        cos = d[:, 0].sum() / np.sqrt(d[:, 1].sum() * d[:, 2].sum())
        names, norms = [str(k) for k in g["grad_names"]], np.asarray(g["grad_norms"], dtype=np.float64)
        pos = norms[norms > 0]
        floor = 0.05 * float(np.median(pos)) if len(pos) else 1.0
        for k, nr in zip(names, norms):
            gr = params[k].grad
            rels.append(abs(float(gr.double().norm()) - nr) / (nr + floor))
        finite = bool(torch.isfinite(store.flat_g).all())
        print(f"{case.name:14s} {str(dtype)[6:]:9s} logits {e_logit:.2e} loss {e_loss:.2e} js {e_js:.2e} grad cos {cos:.4f} "
              f"norm-err median {np.median(rels):.2e} p90 {np.quantile(rels, 0.9):.2e} finite {finite}", flush=True)
        del model, store
