"""CPU experiment (oracle + tests/lowp_emulation.py, no GPU): how does the cosine of the 16-bit gradient against the fp64 gradient DISTRIBUTE
over (init, batch) seeds under emulated precision policies?
  operands   16-bit MFMA operands only, every stored activation fp32 (the floor of any 16-bit-operand implementation)
  store-all  + 16-bit storage of every activation (what the HIP path does)
  fp32-rout  + 16-bit storage in the encoders only: the routing modules keep an fp32 activation stream (VERDICT r3 item 7's proposal)
Same tiny model as tests/test_gpu_model.py::test_default_init_gradients_vs_oracle (2 + 2 layers, batch 4, L 24), default init."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from oracle import d2r_oracle as O
from lowp_emulation import LowpPolicy, STORE_ALL

torch.set_num_threads(8)
lowp = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
layers, B, L = 2, 4, 24
POL = {"operands": (), "store-all": STORE_ALL, "fp32-rout": ("enc", "other")}
res = {k: [] for k in POL}
for init_seed, batch_seed in ((2023, 6), (2023, 7), (2023, 8), (2023, 9), (7, 6), (11, 6), (13, 7), (17, 8)):
    torch.manual_seed(init_seed)
    tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=layers, image_size=96, patch_size=32)
    model = M.UnimoModelF(default_args(), vc, tc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.OracleConfig(text_layers=layers, vision_layers=layers, image_size=96, patch_size=32)
    ids, mask, tt, labels, images = O.synthetic_batch(cfg, B, L, seed=batch_seed)

    def grads(dtype, policy):
        osd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
        if policy is None:
            lo, _, _ = O.forward(osd, cfg, ids, mask, tt, labels, images.to(dtype), train=True)
        else:
            with LowpPolicy(lowp, policy):
                lo, _, _ = O.forward(osd, cfg, ids, mask, tt, labels, images.to(dtype), train=True)
        lo.backward()
        return float(lo), {k: v.grad.double() for k, v in osd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}

    l_ref, g_ref = grads(torch.float64, None)
    line = "init %4d batch %2d:" % (init_seed, batch_seed)
    for name, store in POL.items():
        l, g = grads(torch.float32, store)
        dot = sum(float((g[k] * g_ref[k]).sum()) for k in g_ref if k in g)
        ng = sum(float(g[k].pow(2).sum()) for k in g_ref if k in g)
        nr = sum(float(g_ref[k].pow(2).sum()) for k in g_ref if k in g)
        c = dot / (ng * nr) ** 0.5
        res[name].append(c)
        line += "  %s loss err %.1e cos %.4f" % (name, abs(l - l_ref), c)
    print(line, flush=True)
for name, v in res.items():
    s = sorted(v)
    print("%-10s min %.4f  median %.4f  max %.4f" % (name, s[0], 0.5 * (s[3] + s[4]), s[-1]))
