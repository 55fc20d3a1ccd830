"""Per-shape GEMM timings INSIDE the training step (one stream): the library's launch timer (d2r_gemm_timer: HIP events around every
d2r_gemm / grouped launch, also those issued inside the composite calls) grouped by (kernel family, flops of the launch).

  python tests/probes/gemm_shapes_in_step.py [--dtype fp16]
"""
import argparse
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from d2r_amd import _lib  # noqa: E402
from d2r_amd import modules as M  # noqa: E402
from d2r_amd.config import TextConfig, VisionConfig, default_args  # noqa: E402
from d2r_amd.params import FusedAdamW, ParamStore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--streams", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
torch.manual_seed(2023)
model = M.UnimoModelF(default_args(DR_step=3, num_cells=6), VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16),
                      TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), num_classes=3)
model.to(dev).set_compute_dtype(dtype).train()
model.model.use_streams = bool(a.streams)
store = ParamStore(model, dtype)
opt = FusedAdamW(store, lr=3e-5)
if dtype == torch.float16:
    opt.enable_loss_scaling()
batch = bench.synthetic_batch(32, 128, 224, dev, seed=0)


def step():
    loss, _ = model(*batch)
    opt.backward(loss)
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
NSTEP = 3
_lib.call("d2r_gemm_timer", 1)
for _ in range(NSTEP):
    step()
torch.cuda.synchronize()
_lib.call("d2r_gemm_timer", 0)
lib = _lib.load()
cap = 1 << 15
fam, fl, by, ms = (C.c_int * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), (C.c_float * cap)()
n = lib.d2r_gemm_timer_read(fam, fl, by, ms, cap)
VARIANT = {0: "tiles", 3: "128x128w8", 20: "128x128", 21: "tiles64", 22: "batched16", 30: "skinny", 31: "skinny", 8: "256x256", 28: "256x256"}
g = collections.defaultdict(list)
for i in range(n):
    if fam[i] >= 10000:
        continue
    var, base = fam[i] // 100, fam[i] % 100
    name = "%s_%s%s_%s" % (("f32", "bf16", "f16")[base // 8], ("NT", "NN", "TN")[(base % 8) // 2], "_grouped" if base & 1 else "", VARIANT.get(var, str(var)))
    g[(name, round(fl[i] / 1e6))].append(ms[i])
rows = []
for (name, mf), ts in g.items():
    ts.sort()
    med = ts[len(ts) // 2]
    rows.append((sum(ts) / NSTEP, name, mf, len(ts) / NSTEP, med))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("GEMM launches in one step (one stream=%s), bracketed by events (each bracket adds ~2-4 us): %.2f ms per step" % (not a.streams, tot))
print("%-28s %10s %7s %9s %8s %8s" % ("kernel", "GFLOP", "calls", "median us", "TFLOP/s", "ms/step"))
for t, name, mf, calls, med in rows[:60]:
    print("%-28s %10.2f %7.1f %9.1f %8.0f %8.3f" % (name, mf / 1e3, calls, med * 1e3, mf / 1e6 / (med / 1e3) if med > 0 else 0, t))
