"""Do the two routing modules (text on one stream, image on the other) overlap on the GPU, and if not, who is late: the host or the GPU?

For a few steady-state steps of the bench workload every d2r_interaction_{fwd,bwd} / d2r_head_{fwd,bwd} call is bracketed by
(host clock before / after, HIP event before / after on the call's stream).  The events are placed on one time line with the host
clock through a reference event recorded right after a device synchronisation.  Printed per call: when the host issued it, when the
GPU started and finished it.  GPU start far behind host issue with the other module still running = a GPU-side serialisation;
GPU start == host issue = the host is the late one.

  python tests/probes/module_overlap_probe.py [--dtype fp16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from d2r_amd import _lib  # noqa: E402
from d2r_amd import modules as M  # noqa: E402
from d2r_amd.config import TextConfig, VisionConfig, default_args  # noqa: E402
from d2r_amd.params import FusedAdamW, ParamStore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
torch.manual_seed(2023)
model = M.UnimoModelF(default_args(DR_step=3, num_cells=6), VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16), TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), num_classes=3)
model.to(dev).set_compute_dtype(dtype).train()
store = ParamStore(model, dtype)
opt = FusedAdamW(store, lr=3e-5)
if dtype == torch.float16:
    opt.enable_loss_scaling()
batch = bench.synthetic_batch(32, 128, 224, dev, seed=0)

rec = None
orig = _lib.call
WATCH = ("d2r_interaction_fwd", "d2r_interaction_bwd", "d2r_head_fwd", "d2r_head_bwd", "d2r_adamw_step")


def call(name, *args, meta=None):
    if rec is None or name not in WATCH:
        return orig(name, *args, meta=meta)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    e0.record(st)
    r = orig(name, *args, meta=meta)
    e1.record(st)
    rec.append((name, st.cuda_stream, h0, time.perf_counter(), e0, e1))
    return r


_lib.call = call
import d2r_amd.functional as F  # noqa: E402
import d2r_amd.params as P  # noqa: E402
F._lib.call = call
P._lib.call = call


def step():
    loss, _ = model(*batch)
    opt.backward(loss)
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(4):
    step()
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
torch.cuda.synchronize()
h_ref = time.perf_counter()
rec = []
marks = []
for i in range(a.steps):
    marks.append(time.perf_counter())
    step()
torch.cuda.synchronize()
h_end = time.perf_counter()
print("%d steps: %.2f ms/step (with the brackets)" % (a.steps, (h_end - h_ref) / a.steps * 1e3))
streams = {}
for name, st, h0, h1, e0, e1 in rec:
    streams.setdefault(st, "s%d" % len(streams))
last = [r for r in rec if r[2] >= marks[-1]]
t_step = (marks[-1] - h_ref) * 1e3
print("last step: host begins issuing at %.2f ms" % t_step)
print("%-22s %-3s %10s %10s | %10s %10s %8s" % ("call", "st", "host issue", "host done", "gpu start", "gpu end", "gpu ms"))
for name, st, h0, h1, e0, e1 in last:
    g0, g1 = ref.elapsed_time(e0), ref.elapsed_time(e1)
    print("%-22s %-3s %10.3f %10.3f | %10.3f %10.3f %8.3f" % (name, streams[st], (h0 - h_ref) * 1e3 - t_step, (h1 - h_ref) * 1e3 - t_step, g0 - t_step, g1 - t_step, g1 - g0))
