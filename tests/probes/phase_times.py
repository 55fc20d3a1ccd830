"""GPU time per phase of the UNTRACED training step at the benchmark shape (events on the launching stream at the fork / join
points): encoders forward | routing modules forward | head forward + backward | routing modules backward | encoders backward
(+ weight-gradient flush) | optimiser.  (rocprofv3's kernel trace slows the host enough to serialise the two modules.)
    python tests/probes/phase_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d2r_amd
from d2r_amd import functional as F, modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, ParamStore
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synthetic_batch

d2r_amd.configure_runtime()
dev = torch.device("cuda:0")
dtype = torch.float16 if os.environ.get("DT", "fp16") == "fp16" else torch.bfloat16
torch.manual_seed(2023)
model = M.UnimoModelF(default_args(DR_step=3), VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16),
                      TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
model.to(dev).set_compute_dtype(dtype).train()
store = ParamStore(model, dtype)
opt = FusedAdamW(store, lr=3e-5)
if dtype == torch.float16:
    opt.enable_loss_scaling()
batch = synthetic_batch(int(os.environ.get("BATCH", "32")), 128, 224, dev, seed=0)
marks = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks.append((tag, e))


_fw, _bw = F._StreamJoin.forward, F._StreamJoin.backward
branch_marks = []


def _join_fw(ctx, a, b):
    # when does each branch stream finish its encoder (+ extra self layer)?  (the launching stream waits for both)
    st = model.model._streams
    if st is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in st]
        for e, s_ in zip(ev, st):
            e.record(s_)
        branch_marks.append(ev)
    mark("encoders forward")
    return _fw(ctx, a, b)


F._StreamJoin.forward = staticmethod(_join_fw)
F._StreamJoin.backward = staticmethod(lambda ctx, ga, gb: (mark("modules backward"), _bw(ctx, ga, gb))[1])
_lin = F.lincomb
def lincomb(*a, **k):
    mark("modules forward")
    return _lin(*a, **k)
F.lincomb = lincomb
M.F.lincomb = lincomb
_hb = F._Head.backward
F._Head.backward = staticmethod(lambda ctx, *g: (_hb(ctx, *g), mark("head forward + backward"))[0])

rows = []
branch_rows = []
for it in range(10):
    marks.clear()
    branch_marks.clear()
    mark("start")
    loss, _ = model(*batch)
    (opt.backward(loss) if os.environ.get("FUSED_ROOT", "1") != "0" else opt.scale_loss(loss).backward())
    F.wgrad_join()
    mark("encoders backward + weight-gradient flush")
    opt.step()
    opt.zero_grad()
    mark("optimiser + zero")
    torch.cuda.synchronize()
    if it >= 4 and branch_marks:
        branch_rows.append([marks[0][1].elapsed_time(e) for e in branch_marks[0]])
    if it >= 4:
        rows.append([(marks[i][0], marks[i - 1][1].elapsed_time(marks[i][1])) for i in range(1, len(marks))])
for i, (tag, _) in enumerate(rows[0]):
    v = sorted(r[i][1] for r in rows)
    print(f"{tag:45s} median {v[len(v) // 2]:6.2f} ms  (min {v[0]:.2f}, max {v[-1]:.2f})")
print(f"{'step':45s} median {sorted(sum(x[1] for x in r) for r in rows)[len(rows) // 2]:6.2f} ms")
if branch_rows:
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"branch streams reach the barrier behind the encoders after {med([r[0] for r in branch_rows]):.2f} ms (text) / "
          f"{med([r[1] for r in branch_rows]):.2f} ms (vision) of the step")
