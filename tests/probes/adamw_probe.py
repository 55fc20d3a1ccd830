"""AdamW kernel alone on the step's 353.6 M parameters: default vs non-temporal streams vs grid caps (d2r_adamw_probe_mode)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from d2r_amd import _lib
n = 353_600_000
dev = torch.device("cuda", 0)
w, g, m, v = (torch.randn(n, device=dev) * s for s in (0.02, 1e-3, 1e-4, 1e-6))
v.abs_()
lp = torch.empty(n, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
flag = torch.zeros(1, dtype=torch.int32, device=dev)
def run(nt, blocks, reps=8):
    _lib.load().d2r_adamw_probe_mode(nt, blocks)
    ts = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("d2r_adamw_step", w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lp.data_ptr(), 2, n, 3e-5, 0.9, 0.999, 1e-8, 0.01, i + 1, 1.0, flag.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]
for nt in (0, 1):
    for blocks in (0, 1024, 4096, 8192, 16384):
        t = run(nt, blocks)
        print("nt=%d blocks=%5d: %.3f ms  %.2f TB/s" % (nt, blocks or 2048, t, n * 30 / t / 1e9))
# the overflow scan of the same buffer
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(); _lib.call("d2r_grad_nonfinite", g.data_ptr(), n, flag.data_ptr(), st); e1.record(); torch.cuda.synchronize()
print("nonfinite scan: %.3f ms  %.2f TB/s" % (e0.elapsed_time(e1), n * 4 / e0.elapsed_time(e1) / 1e9))
for _ in range(3):
    e0.record(); g.zero_(); e1.record(); torch.cuda.synchronize()
print("zero fill: %.3f ms  %.2f TB/s" % (e0.elapsed_time(e1), n * 4 / e0.elapsed_time(e1) / 1e9))
