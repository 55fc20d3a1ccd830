"""What does a dependency between two streams cost?  N tiny kernels back to back on one stream, against the same kernels alternating
between two streams with an event record + wait at every hop (the pattern of a fork / join).  GPU time between two events.
    python tests/probes/hop_latency.py"""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 400


def run(two):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        e0.record()
    cur = s1
    for i in range(N):
        nxt = (s2 if cur is s1 else s1) if two else s1
        if nxt is not cur:
            ev = torch.cuda.Event()
            ev.record(cur)
            nxt.wait_event(ev)
        with torch.cuda.stream(nxt):
            x.add_(1.0)
        cur = nxt
    if cur is not s1:
        ev = torch.cuda.Event()
        ev.record(cur)
        s1.wait_event(ev)
    with torch.cuda.stream(s1):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


for _ in range(2):
    a, b = run(False), run(True)
print(f"{N} tiny kernels: one stream {a:.2f} us per kernel; alternating between two streams with an event per hop {b:.2f} us per kernel "
      f"-> a cross-stream dependency costs about {b - a:.1f} us", flush=True)
