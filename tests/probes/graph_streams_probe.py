"""Which multi-stream pattern breaks hipGraph capture on this ROCm?  usage: graph_streams_probe.py <variant>"""
import sys
import torch
variant = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.randn(1024, 1024, device=dev)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
if "norec" in variant:
    torch.Tensor.record_stream = lambda self, s: None
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    x.record_stream(a); x.record_stream(b)
    a.wait_stream(main); b.wait_stream(main)
    with torch.cuda.stream(a):
        ya = x * 2
        ya.record_stream(b)
    with torch.cuda.stream(b):
        yb = x + 1
        yb.record_stream(a)
    if "cross" in variant:
        if "ev" in variant:
            ea, eb = a.record_event(), b.record_event()
            a.wait_event(eb); b.wait_event(ea)
        elif "rejoin" in variant:
            main.wait_stream(a); main.wait_stream(b); a.wait_stream(main); b.wait_stream(main)
        else:
            a.wait_stream(b); b.wait_stream(a)
        with torch.cuda.stream(a):
            za = ya + yb
        with torch.cuda.stream(b):
            zb = ya - yb
    else:
        za, zb = ya, yb
    main.wait_stream(a); main.wait_stream(b)
    za.record_stream(main); zb.record_stream(main)
    out = za * zb
print(variant, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
ref = ((x * 2 + x + 1) * (x * 2 - x - 1)) if "cross" in variant else (x * 2) * (x + 1)
print(variant, "replayed OK", float((out - ref).abs().max()), flush=True)
