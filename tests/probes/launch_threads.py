"""Do kernel launches from two host threads on two streams run in parallel?  (ctypes releases the GIL around the foreign call.)
    python tests/probes/launch_threads.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
fn = _lib._FN["d2r_axpby"]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(8, device=dev); b = torch.zeros(8, device=dev)
N = 4000


def run(buf, st, n):
    p = buf.data_ptr()
    for _ in range(n):
        fn(0, 1.0, p, 1.0, p, 8, st)


for _ in range(2):
    run(a, s1.cuda_stream, 200); run(b, s2.cuda_stream, 200)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(a, s1.cuda_stream, N); run(b, s2.cuda_stream, N); t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"one thread, two streams, {2 * N} launches: {(t1 - t0) / (2 * N) * 1e6:.2f} us per launch")
th = threading.Thread(target=run, args=(b, s2.cuda_stream, N))
t0 = time.perf_counter(); th.start(); run(a, s1.cuda_stream, N); th.join(); t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"two threads, one stream each, {2 * N} launches: {(t1 - t0) / (2 * N) * 1e6:.2f} us per launch (wall)")
