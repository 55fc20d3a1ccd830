"""Where does the fp16 path lose gradient accuracy at the C2 shape?  GPU probe around
tests/test_gpu_bench_shapes.py::fixed_cotangent_gradients: loss-scale sweep and the worst tensors.
    python tests/probes/fixed_cot_probe.py [fp16|bf16] [scales...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_bench_shapes as T  # noqa: E402

dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
scales = [float(x) for x in sys.argv[2:]] or [1024.0]
gpu = torch.device("cuda", 0)
for sc in scales:
    per = {}
    e_s, cos, ratio, cos_enc, n = T.fixed_cotangent_gradients(gpu, dtype, lscale=sc, per_tensor=per)
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'fp16'} loss scale {sc:g}: objective err {e_s:.2e} cosine {cos:.5f} |g|/|ref| {ratio:.4f} encoders {cos_enc:.5f}", flush=True)
    groups = {}
    for k, (c, r, nr) in per.items():
        parts = k.split(".")
        key = ".".join(parts[:4]) if "encoder" in k else ".".join(parts[:5])
        g = groups.setdefault(key, [0.0, 0.0, 0.0])
        g[0] += c * r * nr * nr
        g[1] += (r * nr) ** 2
        g[2] += nr * nr
    for key, (d, g2, r2) in groups.items():
        print(f"    {key:60s} cos {d / max((g2 * r2) ** 0.5, 1e-300):.4f} |g|/|ref| {(g2 / max(r2, 1e-300)) ** 0.5:.3f} |ref| {r2 ** 0.5:.2e}")
