"""Builds ``d2r_amd/libd2r_hip.so`` (hand-written gfx950 kernels + the C ABI of include/d2r_hip.h) with hipcc.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box inside the repo snapshot.  Objects are cached per source hash.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
BUILD_DIR = os.path.join(HERE, "csrc", "build")
LIB_PATH = os.path.join(HERE, "libd2r_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


# per-file additions to FLAGS
EXTRA_FLAGS = {"routing.hip": ["-fno-slp-vectorize"]}
if os.environ.get("D2R_GEMM_PROBES", "0") != "0":  # measurement build: cycle stamps / ablation switches inside the LDS-DMA GEMM kernel
    EXTRA_FLAGS["gemm_glds.hip"] = ["-DD2R_GEMM_PROBES=1"]
if os.environ.get("D2R_X3_PROBES", "0") != "0":  # measurement build of the cross-attention forward kernel (tests/probes/xattn3_probe.py stamps)
    EXTRA_FLAGS["xattn3.hip"] = ["-DD2R_X3_PROBES=1"]
    EXTRA_FLAGS["xattn2.hip"] = ["-DD2R_X3_PROBES=1"]
if os.environ.get("D2R_G8_STAMPS", "0") != "0":  # measurement build of the 256-wide GEMM (tests/probes/gemm8_probe.py stamps)
    EXTRA_FLAGS["gemm8.hip"] = ["-DD2R_G8_STAMPS=1"]

if os.environ.get("D2R_FAST_ACT", "0") != "0":  # hardware exp / rcp and a polynomial erfc in the 16-bit GEMM epilogues (gemm_args.h: -0.25 ms per step, not the default)
    for f_ in ("gemm.hip", "gemm_glds.hip", "gemm8.hip"):
        EXTRA_FLAGS[f_] = EXTRA_FLAGS.get(f_, []) + ["-DD2R_FAST_ACT=1"]

def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libd2r_hip.so cannot be built (ROCm toolchain missing)")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str) -> str:
    h = hashlib.sha256()
    deps = [path, os.path.join(INCLUDE, "d2r_hip.h")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    for dep in deps:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()[:16]


def _compile_one(src: str, verbose: bool) -> str:
    obj = os.path.join(BUILD_DIR, os.path.basename(src)[:-4] + "." + _digest(src) + ".o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(BUILD_DIR):
        if old.startswith(os.path.basename(src)[:-4] + ".") and old.endswith(".o"):
            os.remove(os.path.join(BUILD_DIR, old))
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-I", INCLUDE, "-c", src, "-o", obj]
    if verbose:
        print("[d2r build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, flush=True)
    return obj


def build(verbose: bool = True, force: bool = False) -> str:
    os.makedirs(BUILD_DIR, exist_ok=True)
    if force:
        for f in os.listdir(BUILD_DIR):
            os.remove(os.path.join(BUILD_DIR, f))
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, verbose), srcs))
    stamp = hashlib.sha256(" ".join(objs).encode()).hexdigest()[:16]
    stamp_file = os.path.join(BUILD_DIR, "link.stamp")
    if os.path.exists(LIB_PATH) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB_PATH
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs]
    if verbose:
        print("[d2r build] link", LIB_PATH, flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
