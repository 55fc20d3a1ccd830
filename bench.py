#!/usr/bin/env python3
"""bench.py — samples/s of the D2R hot path (fwd + bwd + gradient all-reduce + fused AdamW) on N MI355X GPUs.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1 is launched through
torch.distributed.run, one rank per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come from the environment).
Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] ("C2"): synthetic MVSA-Single shape — per-GPU batch 32, text length 128,
224x224 images at patch 16 (197 ViT tokens), DR_step 3, 12+12 encoder layers, random-init weights, BERT dropout 0
(BASELINE.md section 3).  Compute dtype of the headline: fp16 (IEEE half MFMA operands and activation storage, fp32
accumulation / statistics / softmax / poolers / Block / loss, fp32 master weights, dynamic loss scale) — the 16-bit mode
whose logits and loss stay within the north star's 1e-3 of the reference (asserted in tests/test_gpu_model.py and
tests/test_gpu_bench_shapes.py; DESIGN.md section 2 shows why no bf16-operand mode can).  The bf16 number is reported
beside it (`bf16_path`), never as `value`.  Weak scaling: the per-GPU batch is fixed as N grows.

Extra objects on the JSON line (tier contract section 4):
  roofline      the dominant kernel family by GPU time, timed live with HIP events on the launching stream in an
                instrumented pass of the SAME step right after the timed region (per-launch events would perturb
                the timed region); `roofline_kernels` lists router pool (K1), route_aggregate (K8) and the
                cross-attention core (K2) against the HBM roofline, and the GEMM families against the MFMA roofline
  cpu_baseline  the pinned CPU oracle (oracle/d2r_oracle.py, a port of the reference's maths — the reference itself
                cannot travel to the GPU box) timed on the host cores for a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3}  # dense peaks (MI355X_MICROARCH.md); fp16 and bf16 share one
PRECISION_NOTE = {
    "fp16": "fp16 MFMA operands + fp16 activation storage; fp32 accumulation, LayerNorm/BatchNorm/softmax statistics, poolers, Block, "
            "loss and master weights; dynamic loss scale with the overflow check inside the step. Logits/loss within 1e-3 of the "
            "reference at this shape and on every golden fixture (asserted, tests/test_gpu_bench_shapes.py, tests/test_gpu_model.py)",
    "bf16": "bf16 MFMA operands + bf16 activation storage, fp32 elsewhere as in the fp16 mode, no loss scale. NOT within 1e-3: "
            "2e-3 on the goldens (bf16 operand rounding alone is 0.9e-3, profiles/precision_policy_goldens_r03.log)",
    "f32": "fp32 everywhere (v_mfma_f32_16x16x4_f32); logits within 3e-7 of the reference",
}


_T0 = time.time()


def log(msg):
    """Progress on stderr (keeps the GPU box's silence watchdog fed; stdout carries only the JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--dr-step", type=int, default=3)
    ap.add_argument("--num-cells", type=int, default=6, help="cells per routing layer (6 = the reference; BASELINE configs[4] names 4)")
    ap.add_argument("--classes", type=int, default=3, help="classes of the head (3 = MVSA; BASELINE configs[3], TumEmo: 7)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "f32"],
                    help="compute dtype; fp16 (default) is the 16-bit mode that meets the 1e-3 logits/loss tolerance")
    ap.add_argument("--no-alt-leg", "--no-fp16-leg", dest="no_alt_leg", action="store_true",
                    help="skip the number of the OTHER 16-bit compute dtype reported beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=32, help="batch of the CPU baseline (BASELINE.md section 3: the benchmark batch)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the batch-2 measurement of the host's enqueue cost per step")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the short fp32-path timing reported next to the headline")
    ap.add_argument("--overlap", action="store_true", help="(default at N > 1) overlap the bucketed grad all-reduce with backward")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-reduce the whole gradient buffer after backward")
    ap.add_argument("--grad-comm", default="f32", choices=["f32", "bf16"], help="N > 1: dtype of the gradient buckets on the links")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: per bucket reduce-scatter of the gradients, AdamW on 1/N of the buffer, all-gather of the weights")
    ap.add_argument("--algorithm", default="all_reduce", choices=["all_reduce", "reduce_scatter_all_gather"],
                    help="N > 1: gradient reduction per bucket (RCCL all-reduce, or reduce-scatter + all-gather issued explicitly)")
    ap.add_argument("--bert-dropout", type=float, default=0.0,
                    help="hidden / attention-probability dropout of the BERT config (BASELINE.md section 3 benchmarks 0)")
    ap.add_argument("--tuning", type=int, action="append", default=[],
                    help="A/B runs: d2r_gemm_tuning tile code(s) applied before the first step (e.g. 120: grouped launches off, 102: 128-wide weight gradients)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("D2R_BENCH_GRAPH", "0")),
                    help="1: replay the step from a captured hipGraph (fwd+bwd[+AdamW at N=1]); 0 (default): eager launches — measured faster on ROCm 7.2, see DESIGN.md")
    return ap.parse_args()


def synthetic_batch(B, L, image_size, device, seed, classes=3):
    """ids ~ U{1000..29999} with [:,0]=101, all-ones mask, images ~ N(0,1), labels ~ U{0..classes-1}."""
    import torch
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, 30000, (B, L), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, L, dtype=torch.long)
    tt = torch.zeros(B, L, dtype=torch.long)
    images = torch.randn(B, 3, image_size, image_size, generator=g)
    labels = torch.randint(0, classes, (B,), generator=g)
    return tuple(t.to(device) for t in (ids, mask, tt, labels, images))


def workload_name(args):
    """Which BASELINE.json configuration the flags describe (per-GPU shard of the 8-GPU configurations)."""
    key = (args.seq, args.image_size, args.patch, args.dr_step, args.num_cells, args.classes)
    if key == (128, 224, 16, 3, 6, 3):
        return "BASELINE configs[1]" if args.batch == 32 else "BASELINE configs[1] shape at another batch"
    if key == (256, 384, 16, 3, 6, 7):
        return "BASELINE configs[3] (TumEmo-scale: global batch 64 on 8 GPUs = 8 per GPU)" if args.batch == 8 else "BASELINE configs[3] shape at another batch"
    if key == (512, 224, 16, 8, 4, 3):
        return "BASELINE configs[4] (stress: global batch 128 on 8 GPUs = 16 per GPU)" if args.batch == 16 else "BASELINE configs[4] shape at another batch"
    return "custom shape"


def cpu_baseline(args, sd_cpu):
    """The pinned oracle (CPU port of the reference's maths; the reference itself cannot travel to the GPU box) timed on
    the host cores by BASELINE.md section 3: the benchmark's own batch (32 samples, same synthetic tensors), fp32, train
    mode, dropout 0, 1 warm-up + the median of 3 timed fwd+bwd iterations."""
    import platform
    import statistics
    import torch
    from oracle import d2r_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a 1-GPU box owns a 16-core CPU share; os.cpu_count() reports the whole host
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(text_layers=args.layers, vision_layers=args.layers, image_size=args.image_size,
                         patch_size=args.patch, DR_step=args.dr_step, num_cells=args.num_cells, num_classes=args.classes)
    nb = args.cpu_samples
    ids, mask, tt, labels, images = synthetic_batch(nb, args.seq, args.image_size, "cpu", seed=0, classes=args.classes)

    def run():
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
              for k, v in sd_cpu.items()}
        t0 = time.time()
        loss, _, _ = O.forward(sd, cfg, ids, mask, tt, labels, images, train=True)
        loss.backward()
        return time.time() - t0

    log(f"cpu baseline: warm-up iteration (batch {nb}, {cores} threads)")
    run()
    times = []
    for i in range(3):
        times.append(run())
        log(f"cpu baseline: timed iteration {i + 1}/3: {times[-1]:.1f} s")
    dt = statistics.median(times)
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), cpu_model)
    except OSError:
        pass
    return {"value": round(nb / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"batch {nb} of the same workload (L={args.seq}, {(args.image_size // args.patch) ** 2 + 1} image tokens, "
                      f"{args.layers}+{args.layers} encoder layers, DR_step {args.dr_step}), fp32, train mode, 1 warm-up + median of 3 "
                      f"timed fwd+bwd iterations ({', '.join(f'{t:.1f}' for t in times)} s), torch CPU {torch.get_num_threads()} threads "
                      f"of {os.cpu_count()} host CPUs ({cpu_model})"}


def self_launch(args):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks as a CHILD process (this process has
    not touched the GPU — torch is not even imported yet — and never re-execs), relay its output and exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist
    from d2r_amd import modules as M
    from d2r_amd._lib import KernelTimer
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    from d2r_amd.dp import DataParallel, init_process_group_from_env
    from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore

    import d2r_amd
    d2r_amd.configure_runtime()
    log("imports done")
    for code in args.tuning:
        from d2r_amd import _lib as _l
        _l.load().d2r_gemm_tuning(1, 1, code)
    rank, world = init_process_group_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)  # (% : gloo rehearsal on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[args.dtype]

    torch.manual_seed(2023)  # reference default seed (run.py:49); identical replicas, then broadcast anyway
    tc = TextConfig(num_hidden_layers=args.layers, hidden_dropout_prob=args.bert_dropout,
                    attention_probs_dropout_prob=args.bert_dropout)
    vc = VisionConfig(num_hidden_layers=args.layers, image_size=args.image_size, patch_size=args.patch)
    model = M.UnimoModelF(default_args(DR_step=args.dr_step, num_cells=args.num_cells), vc, tc, num_classes=args.classes)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    log("model built on host")
    model.to(dev).set_compute_dtype(dtype).train()
    store = ParamStore(model, dtype)
    log(f"model on {dev}: {store.live_numel() / 1e6:.1f} M live parameters in flat buffers")
    opt = FusedAdamW(store, lr=3e-5)
    if dtype == torch.float16:
        opt.enable_loss_scaling()  # scaled loss, device-side overflow check and step skip: all inside the timed step
    total_steps = args.warmup + args.steps + 8
    sched = LinearWarmupSchedule(opt, 0.01 * total_steps, total_steps)
    dp = DataParallel(store, opt, model, overlap=not args.no_overlap,  # (no effect at N = 1)
                      grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else torch.float32, shard_optimizer=args.shard_optimizer,
                      algorithm=args.algorithm)
    dp.broadcast_parameters()
    batch = synthetic_batch(args.batch, args.seq, args.image_size, dev, seed=rank, classes=args.classes)

    def step():
        dp.begin_step()
        loss, logits = model(*batch)
        opt.backward(loss)
        dp.reduce_gradients()
        opt.step()
        dp.gather_parameters()
        sched.step()
        opt.zero_grad()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eager_step = step
    if args.graph and args.bert_dropout > 0.0:
        raise SystemExit("--graph 1 with --bert-dropout > 0: the dropout seeds are host scalars baked into kernel arguments, a "
                         "replayed graph would reuse one mask every step")
    if args.graph:
        # The step is ~1500 short launches; replaying them from one hipGraph removes the host launch cost and lets
        # the two encoder/interaction streams really overlap.  Same kernels, same order, same buffers as eager.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # allocate every workspace / cache outside the graph's private pool
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph, graph_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        static = {}
        was_overlap, dp.overlap = dp.overlap, False  # no readiness hooks / collectives inside the capture
        with torch.cuda.graph(graph):
            static["loss"], static["logits"] = model(*batch)
            opt.backward(static["loss"])  # (fp16: the root gradient is the loss scale's device scalar)
        if world == 1:
            # the optimiser is its own graph: its per-step scalars are uploaded between the two replays, after the overflow flag
            # of the PREVIOUS step has been read (it landed long ago) - the eager protocol, and the host never waits
            with torch.cuda.graph(graph_opt):
                opt.step_captured()
                opt.zero_grad()
        del was_overlap  # at N > 1 the captured fwd+bwd is followed by the plain (non-overlapped) bucketed all-reduce
        log("hipGraph of the step captured")

        def step():
            opt.refresh_root_grad()
            if world == 1:
                graph.replay()
                opt.stage_hyper()
                graph_opt.replay()
                opt.after_replay()
            else:
                dp.begin_step()
                graph.replay()
                dp.reduce_gradients()
                opt.step()
                opt.zero_grad()
            sched.step()
            return static["loss"]

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i + 1}/{args.warmup} done")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3  # host time to ENQUEUE a step (no device wait)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    log(f"timed region: {args.steps} steps, {ms:.2f} ms/step, {value:.1f} samples/s (host enqueue {host_ms:.2f} ms/step)")

    # From here on no rank issues a gradient collective any more (the legs below are local measurements; rank 0 runs
    # one more pass than the others): switch the overlap hooks off on every rank.
    dp.overlap = False

    # SURVEY.md 8d also asks for fwd+bwd WITHOUT the optimiser: a short second timed loop (not `value`)
    def fwd_bwd_only():
        dp.begin_step()
        loss, _ = model(*batch)
        opt.backward(loss)
        opt.zero_grad()

    n2 = max(3, args.steps // 2)
    fwd_bwd_only()
    fence()
    t1 = time.perf_counter()
    for _ in range(n2):
        fwd_bwd_only()
    fence()
    fb_ms = (time.perf_counter() - t1) / n2 * 1e3
    log(f"fwd+bwd without optimiser / all-reduce: {fb_ms:.2f} ms/step")

    out = {
        "metric": "samples/sec fwd+bwd", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "precision": PRECISION_NOTE[args.dtype], "data": "synthetic",
        "headline_dtype_history": "value is the fp16 compute mode since round 3 (BENCH_r01 / r02: bf16); the bf16 mode of this build is bf16_path",
        "config": {"workload": "%s: synthetic MVSA-Single-style batch, per-GPU batch %d, seq_len %d, %d image tokens, "
                               "DR_step %d, %d cells per routing layer, %d classes, %d+%d encoder layers, random-init weights, dropout %g; "
                               "step = fwd+bwd+grad-allreduce+AdamW"
                               % (workload_name(args), args.batch, args.seq, (args.image_size // args.patch) ** 2 + 1, args.dr_step,
                                  args.num_cells, args.classes, args.layers, args.layers, args.bert_dropout),
                   "global_batch": world * args.batch, "seq_len": args.seq, "parallelism": f"dp{world}"},
        "final_loss": round(loss_val, 5), "launch": "hipGraph replay" if args.graph else "eager",
        "host_enqueue_ms_per_step": round(host_ms, 3),
        "fwd_bwd_only": {"ms_per_step_per_rank": round(fb_ms, 3), "samples_per_s": round(world * args.batch / fb_ms * 1e3, 2),
                         "note": "no optimiser step, no gradient all-reduce; measured on rank 0's clock"},
    }

    if rank == 0 and world == 1 and not args.graph and not args.no_host_leg:
        # What the HOST needs to enqueue one step, separated from the GPU: the same model and step at batch 2 - the GPU drains its
        # queue long before the host refills it, so the loop time is host time (launch calls, autograd, Python).  The
        # host_enqueue figure above is taken at the benchmark batch, where the host mostly waits for room in the launch queue.
        try:
            small = synthetic_batch(2, args.seq, args.image_size, dev, seed=7, classes=args.classes)

            def small_step():
                loss, _ = model(*small)
                opt.backward(loss)
                opt.step()
                opt.zero_grad()

            for _ in range(3):
                small_step()
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(10):
                small_step()
            h_ms = (time.perf_counter() - th) / 10 * 1e3
            torch.cuda.synchronize()
            out["host_only_ms_per_step"] = round(h_ms, 2)
            out["host_only_note"] = "same model and step at batch 2: loop time without synchronisation = host time per step"
            log(f"host-only leg: {h_ms:.2f} ms/step")
        except Exception as e:
            out["host_only_ms_per_step"] = None
            out["host_only_note"] = repr(e)

    if rank == 0 and not args.no_roofline:
        # Per-kernel durations: the same step, but launched op by op on ONE stream (whole-layer C calls and the
        # two-stream overlap switched off), so that every launch is bracketed by its own pair of HIP events and runs
        # alone on the GPU — the durations rocprofv3 --kernel-trace reports (it serialises dispatches too).
        def local_step():  # the step without its collective: rank 0 is alone here
            loss, _ = model(*batch)
            opt.backward(loss)
            opt.step()
            sched.step()
            opt.zero_grad()

        saved = (M.COMPOSITE_LAYERS, model.model.use_streams, M.COMPOSITE_ROUTING)
        M.COMPOSITE_LAYERS, model.model.use_streams, M.COMPOSITE_ROUTING = False, False, False
        try:
            local_step()
            torch.cuda.synchronize()
            with KernelTimer() as kt:
                for _ in range(2):
                    local_step()
            summ = kt.summary()
        finally:
            M.COMPOSITE_LAYERS, model.model.use_streams, M.COMPOSITE_ROUTING = saved
        # The GEMM families are re-measured in the launch configuration of the TIMED step (whole-layer / whole-module C calls
        # on, hence the grouped weight-gradient launches at their real group sizes), one stream: the library brackets every
        # d2r_gemm / d2r_gemm_tn_grouped launch - also those issued inside the composite calls - with HIP events on the
        # launching stream (d2r_gemm_timer).  These records replace the GEMM rows of the op-by-op pass above.
        import ctypes as C
        from d2r_amd import _lib
        lib = _lib.load()
        saved_streams = model.model.use_streams
        model.model.use_streams = False
        try:
            local_step()
            torch.cuda.synchronize()
            _lib.call("d2r_gemm_timer", 1)
            for _ in range(2):
                local_step()
            torch.cuda.synchronize()
            _lib.call("d2r_gemm_timer", 0)
        finally:
            model.model.use_streams = saved_streams
        cap = 1 << 15
        fam, fl, by, ms = (C.c_int * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), (C.c_float * cap)()
        n_rec = lib.d2r_gemm_timer_read(fam, fl, by, ms, cap)
        # What one event bracket adds to a launch (the record commands are barriers in the queue: the next kernel cannot start
        # under the tail of the previous one): the same tiny kernel timed as a train of 200 launches inside ONE bracket and as
        # 200 individually bracketed launches; the difference per launch is subtracted from every bracketed duration below.
        from d2r_amd import functional as F_
        one = torch.zeros(8, device=dev)
        tiny = lambda: _lib.call("d2r_axpby", 0, 1.0, one.data_ptr(), 1.0, one.data_ptr(), 8, F_._stream())
        for _ in range(20):
            tiny()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        a0, a1 = ev(), ev()
        a0.record()
        for _ in range(200):
            tiny()
        a1.record()
        pairs = []
        for _ in range(200):
            p0, p1 = ev(), ev()
            p0.record()
            tiny()
            p1.record()
            pairs.append((p0, p1))
        torch.cuda.synchronize()
        each = sorted(p0.elapsed_time(p1) for p0, p1 in pairs)[100]
        bracket_ms = max(each - a0.elapsed_time(a1) / 200.0, 0.0)
        out["event_bracket_overhead_us"] = round(bracket_ms * 1e3, 2)
        for r in summ.values():  # the op-by-op pass (KernelTimer) used the same brackets
            r["ms"] = max(r["ms"] - bracket_ms * r["calls"], 0.25 * r["ms"])
        summ = {k: v for k, v in summ.items() if not k.startswith("gemm_")}
        per_family = {}
        VARIANT = {0: "tiles", 1: "ldsdma128x64", 2: "ldsdma128x128w4", 3: "ldsdma128x128w8", 4: "ldsdma128x192w8", 11: "ldsdma128x64p", 12: "ldsdma128x128w4p",
                   13: "ldsdma128x128w8p", 20: "ldsdma128x128", 21: "tiles64x64", 22: "batched16", 30: "skinny", 31: "skinny",
                   8: "ldsdma256x256", 28: "ldsdma256x256"}
        ATT = {10001: "xattn_core_fwd", 10002: "xattn_core_bwd"}  # recorded by d2r_xattn_{fwd,bwd}_multi in the step's real launch shape
        att = {}
        for i in range(n_rec):
            if fam[i] in ATT:
                att.setdefault(ATT[fam[i]], []).append((ms[i], fl[i], by[i]))
                continue
            var, base = fam[i] // 100, fam[i] % 100
            # one row per KERNEL (template instance family), e.g. gemm_bf16_TN_grouped_ldsdma128x128 = gemm_glds_kernel<__bf16, TN, 128, 2, 1, true>
            name = "gemm_%s_%s%s_%s" % (("f32", "bf16", "f16")[base // 8], ("NT", "NN", "TN")[(base % 8) // 2], "_grouped" if base & 1 else "",
                                        VARIANT.get(var, str(var)))
            per_family.setdefault(name, []).append((ms[i], fl[i], by[i]))
        for name, recs in per_family.items():
            ts = sorted(t for t, _, _ in recs)
            med = ts[len(ts) // 2]
            clip = lambda t: med if (t > 8.0 * med and t > 0.2) else t  # a host hiccup between the two events is not kernel time
            summ[name] = dict(calls=len(recs), ms=sum(max(clip(t) - bracket_ms, 0.25 * clip(t)) for t, _, _ in recs), flops=sum(f for _, f, _ in recs),
                              bytes=sum(b for _, _, b in recs), algo_bytes=0.0, outliers=sum(1 for t in ts if t > 8.0 * med and t > 0.2))
        for name, recs in att.items():  # (three alignment cores per launch in the routing modules: replaces the op-by-op row)
            ts = sorted(t for t, _, _ in recs)
            med = ts[len(ts) // 2]
            clip = lambda t: med if (t > 8.0 * med and t > 0.2) else t
            nb = sum(b for _, _, b in recs)
            summ[name] = dict(calls=len(recs), ms=sum(max(clip(t) - bracket_ms, 0.25 * clip(t)) for t, _, _ in recs), flops=sum(f for _, f, _ in recs),
                              bytes=nb, algo_bytes=nb, lib_timed=True)
        kernels = []
        for name, r in summ.items():
            calls, t_s = r["calls"], r["ms"] * 1e-3
            if t_s <= 0:
                continue
            ent = {"kernel": name, "launches_per_step": calls // 2, "ms_per_step": round(r["ms"] / 2, 4),
                   "avg_us": round(r["ms"] * 1e3 / calls, 2)}
            if name.startswith("gemm_"):
                key = "f32" if "f32" in name else ("fp16" if "_f16_" in name else "bf16")
                ach = r["flops"] / t_s / 1e12
                ent.update(bound="mfma", achieved=round(ach, 2), peak=MFMA_PEAK_TF[key], unit="TFLOP/s",
                           frac=round(ach / MFMA_PEAK_TF[key], 4), algo_bytes_per_launch=round(r["bytes"] / calls))
            else:
                nbytes = r["algo_bytes"] if r["algo_bytes"] > 0 else r["bytes"]
                if nbytes > 0:
                    ach = nbytes / t_s / 1e9
                    if name.startswith("xattn_core") and name.endswith("bwd") and not r.get("lib_timed"):
                        ach = 2 * summ.get("xattn_core_fwd", r)["algo_bytes"] / t_s / 1e9  # bwd = 2x fwd bytes
                    ent.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=round(ach / HBM_PEAK_GBS, 4))
            kernels.append(ent)
        kernels.sort(key=lambda e: -e["ms_per_step"])
        total_ms = sum(e["ms_per_step"] for e in kernels)
        dom = next((e for e in kernels if "bound" in e), None)
        if dom is not None:
            # HBM bytes of that kernel family from the committed rocprofv3 --pmc passes of the real step, per training step,
            # divided by THIS pass's launches per step: comparable with the algorithmic bytes per launch
            traffic = None
            pmc_file = None
            try:
                cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("pmc_traffic_r") and f.endswith(".json"))
                pmc_file = cands[-1] if cands else None
                if pmc_file:
                    with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                        pm = json.load(f)
                    per_step = pm.get(dom["kernel"], {}).get("hbm_bytes_per_step")  # rows are per kernel, named as here
                    if per_step:
                        traffic = round(per_step / max(dom["launches_per_step"], 1))
            except OSError:
                pass
            out["roofline"] = {"kernel": dom["kernel"], "bound": dom["bound"], "achieved": dom["achieved"],
                               "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"], "traffic": traffic,
                               "traffic_source": f"profiles/{pmc_file}: family HBM bytes per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 on gfx950) / launches per step of this pass" if traffic else None,
                               "algorithmic_bytes_per_launch": dom.get("algo_bytes_per_launch"),
                               "avg_launch_us": dom["avg_us"], "share_of_kernel_time": round(dom["ms_per_step"] / max(total_ms, 1e-9), 3)}
        log("instrumented (per-kernel HIP event) pass done")
        # always report the router pool (K1), the aggregation (K8), the cross-attention core (K2) and the
        # multi-head cores (K3/K4) against the HBM roofline, whatever their rank by time
        named = ("router_pool", "route_aggregate", "xattn_core", "mha_core")
        top = kernels[:20]
        out["roofline_kernels"] = top + [e for e in kernels[20:] if e["kernel"].startswith(named)]
        out["kernel_ms_per_step_total"] = round(total_ms, 3)

    if rank == 0 and world == 1 and args.dtype != "f32" and not args.no_fp32_leg:
        # The same step on the fp32 HIP path (exact-f32 MFMA, the parity reference: logits within 1e-7 of the reference),
        # reported NEXT to the headline so that the price of the 16-bit compute dtype is visible; never `value`.
        try:
            model.set_compute_dtype(torch.float32)
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            f_ms = (time.perf_counter() - t2) / 3 * 1e3
            out["fp32_path"] = {"ms_per_step": round(f_ms, 2), "samples_per_s": round(args.batch / f_ms * 1e3, 2),
                                "note": "same step, fp32 activations / weights / MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate)"}
            log(f"fp32 path: {f_ms:.1f} ms/step")
        except Exception as e:
            out["fp32_path"] = {"error": repr(e)}
        finally:
            model.set_compute_dtype(dtype)

    if rank == 0 and world == 1 and args.dtype in ("fp16", "bf16") and not args.no_alt_leg and not args.graph:
        # The same step with the OTHER 16-bit compute dtype (same kernels and MFMA rate; bf16: 8 significant bits and no loss
        # scale, fp16: 11 bits and a scaled loss).  Run as a child process (its own ParamStore and 16-bit shadow); never `value`.
        alt = "bf16" if args.dtype == "fp16" else "fp16"
        try:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", alt, "--steps", str(min(args.steps, 10)), "--warmup", "3",
                   "--batch", str(args.batch), "--seq", str(args.seq), "--layers", str(args.layers), "--image-size", str(args.image_size),
                   "--patch", str(args.patch), "--dr-step", str(args.dr_step), "--num-cells", str(args.num_cells), "--classes", str(args.classes),
                   "--no-cpu-baseline",
                   "--no-fp32-leg", "--no-roofline", "--no-alt-leg", "--no-host-leg"]
            torch.cuda.synchronize()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            child = json.loads(line)
            out[alt + "_path"] = {"ms_per_step": child["ms_per_step"], "samples_per_s": child["value"], "final_loss": child.get("final_loss"),
                                  "note": "same step, " + PRECISION_NOTE[alt]}
            log(f"{alt} path: {child['ms_per_step']:.2f} ms/step")
        except Exception as e:
            out[alt + "_path"] = {"error": repr(e)}

    if sd_cpu is not None:
        try:
            out["cpu_baseline"] = cpu_baseline(args, sd_cpu)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU number because of it
            out["cpu_baseline"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
