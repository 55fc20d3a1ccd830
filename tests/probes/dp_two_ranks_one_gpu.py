"""Two data-parallel ranks sharing cuda:0 over gloo - or, with D2R_PROBE_BACKEND=nccl, ONE rank over RCCL with every collective
issued anyway (DataParallel(single_rank_collectives=True): the only way a one-GPU box executes the RCCL code path) - launched by
tests/test_gpu_trainer.py through torch.distributed.run:
the whole DataParallel machinery on real kernels — broadcast, sharded batch, bucketed all-reduce with and without
overlap (readiness callbacks from the kernels' gradient sinks, deferred grouped weight gradients, whole-layer C calls),
fused AdamW with grad_scale = 1/world.  Writes the final weights of both modes; they must be bit-identical."""
import faulthandler, os, sys
faulthandler.dump_traceback_later(int(os.environ.get("D2R_PROBE_DUMP_S", "150")), exit=True)  # a hang ends in tracebacks, not silence
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

out_dir = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("D2R_PROBE_BACKEND", "gloo")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend)
assert world == (1 if backend == "nccl" else 2), (backend, world)
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.dp import DataParallel, shard_batch
from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
ids = torch.randint(1000, 30000, (4, 16), generator=g); ids[:, 0] = 101
full = (ids, torch.ones(4, 16, dtype=torch.long), torch.zeros(4, 16, dtype=torch.long), torch.randint(0, 3, (4,), generator=g),
        torch.randn(4, 3, 64, 64, generator=g))
batch = tuple(t.to(dev) for t in shard_batch(full, rank, world))
results = {}
MODES = {False: {}, True: dict(overlap=True), "shard": dict(shard_optimizer=True), "bf16": dict(grad_comm_dtype=torch.bfloat16),
         "shard_overlap": dict(shard_optimizer=True, overlap=True), "rs_ag": dict(algorithm="reduce_scatter_all_gather", overlap=True)}
for overlap, kw in MODES.items():
    torch.manual_seed(100 + rank)  # different replicas on purpose: broadcast_parameters must make them identical
    tc = TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=2, image_size=64, patch_size=32)
    model = M.UnimoModelF(default_args(DR_step=3), vc, tc).to(dev)
    model.set_compute_dtype(torch.bfloat16).train()
    # One compute stream: with the two branch streams an open, timing-dependent last-bits difference in the router gradients of the
    # middle routing layer (tests/probes/determinism_probe.py, DESIGN.md "known issues") would make a bit-for-bit comparison of
    # the communication modes flaky for reasons that have nothing to do with the collectives under test here.
    model.model.use_streams = os.environ.get("D2R_PROBE_STREAMS", "0") == "1"
    store = ParamStore(model, torch.bfloat16)
    opt = FusedAdamW(store, lr=1e-3)
    sched = LinearWarmupSchedule(opt, 0, 12)
    dp = DataParallel(store, opt, model, bucket_mb=32, single_rank_collectives=(world == 1), **kw)
    assert dp.world == world and opt.grad_scale == 1.0 / world and dp.active
    if backend == "gloo":
        dp.reducer.poison_stale = True  # stripes this rank does not own: NaN after the (all-reduce standing in for the) reduce-scatter
    else:
        assert dp.reducer._has_reduce_scatter(), "the RCCL run must take reduce_scatter_tensor / all_gather_into_tensor"
    dp.broadcast_parameters()
    w_start = store.flat_w.detach().cpu().clone()  # (identical in every mode: same seeds, same broadcast)
    losses = []
    gsnap = []  # reduced gradient buffer of every step (plain and overlapped mode only): localises a first divergence
    for _ in range(3):  # (overlap: the first step calibrates the per-parameter report counts)
        dp.begin_step()
        loss, _ = model(*batch)
        loss.backward()
        dp.reduce_gradients()
        if overlap in (False, True) and os.environ.get("D2R_PROBE_GRADS", "1") != "0":
            torch.cuda.synchronize()
            gsnap.append(store.flat_g.detach().cpu().clone())
        opt.step()
        dp.gather_parameters()
        sched.step()
        opt.zero_grad()
        losses.append(float(loss))
    torch.cuda.synchronize()
    results[overlap] = (losses, store.flat_w.detach().cpu().clone(), gsnap)
    entries = [(n, o, k) for n, _, o, k, _ in store.entries]
    bounds = dp.reducer.bounds
    if overlap is True:
        multi = sum(1 for v in dp._expect.values() if v > 1)
        print(f"rank {rank}: {len(dp._expect)} reporting parameters, {multi} of them in more than one piece", flush=True)
        if rank == 0 and os.environ.get("D2R_PROBE_VERBOSE"):
            import collections
            names = {id(p): n for n, p, _, _, _ in store.entries}
            print("pieces histogram:", sorted(collections.Counter(dp._expect.values()).items()), flush=True)
            for pid, v in list(dp._expect.items()):
                print("  ", v, names.get(pid, "<fused leaf>"), flush=True)
dist.barrier()
same_modes = torch.equal(results[False][1], results[True][1])
w = results[True][1].to(dev)
other = w.clone()
dist.broadcast(other, src=0)
same_ranks = bool(torch.equal(other, w))
ne = results[False][1] != results[True][1]
csum = torch.cat([torch.zeros(1, dtype=torch.int64), ne.to(torch.int64).cumsum(0)])  # differing elements before each offset
count = lambda a, b: int(csum[b] - csum[a])
bad = [(n, count(o, o + k)) for n, o, k in entries if count(o, o + k)]
first = None
for step_i, (ga, gb) in enumerate(zip(results[False][2], results[True][2])):
    d = ga != gb
    if bool(d.any()):
        cs = torch.cat([torch.zeros(1, dtype=torch.int64), d.to(torch.int64).cumsum(0)])
        cnt = lambda a, b: int(cs[b] - cs[a])
        first = dict(step=step_i, n=int(d.sum()), max_abs=float((ga - gb).abs().max()),
                     tensors=[(n, cnt(o, o + k), float((ga[o:o + k] - gb[o:o + k]).abs().max()), float(ga[o:o + k].abs().max()))
                              for n, o, k in entries if cnt(o, o + k)][:60])
        break
upd = lambda key: (results[key][1] - w_start).double()
cos_bf16 = float((upd("bf16") @ upd(False)) / (upd("bf16").norm() * upd(False).norm()))
torch.save({"first_grad_divergence": first, "same_shard": bool(torch.equal(results[False][1], results["shard"][1])),
            "same_shard_overlap": bool(torch.equal(results[False][1], results["shard_overlap"][1])),
            "same_rs_ag": bool(torch.equal(results[False][1], results["rs_ag"][1])), "cos_bf16": cos_bf16, "losses_bf16": results["bf16"][0],
            "losses": results[True][0], "losses_plain": results[False][0], "same_modes": same_modes, "same_ranks": same_ranks,
            "n_diff": int(ne.sum()), "bad": bad[:40], "n_bad": len(bad), "bounds": bounds,
            "bad_buckets": [(i, count(a, b)) for i, (a, b) in enumerate(bounds) if count(a, b)],
            "max_abs_diff": float((results[False][1] - results[True][1]).abs().max()),
            "finite": bool(torch.isfinite(results[True][1]).all())}, os.path.join(out_dir, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
