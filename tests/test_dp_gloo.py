"""CPU, world_size 2 over gloo: the data-parallel pieces that do not need a GPU — batch sharding, the bucketed flat
gradient all-reduce (d2r_amd.dp.FlatGradReducer) and the equivalence "SUM over ranks x 1/world == mean of the per-shard
gradients" on the routing module (computed with the CPU oracle, which is allowed in tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _routing_grads(shard):
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(DR_step=2)
    sd = O.seeded_state_dict(cfg, seed=5, router_bias="normal", spec=O.interaction_spec(cfg))
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k and not O.is_dead_param("model.itr_module." + k)]
    osd = {"M." + k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    own, other, w = shard
    emb, sim = O.interaction_module(osd, "M", own, other, 2, O.BNState(True))
    ((emb * w).sum() / own.shape[0] + sim.sum() * 1e-3).backward()
    return torch.cat([osd["M." + k].grad.flatten() for k in names])


class _FakeStore:
    """Three 100-element parameters in one flat gradient buffer (what DataParallel needs of a ParamStore)."""

    def __init__(self):
        self.flat_g = torch.zeros(300)
        self.params = [torch.zeros(100, requires_grad=True) for _ in range(3)]
        self.dead = []

    def units(self):
        return [(p, 100 * i, 100) for i, p in enumerate(self.params)]


class _FakeOpt:
    grad_scale = 1.0
    element_ranges = None
    shard_gather = None
    dp_group = None


def _overlap_readiness(rank):
    """Overlapped bucket launches when a parameter's gradient arrives in two pieces per step (one written during
    backward, one by a deferred launch at its end): the first step calibrates, later steps launch each bucket at the LAST
    piece; a piece after the launch is a loud error."""
    from d2r_amd.dp import DataParallel
    store = _FakeStore()
    dp = DataParallel(store, _FakeOpt(), None, overlap=True, bucket_elems=100)
    p0, p1, p2 = store.params
    order = []
    launch = dp.reducer.launch_bucket
    dp.reducer.launch_bucket = lambda i: (order.append(i), launch(i))[1]

    def step(extra_piece=False):
        store.flat_g.zero_()
        order.clear()
        dp.begin_step()
        store.flat_g[200:300] += rank + 1          # first piece of p2
        dp._ready(p2)
        store.flat_g[100:200] += 5 * (rank + 1)
        dp._ready(p1)
        store.flat_g[0:100] += 7 * (rank + 1)
        dp._ready(p0)
        store.flat_g[200:300] += 10 * (rank + 1)   # second piece of p2, e.g. from the grouped launch at the end of backward
        dp._ready(p2)
        if extra_piece:
            with pytest.raises(RuntimeError, match="after its bucket"):
                dp._ready(p2)
        dp.reduce_gradients()
        want = torch.cat([torch.full((100,), 21.0), torch.full((100,), 15.0), torch.full((100,), 33.0)])
        assert torch.equal(store.flat_g, want), store.flat_g[::100]

    step()
    assert order == [0, 1, 2], order            # calibration step: everything after "backward"
    assert dp._expect[id(p2)] == 2 and dp._expect[id(p1)] == 1
    step()
    assert order == [1, 0, 2], order            # overlapped: a bucket goes out at the last piece of its last parameter
    step(extra_piece=True)


def _bf16_buckets_and_sharded_optimizer(rank, world):
    """(i) gradient buckets sent as bf16: the fp32 buffer receives the bf16-rounded sum; (ii) the sharded optimiser's control
    flow: after the scatter a rank holds the summed gradients of its slice (and of the common tail), updates exactly those
    elements, and the all-gather leaves every rank with the same, fully updated weights."""
    from d2r_amd.dp import DataParallel, FlatGradReducer
    from d2r_amd.params import FusedAdamW
    flat = (torch.arange(1003, dtype=torch.float32) / 7.0) * (rank + 1)
    want = (torch.arange(1003, dtype=torch.float32) / 7.0).bfloat16().float() + (torch.arange(1003, dtype=torch.float32) / 7.0 * 2).bfloat16().float()
    red = FlatGradReducer(flat, bucket_elems=256, comm_dtype=torch.bfloat16)
    red.reduce_all()
    assert float((flat - want.bfloat16().float()).abs().max()) <= 2.0 ** -7 * float(want.abs().max()), "bf16 bucket sum"

    class Store(_FakeStore):
        def __init__(self):
            super().__init__()
            self.flat_g = torch.zeros(303)
            self.flat_w = torch.arange(303, dtype=torch.float32).clone()

        def refresh_lowp(self):
            self.refreshed = True

    # buckets of 128 elements (a multiple of 4 * world): [0,128) [128,256) [256,303); the last one: stripes of 20 and a tail of 7
    store, opt = Store(), _FakeOpt()
    dp = DataParallel(store, opt, None, shard_optimizer=True, bucket_elems=130)
    assert dp.reducer.bounds == [(0, 128), (128, 256), (256, 303)] and dp.reducer.mode == "reduce_scatter"
    own = [(64 * rank, 64 * rank + 64), (128 + 64 * rank, 192 + 64 * rank), (256 + 20 * rank, 276 + 20 * rank)]
    assert opt.element_ranges == own + [(296, 303)], opt.element_ranges
    assert opt.grad_scale == 0.5 and opt.shard_gather is not None
    # gloo all-reduces the bucket (no reduce-scatter): the stripes this rank does NOT own would hold the full sum, where RCCL's
    # reduce-scatter leaves them stale.  The rehearsal poisons them with NaN, so that anything downstream that reads a non-owned
    # stripe (a wrong stripe offset, an optimiser range that is too wide) shows up here instead of on the 8-GPU node.
    dp.reducer.poison_stale = True
    store.flat_g += (rank + 1)
    dp.reduce_gradients()
    for a, b in opt.element_ranges:  # summed where this rank will update
        assert torch.equal(store.flat_g[a:b], torch.full((b - a,), 3.0))
    owned_mask = torch.zeros(303, dtype=torch.bool)
    for a, b in opt.element_ranges:
        owned_mask[a:b] = True
    assert bool(torch.isnan(store.flat_g[~owned_mask]).all()) and int((~owned_mask).sum()) == 148, "non-owned stripes must be poisoned"
    owned = FusedAdamW._owned(opt, (100, 300))  # a parameter group's range against this rank's stripes and the tail
    assert owned == ([(128, 192), (256, 276), (296, 300)] if rank == 0 else [(100, 128), (192, 256), (276, 296), (296, 300)]), owned
    assert FusedAdamW._owned(opt, (0, 0)) == []
    for x, y in FusedAdamW._owned(opt, (0, 303)):  # a stand-in optimiser: w -= grad_scale * g on the owned elements only
        store.flat_w[x:y] -= opt.grad_scale * store.flat_g[x:y]
    dp.gather_parameters()
    assert store.refreshed and torch.equal(store.flat_w, torch.arange(303, dtype=torch.float32) - 1.5)

    # sharded AND overlapped: a bucket's scatter goes out at the last gradient piece of its last parameter
    class Store96(Store):  # three 96-element parameters = three buckets of 96 (a multiple of 4 * world)
        def __init__(self):
            super().__init__()
            self.flat_g = torch.zeros(288)

        def units(self):
            return [(p, 96 * i, 96) for i, p in enumerate(self.params)]

    store, opt = Store96(), _FakeOpt()
    dp = DataParallel(store, opt, None, shard_optimizer=True, overlap=True, bucket_elems=96)
    dp.reducer.poison_stale = True
    assert len(dp.reducer.bounds) == 3
    order = []
    launch = dp.reducer.launch_bucket
    dp.reducer.launch_bucket = lambda i: (order.append(i), launch(i))[1]
    p0, p1, p2 = store.params
    for it in range(2):
        store.flat_g.zero_()
        order.clear()
        dp.begin_step()
        for p, i in ((p2, 2), (p0, 0), (p1, 1)):
            store.flat_g[96 * i:96 * i + 96] += (rank + 1) * (i + 1)
            dp._ready(p)
        dp.reduce_gradients()
        assert order == ([0, 1, 2] if it == 0 else [2, 0, 1]), order
        for a, b in opt.element_ranges:
            assert torch.equal(store.flat_g[a:b], torch.full((b - a,), 3.0 * (a // 96 + 1))), (a, b)

    # the explicit two-phase algorithm gives the all-reduce's result
    flat = torch.arange(1003, dtype=torch.float32) * (rank + 1)
    red = FlatGradReducer(flat, bucket_elems=256, mode="reduce_scatter_all_gather")
    red.reduce_all()
    assert torch.equal(flat, torch.arange(1003, dtype=torch.float32) * 3)
    with pytest.raises(ValueError):
        DataParallel(Store(), _FakeOpt(), None, algorithm="ring")
    with pytest.raises(ValueError):
        FlatGradReducer(flat, 256, comm_dtype=torch.bfloat16, mode="reduce_scatter")

    # FusedAdamW.state_dict() under sharding: the moments of the other ranks' stripes are gathered first
    class PStore(Store):
        group_ranges = {0: (0, 200), 3: (200, 303)}

    store = PStore()
    opt = FusedAdamW(store, lr=1e-3)
    dp = DataParallel(store, opt, None, shard_optimizer=True, bucket_elems=128)
    for a, b in opt.element_ranges:
        opt.m[a:b] = 1.0 + rank
        opt.v[a:b] = 10.0 + rank
    sd = opt.state_dict()
    want_m = torch.zeros(303)
    for r in range(world):
        for (oa, ob), _, _ in [__import__("d2r_amd.dp", fromlist=["stripe_bounds"]).stripe_bounds(a, b, r, world) for a, b in dp.reducer.bounds]:
            want_m[oa:ob] = 1.0 + r
    want_m[296:] = 1.0 + rank  # (the tail is every rank's own)
    assert torch.equal(sd["m"], want_m) and sd["loss_scale"] == 1.0


class _Child(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(100))


class _Parent(torch.nn.Module):
    """Uses child.weight directly: the child never runs through __call__, so its pre-hook never registers the counting hook."""

    def __init__(self):
        super().__init__()
        self.child = _Child()
        self.own = torch.nn.Parameter(torch.ones(100))

    def forward(self, x):
        return (x * self.child.weight * self.own).sum()


def _hook_hole(rank):
    """A parameter whose owner never ran in the calibration step, whose gradient is reported by a sink (so its bucket goes out
    early), and that autograd accumulates into as well: without the guard installed at the end of calibration the second piece
    could land after the bucket's collective — it must raise.  A parameter nobody reports at all only keeps its bucket back."""
    from d2r_amd.dp import DataParallel
    for sink_reports in (False, True):
        model = _Parent()

        class Store:
            flat_g = torch.zeros(200)
            dead = []

            def units(self):
                return [(model.own, 0, 100), (model.child.weight, 100, 100)]

        model.own.grad = Store.flat_g[0:100].view(100)
        model.child.weight.grad = Store.flat_g[100:200].view(100)
        dp = DataParallel(Store(), _FakeOpt(), model, overlap=True, bucket_elems=100)
        dp.begin_step()
        model(torch.full((100,), float(rank + 1))).backward()   # child.weight's autograd piece arrives without a counting hook
        if sink_reports:
            dp._ready(model.child.weight)                          # ... and a sink of some kernel reports the same tensor
        dp.reduce_gradients()                                      # calibration: everything is reduced after backward
        assert torch.equal(Store.flat_g, torch.full((200,), 3.0))
        Store.flat_g.zero_()
        dp.begin_step()
        if not sink_reports:
            model(torch.full((100,), float(rank + 1))).backward()  # never reported: its bucket simply waits for the end of backward
            assert not dp._launched[1]
            dp.reduce_gradients()
            assert torch.equal(Store.flat_g, torch.full((200,), 3.0))
            Store.flat_g.zero_()
        else:
            with pytest.raises(RuntimeError, match="owner did not run in the calibration step"):
                model(torch.ones(100)).backward()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from d2r_amd.dp import FlatGradReducer, shard_batch
        torch.manual_seed(0)
        torch.set_num_threads(2)
        # 1) bucket logic on a ragged flat buffer
        flat = torch.arange(1003, dtype=torch.float32) * (rank + 1)
        red = FlatGradReducer(flat, bucket_elems=256)
        assert len(red.bounds) == 4 and red.bounds[-1] == (768, 1003)
        red.reduce_all()
        assert torch.equal(flat, torch.arange(1003, dtype=torch.float32) * 3)
        # 2) sharding + gradient averaging on the routing module
        g = torch.Generator().manual_seed(1)
        full = (torch.randn(4, 6, 768, generator=g), torch.randn(4, 5, 768, generator=g), torch.randn(4, 6, 768, generator=g))
        shard = shard_batch(full, rank, world)
        assert shard[0].shape[0] == 2 and torch.equal(shard[0], full[0][rank * 2:(rank + 1) * 2])
        mine = _routing_grads(shard)
        red2 = FlatGradReducer(mine, bucket_elems=1 << 20)
        red2.reduce_all()
        avg = mine / world  # what the fused AdamW sees through grad_scale = 1/world
        if rank == 0:
            ref = (_routing_grads(shard_batch(full, 0, 2)) + _routing_grads(shard_batch(full, 1, 2))) / 2
            q.put(float((avg - ref).abs().max() / (ref.abs().max() + 1e-12)))
        with pytest.raises(ValueError):
            shard_batch((torch.zeros(3, 2),), rank, world)
        _overlap_readiness(rank)
        _bf16_buckets_and_sharded_optimizer(rank, world)
        _hook_hole(rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    rel = q.get()
    assert rel < 1e-6, rel


def test_dataparallel_world1_is_a_noop_without_process_group():
    from d2r_amd.dp import init_process_group_from_env
    os.environ.pop("WORLD_SIZE", None)
    assert init_process_group_from_env() == (0, 1)
