"""Data parallelism: one process per GPU, minibatch sharded on dim 0, full replica per rank, gradients summed
with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI in buckets of the flat gradient buffer.

The reference has no distributed code at all; semantics are those DDP would give it:
local-batch BatchNorm statistics and local [b,b] JS loss per rank, gradients averaged.
The 1/world factor is folded into the fused AdamW kernel (``grad_scale``), so the collective is a plain SUM.

Buckets are contiguous slices of ``ParamStore.flat_g``.  With ``overlap=True`` a bucket's all-reduce is issued
on a side stream as soon as every parameter in it has its WHOLE gradient, overlapping with the rest of backward;
``finish()`` joins them before the optimiser step.  Gradient pieces report from autograd's post-accumulate hooks, from
the dW-sink callbacks of functional._Linear / the whole-layer C calls and from the deferred grouped launches; a
parameter shared by several call sites reports several times per step, so the first overlapped step only counts the
reports and later steps wait for the last one (DataParallel._ready).

Reduction algorithm (``algorithm=``): "all_reduce" (RCCL picks ring / tree itself) or "reduce_scatter_all_gather" — the two
phases written out per bucket (``ncclReduceScatter`` into the rank's stripe of the bucket, then ``ncclAllGather``), same bytes on
every xGMI link, but each phase is a separate collective the overlap can place.  ``shard_optimizer=True`` keeps only the first
phase for the gradients: every bucket is cut into `world` stripes, rank r receives the summed gradients of stripe r of EVERY
bucket (so the scatter of a bucket can go out as soon as the bucket is complete, overlapped with the rest of backward), runs
AdamW on its stripes (1/world of the optimiser's 30 bytes per parameter) and the updated fp32 weights are all-gathered bucket
by bucket.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .params import ParamStore


def init_process_group_from_env(backend: Optional[str] = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # The host driver of this pool exposes dmabuf IPC only: with the legacy IPC mode RCCL's intra-node transport setup fails in
    # hipIpcGetMemHandle ("invalid argument") before the first collective.  The launcher environment already exports 0; a
    # process started some other way (mp.spawn from a notebook, a bare `python bench.py --gpus N`) gets the same default here.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # D2R_DIST_BACKEND=gloo: rehearsal of the multi-rank control flow with several ranks on ONE GPU (RCCL refuses that)
    backend = backend or os.environ.get("D2R_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_batch(batch, rank: int, world: int):
    """Splits every tensor of a batch tuple on dim 0 into `world` equal shards and returns shard `rank`."""
    out = []
    for t in batch:
        if isinstance(t, torch.Tensor) and t.dim() > 0:
            n = t.shape[0]
            if n % world != 0:
                raise ValueError(f"global batch {n} is not divisible by world size {world}")
            k = n // world
            out.append(t[rank * k:(rank + 1) * k])
        else:
            out.append(t)
    return tuple(out)


ALGORITHMS = ("all_reduce", "reduce_scatter_all_gather")


def stripe_bounds(a: int, b: int, rank: int, world: int):
    """Bucket [a, b) cut into `world` 16-byte-aligned stripes: -> (own stripe, body = all stripes, tail).  The tail (fewer than
    4 * world elements, only where the bucket length is not a multiple of 4 * world) is all-reduced and owned by every rank."""
    per = (b - a) // (4 * world) * 4
    return (a + rank * per, a + (rank + 1) * per), (a, a + per * world), (a + per * world, b)


class FlatGradReducer:
    """Bucketed SUM reduction of a flat fp32 gradient buffer.

    mode "all_reduce": one all-reduce per bucket.  mode "reduce_scatter_all_gather": reduce-scatter into the rank's stripe of
    the bucket, then all-gather of the stripes (the explicit two-phase form of the same sum).  mode "reduce_scatter": the first
    phase only — afterwards a rank holds the summed gradients of its stripe (and of the bucket's tail) and the rest of the
    bucket is stale (sharded optimiser).  gloo has no reduce-scatter: the CPU rehearsal all-reduces the bucket instead, which
    leaves a superset of the same values.
    ``comm_dtype=torch.bfloat16`` sends every bucket as bf16 (half the bytes on the xGMI links; the sum over ranks is then formed
    in bf16 by the collective, the fp32 buffer gets the rounded result) - an optional trade of gradient precision for link time,
    off by default and only with mode "all_reduce"."""

    def __init__(self, flat_g: torch.Tensor, bucket_elems: int, group=None, comm_dtype: torch.dtype = torch.float32,
                 mode: str = "all_reduce"):
        self.flat_g, self.group = flat_g, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert mode in ("all_reduce", "reduce_scatter_all_gather", "reduce_scatter"), mode
        self.mode = mode
        n = flat_g.numel()
        if mode != "all_reduce":  # stripes of 4 * k elements: keep the buckets a multiple of 4 * world so that only the last has a tail
            q = 4 * self.world
            bucket_elems = max(q, bucket_elems // q * q)
        self.bounds = [(a, min(n, a + bucket_elems)) for a in range(0, n, bucket_elems)]
        self.stripes = [stripe_bounds(a, b, self.rank, self.world) for a, b in self.bounds]
        self.handles = []
        self.comm_stream = torch.cuda.Stream() if flat_g.is_cuda else None
        assert comm_dtype in (torch.float32, torch.bfloat16)
        if comm_dtype != torch.float32 and mode != "all_reduce":
            raise ValueError("bf16 gradient buckets are implemented for algorithm='all_reduce' only")
        self.comm_dtype = comm_dtype
        # one staging buffer per bucket (buckets are in flight together when the reduction overlaps with backward)
        self.staging = ([torch.empty(b - a, dtype=comm_dtype, device=flat_g.device) for a, b in self.bounds]
                        if comm_dtype != torch.float32 else None)
        self._unstage = []
        self._native_rs = None
        # Rehearsal aid (tests/test_dp_gloo.py): gloo has no reduce-scatter and all-reduces the bucket instead, which leaves the full
        # sum in every stripe - a reader of a stripe this rank does NOT own (stale under RCCL's reduce-scatter) would go unnoticed.
        # With poison_stale the non-owned stripes of mode "reduce_scatter" are overwritten with NaN after the reduction.
        self.poison_stale = False

    def _has_reduce_scatter(self) -> bool:
        if self._native_rs is None:
            self._native_rs = dist.get_backend(self.group) != "gloo"
        return self._native_rs

    def launch_bucket(self, i: int):
        a, b = self.bounds[i]
        view = self.flat_g[a:b]
        if self.comm_stream is not None:
            # A bucket holds gradients written from every compute stream of the step (the two encoder / routing
            # streams and the launching stream; optionally the weight-gradient side streams): the collective must be
            # ordered after all of them, not only after the stream the last readiness report came from.
            from . import functional as F
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for st in list(F._COMPUTE_STREAMS) + F.wgrad_streams():
                self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                self._issue(i, view)
        else:
            self._issue(i, view)

    def _issue(self, i: int, view: torch.Tensor):
        if self.mode != "all_reduce":
            (oa, ob), (ba, bb), (ta, tb) = self.stripes[i]
            g = self.flat_g
            if bb > ba:
                if self._has_reduce_scatter():
                    # in place: the output stripe aliases its own position inside the input (ncclReduceScatter's in-place form)
                    self.handles.append(dist.reduce_scatter_tensor(g[oa:ob], g[ba:bb], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    if self.mode == "reduce_scatter_all_gather":
                        # same communication stream, issued behind the scatter (collectives of one group execute in issue order)
                        self.handles.append(dist.all_gather_into_tensor(g[ba:bb], g[oa:ob], group=self.group, async_op=True))
                else:
                    self.handles.append(dist.all_reduce(g[ba:bb], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            if tb > ta:
                self.handles.append(dist.all_reduce(g[ta:tb], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self.staging is None:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        buf = self.staging[i]
        buf.copy_(view)  # fp32 -> bf16 on the communication stream
        self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._unstage.append((view, buf))

    def reduce_all(self):
        # reverse order = roughly reverse execution order (head and routing modules first)
        for i in reversed(range(len(self.bounds))):
            self.launch_bucket(i)
        self.finish()

    def finish(self):
        if self.comm_stream is not None and self._unstage:
            with torch.cuda.stream(self.comm_stream):
                for h in self.handles:
                    h.wait()
                for view, buf in self._unstage:
                    view.copy_(buf)
        else:
            for h in self.handles:
                h.wait()
            for view, buf in self._unstage:
                view.copy_(buf)
        self.handles.clear()
        self._unstage.clear()
        if self.poison_stale and self.mode == "reduce_scatter" and not self._has_reduce_scatter():
            ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else contextlib.nullcontext()
            with ctx:
                for (oa, ob), (ba, bb), _ in self.stripes:
                    self.flat_g[ba:oa] = float("nan")
                    self.flat_g[ob:bb] = float("nan")
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def gather_stripes(self, buf: torch.Tensor):
        """All ranks publish their stripes of `buf` (a tensor laid out like the gradient buffer: updated weights, optimiser
        moments): afterwards every rank holds all of it.  Tails are identical on all ranks already."""
        for (oa, ob), (ba, bb), _ in self.stripes:
            if bb > ba:
                dist.all_gather_into_tensor(buf[ba:bb], buf[oa:ob].clone(), group=self.group)


class DataParallel:
    def __init__(self, store: ParamStore, optimizer, model: torch.nn.Module, bucket_mb: int = 128, group=None,
                 overlap: bool = False, bucket_elems: Optional[int] = None, grad_comm_dtype: torch.dtype = torch.float32,
                 shard_optimizer: bool = False, algorithm: str = "all_reduce", single_rank_collectives: bool = False,
                 global_batch_exact: bool = False):
        """grad_comm_dtype: torch.bfloat16 sends the gradient buckets as bf16 (see FlatGradReducer).
        algorithm: "all_reduce" or "reduce_scatter_all_gather" (module docstring).
        shard_optimizer: reduce-scatter of every bucket + AdamW on this rank's stripes + all-gather of the updated weights
        (same bytes on the links as the all-reduce); works with overlap: a bucket's scatter goes out when the bucket is complete.
        single_rank_collectives: with an initialised process group of ONE rank, issue every collective anyway (a one-GPU box
        executes the RCCL code path - in-place reduce_scatter_tensor, all_gather_into_tensor, the communication stream's ordering -
        with results that must equal the plain step bit for bit: tests/test_gpu_trainer.py).  Off: world size 1 is a no-op.
        global_batch_exact: the three batch-coupled points of the reference (BatchNorm1d(1) of the GLAC cells, the [B,B] similarity
        matrices, the batch-softmax JS loss) see the GLOBAL batch (functional.DP_EXACT: two all-reduced doubles per BatchNorm and
        direction, all-gathers of `paths` [b, 36(DR-1)+6] and of the cls vectors [b, 768]); the averaged gradients are then the
        reference's gradients on the global batch.  Default: local-batch statistics and a local JS term per rank (DDP semantics)."""
        self.store, self.opt, self.model, self.group = store, optimizer, model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (bool(single_rank_collectives) and dist.is_initialized())
        from . import functional as F
        F.DP_EXACT = (group, self.world, self.rank) if (global_batch_exact and self.active) else None
        if algorithm not in ALGORITHMS:
            raise ValueError(f"algorithm must be one of {ALGORITHMS}, got {algorithm!r}")
        self.shard_optimizer = bool(shard_optimizer) and self.active
        self.algorithm = "reduce_scatter_all_gather" if self.shard_optimizer else algorithm
        mode = "reduce_scatter" if self.shard_optimizer else algorithm
        self.reducer = (FlatGradReducer(store.flat_g, bucket_elems or bucket_mb * (1 << 20) // 4, group, grad_comm_dtype, mode=mode)
                        if self.active else None)
        optimizer.grad_scale = 1.0 / self.world
        optimizer.dp_group = group
        if self.shard_optimizer:
            # this rank updates its stripe of every bucket, and every rank updates the (tiny) bucket tails
            optimizer.element_ranges = [r for own, _, tail in self.reducer.stripes for r in (own, tail) if r[1] > r[0]]
            optimizer.shard_gather = self.reducer.gather_stripes
        self.overlap = overlap and self.active
        self._pending: List[int] = []
        self._bucket_of = {}
        if self.overlap:
            self._install_hooks()

    # -- start-up: identical replicas ---------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0):
        if not self.active:
            return
        dist.broadcast(self.store.flat_w, src=src, group=self.group)
        for _, p in self.store.dead:
            dist.broadcast(p.data, src=src, group=self.group)
        for b in self.model.buffers():
            dist.broadcast(b, src=src, group=self.group)
        self.store.refresh_lowp()

    def sync_buffers(self):
        """BatchNorm running statistics are updated from each rank's LOCAL batch and drift apart after the initial
        broadcast (DDP would re-broadcast rank 0's every step); before evaluation / checkpointing they are replaced by
        their mean over ranks (integer buffers such as num_batches_tracked: rank 0's value)."""
        if not self.active:
            return
        for b in self.model.buffers():
            if b.is_floating_point():
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                b.div_(self.world)
            else:
                dist.broadcast(b, src=0, group=self.group)

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    # -- readiness counting for overlap ---------------------------------------------------------------
    # A parameter may receive its gradient in several pieces per step (a weight shared by two call sites, one piece
    # written on the spot and one by a deferred grouped launch at the end of backward, an autograd-accumulated piece next
    # to a sink-written one).  Every piece reports through _ready; the FIRST overlapped step only counts the reports per
    # parameter (its buckets are reduced after backward), later steps mark a parameter ready at its last expected
    # report.  A report that arrives after its bucket's collective was issued is a hard error, not a silent wrong sum.
    def _install_hooks(self):
        bounds = self.reducer.bounds
        self._count0 = [0] * len(bounds)
        for p, off, k in self.store.units():
            bi = next(i for i, (a, b) in enumerate(bounds) if a <= off < b)
            # a parameter straddling two buckets belongs to the later one (its tail is reduced there)
            be = next(i for i, (a, b) in enumerate(bounds) if a <= off + k - 1 < b)
            for i in range(bi, be + 1):
                self._count0[i] += 1
            self._bucket_of[id(p)] = (bi, be)
            p._d2r_ready_cb = self._ready
        for leaf, unit in getattr(self.store, "subleaves", ()):  # views of a nested fusion group report for the enclosing unit
            leaf._d2r_ready_cb = (lambda _t, unit=unit: self._ready(unit))
        # Autograd's post-accumulate hooks are registered LAZILY, from a forward pre-hook of the module that owns the tensor,
        # i.e. under the HIP stream that module runs on.  Registering a hook creates the tensor's AccumulateGrad node, and the
        # node accumulates on the stream that was current at its creation: registered here (launching stream), the in-place
        # `grad += piece` of a parameter produced on a branch stream would run on the launching stream, unordered against a
        # second piece that a sink-writing kernel adds on the branch stream - a read-modify-write race on the flat gradient
        # buffer (seen as an intermittent difference between overlapped and plain runs of the two-rank rehearsal).
        owner = {}
        for mod in (self.model.modules() if self.model is not None else ()):
            for t in mod.parameters(recurse=False):
                owner.setdefault(id(t), mod)
            for fz in (getattr(mod, "_fused", None) or {}).values():
                owner.setdefault(id(fz.weight), mod)
                owner.setdefault(id(fz.bias), mod)
        per_module = {}
        for p, _, _ in self.store.units():
            mod = owner.get(id(p))
            if mod is None:  # a tensor no module owns (unit tests drive the reducer with bare tensors): single-stream use
                p.register_post_accumulate_grad_hook(self._ready)
                continue
            per_module.setdefault(mod, []).append(p)
        self._hook_handles = []
        self._unhooked = {}  # module -> (fired box, its tensors): who has no autograd hook yet
        for mod, tensors in per_module.items():
            box = []

            def once(m, args, tensors=tensors, box=box):
                if not box:
                    box.append(True)
                    for t in tensors:
                        t.register_post_accumulate_grad_hook(self._ready)
            self._unhooked[mod] = (box, tensors)
            self._hook_handles.append(mod.register_forward_pre_hook(once))
        self._expect = None  # id(parameter) -> reports per step, learnt in the first overlapped step
        from . import functional as F
        F.EARLY_FLUSH = True  # queued weight gradients of the routing modules go out before the encoders' backward
        self.begin_step()

    def _close_hook_hole(self):
        """End of the calibration step.  A tensor whose owning module never ran through ``__call__`` (every Linear inside a
        whole-layer / whole-module C call, a child whose weight the parent reads directly) has no autograd hook: its gradient
        arrives through the sinks only, which report by themselves.  Should autograd ever accumulate a piece into such a tensor
        later, nobody would count it and its bucket could be reduced before the piece lands - so those tensors (the ones that
        reported from their sinks during calibration) get a hook now that turns the event into an error instead of an
        incomplete sum."""
        names = {id(e[1]): e[0] for e in getattr(self.store, "entries", ()) if len(e) >= 2}
        for mname, mod in (self.model.named_modules() if self.model is not None else ()):
            for key, fz in (getattr(mod, "_fused", None) or {}).items():
                names[id(fz.weight)], names[id(fz.bias)] = f"{mname}: fused group '{key}' weight", f"{mname}: fused group '{key}' bias"

        def guard(name):
            def unexpected(g):
                # (torch runs tensor hooks also when a node handed back None for the tensor - the anchor input of a whole-layer call,
                # a weight whose gradient the kernel wrote into the sink itself: nothing is accumulated then)
                if g is not None:
                    raise RuntimeError("data-parallel overlap: autograd accumulated a gradient into a parameter whose owner did not run in "
                                       f"the calibration step (its pieces were expected from the gradient sinks only): {name}; run with "
                                       "overlap=False")
            return unexpected

        for mod, (box, tensors) in self._unhooked.items():
            if not box:
                box.append(True)  # (the module's own pre-hook must not register the counting hook on top later)
                for t in tensors:
                    # A tensor nobody reported in the calibration step (e.g. the patch-embedding weight, used by its parent through
                    # a plain autograd op) keeps its bucket's count above zero for ever: that bucket is reduced after backward,
                    # which is safe.  The dangerous case is a tensor that DID report from its sinks: its bucket does go out early.
                    if id(t) in self._expect:
                        t.register_hook(guard(names.get(id(t), "<fused leaf>")))
        self._unhooked = {}

    def begin_step(self):
        if self.overlap:
            self._pending = list(self._count0)
            self._got = {}
            self._launched = [False] * len(self._count0)

    def _ready(self, p):
        if not self.overlap:
            return
        pid = id(p)
        n = self._got[pid] = self._got.get(pid, 0) + 1
        if self._expect is None:  # calibration step
            return
        want = self._expect.get(pid, 1)
        if n < want:
            return
        bi, be = self._bucket_of[pid]
        if n > want:
            if any(self._launched[i] for i in range(bi, be + 1)):
                raise RuntimeError("data-parallel overlap: a parameter received a gradient piece after its bucket's all-reduce was "
                                   "issued (the number of pieces per step changed); run with overlap=False")
            return
        for i in range(bi, be + 1):
            self._pending[i] -= 1
            if self._pending[i] == 0:
                self._launched[i] = True
                self.reducer.launch_bucket(i)

    # -- per step -------------------------------------------------------------------------------------
    def reduce_gradients(self):
        if not self.active:
            return
        if self.overlap:
            if self._expect is None:
                self._expect = dict(self._got)
                self._close_hook_hole()
            # buckets whose parameters did not all report (unused this step, or the calibration step) are reduced now
            for i, done in enumerate(self._launched):
                if not done:
                    self._launched[i] = True
                    self.reducer.launch_bucket(i)
            self.reducer.finish()
        else:
            self.reducer.reduce_all()

    def gather_parameters(self):
        """After optimizer.step() of a sharded step: every rank publishes its updated stripes of the fp32 weights; the 16-bit
        shadow is re-derived locally."""
        if not self.shard_optimizer:
            return
        self.reducer.gather_stripes(self.store.flat_w)
        self.store.refresh_lowp()
