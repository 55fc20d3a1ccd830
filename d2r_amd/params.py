"""Flat parameter / gradient storage, the fused AdamW optimiser and the linear warm-up schedule.

Memory layout in HBM (one allocation each, 288 GB/GPU makes this trivially affordable):
    flat_w  fp32 [n_live]   master weights, every live parameter is a view into it
    flat_g  fp32 [n_live]   gradients (``p.grad`` are views; zeroed once per step by one memset; the dW GEMMs
                            accumulate straight into it — see functional._Linear.backward)
    flat_m, flat_v fp32     AdamW moments
    flat_lp 16-bit [n_live] bf16 / fp16 shadow of the weights (16-bit compute dtypes only), rewritten by the AdamW kernel
Live parameters are ordered by the reference's four optimiser groups (modules/train.py:287-322) so each group is
one contiguous range = one kernel launch per group.  The 52.6 M parameters that never receive a gradient in the
reference stay outside the store and are never updated nor all-reduced (AdamW skips ``grad is None`` there too).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import _lib
from .functional import _stream

ALIGN = 8  # elements: 32 B for fp32, 16 B for the bf16 shadow


def is_dead_param(name: str) -> bool:
    """Parameters that are never used by the reference's forward."""
    if "fusion_dense" in name or "vision_post_layernorm" in name or ".text_pooler." in name:
        return True
    if ".path_mapping." in name or name.endswith("itr_module.bn.weight") or name.endswith("itr_module.bn.bias"):
        return True
    return "CrossModalAlignment.fc_1" in name or "CrossModalAlignment.fc_2" in name


def group_of(name: str) -> int:
    """0 other, 1 text (bert lr), 2 vision (vit lr), 3 fc head — modules/train.py:287-322."""
    if name.startswith("fc"):
        return 3
    if "text" in name:
        return 1
    if "vision" in name:
        return 2
    return 0


class FusedLinear:
    """Several nn.Linear modules that read the same input, served by ONE GEMM over their back-to-back weights
    ([sum N, K] view of the flat buffer).  Created by ParamStore for the groups a model lists in fusion_groups()."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, n_members: int):
        self.weight, self.bias, self.n = weight, bias, n_members

    def _wc(self, cdtype):
        return self.weight if cdtype == torch.float32 else self.weight._d2r_lp

    def __call__(self, x: torch.Tensor, cdtype: torch.dtype, act: int = 0):
        """Every member reads the SAME x: one GEMM with N = sum of the members' outputs."""
        from . import functional as F
        return F.linear(x, self.weight, self.bias, self._wc(cdtype), act)

    def grouped(self, x: torch.Tensor, cdtype: torch.dtype, act: int = 0, x_gm: bool = False):
        """Member g reads ITS OWN slice of x ([B, G*K] columns or group-major [G, B, K]): one batched GEMM."""
        from . import functional as F
        return F.grouped_linear(x, self.weight, self.bias, self._wc(cdtype), self.n, act, x_gm)


class ParamStore:
    def __init__(self, model: torch.nn.Module, compute_dtype: torch.dtype = torch.float32):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("model has no trainable parameters")
        device = named[0][1].device
        if device.type != "cuda":
            raise _lib.D2RError("ParamStore needs the model on the GPU (model.to('cuda') first)")
        self.compute_dtype = compute_dtype
        self.device = device
        self.dead = [(n, p) for n, p in named if is_dead_param(n)]
        live = sorted([(n, p) for n, p in named if not is_dead_param(n)], key=lambda np_: group_of(np_[0]))
        for n, p in live:
            if group_of(n) in (1, 2) and "text" in n and "vision" in n:
                raise ValueError(f"parameter {n!r} would fall into two optimiser groups")
        # same-input projections (q|k|v, k|v) are laid out back to back so one GEMM serves them (FusedLinear)
        fusions = list(model.fusion_groups()) if hasattr(model, "fusion_groups") else []
        # Groups may NEST (the k|v pair of one cross-modal alignment inside the module-wide run of every k|v projection of
        # `other`): the largest group fixes the layout, a group wholly inside an earlier one only gets views of it.
        fusions.sort(key=lambda f: -len(f[2]))
        first_of, member_ids, tight, nested = {}, set(), set(), set()
        for gi, (owner, key, linears) in enumerate(fusions):
            ws, bs = [l.weight for l in linears], [l.bias for l in linears]
            ids = [id(t) for t in ws + bs]
            if all(i in member_ids for i in ids):
                nested.add(gi)
                continue
            assert not any(i in member_ids for i in ids), "fusion groups may nest but not overlap partially"
            first_of[id(ws[0])] = ws + bs
            member_ids.update(ids)
            tight.update(id(t) for t in ws[:-1] + bs[:-1])  # members are packed without alignment padding
        name_of = {id(p): n for n, p in live}
        ordered = []
        for n, p in live:
            if id(p) in first_of:
                ordered.extend((name_of[id(t)], t) for t in first_of[id(p)])
            elif id(p) not in member_ids:
                ordered.append((n, p))
        assert len(ordered) == len(live), "fusion groups must consist of live parameters of one optimiser group"
        self.entries = []  # (name, param, offset, numel, group)
        off = 0
        self.group_ranges: Dict[int, List[int]] = {}
        offset_of = {}
        for n, p in ordered:
            g = group_of(n)
            if g not in self.group_ranges:
                self.group_ranges[g] = [off, off]
            self.entries.append((n, p, off, p.numel(), g))
            offset_of[id(p)] = off
            off = off + p.numel() if id(p) in tight else (off + p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.group_ranges[g][1] = (off + ALIGN - 1) // ALIGN * ALIGN
        for g, (a, b) in self.group_ranges.items():  # groups must be contiguous ranges
            assert all(a <= o < b for _, _, o, _, gg in self.entries if gg == g)
        self.n = off
        self.flat_w = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_lp = torch.zeros(off, dtype=compute_dtype, device=device) if compute_dtype in (torch.bfloat16, torch.float16) else None
        self.lp_dtype = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}.get(compute_dtype, _lib.BF16)
        with torch.no_grad():
            for n, p, o, k, g in self.entries:
                w = self.flat_w[o:o + k].view(p.shape)
                w.copy_(p.data)
                p.data = w
                gv = self.flat_g[o:o + k].view(p.shape)
                p.grad = gv
                p._d2r_grad = gv
                p._d2r_lp = None if self.flat_lp is None else self.flat_lp[o:o + k].view(p.shape)
        for n, p in self.dead:
            p.requires_grad_(False)
        # fused leaves: autograd leaves aliasing the members' storage; their gradient sinks alias the members' grads
        self.fused = []  # (leaf tensor, offset, numel, member ids): the top-level groups, i.e. the units of the flat buffer
        self.subleaves = []  # (leaf of a nested group, the top-level leaf whose storage it views)
        top = []  # (weight leaf, bias leaf, weight range, bias range) of the top-level groups
        for gi, (owner, key, linears) in enumerate(fusions):
            ws, bs = [l.weight for l in linears], [l.bias for l in linears]
            K = ws[0].shape[1]
            nrows = sum(w.shape[0] for w in ws)
            ow, ob = offset_of[id(ws[0])], offset_of[id(bs[0])]
            assert all(offset_of[id(w)] == ow + sum(x.numel() for x in ws[:i]) for i, w in enumerate(ws)), "weights not adjacent"
            assert all(offset_of[id(b)] == ob + sum(x.numel() for x in bs[:i]) for i, b in enumerate(bs)), "biases not adjacent"
            fw = self.flat_w[ow:ow + nrows * K].view(nrows, K).detach().requires_grad_(True)
            fb = self.flat_w[ob:ob + nrows].detach().requires_grad_(True)
            fw._d2r_grad = self.flat_g[ow:ow + nrows * K].view(nrows, K)
            fw.grad = fw._d2r_grad
            fb.grad = self.flat_g[ob:ob + nrows]
            fb._d2r_grad = fb.grad
            fw._d2r_lp = None if self.flat_lp is None else self.flat_lp[ow:ow + nrows * K].view(nrows, K)
            if not isinstance(getattr(owner, "_fused", None), dict):
                owner._fused = {}
            owner._fused[key] = FusedLinear(fw, fb, len(linears))
            if gi in nested:  # readiness of its gradient is reported for the enclosing unit (d2r_amd.dp)
                pw, pb = next((tw, tb) for tw, tb, (a, b), _ in top if a <= ow < b)
                self.subleaves += [(fw, pw), (fb, pb)]
                continue
            top.append((fw, fb, (ow, ow + nrows * K), (ob, ob + nrows)))
            self.fused.append((fw, ow, nrows * K, [id(w) for w in ws]))
            self.fused.append((fb, ob, nrows, [id(b) for b in bs]))
        self.refresh_lowp()

    def units(self):
        """(tensor that reports gradient readiness, offset, numel): fused leaves replace their members."""
        hidden = set(i for _, _, _, ids in self.fused for i in ids)
        out = [(p, o, k) for _, p, o, k, _ in self.entries if id(p) not in hidden]
        out.extend((t, o, k) for t, o, k, _ in self.fused)
        return out

    # -- 16-bit shadow -----------------------------------------------------------------------------
    def refresh_lowp(self):
        """Re-derives the 16-bit shadow from the fp32 masters (after load_state_dict / broadcast)."""
        if self.flat_lp is not None:
            _lib.call("d2r_cast", _lib.F32, self.flat_w.data_ptr(), self.lp_dtype, self.flat_lp.data_ptr(), self.n, _stream())

    def zero_grad(self):
        from .functional import wgrad_join
        wgrad_join()  # weight-gradient GEMMs of the step may still be accumulating on their side streams
        self.flat_g.zero_()  # one memset node

    def live_numel(self) -> int:
        return sum(k for _, _, _, k, _ in self.entries)


class FusedAdamW:
    """torch.optim.AdamW semantics (betas 0.9/0.999, eps 1e-8, decoupled weight decay) as one HIP launch per
    parameter group over the flat buffers (K14)."""

    def __init__(self, store: ParamStore, lr: float, fc_lr: float = 5e-2, weight_decay: float = 1e-2,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.store = store
        self.betas, self.eps = betas, eps
        self.m = torch.zeros_like(store.flat_w)
        self.v = torch.zeros_like(store.flat_w)
        self.step_count = 0
        self.grad_scale = 1.0  # 1/world_size under data parallelism (gradients are SUM-reduced)
        self.loss_scale = 1.0  # > 1 after enable_loss_scaling() (fp16 compute dtype)
        self._scaler = None
        # sharded data parallelism (d2r_amd.dp, shard_optimizer=True): this rank updates these element ranges only (its stripe of
        # every gradient bucket + the bucket tails); shard_gather(buffer) publishes the stripes of a flat buffer to all ranks
        self.element_ranges = None
        self.shard_gather = None
        self.dp_group = None
        names = {0: "other", 1: "text", 2: "vision", 3: "fc"}
        self.param_groups = []
        for g, (a, b) in sorted(store.group_ranges.items()):
            base = fc_lr if g == 3 else lr
            self.param_groups.append(dict(name=names[g], range=(a, b), lr=base, initial_lr=base,
                                          weight_decay=weight_decay))
        if store.flat_w.is_cuda:  # created eagerly, not inside a stream capture
            self._hyper_buffers()

    def zero_grad(self, set_to_none: bool = False):
        self.store.zero_grad()

    # -- loss scaling for the fp16 compute dtype (the job torch.cuda.amp.GradScaler does for the reference's users) ----------
    def enable_loss_scaling(self, init_scale: float = 2.0 ** 14, growth_interval: int = 500, max_scale: float = 2.0 ** 24):
        """fp16 activation gradients underflow below 6e-8: the loss is multiplied by ``loss_scale`` before backward
        (``scale_loss``), the AdamW kernel divides it out again (it is folded into grad_scale), a device-side check of the
        reduced gradients drops a step whose gradients overflowed (no host round trip: the kernels read the flag), and the
        host learns of the overflow one step later, halves the scale then, and doubles it after ``growth_interval`` clean steps."""
        dev = self.store.flat_w.device
        self.loss_scale = float(init_scale)
        self._scaler = dict(flag=torch.zeros(1, dtype=torch.int32, device=dev), host=torch.zeros(1, dtype=torch.int32).pin_memory()
                            if dev.type == "cuda" else torch.zeros(1, dtype=torch.int32), event=None, good=0,
                            growth_interval=int(growth_interval), max_scale=float(max_scale), skipped=0, used_scale=self.loss_scale)
        return self

    def scale_loss(self, loss):
        return loss if self._scaler is None else loss * self.loss_scale

    def backward(self, loss):
        """``scale_loss(loss).backward()`` without the three tiny launches it costs on the serial stretch between forward and backward
        (the multiply, autograd's ones_like root gradient, the multiply's backward): the loss scale IS the root gradient, kept in a
        device scalar that is rewritten only when the scale changes.  Same gradients bit for bit (1.0 * scale == scale)."""
        if self._scaler is None or loss.dim() != 0 or not loss.is_cuda:
            self.scale_loss(loss).backward()
            return
        sc = self._scaler
        g = sc.get("root_grad")
        if g is None or sc.get("root_grad_value") != self.loss_scale or g.dtype != loss.dtype or g.device != loss.device:
            g = sc["root_grad"] = torch.full((), self.loss_scale, dtype=loss.dtype, device=loss.device)
            sc["root_grad_value"] = self.loss_scale
        loss.backward(gradient=g)

    def refresh_root_grad(self):
        """Before replaying a captured ``backward``: the graph holds the root gradient's address, so a changed loss scale is
        written into it in place."""
        sc = self._scaler
        if sc is not None and sc.get("root_grad") is not None and sc.get("root_grad_value") != self.loss_scale:
            sc["root_grad"].fill_(self.loss_scale)
            sc["root_grad_value"] = self.loss_scale

    def _scaler_before_step(self):
        """Consumes the previous step's overflow flag (its copy has long landed), then arms the check for this step."""
        self._scaler_consume()
        self._scaler_arm()

    def _scaler_consume(self):
        sc = self._scaler
        if sc["event"] is not None:
            sc["event"].synchronize()
            if int(sc["host"][0]) != 0:
                sc["skipped"] += 1
                sc["good"] = 0
                self.step_count -= 1  # the dropped step does not count towards the bias correction
                self.loss_scale = max(self.loss_scale * 0.5, 1.0)
            else:
                sc["good"] += 1
                if sc["good"] >= sc["growth_interval"]:
                    sc["good"] = 0
                    self.loss_scale = min(self.loss_scale * 2.0, sc["max_scale"])
            sc["event"] = None

    def _scaler_arm(self):
        sc = self._scaler
        st = self.store
        sc["flag"].zero_()
        _lib.call("d2r_grad_nonfinite", st.flat_g.data_ptr(), st.n, sc["flag"].data_ptr(), _stream())
        if self.element_ranges is not None:
            # sharded optimiser: a rank holds the REDUCED gradients of its own stripes only - an overflow in another rank's stripe
            # must drop the step here too, or the replicas diverge
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size(self.dp_group) > 1:
                dist.all_reduce(sc["flag"], op=dist.ReduceOp.MAX, group=self.dp_group)

    def _scaler_after_step(self):
        sc = self._scaler
        sc["host"].copy_(sc["flag"], non_blocking=True)
        sc["event"] = torch.cuda.Event()
        sc["event"].record()

    def _owned(self, rng):
        """The parts of a parameter group's element range this rank updates: all of it, or (sharded optimiser) its
        intersections with the rank's stripes and the bucket tails."""
        a, b = rng
        if b <= a:
            return []
        if self.element_ranges is None:
            return [(a, b)]
        out = []
        for lo, hi in self.element_ranges:
            x, y = max(a, lo), min(b, hi)
            if y > x:
                out.append((x, y))
        return out

    def step(self):
        from .functional import wgrad_join
        wgrad_join()
        used = self.loss_scale  # the scale this step's backward ran with (the host may change it below)
        skip = None
        if self._scaler is not None:
            self._scaler_before_step()
            skip = self._scaler["flag"].data_ptr()
        self.step_count += 1
        st = self.store
        for pg in self.param_groups:
            for a, b in self._owned(pg["range"]):
                lp = None if st.flat_lp is None else st.flat_lp.data_ptr() + 2 * a
                _lib.call("d2r_adamw_step", st.flat_w.data_ptr() + 4 * a, st.flat_g.data_ptr() + 4 * a,
                          self.m.data_ptr() + 4 * a, self.v.data_ptr() + 4 * a, lp, st.lp_dtype, b - a, pg["lr"], self.betas[0],
                          self.betas[1], self.eps, pg["weight_decay"], self.step_count, self.grad_scale / used, skip, _stream())
        if self._scaler is not None:
            self._scaler_after_step()

    # -- hipGraph-capturable stepping: per-step scalars come from a device array ------------------------
    def _hyper_buffers(self):
        if not hasattr(self, "_hyper_dev"):
            n = len(self.param_groups)
            self._hyper_cpu = torch.zeros(n, 4, dtype=torch.float32)  # pageable on purpose, see stage_hyper
            self._hyper_dev = torch.zeros(n, 4, dtype=torch.float32, device=self.store.flat_w.device)
        return self._hyper_cpu, self._hyper_dev

    def stage_hyper(self):
        """Call BEFORE replaying the captured optimiser launches: advances the step count and uploads {lr, bias corrections,
        grad_scale} of every group (one small async H2D copy on the current stream).

        With loss scaling the protocol of a replayed step is the eager one, launch for launch:
            opt.refresh_root_grad(); <replay forward+backward>; opt.stage_hyper(); <replay step_captured + zero_grad>;
            opt.after_replay()
        stage_hyper divides by the scale the backward just ran with, THEN consumes the previous step's overflow flag (which may
        halve the scale for the next step and takes the dropped step out of the count) - exactly what step() does.  Captured as
        two graphs, the host never waits for the device (the flag it reads is a step old); with everything in ONE graph
        stage_hyper comes before the replay and its flag read waits for the previous replay to finish (correct, same
        numbers, but the host cannot run ahead)."""
        used = self.loss_scale
        if self._scaler is not None:
            self._scaler_consume()
        self.step_count += 1
        cpu, dev = self._hyper_buffers()
        b1, b2 = (float(torch.tensor(b, dtype=torch.float32)) for b in self.betas)  # the f32 values the kernel sees
        bc1 = 1.0 - b1 ** self.step_count
        bc2s = math.sqrt(1.0 - b2 ** self.step_count)
        for i, pg in enumerate(self.param_groups):
            cpu[i, 0], cpu[i, 1], cpu[i, 2], cpu[i, 3] = pg["lr"], bc1, bc2s, self.grad_scale / used
        dev.copy_(cpu)  # pageable source: the runtime stages it before returning, so `cpu` may be rewritten at once

    def step_captured(self):
        """The launches recorded into a hipGraph (no host-side scalars).  With loss scaling on, the captured backward must have
        been ``opt.backward(loss)`` (root gradient = the scale's device scalar); the overflow check and the skip flag are part
        of the capture, and the host calls ``after_replay()`` behind every replay to fetch the flag."""
        from .functional import wgrad_join
        wgrad_join()
        st = self.store
        _, dev = self._hyper_buffers()
        skip = None
        if self._scaler is not None:
            if self.element_ranges is not None:
                raise RuntimeError("step_captured: the sharded optimiser's overflow flag needs a collective, which is not captured")
            self._scaler_arm()
            skip = self._scaler["flag"].data_ptr()
        for i, pg in enumerate(self.param_groups):
            a, b = pg["range"]
            if b <= a:
                continue
            lp = None if st.flat_lp is None else st.flat_lp.data_ptr() + 2 * a
            _lib.call("d2r_adamw_step_dev", st.flat_w.data_ptr() + 4 * a, st.flat_g.data_ptr() + 4 * a,
                      self.m.data_ptr() + 4 * a, self.v.data_ptr() + 4 * a, lp, st.lp_dtype, b - a, dev.data_ptr() + 16 * i,
                      self.betas[0], self.betas[1], self.eps, pg["weight_decay"], skip, _stream())

    def after_replay(self):
        """Behind a replay of a captured step: starts the copy of the overflow flag (no-op without loss scaling)."""
        if self._scaler is not None:
            self._scaler_after_step()

    def state_dict(self):
        """Under the sharded optimiser a rank's moments are current for its own stripes only: they are all-gathered first (a
        collective - every rank must call state_dict()), so that the result is the same complete state on every rank."""
        if self.element_ranges is not None and self.shard_gather is not None:
            self.shard_gather(self.m)
            self.shard_gather(self.v)
        sd = dict(m=self.m, v=self.v, step=self.step_count, lrs=[pg["lr"] for pg in self.param_groups], loss_scale=self.loss_scale)
        if self._scaler is not None:
            sd["scaler"] = dict(good=self._scaler["good"], skipped=self._scaler["skipped"])
        return sd

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        for pg, lr in zip(self.param_groups, sd["lrs"]):
            pg["lr"] = lr
        if self._scaler is not None and "loss_scale" in sd:  # an fp16 run resumes at the scale it had reached, not at 2^14
            self.loss_scale = float(sd["loss_scale"])
            self._scaler["used_scale"] = self.loss_scale
            self._scaler["good"] = int(sd.get("scaler", {}).get("good", 0))
            self._scaler["skipped"] = int(sd.get("scaler", {}).get("skipped", 0))


class LinearWarmupSchedule:
    """``transformers.get_linear_schedule_with_warmup`` (modules/train.py:326-328), including the reference's float
    ``num_warmup_steps = warmup_ratio * steps`` and LambdaLR's "lr(0) at construction, then step()" ordering."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: float, num_training_steps: int):
        self.opt, self.warm, self.total = optimizer, num_warmup_steps, num_training_steps
        self.last_epoch = 0
        self._apply()

    def factor(self, step: int) -> float:
        if step < self.warm:
            return float(step) / float(max(1, self.warm))
        return max(0.0, float(self.total - step) / float(max(1, self.total - self.warm)))

    def _apply(self):
        f = self.factor(self.last_epoch)
        for pg in self.opt.param_groups:
            pg["lr"] = pg["initial_lr"] * f

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [pg["lr"] for pg in self.opt.param_groups]
