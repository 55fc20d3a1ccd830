"""GPU: the training surface — fused AdamW + schedule against torch.optim.AdamW driven by the oracle's gradients with the
reference's parameter grouping (modules/train.py:287-328), MSDTrainer end to end on synthetic data (checkpoint with
the reference's key names, evaluate/test metrics), and the CLI counterpart of run.py."""
import logging
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny(dtype, dr=3, layers=1):
    from d2r_amd import modules as M
    from d2r_amd.config import TextConfig, VisionConfig, default_args
    tc = TextConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vc = VisionConfig(num_hidden_layers=layers, image_size=64, patch_size=32)
    args = default_args(DR_step=dr, compute_dtype=dtype, device="cuda:0", num_epochs=2, batch_size=4, warmup_ratio=0.0,
                        save_path=None, lr=3e-5)
    return M.UnimoModelF(args, vc, tc), args


def test_one_optimizer_step_matches_torch_adamw_on_oracle_gradients(gpu):
    from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore, group_of
    from oracle import d2r_oracle as O
    model, args = _tiny(torch.float32)
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    sd = O.seeded_state_dict(cfg, seed=21, router_bias="normal")
    model.load_state_dict(sd, strict=True)
    model.to(gpu).train()
    store = ParamStore(model, torch.float32)
    opt = FusedAdamW(store, lr=args.lr)
    sched = LinearWarmupSchedule(opt, 0.0, 10)
    ids, mask, tt, labels, images = O.synthetic_batch(cfg, 3, 12, seed=4)
    loss, _ = model(ids.to(gpu), mask.to(gpu), tt.to(gpu), labels.to(gpu), images.to(gpu))
    loss.backward()
    opt.step()
    sched.step()
    torch.cuda.synchronize()
    # reference optimiser on the oracle's gradients, same grouping rule
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    lo, _, _ = O.forward(osd, cfg, ids, mask, tt, labels, images, train=True)
    lo.backward()
    live = [(k, v) for k, v in osd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None]
    groups = [dict(params=[v for k, v in live if group_of(k) == g], lr=(5e-2 if g == 3 else args.lr), weight_decay=1e-2)
              for g in range(4)]
    topt = torch.optim.AdamW([g for g in groups if g["params"]])
    before = {k: v.detach().clone() for k, v in live}
    topt.step()
    new = model.state_dict()
    checked = bad = 0
    for k, v in live:
        lr = 5e-2 if group_of(k) == 3 else args.lr
        d_ref = (v.detach() - before[k])
        d_got = new[k].cpu() - before[k]
        # step 1 of Adam moves every element by ~lr*sign(g): compare where the gradient is not at the noise floor
        sel = osd[k].grad.abs() > 1e-3 * osd[k].grad.abs().max()
        checked += int(sel.sum())
        bad += int(((d_got - d_ref).abs()[sel] > 0.05 * lr).sum())
    assert checked > 1e6 and bad / checked < 2e-3, (bad, checked)
    for n, p in store.dead:  # never touched by the optimiser (grad is None in the reference)
        assert torch.equal(p.detach().cpu(), sd[n]), n
    assert abs(opt.param_groups[0]["lr"] - args.lr * 0.9) < 1e-12  # linear decay after one scheduler step of 10


@pytest.mark.parametrize("lowp", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_trainer_end_to_end_synthetic(gpu, tmp_path, lowp):
    from d2r_amd.data import SyntheticMSDDataset, make_loader
    from d2r_amd.train import MSDTrainer
    from oracle import d2r_oracle as O
    torch.manual_seed(0)
    model, args = _tiny(lowp)
    args.save_path = str(tmp_path) + "/"
    args.lr = 1e-4
    mk = lambda n, seed, sh: make_loader(SyntheticMSDDataset(n, 16, 64, 3, seed=seed, num_image_tokens=5), 4, sh, 0, drop_last=sh)
    losses = []
    logger = logging.getLogger("trainer-test")

    class Catch(logging.Handler):
        def emit(self, rec):
            msg = rec.getMessage()
            if msg.startswith("step "):
                losses.append(float(msg.split("loss:")[1].split()[0]))

    logger.addHandler(Catch())
    logger.setLevel(logging.INFO)
    tr = MSDTrainer(train_data=mk(32, 1, True), dev_data=mk(8, 2, False), test_data=mk(8, 3, False), model=model,
                    args=args, logger=logger, writer=None)
    tr.train(None, None)
    assert len(losses) == 8 and all(l == l and abs(l) < 50 for l in losses), losses
    if lowp == torch.float16:  # the trainer switched loss scaling on by itself and no step overflowed at the initial scale
        assert tr.optimizer._scaler is not None and tr.optimizer.loss_scale >= 1024.0 and tr.optimizer._scaler["skipped"] == 0
    assert tr.samples_per_sec and tr.samples_per_sec > 0
    ck = os.path.join(str(tmp_path), "best_model.pth")
    assert os.path.exists(ck)
    saved = torch.load(ck, map_location="cpu")
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    assert set(saved) == set(O.param_spec(cfg)), "checkpoint keys differ from the reference's state-dict names"
    model2, _ = _tiny(torch.bfloat16)
    model2.load_state_dict(saved, strict=True)
    res = tr.evaluate(3)
    assert set(res) == {"eval_accuracy", "precision", "recall", "f_score", "global_step", "loss"}
    assert 0.0 <= res["eval_accuracy"] <= 1.0


def test_fp16_loss_scaling_matches_fp32_steps_and_survives_an_overflow(gpu):
    """fp16 compute dtype with the scaled loss (FusedAdamW.enable_loss_scaling): (i) three optimiser steps move the fp32
    master weights like the fp32 compute path does (same data, same init; AdamW's first steps are +-lr per element whatever
    the gradient's size, so elements whose gradient sits at the rounding floor flip freely: the comparison is the direction of
    the whole update and the loss curve); (ii) a step whose
    gradients overflow is dropped on the device (weights, moments, shadow unchanged), the host halves the scale one step
    later and training continues with finite weights."""
    from d2r_amd.params import FusedAdamW, ParamStore
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    torch.manual_seed(5)
    sd = {k: v.detach().clone() for k, v in _tiny(torch.float32)[0].state_dict().items()}  # the reference's construction-time init
    batch = [t.to(gpu) for t in O.synthetic_batch(cfg, 4, 12, seed=4)]
    final, stores = {}, {}
    for dtype in (torch.float32, torch.float16):
        model, args = _tiny(dtype)
        model.load_state_dict(sd, strict=True)
        model.to(gpu).set_compute_dtype(dtype).train()
        store = ParamStore(model, dtype)
        opt = FusedAdamW(store, lr=1e-4, fc_lr=1e-4)  # small steps: three of them stay in the regime where the two runs are comparable
        if dtype == torch.float16:
            opt.enable_loss_scaling(init_scale=4096.0, growth_interval=2)
        w0 = store.flat_w.clone()
        curve = []
        for _ in range(3):
            loss, _ = model(*batch)
            curve.append(float(loss))
            opt.scale_loss(loss).backward()
            opt.step()
            opt.zero_grad()
        with torch.no_grad():
            curve.append(float(model(*batch)[0]))
        torch.cuda.synchronize()
        final[dtype] = ((store.flat_w - w0).cpu(), curve)
        stores[dtype] = (model, store, opt)
    (d32, c32), (d16, c16) = final[torch.float32], final[torch.float16]
    cos = float((d16.double() @ d32.double()) / (d16.double().norm() * d32.double().norm()))
    print(f"[fp16 vs fp32, three AdamW steps] update cosine {cos:.4f}; loss curves fp32 {c32} fp16 {c16}")
    assert abs(c16[0] - c32[0]) <= 1e-3, (c16[0], c32[0])            # same weights, same batch: the forward agrees
    assert c16[-1] < c16[0] and abs(c16[-1] - c32[-1]) <= 0.15 * abs(c32[0] - c32[-1]) + 2e-2, (c16, c32)  # and it trains alike
    assert cos >= 0.6, cos
    model, store, opt = stores[torch.float16]
    assert opt.loss_scale == 8192.0, opt.loss_scale  # grew once after two clean steps (growth_interval = 2)
    assert store.flat_lp.dtype == torch.float16 and torch.equal(store.flat_lp, store.flat_w.half())  # the shadow follows the masters
    # (ii) an overflow: scale the loss so far up that fp16 activation gradients become inf
    opt.loss_scale = 2.0 ** 40
    before = (store.flat_w.clone(), opt.m.clone(), opt.v.clone(), store.flat_lp.clone())
    count = opt.step_count
    loss, _ = model(*batch)
    opt.scale_loss(loss).backward()
    assert not bool(torch.isfinite(store.flat_g).all()), "this loss scale was meant to overflow"
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()
    for a, b in zip(before, (store.flat_w, opt.m, opt.v, store.flat_lp)):
        assert torch.equal(a, b), "the overflowed step must change nothing"
    opt.loss_scale = 4096.0
    loss, _ = model(*batch)
    opt.scale_loss(loss).backward()
    opt.step()  # learns of the overflow here: halves the scale it finds, does not count the dropped step
    opt.zero_grad()
    torch.cuda.synchronize()
    assert opt._scaler["skipped"] == 1 and opt.loss_scale == 2048.0 and opt.step_count == count + 1, (opt._scaler, opt.loss_scale)
    assert not torch.equal(before[0], store.flat_w) and bool(torch.isfinite(store.flat_w).all())
    # (iii) opt.backward(loss) - the loss scale as the root gradient - gives the gradients of scale_loss(loss).backward() bit for bit,
    #       also after the scale has changed
    for scale in (2048.0, 512.0):
        opt.loss_scale = scale
        grads = []
        for fused in (False, True):
            store.zero_grad()
            loss, _ = model(*batch)
            if fused:
                opt.backward(loss)
            else:
                opt.scale_loss(loss).backward()
            torch.cuda.synchronize()
            grads.append(store.flat_g.clone())
        assert bool(grads[0].abs().max() > 0) and torch.equal(grads[0], grads[1]), scale


def test_cli_smoke(gpu, tmp_path):
    cmd = [sys.executable, "-m", "d2r_amd.run", "--num_epochs", "1", "--train_samples", "16", "--eval_samples", "8",
           "--batch_size", "8", "--encoder_layers", "1", "--image_size", "64", "--max_seq", "16", "--num_workers", "0",
           "--save_path", str(tmp_path) + "/", "--dtype", "bf16"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Test Eval results" in r.stderr and os.path.exists(os.path.join(str(tmp_path), "best_model.pth"))


def _run_steps(gpu, sd, batch, use_graph, use_streams, n=4, layers=1, dtype=torch.bfloat16, init_scale=2.0 ** 10):
    from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore
    model, args = _tiny(dtype, layers=layers)
    model.load_state_dict(sd, strict=True)
    model.to(gpu).train()
    model.set_compute_dtype(dtype)
    model.model.use_streams = use_streams
    store = ParamStore(model, dtype)
    opt = FusedAdamW(store, lr=1e-3)
    if dtype == torch.float16:
        opt.enable_loss_scaling(init_scale=init_scale)
    sched = LinearWarmupSchedule(opt, 2, 10)
    losses = []
    if not use_graph:
        for _ in range(n):
            loss, _ = model(*batch)
            opt.scale_loss(loss).backward()
            opt.step()
            sched.step()
            opt.zero_grad()
            losses.append(float(loss))
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm caches/workspaces; no optimiser step, so the weights are untouched
            loss, _ = model(*batch)
            opt.backward(loss)
            opt.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g, g_opt = torch.cuda.CUDAGraph(), None
        with torch.cuda.graph(g):
            loss, _ = model(*batch)
            opt.backward(loss)  # fp16: the root gradient is the loss scale's device scalar
            if use_graph != "split":
                opt.step_captured()
                opt.zero_grad()
        if use_graph == "split":  # forward+backward and the optimiser as two graphs: the host never waits for the device
            g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_opt):
                opt.step_captured()
                opt.zero_grad()
        for _ in range(n):
            opt.refresh_root_grad()
            if g_opt is None:
                opt.stage_hyper()
                g.replay()
            else:
                g.replay()
                opt.stage_hyper()
                g_opt.replay()
            opt.after_replay()
            sched.step()
            losses.append(float(loss))
    torch.cuda.synchronize()
    if dtype == torch.float16:
        opt._scaler_consume()  # the last step's flag
        losses.append(("skipped", opt._scaler["skipped"], "scale", opt.loss_scale, "count", opt.step_count))
    return losses, store.flat_w.clone(), opt.m.clone(), opt.v.clone(), [(n, o, k) for n, _, o, k, _ in store.entries]


@pytest.mark.parametrize("init_scale", [2.0 ** 10, 2.0 ** 20])
def test_fp16_graph_replay_scales_the_loss_and_skips_like_eager(gpu, init_scale):
    """fp16 steps replayed from a hipGraph carry the whole loss-scaling protocol: scaled root gradient (device scalar, rewritten
    when the scale moves), overflow check and the AdamW skip flag inside the capture, the host halving one step late.  Weights,
    moments, losses, the number of skipped steps and the final scale equal eager stepping bit for bit - at a scale that never
    overflows and at 2^20, where the first steps do (the scale walks down until the gradients fit)."""
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    sd = O.seeded_state_dict(cfg, seed=5, router_bias="normal")
    batch = tuple(t.to(gpu) for t in O.synthetic_batch(cfg, 4, 12, seed=9))
    base = _run_steps(gpu, sd, batch, False, True, n=9, dtype=torch.float16, init_scale=init_scale)
    skipped = base[0][-1][1]
    assert (skipped > 0) == (init_scale > 2.0 ** 15) and skipped < 8, base[0][-1]
    assert base[0][-2] != base[0][0]  # it trains
    for how in (True, "split"):
        got = _run_steps(gpu, sd, batch, how, True, n=9, dtype=torch.float16, init_scale=init_scale)
        assert got[0] == base[0], (how, got[0], base[0])
        for what, a, b in zip(("w", "m", "v"), got[1:4], base[1:4]):
            assert torch.equal(a, b), (how, what)


def test_streams_and_graph_replay_are_bit_identical_to_single_stream_eager(gpu):
    """The two-stream forward/backward, and fwd+bwd+AdamW+zero_grad replayed from one hipGraph (device-side
    hyper-parameters, d2r_adamw_step_dev), must give exactly the weights of single-stream eager launches: same kernels,
    same operands, deterministic reductions - only the launch mechanism differs."""
    from oracle import d2r_oracle as O
    cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32)
    sd = O.seeded_state_dict(cfg, seed=5, router_bias="normal")
    batch = tuple(t.to(gpu) for t in O.synthetic_batch(cfg, 4, 12, seed=9))
    base = _run_steps(gpu, sd, batch, use_graph=False, use_streams=False)
    assert base[0][-1] != base[0][0]
    for use_graph, use_streams in ((False, False), (False, True), (True, False), (True, True)):
        got = _run_steps(gpu, sd, batch, use_graph, use_streams)
        tag = f"graph={use_graph} streams={use_streams}"
        assert got[0] == base[0], (tag, got[0], base[0])
        for what, a, b in zip(("w", "m", "v"), got[1:4], base[1:4]):
            if not torch.equal(a, b):
                bad = (a != b).nonzero().flatten()
                names = sorted({n for n, o, k in got[4] if ((bad >= o) & (bad < o + k)).any()})
                raise AssertionError(f"{tag}: {what} differs in {bad.numel()} elements of {names[:12]} ({len(names)} tensors)")


def test_data_parallel_two_ranks_on_one_gpu(gpu, tmp_path):
    """World size 2 with both ranks on cuda:0 (gloo backend — RCCL refuses two ranks per device): broadcast, sharded batch,
    bucketed gradient all-reduce without and with overlap (bucket readiness reported by the kernels' gradient sinks, the
    deferred grouped weight gradients and the whole-layer C calls), AdamW with grad_scale 1/2.  Overlap must not change a
    single bit, and the two replicas must stay identical."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "probes", "dp_two_ranks_one_gpu.py"), str(tmp_path)]
    # A hang (deadlocked collective, a bucket that never becomes ready) must FAIL: the probe dumps every thread's
    # traceback and exits after 150 s (faulthandler), and whatever is left after 240 s is killed by process group.
    import signal
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                            env=dict(os.environ, D2R_PROBE_DUMP_S="150", D2R_PROBE_VERBOSE="1"))
    try:
        out, err = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail("the two data-parallel ranks did not finish in 300 s (usually 10-20 s): hang.\n--- stdout\n"
                    + out[-3000:] + "\n--- stderr\n" + err[-6000:])
    assert proc.returncode == 0, "two-rank probe failed (a faulthandler dump below means it hung for 150 s)\n--- stdout\n" \
        + out[-3000:] + "\n--- stderr\n" + err[-6000:]
    for rank in (0, 1):
        res = torch.load(os.path.join(str(tmp_path), f"rank{rank}.pt"))
        assert res["finite"] and all(l == l for l in res["losses"]), res
        assert res["same_modes"], ("overlapped all-reduce changed the result: %d differing elements in %d tensors, buckets %s of %d, "
                                   "max |diff| %.3e, first tensors %s" % (res["n_diff"], res["n_bad"], res["bad_buckets"], len(res["bounds"]),
                                                                          res["max_abs_diff"], res["bad"][:6])
                                   + "\nfirst divergence of the reduced gradients: %s" % (res.get("first_grad_divergence"),))
        assert res["same_ranks"], "replicas diverged"
        # reduce-scatter + all-gather with AdamW on this rank's slice: the same sums, the same update -> the same weights
        assert res["same_shard"], "the sharded optimiser (reduce-scatter / all-gather) changed the result"
        assert res["same_shard_overlap"], "the sharded optimiser with overlapped per-bucket scatters changed the result"
        assert res["same_rs_ag"], "algorithm='reduce_scatter_all_gather' changed the result"
        # gradient buckets sent as bf16: a rounded sum, same direction of the three-step update
        assert res["cos_bf16"] >= 0.9 and all(l == l for l in res["losses_bf16"]), (res["cos_bf16"], res["losses_bf16"])


def test_rccl_collectives_execute_on_one_rank(gpu, tmp_path):
    """The RCCL code path on the one GPU of the box: a process group of ONE rank over backend "nccl" (= RCCL) with every collective
    issued anyway (DataParallel(single_rank_collectives=True)) - bucketed all_reduce without and with overlap, the in-place
    reduce_scatter_tensor (output stripe aliasing its position in the input) + all_gather_into_tensor of the two-phase algorithm, the
    sharded optimiser (reduce-scatter, AdamW on the rank's stripes, all-gather of the weights) plain and overlapped on the
    communication stream, bf16 buckets.  A sum over one rank is the identity: every mode must reproduce the plain step bit for bit
    (bf16 buckets: the gradient rounded to bf16, same direction of the update).  The rank is a fresh process started before any GPU
    call (RCCL initialisation, HSA_ENABLE_IPC_MODE_LEGACY=0)."""
    import signal
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "probes", "dp_two_ranks_one_gpu.py"), str(tmp_path)]
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                            env=dict(os.environ, D2R_PROBE_DUMP_S="150", D2R_PROBE_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        out, err = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail("the RCCL rank did not finish in 300 s: hang.\n--- stdout\n" + out[-3000:] + "\n--- stderr\n" + err[-6000:])
    assert proc.returncode == 0, "single-rank RCCL probe failed\n--- stdout\n" + out[-3000:] + "\n--- stderr\n" + err[-6000:]
    res = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    assert res["finite"] and all(l == l for l in res["losses"]), res
    assert res["same_modes"], "overlapped RCCL all-reduce buckets changed the result: %s" % (res["bad"][:6],)
    assert res["same_rs_ag"], "in-place reduce_scatter_tensor + all_gather_into_tensor (one rank) changed the result"
    assert res["same_shard"] and res["same_shard_overlap"], "the sharded optimiser over RCCL (one rank) changed the result"
    assert res["cos_bf16"] >= 0.9 and all(l == l for l in res["losses_bf16"]), (res["cos_bf16"], res["losses_bf16"])


@pytest.mark.parametrize("lowp", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_fifty_step_convergence_of_the_16_bit_paths_follows_the_fp32_path(gpu, lowp):
    """Training fidelity of the benchmarked compute mode beyond single-step gradient statistics: the same model (2 + 2 encoder layers,
    DR_step 3, default init), the same LEARNABLE synthetic data (the label is a function of the first text token and of the image's
    mean brightness; 16 batches of 8 visited in the same order) and the same AdamW settings, trained for 50 steps on the fp32 HIP path
    (the parity reference: its gradients equal the oracle's to 1e-6) and on the 16-bit path (fp16 with the dynamic loss scale, bf16).
    Checked: both runs learn (mean loss of the last ten steps well below the first ten), no fp16 step is dropped after the scale has
    settled, the 16-bit run reaches every level of the descent within four steps of the fp32 run, and the plateaus agree."""
    from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore
    torch.manual_seed(7)
    model0, _ = _tiny(torch.float32, layers=2)
    sd = {k: v.detach().clone() for k, v in model0.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    batches = []
    for _ in range(16):
        ids = torch.randint(1000, 30000, (8, 16), generator=g)
        ids[:, 0] = 101
        key = torch.randint(0, 3, (8,), generator=g)
        ids[:, 1] = 2000 + key * 7919            # the class is written into the second token ...
        images = torch.randn(8, 3, 64, 64, generator=g) + (key.float().view(8, 1, 1, 1) - 1.0)  # ... and into the image's brightness
        batches.append(tuple(t.to(gpu) for t in (ids, torch.ones(8, 16, dtype=torch.long), torch.zeros(8, 16, dtype=torch.long), key, images)))

    def run(dtype):
        model, _ = _tiny(dtype, layers=2)
        model.load_state_dict(sd, strict=True)
        model.to(gpu).train()
        model.set_compute_dtype(dtype)
        store = ParamStore(model, dtype)
        opt = FusedAdamW(store, lr=1e-4, fc_lr=5e-3)
        if dtype == torch.float16:
            opt.enable_loss_scaling(init_scale=2.0 ** 12)
        sched = LinearWarmupSchedule(opt, 5, 60)
        losses = []
        for step in range(50):
            ids, mask, tt, labels, images = batches[step % len(batches)]
            loss, _ = model(ids, mask, tt, labels, images)
            opt.backward(loss)
            opt.step()
            sched.step()
            opt.zero_grad()
            losses.append(float(loss))
        torch.cuda.synchronize()
        skipped = opt._scaler["skipped"] if opt._scaler is not None else 0
        return losses, skipped

    ref, _ = run(torch.float32)
    got, skipped = run(lowp)
    head, tail = sum(ref[:10]) / 10, sum(ref[-10:]) / 10
    print(f"    [{str(lowp)[6:]}] fp32 loss {head:.4f} -> {tail:.4f}; 16-bit {sum(got[:10]) / 10:.4f} -> {sum(got[-10:]) / 10:.4f}; "
          f"max |difference| {max(abs(a - b) for a, b in zip(got, ref)):.4f}; dropped steps {skipped}")
    assert all(l == l and abs(l) < 1e4 for l in got), got
    assert tail < 0.75 * head, f"the fp32 run did not learn: {head:.4f} -> {tail:.4f}"
    assert sum(got[-10:]) / 10 < 0.75 * (sum(got[:10]) / 10), "the 16-bit run did not learn"
    assert skipped <= 2, f"{skipped} fp16 steps dropped by the overflow check"
    # How close.  The loss falls from ~1.0 to ~-0.1 (the JS term enters with a negative weight) within these 50 steps, steeply in the
    # middle (0.1-0.2 per step).  Rounding differences compound through the updates into a LAG of the descent, and the lag is itself
    # sensitive to rounding-level changes of the arithmetic: on MI355X the fp16 run trails the fp32 run by 1-2 steps, the bf16 run by
    # 1-3 (measured with two builds that differ only in the summation order of one K = 13,824 product: pointwise differences 0.16-0.28,
    # ten-step window means up to 0.08-0.13 in the second window, 0.002-0.003 on the plateau).  A pointwise band in the steep part
    # therefore measures the lag times the slope, not fidelity.  Checked instead:
    #   * the LAG: the step at which the trailing five-step mean first falls below 75 / 50 / 25 / 10 % of the drop is at most four
    #     steps later (or earlier) than in the fp32 run;
    #   * the plateau: the means of the last ten steps agree within 2 % of the starting loss;
    #   * no single step leaves the fp32 curve by more than 40 % of the starting loss (a blow-up, not a lag).
    scale = max(head, 0.5)
    worst = max(abs(a - b) for a, b in zip(got, ref))
    wins = [abs(sum(got[i:i + 10]) - sum(ref[i:i + 10])) / 10 for i in range(0, 50, 10)]
    curves = "\n    fp32   " + " ".join(f"{v:.3f}" for v in ref) + "\n    16-bit " + " ".join(f"{v:.3f}" for v in got)

    def first_below(curve, level):
        for i in range(len(curve)):
            w = curve[max(0, i - 4):i + 1]
            if sum(w) / len(w) <= level:
                return i
        return len(curve)

    lags = []
    for frac in (0.75, 0.5, 0.25, 0.1):
        level = tail + frac * (head - tail)
        lags.append(first_below(got, level) - first_below(ref, level))
    print(f"    lag of the 16-bit run at 75 / 50 / 25 / 10 % of the drop: {lags} steps; window means differ by {[round(w, 4) for w in wins]}")
    assert max(abs(l) for l in lags) <= 4, f"the 16-bit run lags the fp32 run by {lags} steps (fp32 starts at {head:.4f})" + curves
    assert wins[-1] <= 0.02 * scale, f"the plateaus differ by {wins[-1]:.4f}" + curves
    assert worst <= 0.4 * scale, f"16-bit loss curve leaves the fp32 curve: max |difference| {worst:.4f} (fp32 starts at {head:.4f})" + curves


def test_global_batch_exact_data_parallelism_matches_the_full_batch_oracle(gpu, tmp_path):
    """SURVEY 8e, the optional exact mode: two ranks (cuda:0, gloo) with half of a global batch of four each reproduce the ORACLE's
    loss, logits and gradients of the four samples together - BatchNorm1d(1) statistics of the GLAC cells over all ranks' samples
    (models/XModules.py:376,381), [B,B] similarity matrices and JS loss over the global batch (models/modeling_unimo.py:845-849) -
    in the fp32 mode (op-by-op path) tightly and in the fp16 mode (whole-module C calls with the bn_sync callback) within its
    rounding; the default local-statistics mode does not (control)."""
    import signal
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "probes", "dp_exact_two_ranks.py"), str(tmp_path)]
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                            env=dict(os.environ, D2R_PROBE_DUMP_S="150"))
    try:
        out, err = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail("the two ranks did not finish in 300 s: hang.\n--- stdout\n" + out[-3000:] + "\n--- stderr\n" + err[-6000:])
    assert proc.returncode == 0, "exact-mode probe failed\n--- stdout\n" + out[-3000:] + "\n--- stderr\n" + err[-6000:]
    print(out)
    for rank in (0, 1):
        res = torch.load(os.path.join(str(tmp_path), f"rank{rank}.pt"))
        f32, f16 = res["f32"], res["fp16"]
        assert f32["e_loss"] <= 2e-5 and f32["e_logit"] <= 2e-5 and f32["cos"] >= 0.9999 and abs(f32["norm_ratio"] - 1.0) <= 1e-3, f32
        assert f16["e_loss"] <= 1e-3 and f16["e_logit"] <= 1e-3 and f16["cos"] >= 0.98, f16
        assert res["local_mode_loss_gap"] >= 1e-4, ("the local-statistics mode reproduced the full-batch loss: vacuous test", res["local_mode_loss_gap"])
