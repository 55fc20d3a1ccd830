"""Global-batch-exact data parallelism (DataParallel(global_batch_exact=True)) with two ranks on cuda:0 over gloo, against the pinned
ORACLE run on the FULL batch (launched by tests/test_gpu_trainer.py through torch.distributed.run): each rank holds half of a global
batch of four; the BatchNorm1d(1) of every GLAC cell normalises with the statistics of all four samples (two all-reduced doubles per
BatchNorm and direction), the [B,B] similarity matrices and the JS loss span all four (all-gathered paths / cls vectors).  Checked per
compute dtype: the mean of the ranks' losses is the oracle's loss on the four samples, each rank's logits are the oracle's rows, and
the all-reduced, averaged gradient is the oracle's gradient.  fp32 takes the op-by-op path (exact BatchNorm through _SafGate), fp16
the whole-module C calls (d2r_interaction_desc.bn_sync called back from inside the library)."""
import faulthandler, os, sys
faulthandler.dump_traceback_later(int(os.environ.get("D2R_PROBE_DUMP_S", "150")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

out_dir = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from d2r_amd import functional as F
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.dp import DataParallel, shard_batch
from d2r_amd.params import FusedAdamW, ParamStore
from oracle import d2r_oracle as O

dev = torch.device("cuda:0")
cfg = O.OracleConfig(text_layers=1, vision_layers=1, image_size=64, patch_size=32, DR_step=3)
sd = O.seeded_state_dict(cfg, seed=3, router_bias="normal")
full = O.synthetic_batch(cfg, 4, 12, seed=4)
# the oracle on the full batch (fp32 on the host)
osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
lo, logits_o, _ = O.forward(osd, cfg, *full, train=True)
lo.backward()
res = {}
for name, dtype in (("f32", torch.float32), ("fp16", torch.float16)):
    model = M.UnimoModelF(default_args(DR_step=3), VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32),
                          TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    model.load_state_dict(sd, strict=True)
    model.to(dev).set_compute_dtype(dtype).train()
    model.model.use_streams = False
    store = ParamStore(model, dtype)
    opt = FusedAdamW(store, lr=1e-3)
    dp = DataParallel(store, opt, model, global_batch_exact=True)
    assert F.DP_EXACT is not None and dp.world == 2
    batch = tuple(t.to(dev) for t in shard_batch(full, rank, world))
    lscale = 256.0 if dtype == torch.float16 else 1.0
    loss, logits = model(*batch)
    (loss * lscale).backward()
    dp.reduce_gradients()
    torch.cuda.synchronize()
    lsum = loss.detach().clone().double()
    dist.all_reduce(lsum)
    e_loss = abs(float(lsum) / world - float(lo))
    b = full[0].shape[0] // world
    e_logit = float((logits.detach().double().cpu() - logits_o.detach().double()[rank * b:(rank + 1) * b]).abs().max())
    dot = ng = nr = 0.0
    worst = ("", 0.0)
    for pname, p in model.named_parameters():
        ref = osd[pname].grad
        if ref is None or getattr(p, "_d2r_grad", None) is None:
            continue
        got = p._d2r_grad.detach().double().cpu() / (world * lscale)
        ref = ref.double()
        dot, ng, nr = dot + float((got * ref).sum()), ng + float(got.pow(2).sum()), nr + float(ref.pow(2).sum())
        rel = float((got - ref).norm() / (ref.norm() + 1e-6 * float(nr) ** 0.5 + 1e-30))
        if rel > worst[1]:
            worst = (pname, rel)
    cos = dot / max((ng * nr) ** 0.5, 1e-300)
    # the BatchNorm running statistics must be those of the global batch too (compare with a local-statistics run: they differ)
    bn = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if "SAF_module.bn.running" in k}
    res[name] = dict(e_loss=e_loss, e_logit=e_logit, cos=cos, worst=worst, bn=bn, norm_ratio=(ng / max(nr, 1e-300)) ** 0.5)
    F.DP_EXACT = None
    print(f"rank {rank} {name}: loss err {e_loss:.2e} logits err {e_logit:.2e} gradient cosine {cos:.6f} |g|/|ref| {res[name]['norm_ratio']:.4f} "
          f"worst tensor {worst[0]} {worst[1]:.2e}", flush=True)
# control: the default mode (local statistics, local JS) must NOT reproduce the full-batch loss - the test would otherwise be vacuous
model = M.UnimoModelF(default_args(DR_step=3), VisionConfig(num_hidden_layers=1, image_size=64, patch_size=32),
                      TextConfig(num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
model.load_state_dict(sd, strict=True)
model.to(dev).set_compute_dtype(torch.float32).train()
model.model.use_streams = False
ParamStore(model, torch.float32)
loss, _ = model(*tuple(t.to(dev) for t in shard_batch(full, rank, world)))
lsum = loss.detach().clone().double()
dist.all_reduce(lsum)
res["local_mode_loss_gap"] = abs(float(lsum) / world - float(lo))
torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
