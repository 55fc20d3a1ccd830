"""Host cost of enqueuing one training step: the C2 model at batch 2 (the GPU finishes long before the host: what is timed is the
host), with a cProfile of where it goes."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d2r_amd
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, LinearWarmupSchedule, ParamStore
d2r_amd.configure_runtime()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L = int(os.environ.get("BATCH", "2")), 128
tc = TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc).to(dev)
model.set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
opt = FusedAdamW(store, lr=3e-5)
sched = LinearWarmupSchedule(opt, 1, 100)
ids = torch.randint(1000, 30000, (B, L), device=dev); ids[:, 0] = 101
batch = (ids, torch.ones(B, L, dtype=torch.long, device=dev), torch.zeros(B, L, dtype=torch.long, device=dev),
         torch.randint(0, 3, (B,), device=dev), torch.randn(B, 3, 224, 224, device=dev))

def step():
    loss, _ = model(*batch)
    loss.backward()
    opt.step()
    sched.step()
    opt.zero_grad()

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {B}: host enqueue {(t1 - t0) / 20 * 1e3:.2f} ms/step, with the GPU drained {(t2 - t0) / 20 * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print(st.getvalue()[:5000])
