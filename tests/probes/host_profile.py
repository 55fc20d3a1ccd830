"""Where does the HOST time of one step go?  cProfile over a few eager steps of the bench workload."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import FusedAdamW, ParamStore
sys.argv = ["bench.py"]
import bench

dev = torch.device("cuda:0")
torch.manual_seed(0)
tc = TextConfig(num_hidden_layers=12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=12, image_size=224, patch_size=16)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc)
model.to(dev).set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
opt = FusedAdamW(store, lr=3e-5)
batch = bench.synthetic_batch(32, 128, 224, dev, 0)


def step():
    loss, _ = model(*batch)
    loss.backward()
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, drained after another {1e3 * (t2 - t1):.2f} ms")
torch.autograd.set_multithreading_enabled(False)  # backward functions run in this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats("functional.py|modules.py", 45)
