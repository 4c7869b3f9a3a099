"""Host-side cost of one update step (launch path only, GPU running behind): how close the step is to being CPU-bound."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
dev = torch.device("cuda", 0)
scenes = [syn.make_scene(i) for i in range(512)]
replay = DeviceReplay(scenes, dev, rcap=6)
torch.manual_seed(1)
model = PlanningModel(radius=120)
model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
model = model.to(dev); model.need_traj = False; model.train()
tr = RLFTTrainer(model, kind="rift")
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 256
idx = [torch.randperm(512)[:BATCH].to(torch.int32).to(dev) for _ in range(64)]
def step(i):
    fb, b = tr.gather(replay, idx[i % 64]); return tr.training_step(fb, b)
for i in range(10): step(i)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for i in range(200): step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(100): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
# true host cost: one step issued into an empty queue (no back-pressure), averaged
import statistics
ts = []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(i); ts.append(time.perf_counter() - t0)
print(f"host cost per step into an empty queue: median {statistics.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms")
