"""Cost of filling and draining the step pipeline: total time of K-step timed regions (synchronize on both sides), several repetitions."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
dev = torch.device("cuda", 0)
scenes = [syn.make_scene(i) for i in range(512)]
replay = DeviceReplay(scenes, dev, rcap=6)
model = PlanningModel(radius=120)
model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
model = model.to(dev); model.need_traj = False; model.train()
tr = RLFTTrainer(model, kind="rift")
idx = [torch.randperm(512)[:256].to(torch.int32).to(dev) for _ in range(64)]
def step(i):
    fb, b = tr.gather(replay, idx[i % 64]); return tr.training_step(fb, b)
for i in range(60): step(i)
def timed(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for n in (1, 2, 4, 8, 20, 20, 20, 50, 200):
    ts = [timed(n) for _ in range(3)]
    print(f"{n:4d} steps: " + "  ".join(f"{t:.3f} ms ({t / n:.4f}/step)" for t in ts))
