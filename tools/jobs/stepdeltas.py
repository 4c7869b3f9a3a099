"""Per-step completion times of the trunk (event after each training_step on the caller's stream) after a full synchronize."""
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
for i in range(5): step(i)
for rep in range(2):
    torch.cuda.synchronize()
    N = 25
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    host = []
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(N):
        step(i); evs[i + 1].record(); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    print("trunk-end deltas (ms):", " ".join(f"{x:.2f}" for x in d))
    print("host issue times (ms):", " ".join(f"{x * 1e3:.2f}" for x in host))
    print(f"total {t_all * 1e3:.2f} ms for {N} steps = {t_all / N * 1e3:.3f} ms/step; sum of deltas {sum(d):.2f}")
