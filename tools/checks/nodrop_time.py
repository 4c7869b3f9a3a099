"""Step time with every drop probability 0 (F_NO_DROP) against the normal train step: what the counter-based RNG costs."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
idx = [torch.randperm(512)[:256].to(torch.int32).to(dev) for _ in range(64)]
def step(i):
    fb, b = replay.collate(tr.engine, idx[i % 64]); return tr.training_step(fb, b)
for nd in (False, True, False, True):
    model._no_drop = nd
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): step(i)
    torch.cuda.synchronize(); print("no_drop" if nd else "drops  ", f"{(time.perf_counter() - t0) * 10:.4f} ms/step")
