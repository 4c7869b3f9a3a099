"""Does the interleaved validation test see the missing serial-step wait?  (emulates the bug: forward_loss without the event record)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
from tests import helpers as H
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
scenes = [syn.make_scene(2000 + i) for i in range(160)]
sd = H.weights()
g = torch.Generator().manual_seed(7)
train_ix = [torch.randperm(160, generator=g)[:128].to(torch.int32).to(dev) for _ in range(12)]
val_ix = torch.arange(128, dtype=torch.int32, device=dev)
res = {}
for mode in ("fixed", "bug"):
    replay = DeviceReplay(scenes, dev, rcap=6)
    model = PlanningModel(radius=120); model.load_state_dict({k: v.clone() for k, v in sd.items()}); model = model.to(dev); model.need_traj = False; model.train()
    tr = RLFTTrainer(model, kind="rift", seed=3)
    if mode == "bug":
        tr.forward_loss = tr._forward_loss
    vals = []
    try:
        for k, ix in enumerate(train_ix):
            fb, b = tr.gather(replay, ix); tr.training_step(fb, b)
            if k % 3 == 2:
                tr.wait_update(); fb, b = replay.collate(tr.engine, val_ix); vals.append(tr.validation_step(fb, b).clone())
        mean = tr.pop_mean_loss()
    except RuntimeError as e:
        mean = "error: " + str(e)[:80]
    torch.cuda.synchronize()
    res[mode] = (mean, [float(v.item()) for v in vals])
    print(mode, res[mode])
