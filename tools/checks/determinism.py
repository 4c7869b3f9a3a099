"""Run-to-run determinism of training steps: the same seeded steps on two fresh engines must give bit-identical losses / parameters.
Usage: python tools/checks/determinism.py [small|bench]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
kind = sys.argv[1] if len(sys.argv) > 1 else "small"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if kind == "small":
    scenes = [syn.make_scene(i, num_agents=12, num_polygons=8, r_min=1, r_max=3) for i in range(48)]
    bss = [int(x) for x in os.environ.get("BSS", "8,8,8,8,8,3,8,5").split(",")]
else:
    scenes = [syn.make_scene(i) for i in range(512)]
    bss = [256] * 6
runs = []
for rep in range(2):
    replay = DeviceReplay(scenes, dev, rcap=6)
    torch.manual_seed(7)
    model = PlanningModel(radius=120)
    model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
    model = model.to(dev); model.need_traj = False; model.train()
    tr = RLFTTrainer(model, kind="rift")
    g = torch.Generator().manual_seed(3)
    losses = []
    for bs in bss:
        idx = torch.randperm(len(scenes), generator=g)[:bs].to(torch.int32).to(dev)
        R_out = 6 if os.environ.get("FIXR") == "1" else int(replay.r_count_cpu[idx.cpu().long()].max())
        fb, b = replay.collate(tr.engine, idx, R_out)
        if os.environ.get("TAPS"):
            tr.forward_loss(fb, b, train=True, backward=False)
            torch.cuda.synchronize()
            losses.append({n: tr.engine.tap(n).double().nan_to_num().sum().item() for n in os.environ["TAPS"].split(",")})
            continue
        tr.training_step(fb, b)
        tr.wait_update()
        losses.append(float(tr.loss.item()))
    torch.cuda.synchronize()
    runs.append((losses, {k: v.detach().cpu().clone() for k, v in tr.params.items()}))
    tr.engine.close()
print("losses run0", runs[0][0])
print("losses run1", runs[1][0])
print("identical losses:", runs[0][0] == runs[1][0], "identical params:", all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1]))
