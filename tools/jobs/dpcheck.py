import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, torch.distributed as dist
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29633")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
scenes = [syn.make_scene(i) for i in range(512)]
replay = DeviceReplay(scenes, dev, rcap=6)
sd = syn.perturbed_state_dict({k: list(v.shape) for k, v in PlanningModel(radius=120).state_dict().items()})
g = torch.Generator().manual_seed(1)
idx = [torch.randperm(512, generator=g)[:256].to(torch.int32).to(dev) for _ in range(12)]
for pipe in ("0", "1"):
    os.environ["RIFT_PIPELINE"] = pipe
    out = {}
    for name, grp in (("plain", None), ("forced", dist.group.WORLD)):
        m = PlanningModel(radius=120); m.load_state_dict(sd); m = m.to(dev); m.need_traj = False; m.train()
        tr = RLFTTrainer(m, kind="rift", process_group=grp, seed=1)
        tr.force_exchange = grp is not None
        ls = []
        for ix in idx:
            fb, b = replay.collate(tr.engine, ix, slot=tr.next_slot())
            tr.training_step(fb, b, shard=(0, 256))
            ls.append(tr.step_loss())
        out[name] = ls
        tr.close()
    print("pipeline", pipe, "max |loss diff| plain vs forced exchange:", max(abs(a - b) for a, b in zip(out["plain"], out["forced"])), out["plain"][:3], out["forced"][:3])
dist.destroy_process_group()
