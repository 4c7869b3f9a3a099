import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
torch.cuda.set_device(0); dev = torch.device("cuda:0")
N = int(os.environ.get("NSCENES", "4096"))
scenes = [syn.make_scene(i) for i in range(N)]
replay = DeviceReplay(scenes, dev, rcap=6)
if os.environ.get("DROP_SCENES"): scenes = None
model = PlanningModel(radius=120).to(dev); model.need_traj = False; model.train()
tr = RLFTTrainer(model, kind="rift", seed=1)
g = torch.Generator().manual_seed(5)
def loop(feat):
    times = []
    for ep in range(16):
        perm = torch.randperm(N, generator=g)[:3686] if N >= 3686 else torch.randint(0, N, (3686,), generator=g)
        if "A" in feat:
            idx_dev = perm.to(torch.int32).to(dev)
        else:
            idx_dev = PRE[ep]
        up = torch.cuda.Event(); up.record()
        nb = 15 if "D" in feat else 14
        for k in range(nb):
            s = k * 256; m = min(256, 3686 - s)
            sl = idx_dev[s:s + m]
            R = int(replay.r_count_cpu[perm[s:s + m]].max()) if "C" in feat else None
            fb, b = tr.gather(replay, sl, R, ready=up if "B" in feat else None)
            tr.training_step(fb, b)
        t = time.perf_counter(); tr.pop_mean_loss(); times.append(round((time.perf_counter() - t) * 1e3, 1))
    return times
PRE = [torch.randperm(N, generator=g)[:3686].to(torch.int32).to(dev) if N >= 3686 else torch.randint(0, N, (3686,), generator=g).to(torch.int32).to(dev) for _ in range(16)]
for feat in sys.argv[1:]:
    loop(feat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ts = loop(feat)
    torch.cuda.synchronize()
    print(f"feat={feat:6s} total {time.perf_counter() - t0:.3f}s pop_mean_loss ms: {ts}")
