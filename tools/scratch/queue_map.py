"""Does the number of streams a process created BEFORE the trainer change the step time (streams -> hardware queues, round robin)?
    python tools/scratch/queue_map.py <n dummy streams> [batch]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
n = int(sys.argv[1]); BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
dummies = [torch.cuda.Stream() for _ in range(n)]
x = torch.zeros(1024, device=dev)
for s in dummies:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
scenes = [syn.make_scene(i) for i in range(512)]
replay = DeviceReplay(scenes, dev, rcap=6)
torch.manual_seed(1)
model = PlanningModel(radius=120)
model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
model = model.to(dev); model.need_traj = False; model.train(); model.compute_precision = "bf16"
tr = RLFTTrainer(model, kind="rift")
idx = [torch.randperm(512)[:BATCH].to(torch.int32).to(dev) for _ in range(64)]
def step(i):
    fb, b = tr.gather(replay, idx[i % 64]); return tr.training_step(fb, b)
for i in range(20): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): step(i)
torch.cuda.synchronize()
print(f"dummy streams {n}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step (batch {BATCH}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')})")
