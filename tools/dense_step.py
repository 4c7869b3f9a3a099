"""Step time of the dense-traffic configuration (BASELINE configs[4] shapes: 128 agents, 40 polygons, 8-16 reference lines) on one GPU.
N = 168 tokens and up to 16 reference lines run on the dense-traffic variants of the wave-private kernels (enc_w_kernel: two passes per layer
over rounds of eight token tiles; dec_w_kernel<., true>: rounds of eight query tiles, hand-over through the query array)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scenes = [syn.make_scene(i, num_agents=128, num_polygons=40, r_min=8, r_max=16) for i in range(max(bs, 256))]
replay = DeviceReplay(scenes, dev, rcap=16)
torch.manual_seed(1)
model = PlanningModel(radius=120)
model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
model = model.to(dev); model.need_traj = False; model.train()
tr = RLFTTrainer(model, kind="rift")
g = torch.Generator().manual_seed(0)
idx = [torch.randperm(len(scenes), generator=g)[:bs].to(torch.int32).to(dev) for _ in range(16)]
def step(i):
    fb, b = tr.gather(replay, idx[i]); return tr.training_step(fb, b)
for i in range(4): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4, 14): loss = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
tr.wait_update(); tr.pipeline = False
tr.engine.prof_enable(True); step(14); rep = tr.engine.prof_report(); tr.engine.prof_enable(False)
top = sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]
print(f"dense bs={bs}: {dt*1e3:.2f} ms/step, {bs/dt:.0f} scenes/s, loss {float(loss):.4f}")
for k, v in top: print(f"  {k:32s} {v['ms']:.3f} ms x{v['count']}")
