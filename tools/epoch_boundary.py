"""What the two validation steps and the epoch bookkeeping cost an update on the device: 16 epochs x 15 training steps of 256 scenes with and
without the epoch's two validation steps (256 + 154 scenes), as RLFTPluto._train issues them.   python tools/epoch_boundary.py  (on the GPU box)"""
import os, sys, time, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay

dev = torch.device("cuda", 0)
scenes = [syn.make_scene(i) for i in range(1024)]
replay = DeviceReplay(scenes, dev, rcap=6)
torch.manual_seed(1)
model = PlanningModel(radius=120)
model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
model = model.to(dev); model.compute_precision = "bf16"; model.train()
tr = RLFTTrainer(model, kind="rift")
idx = [torch.randperm(1024)[:256].to(torch.int32).to(dev) for _ in range(64)]
vidx = [torch.arange(256, dtype=torch.int32, device=dev), torch.arange(154, dtype=torch.int32, device=dev) + 300]
table = torch.zeros(16, 2, dtype=torch.float64, device=dev)


def update(val, book):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e in range(16):
        for i in range(15):
            fb, b = tr.gather(replay, idx[(e * 15 + i) % 64]); tr.training_step(fb, b)
        if book:
            tr.pop_mean_loss_async(table[e, 0])
        if val:
            vl = []
            for v in vidx:
                fb, b = replay.collate(tr.engine, v, None, slot=0)
                vl.append(tr.validation_step(fb, b).clone())
            table[e, 1].copy_(torch.stack(vl).mean())
        if book:
            tr.on_epoch_end()
    tr.wait_update(); torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(2):
    update(True, True)
for name, (val, book) in {"training steps only": (False, False), "+ epoch bookkeeping": (False, True), "+ two validation steps": (True, True)}.items():
    ts = [update(val, book) for _ in range(5)]
    print(f"{name:28s} {statistics.median(ts) * 1e3:8.2f} ms per update ({statistics.median(ts) / 16 * 1e3:.3f} per epoch)")
# a validation forward alone
for n, v in zip((256, 154), vidx):
    fb, b = replay.collate(tr.engine, v, None, slot=0)
    for _ in range(3):
        tr.validation_step(fb, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        tr.validation_step(fb, b)
    torch.cuda.synchronize()
    print(f"validation step of {n} scenes, back to back: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
