"""Step time and per-kernel breakdown of the update step at a given scene shape on one GPU.

    python tools/shape_step.py carla            # rift_pluto.yaml:35-36 shapes: <= 49 agents (max_agent 48 + the CBV), ~60 polygons inside radius 120, R ~ U{1..6}
    python tools/shape_step.py carla-ragged     # the same caps with per-scene agent / polygon counts drawn below them, as a rollout produces
    python tools/shape_step.py bench            # BASELINE configs[2]: 64 agents, 20 polygons
    python tools/shape_step.py dense            # BASELINE configs[4]: 128 agents, 40 polygons, R ~ U{8..16}
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay

SHAPES = {"bench": (64, 64, 20, 20, 1, 6), "carla": (49, 49, 60, 60, 1, 6), "carla-ragged": (12, 49, 35, 60, 1, 6), "dense": (128, 128, 40, 40, 8, 16)}


def make_scenes(shape, n, seed0=0):
    a0, a1, m0, m1, r0, r1 = SHAPES[shape]
    g = torch.Generator().manual_seed(91 + seed0)
    out = []
    for i in range(n):
        A = int(torch.randint(a0, a1 + 1, (1,), generator=g)); Mp = int(torch.randint(m0, m1 + 1, (1,), generator=g))
        out.append(syn.make_scene(seed0 + i, num_agents=A, num_polygons=Mp, r_min=r0, r_max=r1))
    return out


def run(shape, bs=256, precision="bf16", steps=60, nscenes=512, verbose=True):
    dev = torch.device("cuda:0"); torch.cuda.set_device(0)
    scenes = make_scenes(shape, max(bs, nscenes))
    replay = DeviceReplay(scenes, dev, rcap=SHAPES[shape][5])
    torch.manual_seed(1)
    model = PlanningModel(radius=120)
    model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
    model = model.to(dev); model.need_traj = False; model.compute_precision = precision; model.train()
    tr = RLFTTrainer(model, kind="rift")
    g = torch.Generator().manual_seed(0)
    idx = [torch.randperm(len(scenes), generator=g)[:bs].to(torch.int32).to(dev) for _ in range(steps + 12)]

    def step(i):
        fb, b = tr.gather(replay, idx[i]); return tr.training_step(fb, b)
    for i in range(6): step(i)
    tr.wait_update(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6, 6 + steps): loss = step(i)
    tr.wait_update(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    piped, tr.pipeline = tr.pipeline, False
    tr.engine.prof_enable(True); step(steps + 7); rep = tr.engine.prof_report(); tr.engine.prof_enable(False)
    tr.pipeline = piped
    top = sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:10]
    tokens = replay.A + replay.Mp + replay.S
    res = {"shape": shape, "A": replay.A, "Mp": replay.Mp, "tokens": tokens, "batch": bs, "precision": precision, "ms_per_step": dt * 1e3,
           "scenes_per_s": bs / dt, "us_per_scene": dt / bs * 1e6, "sum_of_kernels_ms": sum(v["ms"] for v in rep.values()),
           "per_kernel_ms": {k: round(v["ms"], 4) for k, v in top}, "final_loss": float(loss)}
    tr.close(); model.release_engine()
    if verbose:
        print(f"{shape} (A {replay.A}, Mp {replay.Mp}, {tokens} tokens) bs={bs} {precision}: {dt*1e3:.3f} ms/step, {bs/dt:.0f} scenes/s, sum of serial kernels {res['sum_of_kernels_ms']:.3f} ms")
        for k, v in top: print(f"  {k:32s} {v['ms']:.3f} ms x{v['count']}")
    return res


if __name__ == "__main__":
    shapes = sys.argv[1:] or ["bench", "carla", "carla-ragged"]
    for s in shapes:
        run(s)
