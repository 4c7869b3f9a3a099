"""Which triples of streams make the update pipeline fast?  One process, a pool of 12 streams created first (as the trainer's probe does);
for every stream of the pool two probes against the caller's stream and against each other -- the trainer's (a one-block spin on a, a tiny
kernel on b: do the hardware QUEUES run beside each other?) and a chip-filling one (a 512 MB elementwise kernel on a -- its grid is being
dispatched for its whole duration -- and the tiny kernel on b: does b's DISPATCH get through while a's is busy?) -- then the 256-scene step
time of trainers built on forced triples.      python tools/stream_sets.py        (on the GPU box)"""
import itertools, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft import trainer as T
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay

dev = torch.device("cuda", 0)
main = torch.cuda.current_stream(dev)
pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
big = torch.zeros(128 * 1024 * 1024, device=dev)
x = torch.zeros(64, device=dev)


def beside_big(a, b):
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        e0.record(a); big.mul_(1.0); ea.record(a)
    with torch.cuda.stream(b):
        x.add_(1.0); eb.record(b)
    torch.cuda.synchronize(dev)
    return eb.elapsed_time(ea) / max(e0.elapsed_time(ea), 1e-6)        # fraction of a's kernel that was still ahead when b's finished


names = ["main"] + [f"s{i}" for i in range(12)]
streams = [main] + pool
for _ in range(2):
    beside_big(main, pool[0])
print("tiny probe (1 = runs beside), rows a, columns b:")
tiny = [[1 if (a is b or T._runs_beside(a, b, dev)) else 0 for b in streams] for a in streams]
for n, r in zip(names, tiny):
    print(f"  {n:5s} " + " ".join(str(v) for v in r))
print("chip-filling probe (fraction of a's kernel still ahead when b's tiny kernel finished), rows a, columns b:")
bigm = [[(1.0 if a is b else beside_big(a, b)) for b in streams] for a in streams]
for n, r in zip(names, bigm):
    print(f"  {n:5s} " + " ".join(f"{v:4.2f}" for v in r))

scenes = [syn.make_scene(i) for i in range(512)]
replay = DeviceReplay(scenes, dev, rcap=6)
idx = [torch.randperm(512)[:256].to(torch.int32).to(dev) for _ in range(64)]


def step_time(triple):
    T._STREAMS.clear()
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), "pipeline")
    T._STREAMS[key] = tuple(pool[i] for i in triple)
    torch.manual_seed(1)
    model = PlanningModel(radius=120)
    model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
    model = model.to(dev); model.compute_precision = "bf16"; model.train()
    tr = T.RLFTTrainer(model, kind="rift")
    for i in range(60):
        fb, b = tr.gather(replay, idx[i % 64]); tr.training_step(fb, b)
    tr.wait_update(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        fb, b = tr.gather(replay, idx[i % 64]); tr.training_step(fb, b)
    tr.wait_update(); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 100 * 1e3
    tr.close() if hasattr(tr, "close") else None
    return t


triples = [(0, 1, 2), (1, 2, 3), (2, 3, 4), (3, 4, 5), (4, 5, 6), (5, 6, 7), (6, 7, 8), (0, 2, 4), (1, 3, 5), (0, 3, 6), (0, 4, 8), (1, 5, 9), (2, 6, 10), (3, 7, 11), (0, 1, 5), (0, 1, 9)]
for tri in triples:
    t = step_time(tri)
    ok_t = all(tiny[a + 1][b + 1] and tiny[b + 1][a + 1] for a, b in itertools.combinations(tri, 2)) and all(tiny[0][a + 1] and tiny[a + 1][0] for a in tri)
    mn = min([bigm[a + 1][b + 1] for a, b in itertools.permutations(tri, 2)] + [bigm[0][a + 1] for a in tri] + [bigm[a + 1][0] for a in tri])
    print(f"streams {tri}: {t:.4f} ms/step   tiny probe {'ok ' if ok_t else 'NO '}  chip-filling probe min {mn:4.2f}")
