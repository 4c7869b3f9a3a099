"""Ping-pong between the caller's stream and each stream of a pool (tiny kernels, event hand-overs): do some streams hand over more slowly
(a shared dispatch pipe)?   python tools/scratch/pipe_probe.py [n dummy streams first]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rift_amd  # noqa: F401  (GPU_MAX_HW_QUEUES, before the runtime starts)
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
dummies = [torch.cuda.Stream() for _ in range(n)]
for s in dummies:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(12)]
for s in pool:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()


def pingpong(a, b, rounds=60):
    ev = [torch.cuda.Event() for _ in range(2 * rounds)]
    ya, yb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(rounds):
        with torch.cuda.stream(a):
            if i: a.wait_event(ev[2 * i - 1])
            ya.add_(1); ev[2 * i].record(a)
        with torch.cuda.stream(b):
            b.wait_event(ev[2 * i]); yb.add_(1); ev[2 * i + 1].record(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds * 1e6


for rep in range(2):
    print(f"n={n} rep {rep}: us per round trip main <-> pool[i]:", " ".join(f"{pingpong(main, s):5.1f}" for s in pool), flush=True)
print("pool[0] <-> pool[i]:", " ".join(f"{pingpong(pool[0], s):5.1f}" for s in pool[1:]))
