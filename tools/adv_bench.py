"""Achieved time / algorithmic GB/s of the HBM- and latency-bound kernels of the path (GAE scan, discounted return,
buffer z-score, group z-score, rollout return, replay collate), by torch.cuda events on the launch stream.
Run on the GPU box: python tools/adv_bench.py > gpurun_out/adv_bench.json"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rift_amd import _ffi, synthetic as syn  # noqa: E402
from rift_amd.replay import DeviceReplay  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    eng = _ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(0)
    out = {}

    def rec(name, us, nbytes, note):
        out[name] = {"us": round(us, 2), "algorithmic_bytes": int(nbytes), "GB_per_s": round(nbytes / us / 1e3, 2), "note": note}

    for n in (4096, 65536):
        r = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
        und = (torch.rand(n, generator=g) > 0.02).float().to(dev)
        v, vn = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
        unt = (torch.rand(n, generator=g) > 0.01).float().to(dev)
        rec(f"gae n={n}", timeit(lambda: eng.gae(r, und, v, vn, unt)), n * (8 + 4 * 4 + 4),
            "reverse affine scan, 4096 = one PPO buffer (ppo_datamodule.py:22-37)")
        rec(f"discounted_return n={n}", timeit(lambda: eng.discounted_return(r, und)), n * (8 + 4 + 8), "reinforce_datamodule.py:19-38")
        a = torch.randn(n, generator=g).to(dev)
        rec(f"normalize_advantage n={n}", timeit(lambda: eng.normalize_advantage_(a)), n * 4 * 3, "buffer-wide z-score, two passes")
    for ng, G in ((1, 48), (4096, 72), (4096, 192)):
        ret = torch.randn(ng, G, generator=g, dtype=torch.float64).to(dev)
        rec(f"group_advantage groups={ng} G={G}", timeit(lambda: eng.group_advantage(ret)), ng * G * 16,
            "group z-score (ddof 0, +1e-5), traj_evaluator.py:115-158")
    for G in (48, 192, 4096 * 72):
        Ts = 40
        f = [torch.randn(G, Ts, generator=g).to(dev) for _ in range(6)]
        f[0] = torch.randn(G, 80, generator=g).to(dev)
        col = (torch.rand(G, 40, generator=g) > 0.97).to(dev)
        off = (torch.rand(G, 80, generator=g) > 0.97).to(dev)
        rec(f"rollout_return G={G}", timeit(lambda: eng.rollout_return(f[0], f[1], f[2], f[3], f[4], f[5], col, off)),
            G * (80 * 4 + 5 * Ts * 4 + 40 + 80 + 8), "dense reward + discounted sum with collision break")
    for G, N in ((48, 10), (192, 30), (4096 * 72, 10)):
        cv = torch.randn(G, 80, 4, 2, generator=g).to(dev)
        ov = torch.randn(N, 40, 4, 2, generator=g, dtype=torch.float64).to(dev)
        rec(f"collision_matrix G={G} N={N}", timeit(lambda: eng.collision_matrix(cv, ov, Ts=40)), G * 40 * 32 + N * 40 * 64 + G * 40,
            "envelope overlap of candidate footprints with forecast neighbours (traj_evaluator.py:241-275)")
        rc = (torch.randn(G, 80, 2, generator=g) * 60).to(dev)
        mask = (torch.rand(400, 400, generator=g) > 0.5).to(torch.uint8).to(dev)
        rec(f"off_road_matrix G={G}", timeit(lambda: eng.off_road_matrix(rc, mask, (1.0, 2.0), 0.3)), G * 80 * 9,
            "raster lookup (traj_evaluator.py:299-318)")
    scenes = [syn.make_scene(i) for i in range(512)]
    replay = DeviceReplay(scenes, dev, rcap=6)
    idx = torch.randperm(512, generator=g)[:256].to(torch.int32).to(dev)
    per_scene = replay.nbytes() / 512
    rec("collate 256 scenes", timeit(lambda: replay.collate(eng, idx)), 2 * 256 * per_scene, "replay gather: read + write of the padded scene tensors")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
