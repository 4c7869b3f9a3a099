"""The idle-device ramp (round-5 review, weak #5): a fresh trainer's first 20-step regions are slower than the steady state, and 50 ms of
idleness bring the ramp back.  Clocks (kernel durations shrink over the ramp) or dispatch (durations flat, gaps / overlap change)?

    python tools/ramp_trace.py [out.json]                                                           (GPU box; in-kernel clock probes)
    RAMP_NO_PROBE=1 RIFT_TWO_STREAMS=0 RIFT_PIPELINE=0 rocprofv3 --kernel-trace -d /tmp/ramp -o ramp -- python tools/ramp_trace.py out.json
    python tools/ramp_analyze.py <results.db> out.json                                              (serial step: per-kernel durations per region)

Protocol: 5 warm-up steps, six back-to-back 20-step regions, 0.5 s of host sleep, four more regions, 50 ms of sleep, four more, 200 steps.
Beside every update step one wave on a stream of its own reads s_memtime (shader-clock ticks) against s_memrealtime (constant 100 MHz) over
a 20 us window (tools/ubench/clock_probe.hip -> libclock_probe.so): the effective shader clock, per step, whatever the SIMD is busy with.
Result (profiles/NOTES_r06.md): clocks -- 2.3 GHz on arrival, a dip to 2.15, 2.42 GHz after ~25 ms.  (The sysfs sampler of the first version
read a card that was not the one in use -- 157 MHz throughout -- and was dropped.)"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay


def load_probe():
    """tools/ubench/clock_probe.hip as a ctypes library: one wave that reads s_memtime against s_memrealtime over a 20 us window."""
    import ctypes
    path = os.path.join(REPO, "tools", "ubench", "libclock_probe.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "ramp_host.json")
    dev = torch.device("cuda", 0)
    scenes = [syn.make_scene(i) for i in range(1024)]
    replay = DeviceReplay(scenes, dev, rcap=6)
    torch.manual_seed(1)
    model = PlanningModel(radius=120)
    model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
    model = model.to(dev); model.compute_precision = "bf16"; model.train()
    tr = RLFTTrainer(model, kind="rift")
    idx = [torch.randperm(1024)[:256].to(torch.int32).to(dev) for _ in range(64)]

    def step(i):
        fb, b = tr.gather(replay, idx[i % 64]); return tr.training_step(fb, b)

    nprobe = [0]

    regions = []

    def region(k, tag):
        p0 = nprobe[0]
        tr.wait_update(); torch.cuda.synchronize()
        e0 = time.time_ns(); t0 = time.perf_counter()
        for i in range(k):
            step(i)
        ti = time.perf_counter() - t0
        tr.wait_update(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        regions.append({"tag": tag, "steps": k, "epoch_ns": e0, "t0": t0, "t1": t1, "ms_per_step": (t1 - t0) / k * 1e3, "issue_ms": ti * 1e3, "probe0": p0, "probe1": nprobe[0]})

    probe = load_probe() if os.environ.get("RAMP_NO_PROBE") != "1" else None
    pstream = torch.cuda.Stream(device=dev)
    pbuf = torch.zeros(4096, 3, dtype=torch.int64, device=dev)
    base_step = step

    def step(i):                         # one clock probe per update step, on a stream of its own (one wave: it fits beside any kernel of the step)
        if probe is not None and nprobe[0] < 4096:
            probe.clock_probe_launch(pbuf[nprobe[0]].data_ptr(), 2000, pstream.cuda_stream)
            nprobe[0] += 1
        return base_step(i)
    time.sleep(0.3)                      # the device idle in front of the first region, as in the driver's run (model load, replay upload, then steps)
    for i in range(5):
        step(i)
    for r in range(6):
        region(20, f"fresh.{r}")
    time.sleep(0.5)
    for r in range(4):
        region(20, f"after_0.5s.{r}")
    time.sleep(0.05)
    for r in range(4):
        region(20, f"after_50ms.{r}")
    region(200, "steady200")
    torch.cuda.synchronize()
    pr = pbuf[:nprobe[0]].cpu().numpy()
    mhz = [100.0 * float(c) / max(float(r), 1.0) for _, r, c in pr]          # s_memtime ticks per s_memrealtime tick (100 MHz) -> MHz
    json.dump({"regions": regions, "probe_mhz": mhz, "probe_t0_100mhz": [int(v) for v in pr[:, 0]]}, open(out, "w"))
    for r in regions:
        m = mhz[r["probe0"]:r["probe1"]]
        q = [sum(m[i:i + 5]) / max(len(m[i:i + 5]), 1) for i in range(0, min(len(m), 20), 5)]
        print(f"{r['tag']:16s} {r['ms_per_step']:.4f} ms/step  (issue {r['issue_ms']:.2f} ms)  shader clock by quarters of the region (MHz): " + " ".join(f"{v:7.1f}" for v in q))


if __name__ == "__main__":
    main()
