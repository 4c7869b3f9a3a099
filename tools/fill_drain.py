"""What a timed region of K update steps costs beyond K steady-state steps (the step pipeline's fill and drain): T(K) for several K, the
affine fit T = a + b K, and -- with RIFT_TIMELINE=1 -- where the first and the last step of a 20-step region spend their time.
    python tools/fill_drain.py [batch]        (on the GPU box)"""
import os, sys, time, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import synthetic as syn
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.replay import DeviceReplay


def main():
    dev = torch.device("cuda", 0)
    BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scenes = [syn.make_scene(i) for i in range(1024)]
    replay = DeviceReplay(scenes, dev, rcap=6)
    torch.manual_seed(1)
    model = PlanningModel(radius=120)
    model.load_state_dict(syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}))
    model = model.to(dev); model.compute_precision = "bf16"; model.train()
    tr = RLFTTrainer(model, kind="rift")
    idx = [torch.randperm(1024)[:BATCH].to(torch.int32).to(dev) for _ in range(64)]

    def step(i):
        fb, b = tr.gather(replay, idx[i % 64]); return tr.training_step(fb, b)

    def region(k):
        tr.wait_update(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step(i)
        t_issue = time.perf_counter() - t0
        tr.wait_update(); torch.cuda.synchronize()
        return time.perf_counter() - t0, t_issue

    # what the driver's run sees: W = 5 warm-up steps on a fresh trainer, then ONE 20-step region -- against the same region repeated
    # (RIFT_FD_SPIN_MS=<ms>: that many milliseconds of chip-filling GEMMs first -- separates a clock / power-state ramp of the device
    # from a warm-up of the trainer: the ramp would be gone, the trainer's would not)
    spin_ms = float(os.environ.get("RIFT_FD_SPIN_MS", "0"))
    if spin_ms > 0:
        a = torch.randn(8192, 8192, device=dev, dtype=torch.float16); bm = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        (a @ bm); torch.cuda.synchronize()          # (the first call initialises the BLAS library: not part of the load)
        t0 = time.perf_counter(); n = 0
        while (time.perf_counter() - t0) * 1e3 < spin_ms:
            for _ in range(8): (a @ bm)
            torch.cuda.synchronize(); n += 8
        print(f"{n} 8192^3 fp16 GEMMs in {(time.perf_counter() - t0) * 1e3:.1f} ms ahead of the warm-up steps")
    for i in range(5):
        step(i)
    first = [region(20)[0] for _ in range(6)]
    print("fresh trainer, 5 warm-up steps, then six consecutive 20-step regions (ms/step): " + " ".join(f"{t / 20 * 1e3:.4f}" for t in first))
    # is the slow first region the TRAINER's (gone for good) or the idle DEVICE's (back after every pause, e.g. between two updates of train_cbv)?
    for pause in (0.05, 0.5, 2.0):
        time.sleep(pause)
        again = [region(20)[0] for _ in range(4)]
        print(f"after {pause:.2f} s of idleness, four 20-step regions (ms/step): " + " ".join(f"{t / 20 * 1e3:.4f}" for t in again))
    for i in range(30):
        step(i)
    rows = []
    for k in (1, 2, 3, 5, 10, 20, 40, 100, 200):
        ts = [region(k) for _ in range(7)]
        t = statistics.median(x[0] for x in ts); ti = statistics.median(x[1] for x in ts)
        rows.append((k, t, ti))
        print(f"K = {k:4d}: region {t * 1e3:8.3f} ms  ({t / k * 1e3:.4f} ms/step; host issue done at {ti * 1e3:.3f} ms)")
    (k1, t1, _), (k2, t2, _) = rows[-3], rows[-1]
    b = (t2 - t1) / (k2 - k1)
    print(f"steady-state slope {b * 1e3:.4f} ms/step; intercept (fill + drain) at K = 20: {(rows[5][1] - 20 * b) * 1e3:.3f} ms, at K = 1: {(rows[0][1] - b) * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
