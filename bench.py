"""RIFT policy-update benchmark on MI355X (contract: see the task brief / DESIGN.md section "Measurement").

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one policy-update step of the reference's RIFT trainer on a 256-scene minibatch per GPU:
device-side collation of 256 scenes from the HBM-resident replay arena -> train-mode PlanningModel
forward (dropout / DropPath / state-dropout on, BatchNorm batch statistics) -> RIFT dual-clip loss ->
analytic pi_head backward -> [RCCL all-reduce] -> clip_grad_norm_(0.5) -> AdamW.
Workload = BASELINE.json configs[2]/[3]: 4096-scene synthetic replay, 64 agents x 20 polygons x R~U{1..6}
reference lines x 12 modes, sharded across the N ranks (weak scaling: 256 scenes / GPU / step).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOPS_PER_SCENE = 1.335e9        # reference-equivalent forward FLOPs/scene at (A=64, Mp=20, R=4)
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
BATCH = 256


def cpu_baseline(scenes, sd, steps=6):
    """The reference update step restated on the host cores (oracle = PyTorch-CPU fp32 port of the
    reference algorithm): CPU collate (pad_sequence) -> forward (BatchNorm batch stats, drop p=0) ->
    RIFT loss -> autograd pi_head backward -> clip 0.5 -> AdamW.  Bounded sample (~20 s).
    Threads: 32 (measured fastest on the 256-core GPU box: 4 -> 76, 8 -> 117, 16 -> 131, 32 -> 145, 64 -> 71
    scenes/s forward; the reference's own default is 4, scripts/run.py:133,164)."""
    from oracle import losses, pluto_ref
    from rift_amd import synthetic as syn
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    prefix = "planning_decoder.pi_head."
    params = {k: sd[prefix + k].clone().requires_grad_(True) for k in losses.PI_KEYS}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=1e-5)

    def one_step(chunk):
        batch = syn.collate_scenes(chunk)
        data = batch["cur_pluto_feature_torch"]
        _, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=True, want_taps=True)
        r_pad = ~data["reference_line"]["valid_mask"].any(-1)
        live = dict(sd)
        live.update({prefix + k: v for k, v in params.items()})
        pi = pluto_ref.mlp_layer(taps["q_final"], pluto_ref.SD(live, prefix)).squeeze(-1)
        prob = pi.masked_fill(r_pad.unsqueeze(-1), -1e6)
        loss = losses.rift_loss(prob, r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                batch["group_advantage_mask_torch"])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 0.5)
        opt.step()

    one_step(scenes[:32])          # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    n = 0
    for s in range(steps):
        chunk = scenes[s * BATCH:(s + 1) * BATCH]
        if len(chunk) < BATCH:
            break
        one_step(chunk)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * BATCH / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": f"{n} update steps x {BATCH} scenes (collate+fwd+RIFT loss+bwd+clip+AdamW), PyTorch-CPU fp32 oracle, "
                      f"{dt:.1f}s", "steps_per_sec": n / dt}


def pmc_traffic(label):
    """HBM bytes per launch of the kernel behind an engine profiler label, from the committed PMC passes (None if absent)."""
    path = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    kern = json.load(open(path))["kernels"]
    nat = {"nat_level_kernel_L0": "nat_level_kernel<32,", "nat_level_kernel_L1": "nat_level_kernel<64,", "nat_level_kernel_L2": "nat_level_kernel<128,"}
    key = nat.get(label, label.split(":")[0])
    gemm = {"gemm_bf16_m1n8": "gemm_rows_kernel<true, 1, 8", "gemm_bf16_m4n2": "gemm_rows_kernel<true, 4, 2", "gemm_bf16_m2n6": "gemm_rows_kernel<true, 2, 6",
            "gemm_bf16_m4n4": "gemm_rows_kernel<true, 4, 4"}
    key = gemm.get(key, key)
    hit = [v for k, v in kern.items() if key in k]
    if not hit:
        return None
    n = sum(v["launches"] for v in hit)
    return sum((2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0 * v["launches"] for v in hit) / max(n, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--replay", type=int, default=4096, help="total replay scenes (sharded across ranks)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line (rank 0's JSON): libraries that print banners to fd 1 (RCCL prints its version block at communicator
    # creation) are sent to stderr for the whole run; the saved descriptor is restored for the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or os.environ.get("RIFT_BENCH_FORCE_PG") == "1":   # (the env switch exercises the RCCL path on a single GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
        pg = dist.group.WORLD

    from rift_amd import synthetic as syn
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer, shard_scene_ids
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay

    # ---- replay shard of this rank (seeded per scene index, so the union over ranks is the same 4096 scenes)
    per_rank = max(BATCH, args.replay // world)
    t_gen = time.perf_counter()
    scenes = [syn.make_scene(i) for i in shard_scene_ids(rank, world, per_rank)]
    replay = DeviceReplay(scenes, dev, rcap=6)
    t_gen = time.perf_counter() - t_gen

    torch.manual_seed(20250515)   # identical random-init policy on every rank
    model = PlanningModel(radius=120)
    # non-trivial norm/bias/BatchNorm statistics (random-init has all-zero biases)
    sd_cpu = syn.perturbed_state_dict({k: list(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd_cpu)
    model = model.to(dev)
    model.compute_precision = args.precision
    model.need_traj = False
    model.train()
    trainer = RLFTTrainer(model, kind="rift", process_group=pg)
    eng = trainer.engine

    g = torch.Generator().manual_seed(1000 + rank)
    nsteps = args.warmup + args.steps + 8
    idx = [torch.randperm(per_rank, generator=g)[:BATCH].to(torch.int32).to(dev) for _ in range(nsteps)]

    def step(i):
        fb, b = replay.collate(eng, idx[i])
        return trainer.training_step(fb, b)

    for i in range(args.warmup):
        step(i)
    if pg is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if pg is not None:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if pg is not None:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())

    # ---- companion figure: the same K steps with EVERY output of PlanningModel.forward computed (trajectory / prediction /
    # ref-free heads -- outputs the RLFT losses never read; the reference's training_step computes them)
    model.need_traj = True
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    torch.cuda.synchronize()
    dt_all = time.perf_counter() - t1
    model.need_traj = False

    # ---- roofline leg: per-launch HIP events on the launch stream (separate short pass, rank 0)
    roof = None
    if rank == 0 and not args.no_roofline:
        eng.prof_enable(True)
        nprof = 4
        for i in range(nprof):
            step(args.warmup + args.steps + i)
        rep = eng.prof_report()
        eng.prof_enable(False)
        tot_ms = sum(v["ms"] for v in rep.values())
        dom = max(rep.items(), key=lambda kv: kv[1]["ms"])
        gemm_ms = sum(v["ms"] for k, v in rep.items() if k.startswith("gemm_"))
        gemm_fl = sum(v["flops"] for k, v in rep.items() if k.startswith("gemm_"))
        ach = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12 if dom[1]["ms"] > 0 else 0.0
        roof = {"bound": "mfma", "kernel": dom[0], "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(dom[0]),
                "traffic_source": "profiles/r01_pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch from separate rocprofv3 --pmc "
                                  "passes of this bench (tools/profile_round.sh)",
                "avg_launch_us": dom[1]["ms"] * 1e3 / dom[1]["count"], "launches_per_step": dom[1]["count"] / nprof,
                "kernel_share_of_gpu_time": dom[1]["ms"] / tot_ms,
                "all_gemm_tflops": gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                "gpu_ms_per_step_sum_of_kernels": tot_ms / nprof,
                "per_kernel_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in
                                           sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("RIFT_PROF_TOP", "12"))]}}

    if rank == 0:
        steps_per_sec = args.steps / dt
        scenes_per_sec = steps_per_sec * BATCH * world
        line = {
            "metric": "policy-update scenes/sec (256-scene RIFT update steps on a 4096-scene replay)",
            "value": scenes_per_sec, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "steps_per_sec": steps_per_sec,
            "config": {"workload": "BASELINE configs[2]/[3]: full rift_pluto CBV policy, 4096-scene synthetic replay "
                                   "(64 agents x 21 steps, 20 polygons x 3 x 20 pts, R~U{1..6} x 120 ref pts, 12 modes), "
                                   "RIFT loss, pi_head trainable",
                       "per_gpu_batch": BATCH, "global_batch": BATCH * world, "replay_scenes_per_gpu": per_rank,
                       "parallelism": f"dp{world}", "train_mode": "dropout+droppath+state-dropout, BN batch stats",
                       "outputs": "probability (the trajectory / prediction / hidden heads are dead outputs for the RIFT loss; see all_outputs)"},
            "whole_step_mfma_frac": scenes_per_sec / world * FLOPS_PER_SCENE / (PEAK_BF16_TFLOPS * 1e12),
            "all_outputs": {"ms_per_step": dt_all / args.steps * 1e3, "value": args.steps / dt_all * BATCH * world,
                            "note": "this rank's rate x N with the dead trajectory/prediction/ref-free heads also computed"},
            "final_loss": final_loss, "replay_gen_s": round(t_gen, 2), "replay_hbm_mb": round(replay.nbytes() / 1e6, 1),
        }
        if roof is not None:
            line["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scenes, sd_cpu)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
