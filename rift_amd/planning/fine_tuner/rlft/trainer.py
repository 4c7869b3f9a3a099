"""RLFT trainer core: the policy-update step of the reference's four LightningTrainers
(rift_pluto/rift_trainer.py, grpo_pluto/grpo_trainer.py, ppo_pluto/ppo_trainer.py,
reinforce_pluto/reinforce_trainer.py) on the HIP engine, without Lightning.

One step = train-mode forward (HIP) -> objective + analytic pi_head backward (HIP)
-> [DP: RCCL all-reduce of the unnormalised (grad sums) and (objective sum, count)] -> loss / grads
-> clip_grad_norm_(0.5) -> AdamW.  The optimizer and the collective stay in
PyTorch-ROCm, as the reference's optimizer does (rift_trainer.py:279-362).
"""
import contextlib
import ctypes as C
import itertools
import math
import os
import sys
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from rift_amd import _ffi

PI_HEAD = "planning_decoder.pi_head"
PI_KEYS = ("mlp.0.weight", "mlp.0.bias", "mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias")


class WarmupCosLR:
    """Per-epoch schedule of pluto/optim/warmup_cos_lr.py:39-54."""

    def __init__(self, optimizer, lr, min_lr, warmup_epochs, epochs):
        self.optimizer, self.lr, self.min_lr, self.warmup_epochs, self.epochs = optimizer, lr, min_lr, warmup_epochs, epochs
        self.last_epoch = 0
        self._apply()

    def get_lr(self):
        e = self.last_epoch
        if e < self.warmup_epochs:
            return self.lr * (e + 1) / self.warmup_epochs
        return self.min_lr + 0.5 * (self.lr - self.min_lr) * (
            1 + math.cos(math.pi * (e - self.warmup_epochs) / (self.epochs - self.warmup_epochs)))

    def _apply(self):
        for g in self.optimizer.param_groups:
            g["lr"] = self.get_lr() * g.get("lr_scale", 1.0)

    def step(self):
        self.last_epoch += 1
        self._apply()


def freeze_parameters(model: nn.Module, trainable_layers: List[str]):
    """rift_trainer.py:78-90."""
    for p in model.parameters():
        p.requires_grad = False
    mods = dict(model.named_modules())
    for name in trainable_layers:
        layer = mods.get(name)
        if layer is None:
            raise ValueError(f"Layer {name} not found in the model.")
        for p in layer.parameters():
            p.requires_grad = True


def configure_optimizer(model: nn.Module, lr: float, weight_decay: float):
    """AdamW parameter groups of rift_trainer.py:279-362 for the trainable sets of this path: `planning_decoder.pi_head` (Linear,
    LayerNorm, ReLU, Linear) and, for PPO / RTR, `value_net` (three Linears and the four normalisation constants).  The reference's
    module-type rule comes out as: group 0 (decays) = the weights of the Linear layers, group 1 (no decay) = every bias, the LayerNorm
    weight and the critic's constants; each group sorted by parameter name, as the reference sorts them."""
    linear_weights = {f"{mn}.weight" for mn, m in model.named_modules() if isinstance(m, nn.Linear)}
    norm_weights = {f"{mn}.weight" for mn, m in model.named_modules() if isinstance(m, nn.LayerNorm)}
    pd = {n: p for n, p in model.named_parameters() if p.requires_grad}
    decay = sorted(n for n in pd if n in linear_weights)
    no_decay = sorted(n for n in pd if n not in linear_weights)
    for n in no_decay:      # anything else with a weight (a conv, an attention in-projection, an embedding) is not a layer this path trains
        if n.endswith("weight") and n not in norm_weights:
            raise NotImplementedError(f"trainable parameter {n}: only Linear / LayerNorm layers are trainable on this path")
    groups = [{"params": [pd[n] for n in decay], "weight_decay": weight_decay},
              {"params": [pd[n] for n in no_decay], "weight_decay": 0.0}]
    # fused=True: the same AdamW update as one multi-tensor kernel per group (the host side of the default foreach path costs
    # 0.7 ms per step for six small tensors -- a third of the whole update step)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=bool(pd) and next(iter(pd.values())).is_cuda)


def dp_all_reduce_exchange(xchg: torch.Tensor, group=None):
    """The same exchange as ONE collective: `xchg` = f64 [16897 unnormalised gradient sums | objective sum | valid count], written by
    rift_loss_backward and read back by rift_loss_finalize (RiftLossOut.exchange).  135 KB, latency-bound on xGMI."""
    torch.distributed.all_reduce(xchg, op=torch.distributed.ReduceOp.SUM, group=group)


def split_minibatch(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of an n-scene minibatch for `rank` (256 scenes -> 32 per GPU at 8 GPUs; a remainder
    goes to the first ranks).  Every rank needs at least one scene (it has to join the exchanges of the forward)."""
    if n < world:
        raise ValueError(f"a {n}-scene minibatch cannot be split over {world} ranks")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dp_all_reduce(flat: torch.Tensor, stats: torch.Tensor, group=None):
    """The only exchange step of the data-parallel path.  Every rank holds UNNORMALISED sums over its scene shard:
    `flat` = sum of d(objective)/d(pi_head params) (16,897 f32), `stats` = (objective sum, valid-entry count) in f64.
    After the SUM all-reduce, loss = -stats[0]/stats[1] and grad = -flat/stats[1] on every rank equal the
    single-process result on the concatenated batch (a global masked mean -- not Lightning-DDP's mean of per-rank
    means, which weights ranks with fewer valid candidates more)."""
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM, group=group)


def shard_scene_ids(rank: int, world: int, per_rank: int):
    """Replay shard of `rank`: contiguous scene ids, disjoint across ranks (independent scenes, no data-path exchange)."""
    return range(rank * per_rank, (rank + 1) * per_rank)


_STREAMS = {}


def _shared_stream(dev, role: str) -> torch.cuda.Stream:
    """One update stream and one prefetch stream per device for all trainers of the process.  Streams map onto a handful of hardware queues
    in the order they are created (GPU_MAX_HW_QUEUES, rift_amd/__init__.py): a process that builds trainer after trainer (bench.py's
    companion legs: 0.73 -> 0.84 ms per step by the third) would otherwise end up with its caller / prefetch / update streams sharing
    queues, which serialises what they are there to overlap.  Two live trainers sharing a stream are merely ordered on it."""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), role)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=dev)
    return _STREAMS[key]


def _runs_beside(a: torch.cuda.Stream, b: torch.cuda.Stream, dev, spin_cycles: int = 1_500_000) -> bool:
    """Does work queued on `b` run while `a` is busy?  A spin kernel on a, a tiny kernel on b right behind it: on different hardware queues
    b's kernel finishes almost at once (the spin ends most of its duration later), on one queue it finishes behind the spin."""
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(enable_timing=True)
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(spin_cycles)
        ea.record(a)
    with torch.cuda.stream(b):
        x.add_(1.0)
        eb.record(b)
    torch.cuda.synchronize(dev)
    spin_ms = e0.elapsed_time(ea)
    return eb.elapsed_time(ea) > 0.5 * spin_ms          # b's kernel ended more than half a spin BEFORE the spin did


_PROBE_BUF = {}


def _dispatches_beside(a: torch.cuda.Stream, b: torch.cuda.Stream, dev) -> float:
    """Does a launch on `b` get DISPATCHED while `a` is dispatching?  ONE chip-filling elementwise kernel on a (512 MB: its grid is being dispatched
    for its whole duration, ~0.25 ms; several smaller kernels would not do -- the pipe takes b's packet between two of them: the first version
    of this probe chose a slow set in 24 processes of 24), a tiny kernel on b right behind it; returns the fraction of a's kernel that was still
    ahead when b's finished.  Measured on MI355X (tools/stream_sets.py, 8 hardware queues on 4 dispatch pipes): ~0.8 for queues on different pipes,
    ~0.1 for two queues of ONE pipe (b's kernel waits until a's grids are through the pipe), ~0 for one queue -- and the update pipeline runs at
    0.573 - 0.577 ms per 256-scene step when all six pairs of its four streams are of the first kind, at 0.60 - 0.73 ms otherwise.  The
    one-block spin of `_runs_beside` cannot tell the first two apart (its grid is through the pipe at once)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _PROBE_BUF:
        # 512 MB fill the chip for ~0.25 ms; on a device short of memory a smaller buffer still fills it for its (shorter) duration -- at
        # most an eighth of what is free, at least 32 MB (below that the grid is through the pipe too soon: the caller falls back)
        free = torch.cuda.mem_get_info(dev)[0]
        n = min(128 * 1024 * 1024, int(free // 8) // 4)
        if n < 8 * 1024 * 1024:
            raise RuntimeError("not enough free device memory for the dispatch-pipe probe")
        _PROBE_BUF[key] = (torch.zeros(n, device=dev), torch.zeros(64, device=dev))
    big, x = _PROBE_BUF[key]
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        e0.record(a)
        big.mul_(1.0)
        ea.record(a)
    with torch.cuda.stream(b):
        x.add_(1.0)
        eb.record(b)
    torch.cuda.synchronize(dev)
    return eb.elapsed_time(ea) / max(e0.elapsed_time(ea), 1e-6)


def _pipeline_streams(dev, main: torch.cuda.Stream, probe: bool = True):
    """(update, prefetch, side): three streams that run concurrently with the caller's stream and with each other, chosen by probe from a pool
    created here; one set per device for all trainers of the process.  HIP places streams on a few hardware queues in creation order, so
    what a process created BEFORE its trainer decides whether the four streams of the update pipeline end up on four queues or share one
    -- measured with n unrelated streams created first (tools/scratch/queue_map.py, 256 scenes): n = 0, 1, 4, 7: 0.65 ms per step, n = 2, 3,
    5, 6: 0.72 - 0.80; the same effect made single legs of the round's profile runs 1.4 - 1.8x slow.  The probe removes direct queue sharing:
    with it n = 0, 1, 3, 4, 5, 6 ran at 0.65 and n = 2, 7 still at 0.73 - 0.81 (queues that run beside each other in the probe can still
    share a dispatch pipe; a probe with one stream parked in an event wait moved the bad cases to n = 2, 5 without removing them).
    RIFT_STREAM_PROBE=0: three fresh streams (and the engine's own side stream), unprobed.  `probe=False`: the same -- a trainer with a
    process group takes that route: the communicator's streams come into being AFTER this choice (at the first collective), and a set
    chosen without them measured worse than the unprobed streams (one rank over RCCL, forced exchanges: 0.690 against 0.665 ms)."""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), "pipeline" if probe else "pipeline-unprobed")
    if key in _STREAMS:
        return _STREAMS[key]
    prio = os.environ.get("RIFT_STREAM_PRIO")        # diagnostic: "u,p,s" = priorities of fresh update / prefetch (history chain) / side (map chain) streams
    if prio:
        _STREAMS[key] = tuple(torch.cuda.Stream(device=dev, priority=int(x)) for x in prio.split(","))
        return _STREAMS[key]
    mode = os.environ.get("RIFT_STREAM_PROBE", "1")
    if not probe or mode == "0" or (mode == "queue" and not hasattr(torch.cuda, "_sleep")):      # (torch.cuda._sleep is the QUEUE probe's spin kernel only)
        _STREAMS[key] = (_shared_stream(dev, "update"), _shared_stream(dev, "prefetch"), None)
        return _STREAMS[key]
    pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
    try:
        chosen = _probe_pool(dev, main, pool, mode)
    except Exception as e:      # noqa: BLE001 -- an out-of-memory probe buffer or any runtime error in ~170 diagnostic launches must not abort trainer construction
        _PROBE_BUF.clear()
        print(f"[rift] stream probe failed ({type(e).__name__}: {e}); using the shared unprobed streams", file=sys.stderr)
        _STREAMS[key] = (_shared_stream(dev, "update"), _shared_stream(dev, "prefetch"), None)
        return _STREAMS[key]
    _PROBE_BUF.clear()                        # (the probe buffer goes back to the caching allocator ...
    if os.environ.get("RIFT_PROBE_KEEP_CACHE") != "1":
        torch.cuda.empty_cache()              # ... and from there to the device: half a gigabyte should not stay reserved for a one-time probe)
    _STREAMS[key] = tuple(chosen)
    _STREAMS[(key[0], "pool")] = pool         # (the unused ones stay alive: destroying them would hand their queues to the next stream created)
    return _STREAMS[key]


def _probe_pool(dev, main, pool, mode):
    """The three streams of `pool` that dispatch beside the caller's stream and beside each other (see _pipeline_streams)."""
    # Round 5: the probe that decides is `_dispatches_beside` (dispatch PIPES, not just queues): the residual slow mode of the queue probe --
    # 1 fresh process of 14 at 0.72 instead of 0.58 ms per step, and whole legs of earlier rounds' profile runs -- was two of the four streams
    # on one pipe.  RIFT_STREAM_PROBE=queue restores the one-block probe.
    if mode == "queue":
        ok = lambda a, b: _runs_beside(a, b, dev) and _runs_beside(b, a, dev)
        chosen = []
        for s in pool:
            if len(chosen) == 3:
                break
            if ok(main, s) and all(ok(c, s) for c in chosen):
                chosen.append(s)
        if len(chosen) < 3:                       # (fewer independent queues than the pipeline has streams: take what there is)
            chosen += [s for s in pool if s not in chosen][:3 - len(chosen)]
    else:
        # the whole matrix first (13 x 12 probes of ~0.3 ms; a throw-away pass ahead of it: the first launches of a process or after a pause
        # measure its warm-up), then the triple whose worst pair -- either direction, the caller's stream included -- is best; ties go to the
        # earliest streams.  (A greedy walk over the pool with probes on demand picked conflicting sets: its first probes of each stream were off.)
        allst = [main] + pool
        for b_ in pool:
            _dispatches_beside(main, b_, dev)
        m = [[1.0 if a_ is b_ else _dispatches_beside(a_, b_, dev) for b_ in allst] for a_ in allst]
        pair = lambda i, j: min(m[i][j], m[j][i])
        best, chosen = -1.0, list(pool[:3])
        for i, j, k in itertools.combinations(range(1, len(allst)), 3):
            v = min(pair(0, i), pair(0, j), pair(0, k), pair(i, j), pair(i, k), pair(j, k))
            if v > best + 0.05:
                best, chosen = v, [allst[i], allst[j], allst[k]]
        if os.environ.get("RIFT_STREAM_DEBUG"):
            print(f"[rift] pipeline streams: pool indices {[pool.index(c) for c in chosen]}, worst pair {best:.2f}", file=sys.stderr)
            for r in m:
                print("[rift]   " + " ".join(f"{v:5.2f}" for v in r), file=sys.stderr)
    return chosen


class RLFTTrainer:
    """Update-step driver for kind in {'rift','grpo','ppo','reinforce'}.

    `process_group` (torch.distributed, backend nccl = RCCL on ROCm) enables data parallelism: every minibatch is sharded across the
    ranks and three SUM all-reduces per step make the sharded update equal the single-process update on the whole minibatch
    (SURVEY.md 8(e)): two inside the forward (BatchNorm sums of the PointsEncoders + the padding rows the r2r mask quirk reads, then
    the second BatchNorm; rift_set_dp) and one between backward and finalize (gradient sums, objective sum, valid count).
    `exchange` (callable(tensor) -> in-place sum over ranks) with `dp_rank` / `dp_world` replaces the process group's all-reduce
    (tests drive several emulated ranks through one GPU with it)."""

    def __init__(self, model, kind: str = "rift", lr=1e-4, cl_lr_decay=0.9, weight_decay=1e-5, epochs=16,
                 warmup_epochs=3, trainable_layers=(PI_HEAD,), gradient_clip_val=0.5, process_group=None,
                 clip_epsilon=0.2, lambda_entropy=0.01, seed: int = 0, exchange=None, dp_rank=None, dp_world=None):
        if kind == "rs":     # RS's objective (fine_tuner/sft/rs_pluto/rs_trainer.py:120-170) is REINFORCE's, line for line
            kind = "reinforce"
        if kind not in _ffi.LOSS_KINDS and kind != "rtr":
            raise ValueError(kind)
        want = (PI_HEAD, "value_net") if kind in ("ppo", "rtr") else (PI_HEAD,)
        if tuple(trainable_layers) != want:
            raise NotImplementedError(
                "the HIP backward covers the reference's configured trainable sets: ['planning_decoder.pi_head'] "
                "(rift_training.yaml:26-27) and, for PPO, ['planning_decoder.pi_head', 'value_net'] (ppo_training.yaml:26-28)")
        # "rtr" (fine_tuner/sft/rtr_pluto/rtr_trainer.py:131-171) = lambda_rl * PPO objective + teacher cross entropy: two passes of the
        # loss kernels over the same forward, the second accumulated into .grad
        self.model, self.kind, self.kind_id = model, kind, _ffi.LOSS_KINDS["ppo" if kind == "rtr" else kind]
        self.lambda_rl = 5.0                                     # rtr_trainer.py:153
        # The trajectory / prediction / ref-free heads feed none of the RLFT objectives (dead outputs), so the trainer
        # computes them for sft / rtr only, whose teacher label is read off the candidate trajectories (sft_trainer.py:186-199).  The
        # selection is the TRAINER's (F_NEED_TRAJ of its own forwards): the model's `need_traj` -- what PlanningModel.forward and the
        # rollout side read -- is not touched, so a trainer that is never closed, or two trainers of different kinds on one model, leave
        # every other consumer's outputs alone (round-5 advisor).
        self.need_traj = kind in ("sft", "rtr")
        self.lr, self.epochs, self.warmup_epochs = lr, epochs, warmup_epochs
        self.gradient_clip_val = gradient_clip_val
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else (dp_world or 1)
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else (dp_rank or 0)
        if exchange is None and process_group is not None:
            exchange = lambda t: torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=process_group)  # noqa: E731
        self.exchange = exchange             # None = single process
        self.dp_buf = None                   # exchange buffer of the forward (rift_set_dp), sized at the first step
        freeze_parameters(model, list(trainable_layers))
        self.optimizer = configure_optimizer(model, lr, weight_decay)
        self.scheduler = WarmupCosLR(self.optimizer, lr=lr, min_lr=lr * cl_lr_decay, warmup_epochs=warmup_epochs,
                                     epochs=epochs)
        self.engine = model.engine()
        if hasattr(model, "_engine_users"):
            model._engine_users.add(self)          # (PlanningModel.engine() refuses to swap the context under a live trainer)
        dev = self.engine.device
        head = dict(model.named_modules())[PI_HEAD]
        self.params = {k: dict(head.named_parameters())[k] for k in PI_KEYS}
        for p in self.params.values():
            p.grad = torch.zeros_like(p)
        self.train_params = list(self.params.values())
        # PPO: the critic (every parameter of value_net is trainable in the reference, normalisation constants included)
        self.critic = None
        if kind in ("ppo", "rtr"):
            vn = dict(model.named_modules())["value_net"]
            self.critic = {k: dict(vn.named_parameters())[k] for k in _ffi.CRITIC_KEYS}
            for p in self.critic.values():
                p.grad = torch.zeros_like(p)
            self.train_params += list(self.critic.values())
            self.flat_c = torch.zeros(_ffi.CRITIC_NPARAM, dtype=torch.float32, device=dev)
            self.critic_desc = self.engine.critic_desc(self.critic)
            vn.bind(self.engine)
        # exchange buffers of the DP path (see dp_all_reduce)
        self.flat = torch.zeros(_ffi.PI_NPARAM, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(2, dtype=torch.float64, device=dev)
        # the finalize kernel writes a training step's loss into its own slot of `loss_hist` (summed once per pop_mean_loss()): no
        # accumulate launch per step; validation writes the separate `loss_val`
        self.loss_hist = torch.zeros(self.LOSS_SLOTS, dtype=torch.float64, device=dev)
        self.loss_val = torch.zeros(1, dtype=torch.float64, device=dev)
        self.loss = self.loss_val
        self.lo = _ffi.RiftLossOut()
        self.lo.loss, self.lo.stats, self.lo.flat_grad_sum = self.loss.data_ptr(), self.stats.data_ptr(), self.flat.data_ptr()
        self.xchg = None
        if self.exchange is not None:     # DP: one f64 exchange buffer, one all-reduce per step
            self.xchg = torch.zeros(_ffi.PI_NPARAM + 2, dtype=torch.float64, device=dev)
            self.lo.exchange = self.xchg.data_ptr()
        g = self.params
        self.lo.grad_w1, self.lo.grad_b1 = g["mlp.0.weight"].grad.data_ptr(), g["mlp.0.bias"].grad.data_ptr()
        self.lo.grad_ln_w, self.lo.grad_ln_b = g["mlp.1.weight"].grad.data_ptr(), g["mlp.1.bias"].grad.data_ptr()
        self.lo.grad_w2, self.lo.grad_b2 = g["mlp.3.weight"].grad.data_ptr(), g["mlp.3.bias"].grad.data_ptr()
        self.li = _ffi.RiftLossIn()
        self.li.clip_epsilon, self.li.lambda_entropy = clip_epsilon, lambda_entropy
        self.out = _ffi.RiftOutputs()
        self._prob = None
        self._hidden = None
        self._argmax = None
        self._traj, self._A, self._traj_A = None, 0, 0
        self.step_count = 0
        # dropout / DropPath / state-dropout stream: a fresh stream per fit (the caller mixes carla_episode into `seed`) and per DP rank,
        # as the reference draws fresh torch RNG per fit and per process; the step counter is added on top (forward_loss)
        rank = self.rank
        self.seed_base = (int(seed) * 0x9E3779B1 + rank * 0x85EBCA77) & 0x7FFFFFFF
        self.training = True
        self._clip_list = None
        self._fast_groups = None
        self.fused_tail = os.environ.get("RIFT_FUSED_TAIL", "1") != "0"     # finalize + clip + AdamW in one launch (rift_update_tail)
        self.force_exchange = os.environ.get("RIFT_BENCH_FORCE_PG") == "1"   # run the all-reduce even with one rank (path check)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        # Only pi_head is trainable, so the frozen trunk of step k+1 does not depend on the update of step k: exchange + finalize +
        # clip + AdamW run on a second stream, and the engine waits for their end right before it reads pi_head (rift_set_param_event).
        # Hides the all-reduce latency under the next forward (DP) and the latency-bound update tail (any world size).
        # On by default (RIFT_OVERLAP=0 / RIFT_NO_OVERLAP=1 switch it off): 0.802 -> 0.791 ms per step on one GPU at the end of round 2, same
        # parameters and losses bit for bit (earlier in the round, with a per-step loss accumulation launch on the update stream, it measured
        # 0.3-2 % slower).  The step's loss scalar is then written on the update stream: read it behind wait_update() / pop_mean_loss().
        want = os.environ.get("RIFT_OVERLAP", "1") == "1"
        self.overlap_update = (want and self.critic is None and dev.type == "cuda" and os.environ.get("RIFT_NO_OVERLAP", "0") != "1")
        self.loss_acc = torch.zeros(1, dtype=torch.float64, device=dev)      # sum of training losses since pop_mean_loss()
        self.loss_n = 0
        if self.overlap_update:
            self._probe = (self.exchange is None and os.environ.get("RIFT_BENCH_FORCE_PG") != "1") or os.environ.get("RIFT_STREAM_PROBE_PG") == "1"
            upd, pre, side = _pipeline_streams(dev, torch.cuda.current_stream(dev), self._probe)
            self._side = upd
            if side is not None:
                self.engine.set_side_stream(side)
            self._ev_loss = torch.cuda.Event()
            self._ev_param = torch.cuda.Event()
            self._ev_param.record(torch.cuda.current_stream(dev))            # creates the handle; a passed event is a no-op wait
            self.engine.set_param_event(self._ev_param)
        # Deferred tail (RIFT_PIPELINE=0 switches it off): with the caller taking its batch buffers by `next_slot()`, the WHOLE tail of a step
        # -- policy head, loss, pi_head backward, exchange, finalize, clip, AdamW -- runs on the update stream beside the next step's
        # gather and frozen trunk (rift_forward with RIFT_F_DEFER_HEAD + rift_forward_head; DEFER_SLOTS activation arenas and batch-buffer sets).
        self.pipeline = self.overlap_update and os.environ.get("RIFT_PIPELINE", "1") == "1" and kind in ("rift", "grpo", "reinforce")
        self._slot, self._slot_taken, self._slot_prefetch = 0, False, False
        self._pending_tail = None
        self._prefetch = None
        if self.pipeline:
            self._ev_tail = [torch.cuda.Event() for _ in range(_ffi.DEFER_SLOTS)]      # end of the tail that read slot / arena i
            for e in self._ev_tail:
                e.record(torch.cuda.current_stream(dev))
            # Input prefetch (RIFT_PREFETCH=0 switches it off): the gather of the next batch (DeviceReplay.collate(stream=prefetch_stream)) and
            # the forward's input-only preparation (rift_set_prepare_stream) run on a stream of their own, beside the current step's kernels
            # instead of between two steps
            if os.environ.get("RIFT_PREFETCH", "1") == "1":
                self._prefetch = _pipeline_streams(dev, torch.cuda.current_stream(dev), self._probe)[1]
                self._ev_serial = torch.cuda.Event()                          # end of the last whole step on the caller's stream (forward_loss)
                self._ev_serial.record(torch.cuda.current_stream(dev))

    # ------------------------------------------------------------------------------------
    def _outputs(self, bs, R):
        """Output tensors of the forward / loss for a (bs, R) batch.  Every shape keeps ITS set for the trainer's lifetime: with the tail of
        step k on the update stream, a tensor freed when the next minibatch has another R would go back to the caching allocator (which
        tracks the stream it was allocated on only) and could be handed out again while that tail still reads it."""
        cache = self.__dict__.setdefault("_out_cache", {})
        key = (bs, R)
        if key not in cache:
            dev = self.engine.device
            cache[key] = (torch.empty(bs, R, 12, device=dev), torch.empty(bs, 128, device=dev), torch.zeros(bs, 2, dtype=torch.int64, device=dev))
        if getattr(self, "_out_key", None) != key:
            self._out_key = key
            self._prob, self._hidden, self._argmax = cache[key]
            # `hidden` (pluto_model.py:173-176) feeds only PPO's critic; the other objectives never read it
            self.out.probability = self._prob.data_ptr()
            self.lo.argmax_rm = self._argmax.data_ptr()
            self._traj = None
        want_traj = self.need_traj
        self.out.hidden = self._hidden.data_ptr() if (self.kind in ("ppo", "rtr") or want_traj) else None
        if want_traj:
            # every output of PlanningModel.forward (pluto_model.py:167-223), as the reference's training_step computes them
            tkey = (bs, R, self._A)                # the prediction buffer is sized by the agent count: part of the cache key
            tcache = self.__dict__.setdefault("_traj_cache", {})
            if tkey not in tcache:
                dev, A = self.engine.device, self._A
                tcache[tkey] = (torch.empty(bs, R, 12, 80, 6, device=dev), torch.empty(bs, max(A - 1, 0), 80, 6, device=dev),
                                torch.empty(bs, 80, 4, device=dev))
            self._traj, self._traj_A = tcache[tkey], self._A
            self.out.trajectory, self.out.prediction, self.out.ref_free_trajectory = (t.data_ptr() for t in self._traj)
        else:
            self._traj = None
            self.out.trajectory = self.out.prediction = self.out.ref_free_trajectory = None
        return self._prob

    def set_loss_inputs(self, b: Dict[str, torch.Tensor]):
        p = _ffi._ptr
        self._li_keep = b
        self.li.old_group_logits = p(b.get("old_group_logits"))
        self.li.ref_group_logits = p(b.get("ref_group_logits"))
        self.li.group_advantage = p(b.get("group_advantage"))
        self.li.group_valid_mask = p(b.get("group_valid_mask"))
        self.li.action_mode = p(b.get("action_mode"))
        self.li.advantage = p(b.get("advantage"))
        self.li.old_log_prob = p(b.get("old_log_prob"))
        self.li.returns = p(b.get("returns"))

    def _set_shard(self, fb, shard):
        """Tell the engine which scenes of the global minibatch this forward holds (None: equal shards in rank order)."""
        if self.exchange is None or not (self.world > 1 or self.force_exchange):
            return
        if shard == "replicated":      # every rank runs the same whole batch (PPO's buffer sweeps): nothing to exchange
            if self.dp_buf is not None:
                self.engine.clear_dp()
            return
        lo, gbs = shard if shard is not None else (self.rank * fb.bs, self.world * fb.bs)
        need = gbs * fb.R + 1026
        if self.dp_buf is None or self.dp_buf.numel() < need:
            # (sized for the largest reference-line count the kernels take, 16: the buffer must not be replaced while exchanges of earlier
            # steps are in flight on the side / update streams, and a larger global minibatch arrives only behind a host sync)
            self.wait_update()
            torch.cuda.synchronize(self.engine.device)
            self.dp_buf = torch.zeros(max(need, gbs * 16 + 1026, 4096), dtype=torch.float64, device=self.engine.device)
        self.engine.set_dp(lo, gbs, self.dp_buf, self.exchange)

    def close(self):
        """Detach the data-parallel hooks from the (model-owned) engine."""
        self._flush_tail()
        if self.dp_buf is not None:
            self.engine.clear_dp()
            self.dp_buf = None
        if hasattr(self.model, "_engine_users"):
            self.model._engine_users.discard(self)

    def forward_loss(self, *args, **kwargs):
        """forward + objective (+ pi_head backward into .grad) on the current stream.  Returns the device f64 loss scalar.  With `clip_val`
        (and no critic, i.e. pi_head is the only trainable module) the gradient-norm clip rides in the finalize launch."""
        self._flush_tail()                   # (a data-parallel tail still held back names its forward relative to the latest one)
        if self.pipeline:
            # a whole step on the caller's stream runs in activation arena 0 and rewrites the shared (bs, R) output buffers: tails of earlier
            # pipelined steps still running on the update stream read both -- wait for them (an event wait, free when nothing is in flight)
            torch.cuda.current_stream().wait_event(self._ev_param)
        try:
            return self._forward_loss(*args, **kwargs)
        finally:
            # a whole step on the caller's stream (validation, PPO, a step without next_slot()) runs in activation arena 0 and on batch buffers
            # the prefetch stream knows nothing about: the next prefetched gather waits for it (next_slot)
            if self._prefetch is not None:
                self._ev_serial.record(torch.cuda.current_stream())

    def _forward_loss(self, fb: "_ffi.RiftFeatureBatch", extras: Dict[str, torch.Tensor], train: bool = True,
                      backward: bool = True, flags_extra: int = 0, clip_val: Optional[float] = None, defer_update: bool = False,
                      shard=None):
        eng = self.engine
        self._A = fb.A
        self._outputs(fb.bs, fb.R)
        self._set_shard(fb, shard)
        flags = (_ffi.F_TRAIN if train else 0) | (_ffi.F_NEED_TRAJ if self.need_traj else 0) | \
                 (_ffi.F_FP32 if self.model.compute_precision == "fp32" else 0) | \
                (_ffi.F_NO_DROP if getattr(self.model, "_no_drop", False) else 0) | flags_extra
        self.step_count += 1
        eng.forward_raw(fb, self.out, flags, (self.seed_base + self.step_count) & 0xFFFFFFFF)
        if self.kind == "sft":   # label = (the policy's best line, the teacher's mode); extras["teacher_infos"]: (bs, 5) as SFTDataModule yields
            extras = dict(extras)
            extras["action_mode"] = eng.sft_teacher_mode(self._traj[0][:fb.bs], extras["teacher_infos"])
        self.set_loss_inputs(extras)
        eng.loss_backward_raw(self.kind_id, self.li, self.lo)
        if self.critic is not None:   # value loss half of get_ppo_loss (ppo_trainer.py:175-176,183)
            eng.critic_loss_backward_raw(self.critic_desc, extras["state"], extras["reward_sum"], self.stats, self.flat_c)
        if self.critic is not None and self.xchg is not None:
            self.xchg[_ffi.PI_NPARAM:].copy_(self.stats)   # the value-loss term joined stats after the exchange buffer was filled
        if defer_update:     # the caller runs exchange / finalize / clip / optimizer (on the update stream)
            return self.loss
        if self.kind == "rtr":
            # weight the PPO pass by lambda_rl: loss = -S/cnt and grad = -flat/cnt, so dividing the count does it for the actor, the
            # critic and the reported loss alike (after the exchange: the count is summed over ranks first)
            self._exchange_and_finalize(backward, None, count_scale=1.0 / self.lambda_rl)
            loss_rl = self.loss.clone()
            ex2 = dict(extras)
            ex2["action_mode"] = eng.sft_teacher_mode(self._traj[0][:fb.bs], extras["teacher_infos"])
            self.set_loss_inputs(ex2)
            eng.loss_backward_raw(_ffi.LOSS_KINDS["sft"], self.li, self.lo)
            self._exchange_and_finalize(backward, None, accumulate=1, with_critic=False)
            self.loss.add_(loss_rl)
            return self.loss
        self._exchange_and_finalize(backward, clip_val)
        return self.loss

    def forward_trunk(self, fb: "_ffi.RiftFeatureBatch", shard=None, train: bool = True):
        """The frozen part of a forward (everything up to the decoder output) on the current stream; the policy head follows through
        Engine.forward_head() wherever the caller orders it.  `train` False: the validation forward (BatchNorm running statistics, no drops)."""
        self._A = fb.A
        self._outputs(fb.bs, fb.R)
        self._set_shard(fb, shard)
        flags = (_ffi.F_TRAIN if train else 0) | _ffi.F_DEFER_HEAD | (_ffi.F_NEED_TRAJ if self.need_traj else 0) | \
            (_ffi.F_FP32 if self.model.compute_precision == "fp32" else 0) | (_ffi.F_NO_DROP if getattr(self.model, "_no_drop", False) else 0)
        self.step_count += 1
        self.engine.forward_raw(fb, self.out, flags, (self.seed_base + self.step_count) & 0xFFFFFFFF)

    def _finalize_and_step(self, fused_clip: bool):
        """The tail behind the backward pass: exchange, loss / gradient finalization + clip, AdamW.  Once torch has created the optimizer
        state (first step) and the trainable set is exactly pi_head under the fused clip, the last two are ONE launch (rift_update_tail)."""
        if fused_clip and self._fast_groups and self._adam_list["n"] == 6 and self.fused_tail:
            self._exchange_and_finalize(True, self.gradient_clip_val, fuse_adam=True)
            return
        self._exchange_and_finalize(True, self.gradient_clip_val if fused_clip else None)
        self._optimizer_step()

    def _exchange_and_finalize(self, backward: bool, clip_val: Optional[float], accumulate: int = 0, with_critic: bool = True,
                               count_scale: float = 1.0, fuse_adam: bool = False):
        eng = self.engine
        with_critic = with_critic and self.critic is not None
        if self.exchange is not None and (self.world > 1 or self.force_exchange):
            self.exchange(self.xchg)
            if with_critic:
                self.exchange(self.flat_c)
        if count_scale != 1.0:
            (self.xchg[_ffi.PI_NPARAM + 1:] if (self.xchg is not None and self.lo.exchange) else self.stats[1:]).mul_(count_scale)
        if backward:
            if clip_val and self.critic is None and fuse_adam:
                self._adam_step += 1
                g0 = self.optimizer.param_groups[0]
                eng.update_tail_raw(self.lo, accumulate, float(clip_val), self.grad_norm, self._adam_list, [g["lr"] for g in self._adam_owner],
                                    [g["weight_decay"] for g in self._adam_owner], float(self._adam_step), g0["betas"][0], g0["betas"][1], g0["eps"])
            elif clip_val and self.critic is None:
                eng.loss_finalize_clip_raw(self.lo, accumulate, float(clip_val), self.grad_norm)
            else:
                eng.loss_finalize_raw(self.lo, accumulate)
            if with_critic:
                eng.critic_finalize_raw(self.flat_c, self.stats, [p.grad for p in self.critic.values()])
        else:   # validation: loss only, the .grad buffers are left untouched
            lv = _ffi.RiftLossOut()
            lv.loss, lv.stats, lv.flat_grad_sum, lv.exchange = self.lo.loss, self.lo.stats, self.lo.flat_grad_sum, self.lo.exchange
            eng.loss_finalize_raw(lv, 0)

    def forward_hidden(self, fb: "_ffi.RiftFeatureBatch", seed: int, shard="replicated") -> torch.Tensor:
        """Train-mode forward for the PPO buffer sweeps: returns the `hidden` output (bs, 128) (pluto_model.py:173-176)."""
        self._A = fb.A
        self._outputs(fb.bs, fb.R)
        self._set_shard(fb, shard)
        flags = _ffi.F_TRAIN | (_ffi.F_FP32 if self.model.compute_precision == "fp32" else 0) | \
                (_ffi.F_NO_DROP if getattr(self.model, "_no_drop", False) else 0)
        self.engine.forward_raw(fb, self.out, flags, seed)
        return self._hidden[:fb.bs]

    LOSS_SLOTS = 1024

    def _loss_slot(self, k=None):
        """Point the loss output at slot k of loss_hist (training step k since the last pop / fold), or at loss_val (k None)."""
        self.loss = self.loss_val if k is None else self.loss_hist[k:k + 1]
        self.lo.loss = self.loss.data_ptr()

    @property
    def prefetch_stream(self):
        """The stream the next batch is gathered on (pass it to DeviceReplay.collate(stream=)), or None: gather on the current stream."""
        return self._prefetch

    def next_slot(self, prefetch: bool = False) -> int:
        """Batch-buffer slot of the NEXT training step (pass it to DeviceReplay.collate).  Calling it is what enables the deferred tail:
        the stream the batch is gathered on first waits until the tail that last read this slot (DEFER_SLOTS steps ago) is over, then the caller
        may overwrite the slot's buffers while the previous step's tail is still running.  `prefetch`: the caller gathers on
        `prefetch_stream` (if that is not None) -- see gather()."""
        if not self.pipeline:
            self._slot_prefetch = False
            return 0
        self._slot = (self._slot + 1) % _ffi.DEFER_SLOTS
        self._slot_taken = True
        self._slot_prefetch = prefetch and self.prefetch_stream is not None
        # with prefetch it is the prefetch stream that waits for the slot's (and the activation arena's) last reader: the gather and the
        # forward's input preparation run there (rift_set_prepare_stream), and the forward's own streams wait for the preparation
        st = self.prefetch_stream if self._slot_prefetch else torch.cuda.current_stream()
        st.wait_event(self._ev_tail[self._slot])
        if self._slot_prefetch:
            st.wait_event(self._ev_serial)
        return self._slot

    def gather(self, replay, scene_idx: torch.Tensor, R_out=None, ready: Optional[torch.cuda.Event] = None):
        """The batch of the next training step: DeviceReplay.collate into the next slot, on the prefetch stream when there is one (beside the
        previous step's kernels instead of in front of this step's).  `ready`: an event behind the kernels / copies that produce `scene_idx`,
        if any are still queued on the caller's stream (a pageable-memory upload is); None when the index tensor is complete."""
        slot = self.next_slot(prefetch=True)
        st = self.prefetch_stream if self._slot_prefetch else None
        if st is not None and ready is not None:
            st.wait_event(ready)
        return replay.collate(self.engine, scene_idx, R_out, slot=slot, stream=st)

    def training_step(self, fb, extras, shard=None):
        """One optimizer step (LightningTrainer.training_step + Lightning's clip + optimizer.step).  Returns the device f64 loss
        scalar; with overlap_update it is written on the update stream -- read it through pop_mean_loss() / wait_update()."""
        k = self.loss_n % self.LOSS_SLOTS
        if k == 0 and self.loss_n:          # every slot used: fold them into the accumulator (once per LOSS_SLOTS steps)
            self.wait_update()
            self.loss_acc.add_(self.loss_hist.sum())
        self._loss_slot(k)
        fused_clip = bool(self.gradient_clip_val) and self.critic is None
        if self.pipeline and self._slot_taken:
            # trunk on the current stream, everything behind it on the update stream (which is serial in itself: head k waits for AdamW k-1)
            self._slot_taken = False
            slot = self._slot
            if self._slot_prefetch:
                self.engine.set_prepare_stream(self.prefetch_stream)
            main = torch.cuda.current_stream()
            late = self.exchange is not None and (self.world > 1 or self.force_exchange) and self._slot_prefetch
            if not late:
                # a tail held back by an earlier late-path step names its forward relative to the LATEST one: it has to be issued before this
                # step's trunk, or its head would pick up this step's forward (a caller alternating gather() with next_slot(prefetch=False))
                self._flush_tail()
            try:
                with _ffi.known_stream(main):
                    self.forward_trunk(fb, shard)
            finally:                          # (a forward that raises must not leave later forwards preparing on the prefetch stream)
                if self._slot_prefetch:
                    self.engine.set_prepare_stream(None)
            loss_t, loss_ptr, amax_ptr = self.loss, self.lo.loss, self.lo.argmax_rm      # (what of the loss descriptor belongs to THIS step: a tail
                                                                                         # issued late runs behind the next step's _outputs() / _loss_slot())

            def tail(back):
                """Head, loss, backward, exchange, finalize + clip, AdamW of THIS step on the update stream (`back`: forwards issued since)."""
                self.loss, self.lo.loss, self.lo.argmax_rm = loss_t, loss_ptr, amax_ptr
                with torch.cuda.stream(self._side), _ffi.known_stream(self._side):
                    self._side.wait_event(self._ev_loss)          # (the record behind this step's trunk: the latest one when the tail is issued)
                    self.engine.forward_head(back)
                    if not getattr(extras, "persistent", False):      # (DeviceReplay's cached batch buffers outlive the step: replay.PersistentBatch)
                        for t in extras.values():         # per-step tensors (buffer-wide extras indexed by the minibatch) are read on this stream:
                            if torch.is_tensor(t) and t.is_cuda:      # keep the caching allocator from recycling them under it
                                t.record_stream(self._side)
                    self.set_loss_inputs(extras)
                    self.engine.loss_backward_raw(self.kind_id, self.li, self.lo)
                    self._finalize_and_step(fused_clip)
                    self._ev_param.record(self._side)
                    self._ev_tail[slot].record(self._side)

            if late:
                # data parallel: the tail is ISSUED one step late -- behind the next step's forward.  The all-reduces of a process group
                # execute in the order they are issued; with the tail issued at once, the BatchNorm exchanges of step k + 1 (early in its
                # front) would sit behind the loss exchange of step k, i.e. behind all of step k, and the fronts could not run ahead
                prev, self._pending_tail = self._pending_tail, tail
                if prev is not None:
                    prev(1)
                    self.loss, self.lo.loss, self.lo.argmax_rm = loss_t, loss_ptr, amax_ptr
                self._ev_loss.record(main)
            else:
                self._flush_tail()
                self._ev_loss.record(main)
                tail(0)
            self.loss_n += 1
            return self.loss
        if self.pipeline:                   # a step without next_slot(): the buffers are the caller's single set -- let the last tail finish first
            self.wait_update()
        if self.overlap_update:
            self.forward_loss(fb, extras, train=True, defer_update=True, shard=shard)
            main = torch.cuda.current_stream()
            self._ev_loss.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._ev_loss)
                self._finalize_and_step(fused_clip)
                self._ev_param.record(self._side)
            self.loss_n += 1
            return self.loss
        loss = self.forward_loss(fb, extras, train=True, clip_val=self.gradient_clip_val if fused_clip else None, shard=shard)
        if self.gradient_clip_val and not fused_clip:   # PPO: clip_grad_norm_(pi_head + critic params, 0.5) on the device
            if self._clip_list is None:
                self._clip_list = self.engine.make_clip_list([p.grad for p in self.train_params])
            self.engine.clip_grad_norm_raw(self._clip_list, float(self.gradient_clip_val), self.grad_norm)
        self._optimizer_step()
        self.loss_n += 1
        return loss

    def _flush_tail(self):
        """Issue the tail of the last data-parallel step if it is still held back (see training_step)."""
        prev, self._pending_tail = getattr(self, "_pending_tail", None), None
        if prev is not None:
            keep = (self.loss, self.lo.loss, self.lo.argmax_rm)
            prev(0)
            self.loss, self.lo.loss, self.lo.argmax_rm = keep

    def wait_update(self):
        """Make the current stream wait for the last parameter update (needed before reading loss / parameters / .grad on it)."""
        self._flush_tail()
        if self.overlap_update:
            torch.cuda.current_stream().wait_event(self._ev_param)

    def step_loss(self) -> float:
        """The last training step's loss as a host float (waits for the update stream; one host read -- for tests and debugging)."""
        self.wait_update()
        return float(self.loss.item())

    def pop_mean_loss(self) -> float:
        """Mean training loss since the last call (one host read per epoch instead of one per step)."""
        self.wait_update()
        n, self.loss_n = self.loss_n, 0
        self.check_finite()                 # the reference's isfinite assert on the decoder queries, at the epoch's one host read
        used = n % self.LOSS_SLOTS or (self.LOSS_SLOTS if n else 0)      # slots written since the last fold
        # (summed on the host: one small device-to-host copy, no reduction kernel whose first use would load a module mid-epoch)
        v = (float(self.loss_acc.item()) + float(self.loss_hist[:used].cpu().sum())) / max(n, 1)
        self.loss_acc.zero_()
        return v

    def pop_mean_loss_async(self, out: torch.Tensor, in_update_stream: bool = False):
        """pop_mean_loss() without the host: the mean training loss since the last call is written into the device f64 scalar `out` on the
        current stream (which first joins the update stream: the epoch's parameters are final for whatever is queued behind).  The caller reads
        its epochs back in ONE copy at the end of the update and calls check_finite() there (the flag is sticky).  `in_update_stream`: called
        inside `with update_stream():` -- the current stream IS the update stream, nothing to join."""
        if not in_update_stream:
            self.wait_update()
        n, self.loss_n = self.loss_n, 0
        used = n % self.LOSS_SLOTS or (self.LOSS_SLOTS if n else 0)
        torch.add(self.loss_acc[0], self.loss_hist[:used].sum(), out=out)
        out.div_(max(n, 1))
        self.loss_acc.zero_()

    def check_finite(self):
        """The engine's non-finite flag is per shard: under data parallelism every rank learns whether ANY rank saw a non-finite
        decoder query (one scalar exchange), so that all of them raise together instead of one raising and the others waiting in the
        next all-reduce."""
        err = None
        try:
            self.engine.check_finite()
        except RuntimeError as e:
            err = e
        if self.exchange is not None and self.world > 1:
            flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64, device=self.engine.device)
            self.exchange(flag)
            if float(flag.item()) > 0 and err is None:
                err = RuntimeError("non-finite decoder queries on another data-parallel rank")
        if err is not None:
            raise err

    def _optimizer_step(self):
        """AdamW on the optimizer's own state tensors in ONE native launch (rift_adamw_step) for every parameter group.  The first
        step goes through optimizer.step() so that torch creates the state (exp_avg, exp_avg_sq, device step counters); afterwards
        the native kernel updates exactly those tensors, so optimizer.state_dict() and the scheduler's param_groups['lr'] keep their
        meaning.  torch's own route for six small tensors is four launches per step (two groups x (_foreach_add_ + _fused_adamw_))
        under ~0.5 ms of Optimizer.step Python; it remains the fallback (state layout not as expected, RIFT_TORCH_ADAMW=1)."""
        opt = self.optimizer
        if self._fast_groups is None:
            opt.step()
            fast = all(g.get("fused") for g in opt.param_groups) and os.environ.get("RIFT_TORCH_ADAMW", "0") != "1"
            groups = []
            for g in opt.param_groups:
                ps = [p for p in g["params"] if p.grad is not None]
                st = [opt.state[p] for p in ps]
                if not ps or g.get("amsgrad") or g.get("maximize") or \
                        any("exp_avg" not in s_ or not torch.is_tensor(s_["step"]) or not s_["step"].is_cuda
                            or s_["step"].dtype != torch.float32 or p.dtype != torch.float32 for s_, p in zip(st, ps)):
                    fast = False
                    break
                groups.append((g, ps, st))
            g0 = opt.param_groups[0]
            if fast and (any(g["betas"] != g0["betas"] or g["eps"] != g0["eps"] for g in opt.param_groups)
                         or sum(len(ps) for _, ps, _ in groups) > 16):
                fast = False
            if fast:
                ps = [p for _, gp, _ in groups for p in gp]
                st = [s_ for _, _, gs in groups for s_ in gs]
                self._adam_list = self.engine.make_adam_list(ps, [p.grad for p in ps], [s_["exp_avg"] for s_ in st],
                                                             [s_["exp_avg_sq"] for s_ in st], [s_["step"] for s_ in st])
                self._adam_owner = [g for g, gp, _ in groups for _ in gp]       # the param group of every listed tensor
                self._adam_step = int(st[0]["step"].item())                      # one host read, at setup
            self._fast_groups = groups if fast else False
            return
        if self._fast_groups is False:
            opt.step()
            return
        self._adam_step += 1
        g0 = opt.param_groups[0]
        self.engine.adamw_step_raw(self._adam_list, [g["lr"] for g in self._adam_owner], [g["weight_decay"] for g in self._adam_owner],
                                   float(self._adam_step), g0["betas"][0], g0["betas"][1], g0["eps"])

    @property
    def pipelined_validation(self) -> bool:
        """Whether validation_step(out=...) on a batch taken through gather() rides the step pipeline (single process; under data parallelism
        the tails are issued a step late around the all-reduces and validation stays a whole step on the caller's stream)."""
        return bool(self.pipeline) and not (self.exchange is not None and (self.world > 1 or self.force_exchange)) \
            and os.environ.get("RIFT_PIPELINE_VAL", "1") == "1"

    def validation_step(self, fb, extras, shard=None, out: Optional[torch.Tensor] = None):
        """LightningTrainer.validation_step: the objective in eval mode, no gradients.  Returns the device f64 loss scalar.
        `out` (a one-element f64 device tensor that outlives the update) + a batch taken through gather() + `pipelined_validation`: the step
        runs like a training step without its update -- frozen trunk on the caller's stream in the next activation arena, policy head +
        objective on the update stream behind the tails already queued there (so it sees the parameters of the last update without the caller's
        stream waiting for them), loss written to `out` on that stream: read it behind wait_update() / inside update_stream()."""
        if out is not None and self._slot_taken and self.pipelined_validation:
            self._slot_taken = False
            slot = self._slot
            if self._slot_prefetch:
                self.engine.set_prepare_stream(self.prefetch_stream)
            main = torch.cuda.current_stream()
            self._flush_tail()
            try:
                with _ffi.known_stream(main):
                    self.forward_trunk(fb, shard, train=False)
            finally:
                if self._slot_prefetch:
                    self.engine.set_prepare_stream(None)
            self._ev_loss.record(main)
            self.loss, self.lo.loss = out, out.data_ptr()
            with torch.cuda.stream(self._side), _ffi.known_stream(self._side):
                self._side.wait_event(self._ev_loss)
                self.engine.forward_head(0)
                if not getattr(extras, "persistent", False):
                    for t in extras.values():
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(self._side)
                self.set_loss_inputs(extras)
                self.engine.loss_backward_raw(self.kind_id, self.li, self.lo)
                self._exchange_and_finalize(False, None)
                self._ev_param.record(self._side)
                self._ev_tail[slot].record(self._side)
            return out
        self.wait_update()              # (a deferred tail may still be reading the activations this forward is about to overwrite)
        self._loss_slot(None)
        loss = self.forward_loss(fb, extras, train=False, backward=False, shard=shard)
        if out is not None:
            out.copy_(loss.reshape(out.shape))
            return out
        return loss

    @contextlib.contextmanager
    def update_stream(self):
        """Device work issued inside (epoch bookkeeping that reads the step losses or the trained parameters) is queued on the update stream,
        behind the tails issued so far and ahead of the next one, WITHOUT the caller's stream waiting for them; wait_update() afterwards
        covers it.  Without an update stream: the caller's stream, behind wait_update()."""
        self._flush_tail()
        if not self.overlap_update:
            yield
            return
        with torch.cuda.stream(self._side):
            yield
            self._ev_param.record(self._side)

    def on_epoch_end(self):
        self.scheduler.step()
