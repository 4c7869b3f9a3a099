"""Data-parallel update on the HIP path: a minibatch sharded over ranks, with the exchanges of rift_set_dp
(BatchNorm sums + r2r quirk masks inside the forward) and of the loss (gradient sums, objective sum, count), must equal the
single-process update on the whole minibatch -- loss, pi_head gradients and BatchNorm running statistics.

One GPU is enough to prove it: the ranks are run one after the other through the same C-ABI with a loop-back exchange that replays,
at exchange call k, the sum over ranks of what they contributed at call k (one extra pass per exchange call, since a later exchange
depends on the earlier ones being the reduced values).  The RCCL transport itself is exercised with one rank in-process and, when
the box has two GPUs, with two torchrun ranks.
"""
import os
import subprocess
import sys

import pytest
import torch

from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BN_STATS = [f"{p}.{m}.1.{s}" for p in ("map_encoder.polygon_encoder", "planning_decoder.r_encoder")
            for m in ("first_mlp", "second_mlp") for s in ("running_mean", "running_var")]


def _model(precision):
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    m = PlanningModel(radius=120, drop_path=0.0, dropout=0.0, state_dropout=0.0)       # drops off: the RNG streams are per rank
    m.load_state_dict(H.weights())
    m = m.to("cuda:0")
    m.compute_precision = precision
    m.need_traj = False
    m.train()
    return m


class Loopback:
    """Sequential emulation of `world` ranks: exchange call k of every rank yields sum_r(contribution of rank r at call k)."""

    def __init__(self, world):
        self.world, self.total, self.pending = world, [], {}
        self.k = [0] * world

    def start_pass(self):
        self.k = [0] * self.world
        self.pending = {}

    def fn(self, r):
        def exchange(t):
            k = self.k[r]
            self.k[r] += 1
            if k < len(self.total):
                t.copy_(self.total[k])
            elif k == len(self.total):
                self.pending[r] = t.clone()
        return exchange

    def end_pass(self):
        """True when every exchange call of the pass was a replay (the results of this pass are final)."""
        if len(self.pending) == self.world:
            self.total.append(sum(self.pending.values()))
            return False
        assert not self.pending
        return True


@pytest.mark.parametrize("precision,sizes,dense", [("fp32", (6, 6), False), ("bf16", (6, 6), False), ("fp32", (5, 4, 3), False), ("bf16", (5, 4, 3), False),
                                                  ("fp16", (6, 6), False), ("fp16", (5, 4, 3), False),
                                                  ("bf16", (3, 3), True), ("fp16", (2, 3, 1), True)])      # dense: BASELINE configs[4] shapes (enc_w / dec_w<., true>)
def test_sharded_step_equals_single_process_step(precision, sizes, dense):
    from rift_amd import _ffi
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.replay import DeviceReplay
    n, world = sum(sizes), len(sizes)
    # heterogeneous reference-line counts: the r2r quirk couples scenes
    scenes = [syn.make_scene(300 + i, 128, 40, 8, 16) if dense else syn.make_scene(300 + i) for i in range(n)]
    replay = DeviceReplay(scenes, "cuda:0")
    assert len(set(replay.r_count_cpu.tolist())) > 1
    R = replay.Rcap
    idx = torch.arange(n, dtype=torch.int32, device="cuda:0")

    single = RLFTTrainer(_model(precision), kind="rift")
    fb, b = replay.collate(single.engine, idx, R)
    want_loss = float(single.forward_loss(fb, b, train=True).item())
    want_grads = {k: p.grad.clone() for k, p in single.params.items()}
    want_stats = {k: v.clone() for k, v in single.model.state_dict().items() if k in BN_STATS}

    lb = Loopback(world)
    ranks = [RLFTTrainer(_model(precision), kind="rift", exchange=lb.fn(r), dp_rank=r, dp_world=world) for r in range(world)]
    lo = [sum(sizes[:r]) for r in range(world)]
    for _ in range(8):
        lb.start_pass()
        for r, tr in enumerate(ranks):
            fbr, br = replay.collate(tr.engine, idx[lo[r]:lo[r] + sizes[r]], R)
            tr.forward_loss(fbr, dict(br), train=True, flags_extra=_ffi.F_NO_BN_UPDATE, shard=(lo[r], n))
        if lb.end_pass():
            break
    else:
        raise AssertionError("the loop-back exchange did not converge")
    n_calls = len(lb.total)
    assert n_calls == (5 if precision == "fp32" else 3)            # fused: 2 BatchNorm points + loss; layer-wise fp32: 4 + loss
    # final pass once more WITH the running-statistics update (every exchange replays the reduced values)
    lb.start_pass()
    for r, tr in enumerate(ranks):
        fbr, br = replay.collate(tr.engine, idx[lo[r]:lo[r] + sizes[r]], R)
        tr.forward_loss(fbr, dict(br), train=True, shard=(lo[r], n))
    assert lb.end_pass()
    torch.cuda.synchronize()
    # Same arithmetic per scene; what differs is the grouping of the fp32 per-tile BatchNorm partial sums (a tile = 120 point rows does not
    # end at a scene boundary unless the shard does: the (6, 6) split is bit-identical, the (5, 4, 3) split is not) -> BatchNorm scale /
    # shift differ in the last fp32 bit.  The exact-fp32 trunk carries that through as ~1e-7; the bf16 trunk can turn it into a flipped
    # operand rounding (2^-9 relative on one activation), measured 3.7e-5 on this 12-scene loss -- and a logit that moves by that much can
    # carry one of the ~500 candidates across a clip boundary of the piecewise RIFT objective, which switches its whole gradient
    # contribution (measured 9.5 % of the largest gradient entry).  Hence two bars; the aligned split stays bit-identical in both modes.
    # (Round 4: the reference-line rounds of pe_w_kernel are PACKED from the lines' valid-prefix tiles, so which lines share a round -- and
    # with it the grouping of the BatchNorm-2 partial sums -- depends on the batch a rank holds: no split of a 16-bit-operand step is
    # bit-identical to the single-process step any more, every one is "equal up to the fp32 summation order of the statistics", as the
    # unaligned splits -- among them 256 scenes over 8 ranks, 32 x 20 polygons = 53.3 rounds -- always were.  RIFT_PE_PACK=0 restores the
    # bit-identical aligned split.)
    n_poly = 40 if dense else 20
    aligned = os.environ.get("RIFT_PE_PACK") == "0" and all((sum(sizes[:r]) * n_poly) % 12 == 0 for r in range(world))     # shards start on a PointsEncoder round (12 polygons of 20 points)
    tol, gtol = (1e-6, 1e-4) if (precision == "fp32" or aligned) else (2e-4, 0.25)
    for r, tr in enumerate(ranks):
        assert abs(float(tr.loss.item()) - want_loss) < tol, (r, float(tr.loss.item()), want_loss)
        for k, p in tr.params.items():
            ref = want_grads[k]
            assert float((p.grad - ref).abs().max()) < 1e-6 + gtol * float(ref.abs().max()), (r, k)
        sd = tr.model.state_dict()
        for k, ref in want_stats.items():
            assert float((sd[k] - ref).abs().max()) < 1e-6 + 1e-5 * float(ref.abs().max()), (r, k)
        tr.close()


def test_sharded_eval_forward_uses_global_quirk_masks():
    """Validation step under data parallelism: no BatchNorm exchange, but the r2r quirk still reads the global padding rows."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.replay import DeviceReplay
    sizes, n = (4, 5), 9
    scenes = [syn.make_scene(700 + i) for i in range(n)]
    replay = DeviceReplay(scenes, "cuda:0")
    R = replay.Rcap
    idx = torch.arange(n, dtype=torch.int32, device="cuda:0")
    single = RLFTTrainer(_model("fp32"), kind="rift")
    fb, b = replay.collate(single.engine, idx, R)
    want = float(single.validation_step(fb, b).item())
    lb = Loopback(2)
    ranks = [RLFTTrainer(_model("fp32"), kind="rift", exchange=lb.fn(r), dp_rank=r, dp_world=2) for r in range(2)]
    lo = [0, sizes[0]]
    for _ in range(4):
        lb.start_pass()
        for r, tr in enumerate(ranks):
            fbr, br = replay.collate(tr.engine, idx[lo[r]:lo[r] + sizes[r]], R)
            tr.validation_step(fbr, dict(br), shard=(lo[r], n))
        if lb.end_pass():
            break
    assert len(lb.total) == 2                      # quirk masks, loss sums
    for tr in ranks:
        assert abs(float(tr.loss.item()) - want) < 1e-6
        tr.close()


def test_rccl_transport_with_one_rank():
    """The product path with a real RCCL process group (world size 1, exchanges forced): same step as without a group."""
    import torch.distributed as dist
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.replay import DeviceReplay
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        replay = DeviceReplay([syn.make_scene(40 + i) for i in range(16)], "cuda:0")
        idx = torch.arange(16, dtype=torch.int32, device="cuda:0")
        plain = RLFTTrainer(_model("bf16"), kind="rift")
        fb, b = replay.collate(plain.engine, idx)
        plain.training_step(fb, b)
        l0 = plain.step_loss()              # (the loss is written on the update stream: read behind wait_update())
        dp = RLFTTrainer(_model("bf16"), kind="rift", process_group=dist.group.WORLD)
        dp.force_exchange = True
        fb, b = replay.collate(dp.engine, idx)
        dp.training_step(fb, b)
        l1 = dp.step_loss()
        torch.cuda.synchronize()
        assert abs(l0 - l1) < 1e-9
        for k in plain.params:
            assert torch.equal(plain.params[k], dp.params[k]), k
        dp.close()
        # the deferred tail with the exchanges in flight on several streams (BatchNorm sums inside the map chain, the loss sums on the update
        # stream) over the same communicator: nine steps equal nine steps without a group, bit for bit -- gathered on the caller's stream
        # (next_slot) and on the prefetch stream (gather: the data-parallel tail is then issued one step late, behind the next forward, and
        # the fronts run ahead), with a host read of the loss in the middle (the held-back tail is flushed)
        g = torch.Generator().manual_seed(5)
        # (minibatches of changing size and reference-line count: a tail issued late must carry its own step's loss descriptor)
        sizes = [8, 12, 6, 8, 16, 6, 12, 8, 10]
        order = [torch.randperm(16, generator=g)[:n].to(torch.int32).to("cuda:0") for n in sizes]
        rmax = [int(replay.r_count_cpu[ix.cpu().long()].max()) for ix in order]
        assert len(set(rmax)) > 1
        torch.cuda.synchronize()
        res = []
        for grp, pre in ((None, False), (dist.group.WORLD, False), (dist.group.WORLD, True), (None, True)):
            tr = RLFTTrainer(_model("bf16"), kind="rift", process_group=grp)
            tr.force_exchange = grp is not None
            assert tr.pipeline
            mid = None
            for n, ix in enumerate(order):
                if pre:
                    fb, b = tr.gather(replay, ix, rmax[n])
                else:
                    fb, b = replay.collate(tr.engine, ix, rmax[n], slot=tr.next_slot())
                tr.training_step(fb, b)
                if n == 4:
                    mid = tr.step_loss()
            mean = tr.pop_mean_loss()
            torch.cuda.synchronize()
            res.append((mid, mean, {k: v.detach().clone() for k, v in tr.params.items()}))
            tr.close()
        for other in res[1:]:
            assert other[0] == res[0][0] and other[1] == res[0][1], (other[:2], res[0][:2])
            for k in res[0][2]:
                assert torch.equal(res[0][2][k], other[2][k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the box")
def test_two_rccl_ranks_equal_single_process():
    """Two torchrun ranks over RCCL: the sharded update (3 optimizer steps) leaves the same pi_head on both ranks as one process."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", os.path.join(REPO, "tests", "dp_worker.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP_WORKER_OK" in out.stdout


@pytest.mark.parametrize("ranks", [2, 3])
def test_real_ranks_sharing_one_gpu_over_gloo(ranks):
    """Two / three REAL ranks -- processes of their own, every one on GPU 0, exchanging over gloo (RCCL refuses two ranks on one device) --
    run tests/dp_worker.py: the sharded fp32 update against the single-process one, and the data-parallel step pipeline (batches gathered on
    the prefetch stream, the tail issued one step late behind the next forward, host reads in between) against the single-process pipeline.
    What one rank cannot show: that the order in which the ranks issue their all-reduces is the same everywhere and free of cycles."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29640 + ranks), os.path.join(REPO, "tests", "dp_worker.py")], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, RIFT_DP_SAME_GPU="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "DP_WORKER_OK" in out.stdout


def test_bench_two_ranks_sharing_one_gpu():
    """bench.py's N > 1 code (both scaling legs, the max-over-ranks timing, the JSON line of rank 0) with two ranks on GPU 0 over gloo: it
    completes and reports both legs.  (The numbers of such a run mean nothing: the all-reduces go through the host.)"""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29657", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--replay", "1024"],
                         capture_output=True, text=True, timeout=1200, env=dict(os.environ, RIFT_BENCH_SAME_GPU="1"))
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 6
    for leg in ("weak", "strong"):
        assert line[leg]["ms_per_step"] > 0 and line[leg]["global_batch"] == (512 if leg == "weak" else 256)
    # the headline is SURVEY.md 8(e)'s partition: the 256-scene minibatch split over the ranks (strong scaling), with the roofline and the
    # CPU baseline in the line whatever N is
    assert line["scaling"] == "strong" and line["value"] == pytest.approx(line["strong"]["value"])
    assert line["config"]["global_batch"] == 256 and line["config"]["per_gpu_batch"] == 128
    assert line["roofline"]["kernel"] and line["cpu_baseline"]["value"] > 0


def test_library_owned_communicator_with_one_rank():
    """`rift_comm_init(ctx, ncclUniqueId, rank, world)` -- a communicator the LIBRARY owns (RCCL through dlopen), for hosts without
    torch.distributed.  One rank on this one-GPU box: the unique id round trip, ncclCommInitRank, an all-reduce of the loss exchange buffer
    (identity with one rank) and a sharded-descriptor training forward whose three exchanges go over that communicator (rift_set_dp with
    exchange = NULL) -- loss and gradients equal the step without a group, bit for bit.  A second rift_comm_init on the same context and a
    descriptor without callback or communicator are refused."""
    from rift_amd import _ffi
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.replay import DeviceReplay
    n = 6
    scenes = [syn.make_scene(300 + i) for i in range(n)]
    replay = DeviceReplay(scenes, "cuda:0")
    idx = torch.arange(n, dtype=torch.int32, device="cuda:0")
    R = replay.Rcap
    single = RLFTTrainer(_model("fp16"), kind="rift")
    fb, b = replay.collate(single.engine, idx, R)
    want = float(single.forward_loss(fb, b, train=True).item())
    want_grads = {k: p.grad.clone() for k, p in single.params.items()}
    single.close()

    tr = RLFTTrainer(_model("fp16"), kind="rift")
    eng = tr.engine
    xchg = torch.zeros(n * 16 + 1026, dtype=torch.float64, device="cuda:0")
    with pytest.raises(RuntimeError, match="no exchange callback and no library communicator"):
        eng.set_dp_library_comm(0, n, xchg)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(uid, 0, 1)
    with pytest.raises(RuntimeError, match="has a communicator already"):
        eng.comm_init(uid, 0, 1)
    t = torch.arange(16899, dtype=torch.float64, device="cuda:0")
    eng.comm_all_reduce(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(16899, dtype=torch.float64))
    eng.set_dp_library_comm(0, n, xchg)
    fb, b = replay.collate(eng, idx, R)
    got = float(tr.forward_loss(fb, b, train=True).item())          # (the trainer has no exchange of its own: the loss sums stay local -- one rank)
    assert got == want
    for k, p in tr.params.items():
        assert torch.equal(p.grad, want_grads[k]), k
    eng.clear_dp()
    eng.comm_destroy()
    eng.comm_destroy()                                               # idempotent
    tr.close()
