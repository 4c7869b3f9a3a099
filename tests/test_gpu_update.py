"""A 30-step policy update as a TRAJECTORY: the HIP RLFTTrainer (device collate -> train-mode forward with BatchNorm batch statistics ->
RIFT loss -> analytic pi_head backward -> device clip 0.5 -> native AdamW, WarmupCosLR stepping between epochs) against the CPU
restatement of the reference's update -- oracle forward + autograd through pi_head + torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW with the reference's parameter groups (rift_trainer.py:279-362) and the per-epoch schedule
(warmup_cos_lr.py:39-54).  Drops are off (the RNG streams cannot match).  5 epochs x 6 minibatches of 16 scenes.

What can and cannot agree after 30 AdamW steps.  Adam's step is lr * m / (sqrt(v) + eps): it NORMALISES the gradient, so an element
whose gradient is at the rounding-noise level moves by a fraction of lr per step in a direction set by the last bits of whichever
implementation computed it -- in the reference too.  pi_head has such elements by construction: `mlp.3.bias` and the LayerNorm bias of
every channel whose ReLU is active on all rows shift all logits of a scene by a constant, the log-softmax is invariant to that, their
true gradient is identically zero.  So the trajectory is held to:
  * every step's loss                                    fp32 1e-5 (measured 2.0e-6)      fp16 1.8e-4 (1.2e-4)   bf16 2e-3 (4.5e-4)
  * parameters whose gradient stayed >= 1e-2 of the largest gradient entry on all 30 steps ("well conditioned"; movement 2.0e-3)
                                                         fp32 1e-5 abs (measured 1.9e-6)  fp16 2.5e-4 (1.6e-4)   bf16 6e-4 (2.8e-4)
  * parameters with gradients >= 1e-4 of the largest      fp32 1e-4 (measured 4.0e-5)      fp16 1.5e-3 (5.6e-4)   bf16 2e-3 (6.9e-4)
  * the POLICY the final parameters define: log-probabilities of a held-out batch, evaluated by the oracle with HIP's final pi_head
    vs the oracle's own -- the functional statement of "same update", blind to the shift-invariant directions.  The 30 steps move
    these log-probabilities by 0.53:                     fp32 2e-4 (measured 7.5e-5)      fp16 1.1e-2 (6.9e-3)   bf16 8e-2 (5.0e-2)
  * direction of the whole displacement (cosine)         fp32 > 0.9995 (measured 0.99975)  fp16 > 0.998 (0.99901) bf16 > 0.97 (0.9921)
bf16 / fp16 rows: the trunk's q_final carries the operand rounding of ~50 chained contractions (bf16 ~5e-2 abs, fp16 ~8e-3;
tests/diagnostics/precision_study.py reproduces both on the CPU from operand rounding alone and shows that 16 significand bits in every
contraction would be needed for 1e-5), which perturbs every gradient by 5-15 % (bf16) / 0.5-4 % (fp16).
The 16-bit figures are the MAXIMUM over the kernel variants with a different summation order (round 4: reference-line rounds of the
PointsEncoder packed / two whole lines per round, RIFT_PE_PACK=1 / 0: fp16 step loss 1.19e-4 / 8.6e-5, well-conditioned parameters 1.63e-4 /
8.6e-5, policy 4.5e-3 / 6.9e-3), every bar >= 1.5x that maximum.  Round 3 quoted the fp16 step losses as "within 1e-4 (8.7e-5)": that
held for ONE arithmetic order; across orders the honest figure is 1.2e-4.
"""
import os

import numpy as np
import pytest
import torch

from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu

EPOCHS, STEPS_PER_EPOCH, BATCH, N_SCENES = 5, 6, 16, 96
PI = "planning_decoder.pi_head."


def _oracle_trajectory(sd, scenes, order):
    """The reference's update loop on the host: returns the pi_head parameters after every step and the step losses."""
    from oracle import advantage as oadv, losses, pluto_ref
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    params = {k: sd[PI + k].clone().requires_grad_(True) for k in losses.PI_KEYS}
    groups = [{"params": [params["mlp.0.weight"], params["mlp.3.weight"]], "weight_decay": 1e-5},       # nn.Linear weights decay
              {"params": [params[k] for k in ("mlp.0.bias", "mlp.1.bias", "mlp.1.weight", "mlp.3.bias")], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-5)
    step_losses = []
    gmin = {k: torch.full_like(v, float("inf")) for k, v in params.items()}      # smallest |gradient| an element saw over the 30 steps
    gmax = 0.0
    for epoch in range(EPOCHS):
        lr = oadv.warmup_cos_lr(epoch, 1e-4, 1e-4 * 0.9, 2, EPOCHS)
        for g in opt.param_groups:
            g["lr"] = lr
        for s in range(STEPS_PER_EPOCH):
            pick = order[epoch * STEPS_PER_EPOCH + s]
            batch = syn.collate_scenes([scenes[i] for i in pick])
            data = batch["cur_pluto_feature_torch"]
            _, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
            r_pad = ~data["reference_line"]["valid_mask"].any(-1)
            live = {PI + k: v for k, v in params.items()}
            pi = pluto_ref.mlp_layer(taps["q_final"], pluto_ref.SD(live, PI)).squeeze(-1)
            prob = pi.masked_fill(r_pad.unsqueeze(-1), -1e6)
            loss = losses.rift_loss(prob, r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                    batch["group_advantage_mask_torch"])
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(list(params.values()), 0.5)
            for k, v in params.items():
                gmin[k] = torch.minimum(gmin[k], v.grad.abs())
                gmax = max(gmax, float(v.grad.abs().max()))
            opt.step()
            step_losses.append(float(loss))
    return {k: v.detach().clone() for k, v in params.items()}, step_losses, {k: v / gmax for k, v in gmin.items()}


@pytest.fixture(scope="module")
def oracle_run():
    sd = H.weights()
    scenes = [syn.make_scene(2000 + i, 24, 10, 1, 5) for i in range(N_SCENES)]
    g = torch.Generator().manual_seed(77)
    order = [torch.randperm(N_SCENES, generator=g)[:BATCH].tolist() for _ in range(EPOCHS * STEPS_PER_EPOCH)]
    final, step_losses, gcond = _oracle_trajectory(sd, scenes, order)
    return sd, scenes, order, final, step_losses, gcond


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_thirty_step_update_trajectory(oracle_run, precision):
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    sd, scenes, order, want, want_losses, gcond = oracle_run
    model = PlanningModel(radius=120, drop_path=0.0, dropout=0.0, state_dropout=0.0)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    model.compute_precision, model.need_traj = precision, False
    model.train()
    tr = RLFTTrainer(model, kind="rift", lr=1e-4, cl_lr_decay=0.9, weight_decay=1e-5, epochs=EPOCHS, warmup_epochs=2)
    replay = DeviceReplay(scenes, "cuda:0")
    got_losses = []
    for epoch in range(EPOCHS):
        for s in range(STEPS_PER_EPOCH):
            pick = order[epoch * STEPS_PER_EPOCH + s]
            idx = torch.tensor(pick, dtype=torch.int32, device="cuda:0")
            fb, b = replay.collate(tr.engine, idx, int(replay.r_count_cpu[pick].max()))
            tr.training_step(fb, b)
            got_losses.append(tr.step_loss())           # (written on the update stream: read behind wait_update())
        mean = tr.pop_mean_loss()           # the epoch's mean of the per-step device losses (one host read; slots of loss_hist)
        assert abs(mean - sum(got_losses[-STEPS_PER_EPOCH:]) / STEPS_PER_EPOCH) < 1e-12
        tr.on_epoch_end()
    torch.cuda.synchronize()
    from oracle import pluto_ref
    start = {k: sd[PI + k] for k in want}
    got = {k: tr.params[k].detach().cpu() for k in want}
    well, mid, move = 0.0, 0.0, 0.0
    dots = [0.0, 0.0, 0.0]
    for k, ref in want.items():
        d = (got[k] - ref).abs()
        print(f"   {k:14s} max |param - oracle| {float(d.max()):.3e}   oracle movement {float((ref - start[k]).abs().max()):.3e}   by conditioning: " +
              "  ".join(f">{thr:.0e}: {float((d * (gcond[k] > thr)).max()):.2e} (n={int((gcond[k] > thr).sum())})" for thr in (1e-2, 1e-3, 1e-4, 1e-5)))
        well = max(well, float((d * (gcond[k] > 1e-2)).max()))
        mid = max(mid, float((d * (gcond[k] > 1e-4)).max()))
        move = max(move, float(((ref - start[k]).abs() * (gcond[k] > 1e-2)).max()))
        a, c = (got[k] - start[k]).double().flatten(), (ref - start[k]).double().flatten()
        dots[0] += float(a @ c); dots[1] += float(a @ a); dots[2] += float(c @ c)
    cosine = dots[0] / (dots[1] * dots[2]) ** 0.5
    loss_err = max(abs(a - c) for a, c in zip(got_losses, want_losses))
    # the policy on a held-out batch (oracle trunk, either pi_head): log-softmax over each scene's candidates
    held = syn.collate_scenes([syn.make_scene(2500 + i, 24, 10, 1, 5) for i in range(16)])
    data = held["cur_pluto_feature_torch"]
    _, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)

    def logp(params):
        pi = pluto_ref.mlp_layer(taps["q_final"], pluto_ref.SD({PI + k: v for k, v in params.items()}, PI)).squeeze(-1)
        return torch.log_softmax(pi.masked_fill(r_pad.unsqueeze(-1), -1e8).view(pi.shape[0], -1), dim=1)[~r_pad.repeat_interleave(12, dim=1)]
    policy_err = float((logp(got) - logp(want)).abs().max())
    policy_move = float((logp(want) - logp(start)).abs().max())
    print(f"30-step update [{precision}]: well-conditioned params {well:.3e} (movement {move:.3e}), gradient > 1e-4 params {mid:.3e}, cosine {cosine:.6f}, "
          f"max step-loss error {loss_err:.3e}, held-out log-prob error {policy_err:.3e} (the update moved them by {policy_move:.3e})")
    assert move > 1e-3 and policy_move > 1e-3            # the parameters and the policy really moved
    if precision == "fp32":
        assert loss_err < 1e-5 and well < 1e-5 and mid < 1e-4 and policy_err < 2e-4 and cosine > 0.9995
    elif precision == "fp16":      # fp16 operands: step losses within 1.2e-4 (measured, worst variant), the learnt policy within 7e-3; bars 1.5x
        assert loss_err < 1.8e-4 and well < 2.5e-4 and mid < 1.5e-3 and policy_err < 1.1e-2 and cosine > 0.998
    else:
        assert loss_err < 2e-3 and well < 6e-4 and mid < 2e-3 and policy_err < 8e-2 and cosine > 0.97


@pytest.mark.gpu
def test_mean_loss_accounting_across_slot_folds(monkeypatch):
    """pop_mean_loss() = mean of the step losses also when the steps outnumber the loss slots (they are folded into the accumulator
    every LOSS_SLOTS steps) and with validation steps in between (validation writes its own scalar)."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    from rift_amd import synthetic as syn
    from tests import helpers as H
    monkeypatch.setattr(RLFTTrainer, "LOSS_SLOTS", 4)
    model = PlanningModel(radius=120, drop_path=0.0, dropout=0.0, state_dropout=0.0)
    model.load_state_dict(H.weights())
    model = model.to("cuda:0")
    model.compute_precision, model.need_traj = "bf16", False
    model.train()
    tr = RLFTTrainer(model, kind="rift", lr=1e-4, cl_lr_decay=0.9, weight_decay=1e-5, epochs=4, warmup_epochs=1)
    scenes = [syn.make_scene(40 + i, r_min=3, r_max=5) for i in range(8)]
    replay = DeviceReplay(scenes, "cuda:0")
    idx = torch.arange(8, dtype=torch.int32, device="cuda:0")
    for n in (3, 4, 9, 1):
        losses = []
        for i in range(n):
            fb, b = replay.collate(tr.engine, idx)
            tr.training_step(fb, b)
            losses.append(tr.step_loss())
            if i % 2 == 0:
                fb, b = replay.collate(tr.engine, idx)
                tr.validation_step(fb, b)
        assert abs(tr.pop_mean_loss() - sum(losses) / n) < 1e-12
    assert tr.pop_mean_loss() == 0.0


@pytest.mark.parametrize("nscene", [128, 48])
def test_validation_between_prefetched_training_steps(monkeypatch, nscene):
    """Training steps whose batch is gathered on the prefetch stream (RLFTTrainer.gather, four batch-buffer sets / activation arenas), with a
    validation step on the caller's stream -- batch-buffer set 0, activation arena 0 -- after every third one, as the update loop interleaves
    them: the host runs ahead, so the gather of the step behind a validation is issued while that validation is still running and must wait
    for it (the trainer's serial-step event).  Losses and parameters equal the run without the prefetch stream bit for bit, and the run
    without any second stream; 128-scene batches, so that a validation forward is long enough to be overrun, and 48-scene batches, where the
    pipelined steps leave their planning decoder to the deferred head (engine.hip: dec_defer_max) and the validation forwards do not."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(2000 + i) for i in range(160)]
    sd = H.weights()
    g = torch.Generator().manual_seed(7)
    train_ix = [torch.randperm(160, generator=g)[:nscene].to(torch.int32).to(dev) for _ in range(12)]
    val_ix = torch.arange(nscene, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    runs = {}
    for mode, (pipeline, prefetch) in {"serial": ("0", "0"), "tail": ("1", "0"), "prefetch": ("1", "1")}.items():
        monkeypatch.setenv("RIFT_PIPELINE", pipeline)
        monkeypatch.setenv("RIFT_PREFETCH", prefetch)
        replay = DeviceReplay(scenes, dev, rcap=6)
        model = PlanningModel(radius=120)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model = model.to(dev)
        model.need_traj = False
        model.train()
        tr = RLFTTrainer(model, kind="rift", seed=3)
        assert (tr.prefetch_stream is not None) == (mode == "prefetch")
        vals = []
        for k, ix in enumerate(train_ix):
            fb, b = tr.gather(replay, ix)
            tr.training_step(fb, b)
            if k % 3 == 2:
                tr.wait_update()
                fb, b = replay.collate(tr.engine, val_ix)
                vals.append(tr.validation_step(fb, b).clone())
        mean = tr.pop_mean_loss()
        torch.cuda.synchronize()
        runs[mode] = (mean, [float(v.item()) for v in vals], {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith(PI)})
        tr.close()
    for other in ("tail", "prefetch"):
        assert runs[other][0] == runs["serial"][0] and runs[other][1] == runs["serial"][1], (other, runs[other][:2], runs["serial"][:2])
        for k, v in runs["serial"][2].items():
            assert torch.equal(v, runs[other][2][k]), (other, k)


def test_pipeline_with_changing_batch_shapes(monkeypatch):
    """Consecutive pipelined steps whose batches differ in size and in reference-line count (8 x R 2, 96 x R 6, 24 x R 4, ...): the arena-sizing
    pass of rift_forward is skipped only between forwards of one shape, an arena that has to grow does so behind a device-wide synchronize (its
    last readers sit on the update / prepare streams), every (size, R) keeps its own batch buffers and outputs.  Parameters and the mean loss
    equal the serial run bit for bit."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(2600 + i) for i in range(128)]
    sd = H.weights()
    g = torch.Generator().manual_seed(9)
    plan = [(8, 2), (96, 6), (24, 4), (96, 6), (8, 2), (128, 5), (24, 4), (128, 6), (8, 3), (96, 6), (96, 6), (24, 4)]
    picks = [torch.randperm(128, generator=g)[:n].to(torch.int32).to(dev) for n, _ in plan]
    torch.cuda.synchronize()
    runs = {}
    for mode, (pipeline, prefetch) in {"serial": ("0", "0"), "pipeline": ("1", "1")}.items():
        monkeypatch.setenv("RIFT_PIPELINE", pipeline)
        monkeypatch.setenv("RIFT_PREFETCH", prefetch)
        replay = DeviceReplay(scenes, dev, rcap=6)
        model = PlanningModel(radius=120)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model = model.to(dev)
        model.need_traj = False
        model.train()
        tr = RLFTTrainer(model, kind="rift", seed=5)
        for ix, (_, R) in zip(picks, plan):
            fb, b = tr.gather(replay, ix, R)
            tr.training_step(fb, b)
        mean = tr.pop_mean_loss()
        torch.cuda.synchronize()
        runs[mode] = (mean, {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith(PI)})
        tr.close()
        model.release_engine()
    assert runs["pipeline"][0] == runs["serial"][0], (runs["pipeline"][0], runs["serial"][0])
    for k, v in runs["serial"][1].items():
        assert torch.equal(v, runs["pipeline"][1][k]), k


@pytest.mark.parametrize("pipeline,lr", [("0", 1e-4), ("1", 1e-4), ("1", 1e-3), ("0", 3e-3)])
def test_fused_update_tail_equals_the_two_launch_tail_bit_for_bit(monkeypatch, pipeline, lr):
    """rift_update_tail (finalize + gradient-norm clip + AdamW of the six pi_head tensors in ONE launch) against rift_loss_finalize_clip followed by
    rift_adamw_step: after 40 steps over two learning rates -- the first one through torch's own optimizer.step(), which creates the state --
    the parameters, the clipped .grad tensors, exp_avg / exp_avg_sq / the device step counters, the gradient norm and the mean loss are
    identical bit for bit, and the fused run did issue the fused launch (prof names)."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    monkeypatch.setenv("RIFT_PIPELINE", pipeline)
    monkeypatch.setenv("RIFT_PREFETCH", pipeline)
    scenes = [syn.make_scene(3100 + i) for i in range(64)]
    sd = H.weights()
    g = torch.Generator().manual_seed(11)
    picks = [torch.randperm(64, generator=g)[:32].to(torch.int32).to(dev) for _ in range(40)]
    torch.cuda.synchronize()
    runs = {}
    for mode in ("two", "fused"):
        monkeypatch.setenv("RIFT_FUSED_TAIL", "1" if mode == "fused" else "0")
        replay = DeviceReplay(scenes, dev, rcap=6)
        model = PlanningModel(radius=120)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model = model.to(dev)
        model.need_traj = False
        model.train()
        tr = RLFTTrainer(model, kind="rift", seed=5, lr=lr)
        assert tr.fused_tail == (mode == "fused")
        for k, ix in enumerate(picks):
            if k == 7:
                tr.wait_update()
                tr.on_epoch_end()          # the scheduler moves param_groups['lr']: the fused call reads it per step like the separate one
            fb, b = tr.gather(replay, ix)
            tr.training_step(fb, b)
        mean = tr.pop_mean_loss()
        tr.wait_update()
        torch.cuda.synchronize()
        state = {}
        for name, p in model.named_parameters():
            if name.startswith(PI):
                st = tr.optimizer.state[p]
                state[name] = (p.detach().cpu().clone(), p.grad.detach().cpu().clone(), st["exp_avg"].cpu().clone(), st["exp_avg_sq"].cpu().clone(),
                               st["step"].cpu().clone())
        runs[mode] = (mean, float(tr.grad_norm.item()), state, tr._adam_step)
        tr.close()
        model.release_engine()
    assert runs["fused"][0] == runs["two"][0] and runs["fused"][1] == runs["two"][1] and runs["fused"][3] == runs["two"][3] == 40
    assert len(runs["two"][2]) == 6
    for name, ref in runs["two"][2].items():
        for what, a, b in zip(("param", "grad", "exp_avg", "exp_avg_sq", "step"), ref, runs["fused"][2][name]):
            assert torch.equal(a, b), (name, what, float((a - b).abs().max()))
        assert float(ref[4]) == 40.0


def test_validation_through_the_step_pipeline_equals_validation_on_the_callers_stream(monkeypatch):
    """RLFTTrainer.validation_step(out=...) on batches taken through gather(): eval-mode trunk in the next activation arena on the caller's
    stream, head + objective on the update stream behind the tails queued there -- three epochs of 5 training steps + two validation batches
    (40 and 23 scenes), bookkeeping inside update_stream() as RLFTPluto._train issues it.  Validation losses, mean training losses and the
    final parameters equal the run whose validation steps are whole steps on the caller's stream (the data-parallel path), bit for bit."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(3300 + i) for i in range(128)]
    sd = H.weights()
    g = torch.Generator().manual_seed(13)
    picks = [torch.randperm(128, generator=g)[:48].to(torch.int32).to(dev) for _ in range(15)]
    vix = [torch.arange(40, dtype=torch.int32, device=dev), torch.arange(23, dtype=torch.int32, device=dev) + 90]
    torch.cuda.synchronize()
    runs = {}
    for mode in ("serial", "piped"):
        monkeypatch.setenv("RIFT_PIPELINE_VAL", "1" if mode == "piped" else "0")
        replay = DeviceReplay(scenes, dev, rcap=6)
        model = PlanningModel(radius=120)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model = model.to(dev)
        model.need_traj = False
        model.train()
        tr = RLFTTrainer(model, kind="rift", seed=9, lr=1e-3)
        assert tr.pipelined_validation == (mode == "piped")
        table = torch.zeros(3, 2, dtype=torch.float64, device=dev)
        vtab = torch.zeros(3, 2, dtype=torch.float64, device=dev)
        for e in range(3):
            for k in range(5):
                fb, b = tr.gather(replay, picks[e * 5 + k])
                tr.training_step(fb, b)
            if mode == "piped":
                for j, ix in enumerate(vix):
                    fb, b = tr.gather(replay, ix)
                    tr.validation_step(fb, b, out=vtab[e, j:j + 1])
                with tr.update_stream():
                    tr.pop_mean_loss_async(table[e, 0], in_update_stream=True)
                    table[e, 1].copy_(vtab[e].mean())
            else:
                tr.pop_mean_loss_async(table[e, 0])
                for j, ix in enumerate(vix):
                    fb, b = replay.collate(tr.engine, ix, None, slot=0)
                    tr.validation_step(fb, b, out=vtab[e, j:j + 1])
                table[e, 1].copy_(vtab[e].mean())
            tr.on_epoch_end()
        tr.wait_update()
        torch.cuda.synchronize()
        runs[mode] = (table.cpu(), vtab.cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith(PI) or "running_" in k})
        tr.close()
        model.release_engine()
    assert torch.equal(runs["piped"][0], runs["serial"][0]) and torch.equal(runs["piped"][1], runs["serial"][1]), (runs["piped"][:2], runs["serial"][:2])
    assert float(runs["serial"][1].min()) > 0 and len(runs["serial"][2]) > 6
    for k, v in runs["serial"][2].items():
        assert torch.equal(v, runs["piped"][2][k]), k
