"""A 30-step policy update as a TRAJECTORY: the HIP RLFTTrainer (device collate -> train-mode forward with BatchNorm batch statistics ->
RIFT loss -> analytic pi_head backward -> device clip 0.5 -> native AdamW, WarmupCosLR stepping between epochs) against the CPU
restatement of the reference's update -- oracle forward + autograd through pi_head + torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW with the reference's parameter groups (rift_trainer.py:279-362) and the per-epoch schedule
(warmup_cos_lr.py:39-54).  Drops are off (the RNG streams cannot match).

What is compared is the trainable state after 30 optimizer steps (5 epochs x 6 minibatches of 16 scenes): the six pi_head tensors.
Tolerances, with what decides them:
  fp32 mode: 1e-5 abs (the bar VERDICT r1 set).  Adam's update is lr * m / (sqrt(v) + eps): with lr <= 1e-4 thirty steps move a weight
             by <= 3e-3, and the fp32 trunk reproduces q_final to ~1e-6, so the difference is rounding noise of the optimizer arithmetic.
  bf16 mode: the trunk's q_final carries bf16 operand rounding (~5e-2 abs, tests/diagnostics/precision_study.py), which perturbs every
             gradient by 5-15 %; Adam normalises gradient magnitudes, so each step can still differ by a fraction of lr per element.
             Measured on MI355X: 2.4e-4 abs after 30 steps (8 % of the largest parameter movement).  Held to 6e-4, with the direction
             of the update checked separately (cosine > 0.97 between the two parameter displacements).
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
            opt.step()
            step_losses.append(float(loss))
    return {k: v.detach().clone() for k, v in params.items()}, step_losses


@pytest.fixture(scope="module")
def oracle_run():
    sd = H.weights()
    scenes = [syn.make_scene(2000 + i, 24, 10, 1, 5) for i in range(N_SCENES)]
    g = torch.Generator().manual_seed(77)
    order = [torch.randperm(N_SCENES, generator=g)[:BATCH].tolist() for _ in range(EPOCHS * STEPS_PER_EPOCH)]
    final, step_losses = _oracle_trajectory(sd, scenes, order)
    return sd, scenes, order, final, step_losses


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_thirty_step_update_trajectory(oracle_run, precision):
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    sd, scenes, order, want, want_losses = oracle_run
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
            got_losses.append(float(tr.training_step(fb, b).item()))
        tr.pop_mean_loss()
        tr.on_epoch_end()
    torch.cuda.synchronize()
    start = {k: sd[PI + k] for k in want}
    worst, move = 0.0, 0.0
    dots = [0.0, 0.0, 0.0]
    for k, ref in want.items():
        got = tr.params[k].detach().cpu()
        worst = max(worst, float((got - ref).abs().max()))
        move = max(move, float((ref - start[k]).abs().max()))
        a, c = (got - start[k]).double().flatten(), (ref - start[k]).double().flatten()
        dots[0] += float(a @ c); dots[1] += float(a @ a); dots[2] += float(c @ c)
    cosine = dots[0] / (dots[1] * dots[2]) ** 0.5
    loss_err = max(abs(a - c) for a, c in zip(got_losses, want_losses))
    print(f"30-step update [{precision}]: max |param - oracle| {worst:.3e} (largest movement {move:.3e}), cosine {cosine:.6f}, "
          f"max step-loss error {loss_err:.3e}")
    assert move > 1e-3                                   # the parameters really moved
    if precision == "fp32":
        assert worst < 1e-5 and loss_err < 1e-5 and cosine > 0.99999
    else:
        assert worst < 6e-4 and loss_err < 2e-3 and cosine > 0.97
