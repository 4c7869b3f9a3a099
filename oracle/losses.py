"""CPU oracle (test infrastructure): the four RLFT objectives of the reference,
restated in PyTorch-CPU with the reference's dtype promotion (fp64 advantage x
fp32 ratio), plus autograd gradients of the only trainable module
(`planning_decoder.pi_head`, rift_training.yaml:26-27).

R/ = rift/cbv/planning/fine_tuner/rlft/
"""
from typing import Dict

import torch
import torch.nn.functional as F

from .pluto_ref import SD, mlp_layer

PI_KEYS = ("mlp.0.weight", "mlp.0.bias", "mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias")


def _masked_log_softmax(logits, r_pad):
    """R/rift_pluto/rift_trainer.py:153-154 : fill padded ref lines with -1e8, row log-softmax over G."""
    bs = logits.shape[0]
    return F.log_softmax(logits.masked_fill(r_pad.unsqueeze(-1), -1e8).view(bs, -1), dim=1)


def rift_loss(probability, r_pad, old_logits, advantage, valid_mask):
    """R/rift_pluto/rift_trainer.py:140-182 (dual-clip group-relative objective)."""
    bs = probability.shape[0]
    lp = _masked_log_softmax(probability, r_pad)
    lp_old = _masked_log_softmax(old_logits, r_pad)
    adv = advantage.view(bs, -1)
    ratio = torch.exp(lp - lp_old)
    unclipped = adv * ratio
    clipped = adv * torch.clamp(ratio, 0.8, 1.2)
    mn = torch.min(unclipped, clipped)
    mx = torch.max(mn, adv * 3.0)
    obj = torch.where(adv < 0, mx, mn)
    v = obj[valid_mask.view(bs, -1)]
    if v.numel() == 0:
        return torch.tensor(0.0)
    return -v.mean()


def grpo_loss(probability, r_pad, old_logits, ref_logits, advantage, valid_mask):
    """R/grpo_pluto/grpo_trainer.py:140-194 (clip objective - 0.2 * KL to the frozen ref policy)."""
    bs = probability.shape[0]
    lp = _masked_log_softmax(probability, r_pad)
    lp_old = _masked_log_softmax(old_logits, r_pad)
    ref_p = F.softmax(ref_logits.masked_fill(r_pad.unsqueeze(-1), -1e8).view(bs, -1), dim=1)
    kl = F.kl_div(input=lp, target=ref_p, reduction="none", log_target=False)
    adv = advantage.view(bs, -1)
    ratio = torch.exp(lp - lp_old)
    obj = torch.min(adv * ratio, adv * torch.clamp(ratio, 0.8, 1.2)) - 0.2 * kl
    v = obj[valid_mask.view(bs, -1)]
    if v.numel() == 0:
        return torch.tensor(0.0)
    return -v.mean()


def ppo_actor_loss(probability, r_pad, action_mode, advantage, old_log_prob,
                   clip_epsilon=0.2, lambda_entropy=0.01):
    """Actor part of R/ppo_pluto/ppo_trainer.py:161-183 (value loss is the
    CriticPPO SmoothL1 term, handled separately)."""
    bs, R, M = probability.shape
    lp = _masked_log_softmax(probability, r_pad).view(bs, R, M)
    cur = lp[torch.arange(bs), action_mode[:, 0], action_mode[:, 1]]
    entropy = -torch.sum(torch.exp(lp) * lp, dim=(1, 2))
    ratio = (cur - old_log_prob).exp()
    l1 = advantage * ratio
    l2 = advantage * torch.clamp(ratio, 1.0 - clip_epsilon, 1.0 + clip_epsilon)
    return -(torch.min(l1, l2).mean() + entropy.mean() * lambda_entropy)


def reinforce_loss(probability, r_pad, returns):
    """R/reinforce_pluto/reinforce_trainer.py:125-170; also returns the bit-exact
    integer (r, m) argmax indices (:130-132)."""
    bs, R, M = probability.shape
    p = probability.masked_fill(r_pad.unsqueeze(-1), -1e8)
    idx = torch.argmax(p.view(bs, -1), dim=1)
    r_idx, m_idx = idx // M, idx % M
    lp = F.log_softmax(p.view(bs, -1), dim=1).view(bs, R, M)
    chosen = lp[torch.arange(bs), r_idx, m_idx]
    return -torch.mean(chosen * returns.detach()), r_idx, m_idx


def pi_head_loss_and_grads(sd_flat: Dict[str, torch.Tensor], q_final, kind: str, batch: Dict, r_pad):
    """Run pi_head (planning_decoder.py:184) on `q_final` (bs,R,M,128) with autograd,
    apply the -1e6 model mask (pluto_model.py:203) and the chosen objective; return
    (loss, {param: grad}, logits)."""
    prefix = "planning_decoder.pi_head."
    params = {k: sd_flat[prefix + k].clone().requires_grad_(True) for k in PI_KEYS}
    sd = SD({prefix + k: v for k, v in params.items()}, prefix)
    pi = mlp_layer(q_final, sd).squeeze(-1)
    prob = pi.masked_fill(r_pad.unsqueeze(-1), -1e6)
    if kind == "rift":
        loss = rift_loss(prob, r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                         batch["group_advantage_mask_torch"])
    elif kind == "grpo":
        loss = grpo_loss(prob, r_pad, batch["old_group_logits_torch"], batch["ref_group_logits_torch"],
                         batch["group_advantage_torch"], batch["group_advantage_mask_torch"])
    elif kind == "ppo":
        loss = ppo_actor_loss(prob, r_pad, batch["action_mode_torch"], batch["advantage_torch"],
                              batch["old_log_prob_torch"])
    elif kind == "reinforce":
        loss, _, _ = reinforce_loss(prob, r_pad, batch["return_torch"])
    elif kind == "sft":
        loss, _, _ = sft_loss(prob, r_pad, batch["trajectory_torch"], batch["teacher_infos_torch"])
    else:
        raise ValueError(kind)
    loss.backward()
    return loss.detach(), {k: v.grad.detach().clone() for k, v in params.items()}, prob.detach()


# ---- SFT teacher objective (rift/cbv/planning/fine_tuner/sft/sft_trainer.py:123-199) -------------------------------------------------
def sft_global_to_local(candidate_trajectories, origin, heading, step_interval=10):
    """sft/utils.py:10-32."""
    bs, R, M, T, _ = candidate_trajectories.shape
    if T < step_interval:
        local_traj = candidate_trajectories[:, :, :, -1:, :2]
    else:
        local_traj = candidate_trajectories[:, :, :, step_interval - 1::step_interval, :2]
    origin = origin.view(bs, 1, 1, 1, 2)
    rot_mat = torch.stack([torch.stack([torch.cos(heading), -torch.sin(heading)], dim=1),
                           torch.stack([torch.sin(heading), torch.cos(heading)], dim=1)], dim=1).view(bs, 1, 1, 1, 2, 2)
    return torch.einsum('brmtc,brmtcd->brmtd', local_traj - origin, rot_mat)


def sft_target_speed(local_traj):
    """PIDController.batch_control_pid, the target-speed half (pluto/controller/pid_controller.py:108-125)."""
    if local_traj.shape[3] == 1:
        return local_traj[..., 0, :].norm(dim=-1, p=2)
    diff = local_traj[..., 1:, :] - local_traj[..., :-1, :]
    return diff.norm(dim=-1, p=2).mean(dim=-1)


def sft_teacher_mode(candidate_trajectories, teacher_infos, frame_rate=10):
    """generate_target_label :186-196: (r, m) of the candidate closest in target speed to the teacher (padded lines included)."""
    bs, R, M = candidate_trajectories.shape[:3]
    local = sft_global_to_local(candidate_trajectories, teacher_infos[:, 1:3], teacher_infos[:, 3], frame_rate)
    speed = sft_target_speed(local)
    idx = torch.argmin((speed - teacher_infos[:, 0][:, None, None]).abs().view(bs, -1), dim=1)
    return idx // M, idx % M


def sft_loss(probability, r_pad, candidate_trajectories, teacher_infos, frame_rate=10):
    """_compute_objectives + get_teacher_loss (:123-184): mask -1e8, argmax -> best_r, one-hot label at (best_r, teacher m), F.cross_entropy
    with a float one-hot target (mean over the batch).  Returns (loss, best_r, teacher_m)."""
    bs, R, M = probability.shape
    prob = probability.masked_fill(r_pad.unsqueeze(-1), -1e8)
    max_idx = torch.argmax(prob.reshape(bs, -1), dim=1)
    best_r = max_idx // M
    _, m_idx = sft_teacher_mode(candidate_trajectories, teacher_infos, frame_rate)
    target = torch.zeros_like(prob)
    target[torch.arange(bs), best_r, m_idx] = 1
    loss = torch.nn.functional.cross_entropy(prob.reshape(bs, -1), target.reshape(bs, -1).detach())
    return loss, best_r, m_idx
