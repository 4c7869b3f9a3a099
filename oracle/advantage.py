"""CPU oracle (test infrastructure): advantage / return computations of the RLFT
update, restated with the reference's sequential loops and dtype promotion.

R/ = rift/cbv/planning/fine_tuner/rlft/
"""
import math

import numpy as np
import torch


def get_advantages_gae(rewards, undones, values, next_values, unterminated, gamma=0.98, lambda_gae_adv=0.98):
    """R/ppo_pluto/ppo_datamodule.py:22-37.  `rewards` is float64 (np.stack of python floats),
    the rest float32; 0-dim tensor arithmetic promotes the recurrence to float64 and the
    store into the float32 `advantages` rounds once per step."""
    advantages = torch.empty_like(values)
    advantage = torch.zeros_like(values[0])
    for t in range(rewards.shape[0] - 1, -1, -1):
        delta = rewards[t] + unterminated[t] * gamma * next_values[t] - values[t]
        advantages[t] = advantage = delta + undones[t] * gamma * lambda_gae_adv * advantage
    return advantages


def normalize_advantage(adv):
    """R/ppo_pluto/ppo_datamodule.py:166 (torch.std is unbiased)."""
    return (adv - adv.mean()) / (adv.std(dim=0) + 1e-5)


def compute_return(rewards, dones, gamma=0.98):
    """R/reinforce_pluto/reinforce_datamodule.py:19-38 (the trailing normalisation line acts on a
    scalar and is discarded -> returns are raw)."""
    returns = torch.zeros_like(rewards)
    episode_return = 0
    for t in range(len(rewards) - 1, -1, -1):
        if dones[t] == 1:
            episode_return = rewards[t]
        else:
            episode_return = rewards[t] + gamma * episode_return
        returns[t] = episode_return
    return returns


def group_zscore(returns):
    """R/traj_eval/traj_evaluator.py:467-470: np.std (ddof 0) + 1e-5, float64."""
    returns = np.asarray(returns, dtype=np.float64)
    return (returns - np.mean(returns)) / (np.std(returns) + 1e-5)


# DenseRewardModel (rift/gym_carla/reward/reward_model.py:22-48)
P = dict(alpha_collision=20.0, alpha_boundary=5.0, alpha_comfort=0.8, alpha_l_align=0.5, alpha_vel_align=0.05,
         alpha_l_center=0.6, alpha_center_bias=0.0, alpha_velocity=0.1, alpha_timestep=0.1)


def dense_reward(delta_dis, delta_angle, speed, acc, angular_speed, angular_acc, collision, offroad):
    """reward_model.py:34-50 with the scalar dtype promotion of the reference environment
    (numpy==1.24.4, requirements.txt:19): np.float32 (op) python float -> float64,
    np.float32 (op) np.float32 -> float32.  Inputs are np.float32 scalars; made explicit here so
    that the oracle does not depend on the installed numpy's promotion rules."""
    f32, f64 = np.float32, np.float64
    delta_dis, delta_angle, speed, acc, angular_acc = map(f32, (delta_dis, delta_angle, speed, acc, angular_acc))
    r_collision = -(f64(P["alpha_collision"]) + f64(abs(speed))) * f64(collision)
    r_offroad = -P["alpha_boundary"] * offroad
    r_comfort = -P["alpha_comfort"] * (int(abs(acc) > 4) + int(abs(angular_acc) > 4))
    c = np.cos(delta_angle)                                  # float32
    cs = f32(c * speed)                                      # float32 * float32
    r_l_align = P["alpha_l_align"] * (f64(min(c, f32(0))) + P["alpha_vel_align"] * f64(min(cs, f32(0)))
                                      + 0.25 * (1 - f64(abs(delta_angle)) / (np.pi / 2)))
    dd = abs(f64(delta_dis) - P["alpha_center_bias"])
    r_l_center = -P["alpha_l_center"] * int(c > 0.5) * (dd - 0.05 / np.exp(dd - 0.5))
    r_velocity = P["alpha_velocity"] * f64(max(c, f32(0))) * int(3 < abs(speed) < 20) * f64(abs(speed))
    r_timestep = -P["alpha_timestep"] * int(abs(speed) > 0 or abs(acc) > 0)
    return r_collision + r_offroad + r_comfort + r_l_align + r_l_center + r_velocity + r_timestep


def rollout_return(delta_dis, delta_angle, speed, acc, ang_vel, ang_acc, collision, off_road, gamma=0.98):
    """R/traj_eval/traj_evaluator.py:333-370."""
    G, Ts = delta_angle.shape
    out = np.zeros((G,), dtype=np.float64)
    for i in range(G):
        for j in range(Ts):
            col, off = collision[i, j], off_road[i, j]
            out[i] += dense_reward(abs(delta_dis[i, j]), abs(delta_angle[i, j]), speed[i, j], acc[i, j],
                                   ang_vel[i, j], ang_acc[i, j], int(col), int(off)) * gamma ** j
            if col:
                break
    return out


def warmup_cos_lr(epoch, lr, min_lr, warmup_epochs, epochs):
    """pluto/optim/warmup_cos_lr.py:39-54 (per-epoch schedule)."""
    if epoch < warmup_epochs:
        return lr * (epoch + 1) / warmup_epochs
    return min_lr + 0.5 * (lr - min_lr) * (1 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))
