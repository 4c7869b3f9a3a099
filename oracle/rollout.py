"""CPU oracle (test infrastructure): candidate closed-loop rollout of the GRPO advantage pipeline --
reference-line deviation, PID + kinematic-bicycle propagation, Savitzky-Golay kinematics and box corners --
restated with the same PyTorch-CPU fp32 operations as the reference, so that it matches it bit for bit.

T/ = rift/cbv/planning/fine_tuner/rlft/traj_eval/
"""
import math
from typing import List

import numpy as np
import torch
import torch.nn.functional as F


def ref_line_info(trajectories, ref_line_pos: List[torch.Tensor], ref_line_angle: List[torch.Tensor]):
    """T/traj_evaluator.py:372-420.  trajectories (R, M, Ts, C>=4); per-line ragged reference points.
    Returns delta_dis (G, Ts) f32, delta_angle (G, Ts) f32, closest_idx (G, Ts) int64."""
    R, M, Ts, _ = trajectories.shape
    dd = torch.zeros((R, M, Ts), dtype=trajectories.dtype)
    da = torch.zeros((R, M, Ts), dtype=trajectories.dtype)
    ci = torch.zeros((R, M, Ts), dtype=torch.long)
    for r in range(R):
        ref_pos, ref_angle = ref_line_pos[r], ref_line_angle[r]
        traj = trajectories[r]
        cand_pos = traj[..., :2]
        cand_angle = torch.atan2(traj[..., 3], traj[..., 2])
        diff = cand_pos.unsqueeze(2) - ref_pos.unsqueeze(0).unsqueeze(0)
        dist = torch.norm(diff, dim=-1)
        closest_idx = torch.argmin(dist, dim=-1)
        closest_angle = ref_angle[closest_idx]
        angle_diff = cand_angle - closest_angle
        da[r] = torch.atan2(torch.sin(angle_diff), torch.cos(angle_diff))
        rel = cand_pos - ref_pos[closest_idx]
        tang = torch.stack([torch.cos(closest_angle), torch.sin(closest_angle)], dim=-1)
        dd[r] = -(rel[..., 0] * tang[..., 1] - rel[..., 1] * tang[..., 0])
        ci[r] = closest_idx
    return dd.view(-1, Ts), da.view(-1, Ts), ci.view(-1, Ts)


def to_global(trajectories, center_pos, center_heading):
    """T/traj_evaluator.py:115-153: candidate (x, y, heading) -> global reference trajectory of the rollout."""
    heading = torch.atan2(trajectories[..., 3], trajectories[..., 2])
    out = torch.cat([trajectories[..., :2], heading[..., None]], dim=-1)
    R, M, T, C = out.shape
    out = out.reshape(-1, T, C).clone()
    # Reference quirk (traj_evaluator.py:137-139): `first_points` is a VIEW of `out_trajectory`, and the in-place
    # `out_trajectory[:, :, :2] -= first_points.unsqueeze(1)` therefore zeroes the first point and then subtracts
    # that zero from every later point (PyTorch-CPU elementwise order) -- only point 0 is moved to the origin.
    out[:, 0, :2] = 0.0
    cos_h, sin_h = torch.cos(center_heading), torch.sin(center_heading)
    rot = torch.stack((torch.stack([cos_h, sin_h], dim=-1), torch.stack([-sin_h, cos_h], dim=-1)), dim=-2)
    pos = torch.matmul(out[..., :2], rot) + center_pos
    return pos, out[..., 2] + center_heading


class PID:
    """T/track_propogate.py:318-400 BatchPIDTorch (state persists across calls)."""

    def __init__(self, kp, ki, kd, n=20):
        self.kp, self.ki, self.kd, self.n = kp, ki, kd, n
        self.buf = None

    def ensure(self, B):
        if self.buf is None:
            self.buf = torch.zeros(B, self.n)
            self.ptr = torch.zeros(B, dtype=torch.long)
            self.len = torch.zeros(B, dtype=torch.long)
        elif self.buf.shape[0] < B:
            e = B - self.buf.shape[0]
            self.buf = torch.cat([self.buf, torch.zeros(e, self.n)])
            self.ptr = torch.cat([self.ptr, torch.zeros(e, dtype=torch.long)])
            self.len = torch.cat([self.len, torch.zeros(e, dtype=torch.long)])

    def step(self, error):
        B = error.shape[0]
        self.ensure(B)
        ar = torch.arange(B)
        idx = self.ptr[:B]
        prev = self.buf[ar, idx]
        self.buf[ar, idx] = error
        self.ptr[:B] = (idx + 1) % self.n
        self.len[:B] = torch.clamp(self.len[:B] + 1, max=self.n)
        integral = self.buf[:B].sum(dim=1) / self.len[:B].clamp(min=1).to(torch.float32)
        return self.kp * error + self.ki * integral + self.kd * (error - prev)


class Rollout:
    """TrackPropagate (T/track_propogate.py:599-780) with its BatchPIDController (:403-491) and
    BatchKinematicBicycleModel.forward (:214-279)."""
    LF, LR, GAIN = -0.090769015, 1.4178275, 0.36848336
    BRAKE = [9.31711370e-03, 8.20967431e-02, -2.83832427e-03, 5.06587474e-05, -4.90357228e-07, 2.44419284e-09,
             -4.91381935e-12]
    THROTTLE = [9.63873001e-01, 4.37535692e-04, -3.80192912e-01, 1.74950069e+00, 9.16787414e-02, -7.05461530e-02,
                -1.05996152e-03, 6.71079346e-04]

    def __init__(self, dt=0.1, rollout_length=80):
        self.dt, self.rollout_length = dt, rollout_length
        self.turn, self.speed = PID(1.25, 0.75, 0.3), PID(5.0, 0.5, 1.0)
        f = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
        self.t_dt, self.Lf, self.Lr, self.gain = f(dt), f(self.LF), f(self.LR), f(self.GAIN)
        self.brake_values, self.throttle_values, self.thr = f(self.BRAKE), f(self.THROTTLE), f(0.3)

    def control_pid(self, local_pos, speed, trace):
        B = local_pos.shape[0]
        rs = local_pos[:, 9::10]
        seg = rs[:, 1:] - rs[:, :-1]
        desired_v = seg.norm(dim=2).mean(dim=1)
        aim_dist = torch.clamp(0.5 * speed + 2.5, min=5.0, max=8.0)
        norms = rs[:, :-1].norm(dim=2)
        idx = (norms - aim_dist[:, None]).abs().argmin(dim=1)
        trace.append(idx)
        aim = rs[torch.arange(B), idx]
        brake = (desired_v < 0.4) | ((speed / desired_v.clamp(min=1e-4)) > 1.1)
        delta = torch.clamp(desired_v - speed, min=0.0, max=0.25)
        throttle = torch.clamp(self.speed.step(delta), min=0.0, max=0.75) * (~brake)
        angle = torch.rad2deg(torch.atan2(aim[:, 1], aim[:, 0])) / 90.0
        angle = torch.where((speed < 0.01) | brake, torch.zeros_like(angle), angle)
        steer = torch.clamp(self.turn.step(angle), min=-1.0, max=1.0)
        return torch.stack([throttle, steer, brake], dim=1)

    def bicycle(self, loc, head, speeds, actions):
        throttle, steer, brake = actions.unbind(dim=1)
        brake_bool = brake.round().bool()
        wheel = self.gain * steer
        slip = torch.atan((self.Lr / (self.Lf + self.Lr)) * torch.tan(wheel))
        dx = speeds * torch.cos(head + slip) * self.t_dt
        dy = speeds * torch.sin(head + slip) * self.t_dt
        nh = head + (speeds / self.Lr) * torch.sin(slip) * self.t_dt
        nl = loc.clone()
        nl[:, 0] += dx
        nl[:, 1] += dy
        kph = speeds * 3.6
        vp = torch.stack([kph.pow(i) for i in range(1, 8)], dim=1)
        nb = vp @ self.brake_values
        v, t = kph, throttle
        v2, t2 = v * v, t * t
        feats = torch.stack([v, v2, t, t2, v * t, v * t2, v2 * t, v2 * t2], dim=1)
        nt = feats @ self.throttle_values
        tmask = (~brake_bool) & (throttle >= self.thr)
        nk = torch.where(brake_bool, nb, kph)
        nk = torch.where(tmask, nt, nk)
        return nl, nh, torch.clamp(nk / 3.6, min=0.0)

    def local_traj(self, ref, closest, pos, heading, future_len=30):
        G, T, _ = ref.shape
        idx = closest.unsqueeze(1) + torch.arange(future_len).unsqueeze(0)
        idc = idx.clamp(max=T - 1)
        g = ref[torch.arange(G).unsqueeze(1).expand(-1, future_len), idc]
        pad = (idx >= T).unsqueeze(-1).expand(-1, -1, 2)
        g = torch.where(pad, ref[:, -1:, :].expand(-1, future_len, -1), g)
        loc = g - pos.unsqueeze(1)
        c, s = torch.cos(heading), torch.sin(heading)
        rot = torch.stack((torch.stack((c, -s), dim=-1), torch.stack((s, c), dim=-1)), dim=-2)
        return torch.matmul(loc, rot)

    @torch.no_grad()
    def propagate(self, ref_pos, ref_heading, init_speed, width, length):
        """Returns dict of (G, 80, ...) tensors + integer traces closest_index (G, 79), aim_idx (G, 79)."""
        G = ref_pos.shape[0]
        centers, angles = [ref_pos[:, 0, :]], [ref_heading[:, 0]]
        speeds = [torch.full((G,), float(init_speed), dtype=ref_pos.dtype)]
        closest = torch.zeros(G, dtype=torch.long)
        ci_trace, aim_trace = [], []
        for _ in range(self.rollout_length - 1):
            local = self.local_traj(ref_pos, closest, centers[-1], angles[-1])
            act = self.control_pid(local, speeds[-1], aim_trace)
            nl, nh, ns = self.bicycle(centers[-1], angles[-1], speeds[-1], act)
            centers.append(nl); angles.append(nh); speeds.append(ns)
            diff = ref_pos - nl[:, None, :]
            closest = torch.argmin((diff ** 2).sum(dim=-1), dim=1)
            ci_trace.append(closest)
        center, angle, speed = torch.stack(centers, 1), torch.stack(angles, 1), torch.stack(speeds, 1)
        sp, acc, yr, ya = derive_kinematics(angle, speed, self.dt)
        shape = torch.tensor([width, length], dtype=ref_pos.dtype).expand(G, -1)
        return {"center": center, "angle": angle, "speed": sp, "acc": acc, "ang_vel": yr, "ang_acc": ya,
                "vertices": vertices(center, angle, shape),
                "closest_index": torch.stack(ci_trace, 1), "aim_idx": torch.stack(aim_trace, 1)}


def sg_kernel(window=5, order=2):
    """T/track_propogate.py:128-141 (computed exactly as the reference does, incl. its fp32 pinv)."""
    h = window // 2
    t = torch.arange(-h, h + 1, dtype=torch.float32)
    A = torch.stack([t ** i for i in range(order + 1)], dim=1)
    pinv = torch.linalg.pinv(A.T @ A) @ A.T
    return pinv[0].flip(0).view(1, 1, -1)


def sg_smooth(x):
    k = sg_kernel()
    return F.conv1d(F.pad(x.unsqueeze(1), (2, 2), mode="reflect"), k).squeeze(1)


def central_diff(x, dt):
    mid = (x[:, 2:] - x[:, :-2]) / (2.0 * dt)
    return torch.cat([(x[:, 1:2] - x[:, :1]) / dt, mid, (x[:, -1:] - x[:, -2:-1]) / dt], dim=1)


def wrap(d):
    return torch.atan2(torch.sin(d), torch.cos(d))


def derive_kinematics(headings, speed, dt=0.1):
    """T/track_propogate.py:500-596 with speed given, window 5 / order 2."""
    sp = sg_smooth(speed)
    acc = central_diff(sp.unsqueeze(-1), dt).squeeze(-1)
    hp = sg_smooth(headings)
    yr = torch.zeros_like(headings)
    yr[:, 1:-1] = wrap(hp[:, 2:] - hp[:, :-2]) / (2 * dt)
    yr[:, 0] = wrap(hp[:, 1] - hp[:, 0]) / dt
    yr[:, -1] = wrap(hp[:, -1] - hp[:, -2]) / dt
    ya = central_diff(yr.unsqueeze(-1), dt).squeeze(-1)
    return sp, acc, yr, ya


def vertices(center, heading, shape):
    """T/track_propogate.py:16-74: corners FL, RL, RR, FR."""
    shape = shape.unsqueeze(-2).expand(*center.shape[:-2], center.shape[-2], 2)
    hw, hl = 0.5 * shape[..., 0], 0.5 * shape[..., 1]
    dx = torch.stack((hl, -hl, -hl, hl), dim=-1)
    dy = torch.stack((hw, hw, -hw, -hw), dim=-1)
    c, s = torch.cos(heading).unsqueeze(-1), torch.sin(heading).unsqueeze(-1)
    return torch.stack((dx * c - dy * s, dx * s + dy * c), dim=-1) + center.unsqueeze(-2)
