"""Waypoint follower for one CBV: planned local path + current speed -> (throttle, steer, brake).

Behaviour of the reference's rollout-side controller (rift/cbv/planning/pluto/controller/pid_controller.py:13-100), pinned by
tests/golden/inference.npz; stays on the host because it is scalar per-CBV per-tick work.  (The batched closed-loop controller of
the advantage rollout is the HIP kernel in csrc/rollout.h.)

The control law, stated once:
  * the path is thinned to every `sample_interval`-th waypoint; the mean spacing of the thinned path is the target speed;
  * aim point = the thinned waypoint (last one excluded) whose range is closest to clamp(0.5 v + 2.5, 5, 8) metres;
  * brake when the target speed is below `brake_speed` or the vehicle is more than `brake_ratio` times too fast;
  * throttle = windowed PI-D on clamp(target - v, 0, clip_delta), limited to [0, max_throttle], zero while braking;
  * steer = windowed PI-D on the aim bearing in quarter turns (negated: CARLA's y axis), held at zero error while standing or braking.
"""
import numpy as np


class WindowedPID:
    """u = Kp e + Ki mean(last n errors) + Kd (e - previous e); the window starts as n zeros, so the mean is always over n samples."""

    def __init__(self, kp: float, ki: float, kd: float, n: int = 20):
        self.kp, self.ki, self.kd = kp, ki, kd
        self._ring = [0.0] * n
        self._head = 0               # slot of the OLDEST sample = next slot to overwrite
        self._last = 0.0

    def step(self, error: float) -> float:
        n = len(self._ring)
        self._ring[self._head] = error
        self._head = (self._head + 1) % n
        chronological = self._ring[self._head:] + self._ring[:self._head]
        out = self.kp * error + self.ki * np.mean(chronological) + self.kd * (error - self._last)
        self._last = error
        return out


PID = WindowedPID      # the reference's class name


class PIDController:
    TURN_GAINS = (1.25, 0.75, 0.3)
    SPEED_GAINS = (5.0, 0.5, 1.0)
    AIM_SLOPE, AIM_OFFSET, AIM_MIN, AIM_MAX = 0.5, 2.5, 5.0, 8.0

    def __init__(self, sample_interval=10, max_throttle=1.0, brake_speed=0.4, brake_ratio=1.1, clip_delta=1.0):
        self.sample_interval = int(sample_interval)
        self.max_throttle, self.brake_speed, self.brake_ratio, self.clip_delta = max_throttle, brake_speed, brake_ratio, clip_delta
        self.turn_controller = WindowedPID(*self.TURN_GAINS)
        self.speed_controller = WindowedPID(*self.SPEED_GAINS)
        self.desired_speed = None      # diagnostics of the last call, as the reference exposes them
        self.delta_angle = None

    def _thin(self, path: np.ndarray) -> np.ndarray:
        k = self.sample_interval
        return path[k - 1::k] if len(path) >= k else path[-1:]

    def control_pid(self, local_pos: np.ndarray, speed: float):
        pts = self._thin(local_pos)
        seg = np.diff(pts, axis=0)
        target_speed = np.sqrt((seg * seg).sum(axis=1)).mean()
        aim_range = min(max(self.AIM_SLOPE * speed + self.AIM_OFFSET, self.AIM_MIN), self.AIM_MAX)
        ranges = np.sqrt((pts[:-1] * pts[:-1]).sum(axis=1))
        aim_x, aim_y = pts[np.argmin(np.abs(ranges - aim_range))]
        brake = bool(target_speed < self.brake_speed or speed / target_speed > self.brake_ratio)

        gas = self.speed_controller.step(min(max(target_speed - speed, 0.0), self.clip_delta))
        throttle = 0.0 if brake else min(max(gas, 0.0), self.max_throttle)

        bearing = 0.0 if (brake or speed < 0.01) else np.degrees(-np.arctan2(aim_y, aim_x)) / 90
        steer = min(max(self.turn_controller.step(bearing), -1.0), 1.0)
        self.desired_speed, self.delta_angle = target_speed, bearing
        return throttle, steer, brake
