"""Waypoint-following PID controller -- host mirror of pluto/controller/pid_controller.py:13-100 (per-CBV, per-tick, scalar work:
stays on the host; the batched closed-loop version used by the advantage rollout is the HIP kernel in csrc/rollout.h)."""
from collections import deque

import numpy as np


class PID:
    """Error window of n samples: P on the error, I on the window mean, D on the last difference (pid_controller.py:13-36)."""

    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        self._K_P, self._K_I, self._K_D = K_P, K_I, K_D
        self._window = deque([0 for _ in range(n)], maxlen=n)
        self._max = 0.0
        self._min = 0.0

    def step(self, error):
        self._window.append(error)
        self._max = max(self._max, abs(error))
        self._min = -abs(self._max)
        integral = np.mean(self._window) if len(self._window) >= 2 else 0.0
        derivative = (self._window[-1] - self._window[-2]) if len(self._window) >= 2 else 0.0
        return self._K_P * error + self._K_I * integral + self._K_D * derivative


class PIDController:
    def __init__(self, sample_interval=10, max_throttle=1.0, brake_speed=0.4, brake_ratio=1.1, clip_delta=1.0):
        self.sample_interval = int(sample_interval)
        self.turn_controller = PID(K_P=1.25, K_I=0.75, K_D=0.3, n=20)
        self.speed_controller = PID(K_P=5.0, K_I=0.5, K_D=1.0, n=20)
        self.alpha, self.beta = 0.5, 2.5
        self.min_aim_dis, self.max_aim_dis = 5.0, 8.0
        self.max_throttle, self.brake_speed, self.brake_ratio, self.clip_delta = max_throttle, brake_speed, brake_ratio, clip_delta
        self.desired_speed = None
        self.delta_angle = None

    def control_pid(self, local_pos: np.ndarray, speed: float):
        """(throttle, steer, brake) for planned local waypoints (T, 2) and the current speed (pid_controller.py:56-100)."""
        k = self.sample_interval
        pts = local_pos[k - 1::k] if local_pos.shape[0] >= k else local_pos[-1:]
        desired_speed = np.linalg.norm(np.diff(pts, axis=0), axis=1).mean()
        aim_dist = np.clip(self.alpha * speed + self.beta, self.min_aim_dis, self.max_aim_dis)
        aim = pts[np.abs(np.linalg.norm(pts[:-1], axis=1) - aim_dist).argmin()]
        brake = desired_speed < self.brake_speed or (speed / desired_speed) > self.brake_ratio
        throttle = np.clip(self.speed_controller.step(np.clip(desired_speed - speed, 0.0, self.clip_delta)), 0.0, self.max_throttle)
        throttle = throttle if not brake else 0.0
        angle = np.degrees(-np.arctan2(aim[1], aim[0])) / 90
        if speed < 0.01 or brake:       # no integral wind-up while standing or braking
            angle = 0.0
        steer = np.clip(self.turn_controller.step(angle), -1.0, 1.0)
        self.desired_speed, self.delta_angle = desired_speed, angle
        return throttle, steer, brake
