"""Minimal TrajectorySampling (nuplan_plugin/trajectory/trajectory_sampling.py:13-60): the model
only constructs it with all three values (pluto_model.py:19), so only that case is kept."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class TrajectorySampling:
    num_poses: Optional[int] = None
    time_horizon: Optional[float] = None
    interval_length: Optional[float] = None

    def __post_init__(self) -> None:
        if self.num_poses is not None and not isinstance(self.num_poses, int):
            raise ValueError(f"num_poses was defined but it is not int. Instead {type(self.num_poses)}!")
        if self.time_horizon is not None:
            self.time_horizon = float(self.time_horizon)
        if self.interval_length is not None:
            self.interval_length = float(self.interval_length)
        if self.num_poses and self.time_horizon and not self.interval_length:
            self.interval_length = self.time_horizon / self.num_poses
        elif self.num_poses and self.interval_length and not self.time_horizon:
            self.time_horizon = self.num_poses * self.interval_length
        elif self.time_horizon and self.interval_length and not self.num_poses:
            self.num_poses = int(round(self.time_horizon / self.interval_length))
