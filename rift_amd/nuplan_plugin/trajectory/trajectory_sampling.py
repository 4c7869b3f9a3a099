"""TrajectorySampling as PlanningModel uses it (nuplan_plugin/trajectory/trajectory_sampling.py:13-60; pluto_model.py:19 constructs it with
all three values and TorchModuleWrapper only stores it)."""
from dataclasses import dataclass


@dataclass
class TrajectorySampling:
    num_poses: int
    time_horizon: float
    interval_length: float
