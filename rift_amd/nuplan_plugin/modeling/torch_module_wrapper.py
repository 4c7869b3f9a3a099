"""TorchModuleWrapper (nuplan_plugin/modeling/torch_module_wrapper.py:11-47): the base class the
policy model is typed against."""
import abc
from typing import Any, List

import torch

from rift_amd.nuplan_plugin.modeling.types import FeaturesType, TargetsType
from rift_amd.nuplan_plugin.trajectory.trajectory_sampling import TrajectorySampling


class TorchModuleWrapper(torch.nn.Module):
    def __init__(self, future_trajectory_sampling: TrajectorySampling, feature_builders: List[Any],
                 target_builders: List[Any]):
        super().__init__()
        self.future_trajectory_sampling = future_trajectory_sampling
        self.feature_builders = feature_builders
        self.target_builders = target_builders

    def get_list_of_required_feature(self) -> List[Any]:
        return self.feature_builders

    def get_list_of_computed_target(self) -> List[Any]:
        return self.target_builders

    @abc.abstractmethod
    def forward(self, features: FeaturesType) -> TargetsType:
        pass
