"""Feature / target type aliases (nuplan_plugin/modeling/types.py:14-29)."""
from typing import Any, Dict, List

import torch


class MissingFeature(Exception):
    """Raised when a feature is not present."""


FeaturesType = Dict[str, Any]
TargetsType = Dict[str, Any]
ScenarioListType = List[Any]
TensorFeaturesType = Dict[str, torch.Tensor]


def move_features_type_to_device(batch: FeaturesType, device: torch.device) -> FeaturesType:
    return {key: value.to_device(device) for key, value in batch.items()}
