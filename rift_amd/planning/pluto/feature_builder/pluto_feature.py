"""PlutoFeature -- host mirror of pluto/feature_builder/pluto_feature.py:18-150 (container + collate +
tensor / device conversion; the CARLA -> feature builder itself is out of scope)."""
from dataclasses import dataclass
from typing import Any, Dict, List

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence


def to_tensor(data):   # pluto/utils/utils.py:12-30
    if isinstance(data, dict):
        return {k: to_tensor(v) for k, v in data.items()}
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data).float() if data.dtype == np.float64 else torch.from_numpy(data)
    if isinstance(data, np.number):
        return torch.tensor(data).float()
    if isinstance(data, (list, torch.Tensor)):
        return data
    if isinstance(data, int):
        return torch.tensor(data)
    if isinstance(data, tuple):
        return to_tensor(data[0])
    raise NotImplementedError(type(data))


def to_device(data, device):   # pluto/utils/utils.py:56-62
    if isinstance(data, dict):
        return {k: to_device(v, device) for k, v in data.items()}
    if isinstance(data, torch.Tensor):
        return data.to(device)
    raise NotImplementedError(type(data))


@dataclass
class PlutoFeature:
    data: Dict[str, Any]
    data_p: Dict[str, Any] = None
    data_n: Dict[str, Any] = None
    data_n_info: Dict[str, Any] = None

    @classmethod
    def collate(cls, feature_list: List['PlutoFeature']) -> 'PlutoFeature':
        """Zero-pad dim 0 of every tensor of the pad groups to the batch maximum, stack the rest (:25-96).
        (The contrastive data_p / data_n variants of the SFT trainers are not part of the RLFT path.)"""
        if feature_list[0].data_p is not None or feature_list[0].data_n is not None:
            raise NotImplementedError("contrastive (data_p / data_n) collation belongs to the SFT trainers")
        batch = {}
        pad_keys, stack_keys = ["agent", "map"], ["current_state", "origin", "angle"]
        first = feature_list[0].data
        if "reference_line" in first:
            pad_keys.append("reference_line")
        if "static_objects" in first:
            pad_keys.append("static_objects")
        if "cost_maps" in first:
            stack_keys.append("cost_maps")
        for key in pad_keys:
            batch[key] = {k: pad_sequence([f.data[key][k] for f in feature_list], batch_first=True) for k in first[key].keys()}
        for key in stack_keys:
            batch[key] = torch.stack([f.data[key] for f in feature_list], dim=0)
        return PlutoFeature(data=batch)

    def to_feature_tensor(self) -> 'PlutoFeature':
        return PlutoFeature(data={k: to_tensor(v) for k, v in self.data.items()})

    def to_device(self, device) -> 'PlutoFeature':
        return PlutoFeature(data={k: to_device(v, device) for k, v in self.data.items()})

    def serialize(self) -> Dict[str, Any]:
        return {"data": self.data}

    @classmethod
    def deserialize(cls, data: Dict[str, Any]) -> 'PlutoFeature':
        return PlutoFeature(data=data["data"])
