"""PlutoFeature -- host mirror of pluto/feature_builder/pluto_feature.py:18-263: container, collate, tensor / numpy / device conversion
and `normalize` (global frame -> CBV frame, map crop), i.e. everything between the CARLA-bound feature builder and the model.  The
builder itself (actors / HD map -> raw arrays) is out of scope, SURVEY.md section 2 row 7."""
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


def to_numpy(data):    # pluto/utils/utils.py:33-53
    if isinstance(data, dict):
        return {k: to_numpy(v) for k, v in data.items()}
    if isinstance(data, torch.Tensor):
        return data.detach().cpu().numpy()
    return data


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
        def pad(ts):      # pad_sequence(batch_first=True) -- or, when nothing needs padding (a single CBV; equal counts), the stack it equals
            n0 = ts[0].shape[0]
            for t in ts:
                if t.shape[0] != n0:
                    return pad_sequence(ts, batch_first=True)
            return torch.stack(ts, dim=0)                    # (a third of pad_sequence's host time: 31 calls per rollout tick)
        for key in pad_keys:
            batch[key] = {k: pad([f.data[key][k] for f in feature_list]) for k in first[key].keys()}
        for key in stack_keys:
            batch[key] = torch.stack([f.data[key] for f in feature_list], dim=0)
        return PlutoFeature(data=batch)

    def to_feature_tensor(self) -> 'PlutoFeature':
        return PlutoFeature(data={k: to_tensor(v) for k, v in self.data.items()})

    def to_numpy(self) -> 'PlutoFeature':
        return PlutoFeature(data={k: to_numpy(v) for k, v in self.data.items()})

    @property
    def is_valid(self) -> bool:
        """pluto_feature.py:159-164: a scene with reference lines needs one valid point; otherwise it needs map polygons."""
        if "reference_line" in self.data:
            return bool(self.data["reference_line"]["valid_mask"].any())
        return self.data["map"]["point_position"].shape[0] > 0

    # how every geometric array moves into the CBV frame (x' = (x - c) R, v' = v R, a' = a - theta): (group, key, kind); kind "pose" =
    # (x, y, angle) in the last axis
    _FRAME_TABLE = (
        ("agent", "position", "point"), ("agent", "velocity", "vector"), ("agent", "heading", "angle"),
        ("map", "point_position", "point"), ("map", "point_vector", "vector"), ("map", "point_orientation", "angle"),
        ("map", "polygon_center", "pose"), ("map", "polygon_position", "point"), ("map", "polygon_orientation", "angle"),
        ("static_objects", "position", "point"), ("static_objects", "heading", "angle"), ("route", "position", "point"),
        ("reference_line", "position", "point"), ("reference_line", "vector", "vector"), ("reference_line", "orientation", "angle"),
    )

    @classmethod
    def normalize(cls, data, first_time=False, radius=None, hist_steps=21) -> 'PlutoFeature':
        """Move a raw (global-frame, numpy float64) feature dict into the frame of its CBV (pluto_feature.py:166-263): translate by the
        CBV position, rotate by its heading, subtract the heading from every angle; future-motion targets relative to the last history
        step; on the first call also crop the map to +-radius around the CBV (per point of the centre line; polygons without a point left
        are removed) and remember the frame as `origin` / `angle`.  Mutates and returns `data`, as the reference does."""
        state = data["current_state"]
        centre, theta = state[:2].copy(), state[2].copy()
        c, s = np.cos(theta), np.sin(theta)
        rot = np.array([[c, -s], [s, c]], dtype=np.float64)
        state[:3] = 0

        def move(arr, kind):
            if kind == "point":
                return np.matmul(arr - centre, rot)
            if kind == "vector":
                return np.matmul(arr, rot)
            if kind == "angle":
                arr -= theta
                return arr
            arr[..., :2] = np.matmul(arr[..., :2] - centre, rot)            # pose: in place, like the reference
            arr[..., 2] -= theta
            return arr

        for group, key, kind in cls._FRAME_TABLE:
            if group in data and key in data[group]:
                data[group][key] = move(data[group][key], kind)
        if "causal" in data and len(data["causal"]["free_path_points"]) > 0:
            data["causal"]["free_path_points"] = move(data["causal"]["free_path_points"], "pose")

        ag = data["agent"]
        last = hist_steps - 1
        target = np.concatenate([ag["position"][:, hist_steps:] - ag["position"][:, last][:, None],
                                 (ag["heading"][:, hist_steps:] - ag["heading"][:, last][:, None])[..., None]], -1)
        target[~ag["valid_mask"][:, hist_steps:]] = 0
        ag["target"] = target

        if first_time:
            xy = data["map"]["point_position"][:, 0]                         # centre line of every polygon: (M, P, 2)
            inside = (np.abs(xy[..., 0]) < radius) & (np.abs(xy[..., 1]) < radius)
            keep = inside.any(-1)
            data["map"]["valid_mask"] = inside
            data["map"] = {k: v[keep] for k, v in data["map"].items()}
            if "causal" in data:
                data["causal"]["ego_care_red_light_mask"] = data["causal"]["ego_care_red_light_mask"][keep]
            data["origin"], data["angle"] = centre, theta
        return PlutoFeature(data=data)

    def to_device(self, device) -> 'PlutoFeature':
        return PlutoFeature(data={k: to_device(v, device) for k, v in self.data.items()})

    def serialize(self) -> Dict[str, Any]:
        return {"data": self.data}

    @classmethod
    def deserialize(cls, data: Dict[str, Any]) -> 'PlutoFeature':
        return PlutoFeature(data=data["data"])
