"""Device-resident replay arena + on-device collation.

Replaces the host hot loop of the reference update (256x ``CBVRolloutBuffer.sample`` +
~40 ``pad_sequence`` calls per step on the main thread, then H2D;
gym_carla/buffer/cbv_rollout_buffer.py:124-138, pluto_feature.py:83-94,
rift_datamodule.py:33-49): every scene is stored once in HBM, padded to
(A, Mp, Rcap, S), and a minibatch is ONE gather kernel (``rift_collate``).
"""
import ctypes as C
from typing import Dict, List, Optional

import torch

from rift_amd import _ffi

_F32, _I8, _U8 = torch.float32, torch.int8, torch.bool

# (struct field, group, key, dtype, ragged-over-R)
_FIELDS = [
    ("agent_position", "agent", "position", _F32, False), ("agent_heading", "agent", "heading", _F32, False),
    ("agent_velocity", "agent", "velocity", _F32, False), ("agent_shape", "agent", "shape", _F32, False),
    ("agent_category", "agent", "category", _I8, False), ("agent_valid_mask", "agent", "valid_mask", _U8, False),
    ("map_point_position", "map", "point_position", _F32, False), ("map_point_vector", "map", "point_vector", _F32, False),
    ("map_point_orientation", "map", "point_orientation", _F32, False),
    ("map_polygon_center", "map", "polygon_center", _F32, False), ("map_polygon_type", "map", "polygon_type", _I8, False),
    ("map_polygon_on_route", "map", "polygon_on_route", _U8, False),
    ("map_polygon_tl_status", "map", "polygon_tl_status", _I8, False),
    ("map_polygon_has_speed_limit", "map", "polygon_has_speed_limit", _U8, False),
    ("map_polygon_speed_limit", "map", "polygon_speed_limit", _F32, False),
    ("map_valid_mask", "map", "valid_mask", _U8, False),
    ("ref_position", "reference_line", "position", _F32, True), ("ref_vector", "reference_line", "vector", _F32, True),
    ("ref_orientation", "reference_line", "orientation", _F32, True),
    ("ref_valid_mask", "reference_line", "valid_mask", _U8, True),
    ("static_position", "static_objects", "position", _F32, False),
    ("static_heading", "static_objects", "heading", _F32, False), ("static_shape", "static_objects", "shape", _F32, False),
    ("static_category", "static_objects", "category", _I8, False),
    ("static_valid_mask", "static_objects", "valid_mask", _U8, False),
]


def _pad_stack(ts: List[torch.Tensor], n: int, dtype) -> torch.Tensor:
    """Stack per-scene tensors, zero-padding dim 0 to n (pad_sequence semantics)."""
    out = torch.zeros((len(ts), n) + tuple(ts[0].shape[1:]), dtype=dtype)
    for i, t in enumerate(ts):
        if t.shape[0]:
            out[i, : t.shape[0]] = t.to(dtype)
    return out


class DeviceReplay:
    """Replay arena in HBM.  `scenes` = list of {'feature': per-scene PlutoFeature.data, 'extras': {...}}."""

    def __init__(self, scenes: List[Dict], device, rcap: Optional[int] = None):
        self.device = torch.device(device)
        feats = [s["feature"] for s in scenes]
        ex = [s["extras"] for s in scenes]
        n = len(scenes)
        self.n = n
        self.A = max(f["agent"]["position"].shape[0] for f in feats)
        self.Mp = max(f["map"]["point_position"].shape[0] for f in feats)
        self.S = max(f["static_objects"]["position"].shape[0] for f in feats) if "static_objects" in feats[0] else 0
        self.T = feats[0]["agent"]["position"].shape[1]
        self.r_count_cpu = torch.tensor([f["reference_line"]["position"].shape[0] for f in feats], dtype=torch.int32)
        self.Rcap = int(rcap or self.r_count_cpu.max())
        dims = {"agent": self.A, "map": self.Mp, "reference_line": self.Rcap, "static_objects": self.S}
        self.t: Dict[str, torch.Tensor] = {}
        for name, grp, key, dt, _ in _FIELDS:
            if grp == "static_objects" and self.S == 0:
                continue
            self.t[name] = _pad_stack([f[grp][key] for f in feats], dims[grp], dt).to(self.device)
        self.t["current_state"] = torch.stack([f["current_state"] for f in feats]).float().to(self.device)
        self.cs_ld = self.t["current_state"].shape[1]
        self.t["old_group_logits"] = _pad_stack([e["old_group_logits"] for e in ex], self.Rcap, _F32).to(self.device)
        self.t["group_advantage"] = _pad_stack([e["group_advantage"] for e in ex], self.Rcap, torch.float64).to(self.device)
        self.t["group_valid_mask"] = _pad_stack([e["group_advantage_mask"] for e in ex], self.Rcap, _U8).to(self.device)
        if "ref_group_logits" in ex[0]:
            self.t["ref_group_logits"] = _pad_stack([e["ref_group_logits"] for e in ex], self.Rcap, _F32).to(self.device)
        self.r_count = self.r_count_cpu.to(self.device)
        ar = _ffi.RiftReplayArena()
        ar.n_scenes, ar.A, ar.Mp, ar.Rcap, ar.S, ar.T, ar.cs_ld = n, self.A, self.Mp, self.Rcap, self.S, self.T, self.cs_ld
        ar.scenes.bs, ar.scenes.A, ar.scenes.Mp, ar.scenes.R, ar.scenes.S, ar.scenes.T = n, self.A, self.Mp, self.Rcap, self.S, self.T
        for name, *_ in _FIELDS:
            setattr(ar.scenes, name, self.t[name].data_ptr() if name in self.t else None)
        ar.scenes.current_state = self.t["current_state"].data_ptr()
        ar.scenes.cs_ld = self.cs_ld
        ar.r_count = self.r_count.data_ptr()
        ar.old_group_logits = self.t["old_group_logits"].data_ptr()
        ar.ref_group_logits = self.t["ref_group_logits"].data_ptr() if "ref_group_logits" in self.t else None
        ar.group_advantage = self.t["group_advantage"].data_ptr()
        ar.group_valid_mask = self.t["group_valid_mask"].data_ptr()
        self.arena = ar
        self._out = {}

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())

    def _buffers(self, bs: int, R: int, slot: int = 0):
        key = (bs, R, slot)
        if key in self._out:
            return self._out[key]
        dev = self.device
        shp = {"agent": self.A, "map": self.Mp, "reference_line": R, "static_objects": self.S}
        b: Dict[str, torch.Tensor] = {}
        for name, grp, _, dt, _ in _FIELDS:
            if name not in self.t:
                continue
            b[name] = torch.zeros((bs, shp[grp]) + tuple(self.t[name].shape[2:]), dtype=dt, device=dev)
        b["current_state"] = torch.zeros(bs, self.cs_ld, device=dev)
        b["old_group_logits"] = torch.zeros(bs, R, 12, device=dev)
        b["ref_group_logits"] = torch.zeros(bs, R, 12, device=dev) if "ref_group_logits" in self.t else None
        b["group_advantage"] = torch.zeros(bs, R, 12, dtype=torch.float64, device=dev)
        b["group_valid_mask"] = torch.zeros(bs, R, 12, dtype=torch.bool, device=dev)
        fb = _ffi.RiftFeatureBatch()
        fb.bs, fb.A, fb.Mp, fb.R, fb.S, fb.T = bs, self.A, self.Mp, R, self.S, self.T
        for name, *_ in _FIELDS:
            setattr(fb, name, b[name].data_ptr() if name in b else None)
        fb.current_state = b["current_state"].data_ptr()
        fb.cs_ld = self.cs_ld
        self._out[key] = (fb, b)
        return self._out[key]

    def collate(self, engine: "_ffi.Engine", scene_idx: torch.Tensor, R_out: Optional[int] = None, slot: int = 0,
                stream: Optional[torch.cuda.Stream] = None):
        """Gather `scene_idx` (int32, device) into the (cached) batch buffers.
        Returns (RiftFeatureBatch, dict of batch tensors incl. the RIFT/GRPO extras).
        `slot`: which of the cached buffer sets to fill -- a trainer that runs the loss of step k beside the forward of step k + 1
        (RLFTTrainer.next_slot) cycles through _ffi.DEFER_SLOTS of them, so that step k + 1's gather does not overwrite what step k's loss still reads.
        `stream`: gather on this stream instead of the current one (RLFTTrainer.prefetch_stream: the gather then runs beside the previous
        step's kernels; the trainer's forward waits for it).  `scene_idx` must be complete when the call is made -- an index tensor built
        by kernels still queued on another stream needs its own event."""
        bs = scene_idx.numel()
        if R_out is None:
            R_out = self.Rcap
        if stream is not None:
            with torch.cuda.stream(stream):       # (a slot's buffers are allocated once, from the pool of the stream that writes them)
                fb, b = self._buffers(bs, R_out, slot)
        else:
            fb, b = self._buffers(bs, R_out, slot)
        rc = engine.lib.rift_collate(
            engine.ctx, C.byref(self.arena), C.c_void_p(scene_idx.data_ptr()), bs, R_out, C.byref(fb),
            _ffi._ptr(b["old_group_logits"]), _ffi._ptr(b["ref_group_logits"]), _ffi._ptr(b["group_advantage"]),
            _ffi._ptr(b["group_valid_mask"]), C.c_void_p(stream.cuda_stream) if stream is not None else _ffi._stream())
        if stream is not None:
            scene_idx.record_stream(stream)
        if rc != 0:
            engine._check(rc, "rift_collate")
        return fb, b

    def batch_dict(self, b: Dict[str, torch.Tensor]) -> Dict:
        """View the batch buffers as the collated dict layout of the reference (for tests / generic callers)."""
        data = {"agent": {}, "map": {}, "reference_line": {}, "static_objects": {}}
        for name, grp, key, _, _ in _FIELDS:
            if name in b:
                data[grp][key] = b[name]
        data["current_state"] = b["current_state"]
        return data


# ---- on-disk scene dump ---------------------------------------------------------------------------
# The reference keeps rollout scenes only as Python objects inside CBVRolloutBuffer (cbv_rollout_buffer.py:16-138); BASELINE config 0
# speaks of "pre-dumped CARLA rollout scenes", so the dump format is defined here: ONE .npz per replay, every per-scene tensor of the
# Appendix-A schema stored under "<scene index>/<group>/<key>" ("<i>/current_state", "<i>/extras/<key>" for the RLFT extras), dtypes
# as PlutoFeature holds them (fp32 / fp64 features, bool masks, int8 categories).  `buffer_to_scenes` output goes in, the same list
# comes out, so a replay dumped next to CARLA can be trained on without it.
DUMP_VERSION = 1


def save_scenes(path: str, scenes: List[Dict]) -> None:
    import numpy as np
    out = {"__rift_scene_dump__": np.array([DUMP_VERSION, len(scenes)], dtype=np.int64)}
    for i, s in enumerate(scenes):
        for grp, v in s["feature"].items():
            if isinstance(v, dict):
                for k, t in v.items():
                    out[f"{i}/{grp}/{k}"] = torch.as_tensor(t).cpu().numpy()
            else:
                out[f"{i}/{grp}"] = torch.as_tensor(v).cpu().numpy()
        for k, t in s.get("extras", {}).items():
            out[f"{i}/extras/{k}"] = torch.as_tensor(t).cpu().numpy()
    np.savez_compressed(path, **out)


def load_scenes(path: str) -> List[Dict]:
    import numpy as np
    z = np.load(path)
    if "__rift_scene_dump__" not in z.files or int(z["__rift_scene_dump__"][0]) != DUMP_VERSION:
        raise ValueError(f"{path}: not a rift_amd scene dump of version {DUMP_VERSION}")
    n = int(z["__rift_scene_dump__"][1])
    scenes: List[Dict] = [{"feature": {}, "extras": {}} for _ in range(n)]
    for name in z.files:
        if name.startswith("__"):
            continue
        parts = name.split("/")
        i, t = int(parts[0]), torch.from_numpy(z[name])
        if parts[1] == "extras":
            scenes[i]["extras"][parts[2]] = t
        elif len(parts) == 2:
            scenes[i]["feature"][parts[1]] = t
        else:
            scenes[i]["feature"].setdefault(parts[1], {})[parts[2]] = t
    return scenes
