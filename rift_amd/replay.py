"""Device-resident replay arena + on-device collation.

Replaces the host hot loop of the reference update (256x ``CBVRolloutBuffer.sample`` +
~40 ``pad_sequence`` calls per step on the main thread, then H2D;
gym_carla/buffer/cbv_rollout_buffer.py:124-138, pluto_feature.py:83-94,
rift_datamodule.py:33-49): every scene is stored once in HBM, padded to
(A, Mp, Rcap, S), and a minibatch is ONE gather kernel (``rift_collate``).
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
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


_EXTRA_FIELDS = (("old_group_logits", _F32), ("group_advantage", torch.float64), ("group_valid_mask", _U8), ("ref_group_logits", _F32))
_GROUP_OF = {name: grp for name, grp, *_ in _FIELDS}
_DIM_OF = {"agent": "A", "map": "Mp", "reference_line": "R", "static_objects": "S"}


def _np(t):
    """numpy view of a per-scene tensor (torch CPU tensor or ndarray); no copy for contiguous inputs."""
    return t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


class HostReplay:
    """The replay as ONE pinned host structure-of-arrays, written a committed transition at a time.

    The reference keeps a replay entry as Python objects and pays for it in every training step (256 x `CBVRolloutBuffer.sample` +
    ~40 `pad_sequence`, cbv_rollout_buffer.py:124-138, rift_datamodule.py:33-49).  Rounds 1-3 of this build moved that cost to one Python
    pass per update (`DeviceReplay(scenes)`: 25 fields x 4096 copies inside `RLFTPluto.train`, seconds).  Here the transition is laid
    into the arena's layout WHEN IT IS COMMITTED (`CBVRolloutBuffer.store`, during the rollout, where the simulator is the bottleneck):
    one preallocated page-locked array per feature key, scene-major, every ragged leading dimension (agents, polygons, reference lines,
    static objects) padded to a capacity that doubles when a scene exceeds it.  `DeviceReplay.upload` then moves the arena to HBM as
    ~30 asynchronous host-to-device copies of contiguous row ranges and the update starts behind them.

    `caps`: initial capacities {"A", "Mp", "R", "S"} (rift_pluto.yaml:35-36: max_agent 48 -> A <= 49 with the CBV itself; the map
    crop of radius 120 gives Mp of a few dozen polygons; R <= 6 reference lines)."""

    DEFAULT_CAPS = {"A": 49, "Mp": 64, "R": 6, "S": 8}

    def __init__(self, capacity: int, caps: Optional[Dict[str, int]] = None, pin: Optional[bool] = None):
        self.capacity = int(capacity)
        self.caps = dict(self.DEFAULT_CAPS)
        self.caps.update(caps or {})
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self.T: Optional[int] = None            # time extent of the agent tensors, fixed by the first scene
        self.cs_ld: Optional[int] = None
        self.t: Dict[str, torch.Tensor] = {}    # name -> (capacity, cap, ...) page-locked tensor
        self.v: Dict[str, np.ndarray] = {}      # name -> numpy view of the same memory (what put() writes through)
        self.dims = {"A": 0, "Mp": 0, "R": 0, "S": 0}       # maxima over the stored scenes = the padded sizes of a collated batch
        self.r_count = torch.zeros(self.capacity, dtype=torch.int32)
        self._rc = self.r_count.numpy()
        self.has_ref_logits = False
        self.version = 0                        # bumped by every write: DeviceReplay.upload skips an arena it already holds
        self.grown = 0

    # ---- storage -----------------------------------------------------------------------------------------------------------
    def _alloc(self, name: str, tail: Sequence[int], dtype, cap: Optional[int]):
        shape = (self.capacity,) + ((cap,) if cap is not None else ()) + tuple(tail)
        t = torch.zeros(shape, dtype=dtype, pin_memory=self.pin)
        self.t[name] = t
        self.v[name] = t.numpy() if dtype != torch.bool else t.view(torch.uint8).numpy()

    def _grow(self, dim: str, need: int):
        """Double the capacity of one ragged dimension (rare: a scene with more agents / polygons / lines than any before)."""
        new = max(2 * self.caps[dim], need)
        for name in [n for n in self.t if self._dim_of(n) == dim]:
            old = self.t[name]
            self._alloc(name, old.shape[2:], old.dtype, new)
            self.t[name][:, :old.shape[1]] = old
        self.caps[dim] = new
        self.grown += 1

    @staticmethod
    def _dim_of(name: str) -> Optional[str]:
        if name in _GROUP_OF:
            return _DIM_OF[_GROUP_OF[name]]
        return "R" if name in ("old_group_logits", "group_advantage", "group_valid_mask", "ref_group_logits") else None

    def _put(self, i: int, name: str, a: np.ndarray, dtype, dim: Optional[str]):
        if dim is None:
            if name not in self.t:
                self._alloc(name, a.shape, dtype, None)
            self.v[name][i] = a
            return
        n = a.shape[0]
        if n > self.caps[dim]:
            self._grow(dim, n)
        if name not in self.t:
            self._alloc(name, a.shape[1:], dtype, self.caps[dim])
        row = self.v[name][i]
        if n:
            row[:n] = a                 # numpy casts on assignment (fp64 -> fp32 as PlutoFeature.to_feature_tensor does, bool -> u8)
        row[n:] = 0                     # pad_sequence's zeros; also clears what an earlier update left in this slot
        if n > self.dims[dim]:
            self.dims[dim] = n

    def put(self, i: int, feature: Dict, extras: Optional[Dict] = None):
        """Lay scene `i` (a per-scene PlutoFeature.data dict + the RLFT extras of `buffer_to_scenes`) into the arena."""
        if not 0 <= i < self.capacity:
            raise IndexError(i)
        if self.T is None:
            self.T = int(feature["agent"]["position"].shape[1])
        for name, grp, key, dt, _ in _FIELDS:
            g = feature.get(grp)
            if g is None or key not in g:
                continue
            self._put(i, name, _np(g[key]), dt, _DIM_OF[grp])
        cs = _np(feature["current_state"])
        self.cs_ld = int(cs.shape[0])
        self._put(i, "current_state", cs, _F32, None)
        R = int(feature["reference_line"]["position"].shape[0])
        self._rc[i] = R
        ex = extras or {}
        adv = ex.get("group_advantage")
        self._put(i, "group_advantage", _np(adv) if adv is not None else np.zeros((R, 12)), torch.float64, "R")
        msk = ex.get("group_advantage_mask")
        self._put(i, "group_valid_mask", _np(msk) if msk is not None else np.ones((R, 12), dtype=np.bool_), _U8, "R")
        old = ex.get("old_group_logits")
        self._put(i, "old_group_logits", _np(old) if old is not None else np.zeros((R, 12), dtype=np.float32), _F32, "R")
        ref = ex.get("ref_group_logits")
        if ref is not None:
            self.has_ref_logits = True
            self._put(i, "ref_group_logits", _np(ref), _F32, "R")
        self.version += 1

    def reset(self):
        """New replay generation: the slots are overwritten (each put() clears its own padding), the batch dimensions start over."""
        self.dims = {"A": 0, "Mp": 0, "R": 0, "S": 0}
        self.has_ref_logits = False
        self.version += 1

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())


class PersistentBatch(dict):
    """The batch tensors of one cached buffer set (DeviceReplay.collate): they live as long as the replay does, so a consumer on another
    stream need not `record_stream` them against the caching allocator (26 calls per update step, a tenth of a small-batch step's host
    time -- RLFTTrainer checks `persistent`)."""
    persistent = True


class DeviceReplay:
    """Replay arena in HBM.  `scenes` = list of {'feature': per-scene PlutoFeature.data, 'extras': {...}} (tests, dumps, bench), or
    -- the product path -- `DeviceReplay.from_host(HostReplay)`: the arena the rollout buffer filled, uploaded as it lies."""

    def __init__(self, scenes: Optional[List[Dict]], device, rcap: Optional[int] = None):
        self.device = torch.device(device)
        self._out = {}
        self.t: Dict[str, torch.Tensor] = {}
        self._host_version = None
        if scenes is None:
            return
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
        self._describe(self.A, self.Mp, self.Rcap, self.S)

    def _describe(self, capA: int, capMp: int, capR: int, capS: int):
        """The RiftReplayArena descriptor: capacities of the stored rows (the gather's source strides).  The batch a gather writes is
        padded to (self.A, self.Mp, R_out, self.S) <= the capacities: every ragged dimension leads its per-scene block, so cropping is
        a prefix copy (rift_collate)."""
        ar = _ffi.RiftReplayArena()
        ar.n_scenes, ar.A, ar.Mp, ar.Rcap, ar.S, ar.T, ar.cs_ld = self.n, capA, capMp, capR, capS, self.T, self.cs_ld
        ar.scenes.bs, ar.scenes.A, ar.scenes.Mp, ar.scenes.R, ar.scenes.S, ar.scenes.T = self.n, capA, capMp, capR, capS, self.T
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

    # ---- the product path: the arena comes up from the rollout buffer's pinned host mirror ----------------------------------------
    @classmethod
    def from_host(cls, host: HostReplay, device, n: Optional[int] = None) -> "DeviceReplay":
        self = cls(None, device)
        self.upload(host, n)
        return self

    def upload(self, host: HostReplay, n: Optional[int] = None):
        """Bring the first `n` scenes of `host` to HBM: one asynchronous copy per stored key (<= 30) on the current stream, out of
        page-locked memory -- the host returns at once, and whatever is queued behind (index uploads, the prefetch stream's first gather
        through its `ready` event, PPO's sweeps) runs when the bytes are there.  Device tensors are kept across updates as long as the
        host arena keeps its capacities; the batch-buffer sets are kept as long as the batch dimensions do."""
        n = host.capacity if n is None else int(n)
        if host.T is None:
            raise ValueError("the host replay is empty")
        if self._host_version == (id(host), host.version, n):
            return
        caps, dims = host.caps, host.dims
        geom = (n, caps["A"], caps["Mp"], caps["R"], caps["S"], host.T, host.cs_ld, host.has_ref_logits, dims["S"] > 0)
        if getattr(self, "_geom", None) != geom:
            self.t = {}
            for name, src in host.t.items():
                if name == "ref_group_logits" and not host.has_ref_logits:
                    continue
                if _GROUP_OF.get(name) == "static_objects" and dims["S"] == 0:
                    continue
                self.t[name] = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device)
            self.r_count = torch.empty(n, dtype=torch.int32, device=self.device)
            self._geom = geom
        out_dims = (dims["A"], dims["Mp"], dims["S"])
        if getattr(self, "_out_dims", None) != out_dims:
            self._out = {}
            self._out_dims = out_dims
        self.n, self.T, self.cs_ld = n, host.T, host.cs_ld
        self.A, self.Mp, self.S, self.Rcap = dims["A"], dims["Mp"], dims["S"], caps["R"]
        for name, dst in self.t.items():
            dst.copy_(host.t[name][:n], non_blocking=True)
        self.r_count.copy_(host.r_count[:n], non_blocking=True)
        self.r_count_cpu = host.r_count[:n].clone()
        self._describe(caps["A"], caps["Mp"], caps["R"], caps["S"] if dims["S"] else 0)
        self._host_version = (id(host), host.version, n)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())

    def _buffers(self, bs: int, R: int, slot: int = 0):
        key = (bs, R, slot)
        if key in self._out:
            return self._out[key]
        dev = self.device
        shp = {"agent": self.A, "map": self.Mp, "reference_line": R, "static_objects": self.S}
        b: Dict[str, torch.Tensor] = PersistentBatch()
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
        cached = self._out.get((bs, R_out, slot))
        if cached is not None:
            fb, b = cached
        elif stream is not None:
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
