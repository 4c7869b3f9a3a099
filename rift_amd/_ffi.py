"""ctypes binding of ``librift_hip.so`` (C-ABI in ``include/rift_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or fails
to load, importing the engine raises immediately.
"""
import ctypes as C
import threading
import os
from typing import Dict, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librift_hip.so")

# forward flags / loss kinds (include/rift_hip.h)
F_TRAIN, F_NEED_TRAJ, F_FP32, F_NO_DROP, F_NO_BN_UPDATE, F_DEFER_HEAD = 1, 2, 4, 8, 16, 32
DEFER_SLOTS = 4        # RIFT_DEFER_SLOTS: activation arenas the deferred-head forwards cycle through
LOSS_KINDS = {"rift": 0, "grpo": 1, "ppo": 2, "reinforce": 3, "sft": 4}
PI_NPARAM = 16897

vp = C.c_void_p


class RiftTensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", vp), ("numel", C.c_int64), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


_FB_PTRS = [
    "agent_position", "agent_heading", "agent_velocity", "agent_shape", "agent_category", "agent_valid_mask",
    "map_point_position", "map_point_vector", "map_point_orientation", "map_polygon_center", "map_polygon_type",
    "map_polygon_on_route", "map_polygon_tl_status", "map_polygon_has_speed_limit", "map_polygon_speed_limit",
    "map_valid_mask", "ref_position", "ref_vector", "ref_orientation", "ref_valid_mask", "static_position",
    "static_heading", "static_shape", "static_category", "static_valid_mask", "current_state",
]


class RiftFeatureBatch(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("bs", "A", "Mp", "R", "S", "T")] + [(n, vp) for n in _FB_PTRS]
                + [("cs_ld", C.c_int32)])


class RiftOutputs(C.Structure):
    _fields_ = [(n, vp) for n in ("probability", "hidden", "trajectory", "prediction", "ref_free_trajectory")]


class RiftLossIn(C.Structure):
    _fields_ = [(n, vp) for n in ("old_group_logits", "ref_group_logits", "group_advantage", "group_valid_mask",
                                  "action_mode", "advantage", "old_log_prob", "returns")] + \
               [("clip_epsilon", C.c_float), ("lambda_entropy", C.c_float)]


class RiftLossOut(C.Structure):
    _fields_ = [(n, vp) for n in ("loss", "stats", "flat_grad_sum", "grad_w1", "grad_b1", "grad_ln_w", "grad_ln_b",
                                  "grad_w2", "grad_b2", "argmax_rm", "exchange")]


class RiftCritic(C.Structure):
    _fields_ = [(n, vp) for n in ("w0", "b0", "w1", "b1", "w2", "b2", "state_avg", "state_std", "value_avg", "value_std")]


class RiftReplayArena(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_scenes", "A", "Mp", "Rcap", "S", "T", "cs_ld")] + \
               [("scenes", RiftFeatureBatch)] + \
               [(n, vp) for n in ("r_count", "old_group_logits", "ref_group_logits", "group_advantage",
                                  "group_valid_mask")]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)


class RiftDp(C.Structure):
    _fields_ = [("scene_offset", C.c_int32), ("global_bs", C.c_int32), ("xchg", vp), ("xchg_len", C.c_int64),
                ("exchange", EXCHANGE_FN), ("user", vp)]


class RiftRolloutIO(C.Structure):
    _fields_ = [("trajectories", vp), ("G", C.c_int32), ("Tfull", C.c_int32), ("G_per_group", C.c_int32), ("center_state", vp),
                ("turn_buf", vp), ("turn_ptr", vp), ("turn_len", vp), ("speed_buf", vp), ("speed_ptr", vp), ("speed_len", vp),
                ("center", vp), ("angle", vp), ("speed", vp), ("acc", vp), ("ang_vel", vp), ("ang_acc", vp), ("vertices", vp),
                ("closest_index", vp), ("aim_idx", vp)]


class RiftTickCBV(C.Structure):        # include/rift_hip.h: one CBV of rift_group_advantage_tick (every pointer a device pointer)
    _fields_ = [("batch_index", C.c_int32), ("R", C.c_int32), ("Pmax", C.c_int32), ("n_actors", C.c_int32), ("center_state", vp),
                ("ref_pos", vp), ("ref_angle", vp), ("ref_len", vp), ("actors", vp), ("off_road_mask", vp), ("H", C.c_int32), ("W", C.c_int32),
                ("pose", C.c_double * 3)]


OPERANDS = {"bf16": 0, "fp16": 1}       # RIFT_OPERANDS_* of include/rift_hip.h: the 16-bit MFMA operand format of a context's fused kernels
EXPORTS = [
    "rift_ctx_create", "rift_ctx_create_ex", "rift_ctx_operand_format", "rift_ctx_destroy", "rift_last_error", "rift_model_load", "rift_forward", "rift_forward_head", "rift_forward_head_back", "rift_loss_backward",
    "rift_loss_finalize", "rift_loss_finalize_clip", "rift_set_param_event", "rift_tap", "rift_op_linear", "rift_gae", "rift_discounted_return",
    "rift_normalize_advantage", "rift_group_advantage", "rift_rollout_return", "rift_collate",
    "rift_prof_enable", "rift_prof_report", "rift_op_linear_bench", "rift_ref_line_info", "rift_rollout",
    "rift_critic_forward", "rift_critic_loss_backward", "rift_critic_finalize", "rift_clip_grad_norm", "rift_adamw_step", "rift_update_tail", "rift_collision_matrix", "rift_off_road_matrix", "rift_other_vehicle_rollout", "rift_sft_teacher_mode",
    "rift_check_finite", "rift_set_dp", "rift_set_prepare_stream", "rift_set_side_stream",
    "rift_comm_unique_id", "rift_comm_init", "rift_comm_all_reduce", "rift_comm_destroy", "rift_group_advantage_tick",
]
CRITIC_NPARAM = 99331
CRITIC_KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
               "state_avg", "state_std", "value_avg", "value_std")

_libs = {}


def load_library(variant: str = "") -> C.CDLL:
    """Load librift_hip.so (variant "stats": librift_hip_stats.so, the diagnostic twin with the drop-decision counters of
    csrc/dropstats.h); raise loudly when it is absent (no fallback path exists)."""
    if variant in _libs:
        return _libs[variant]
    path = LIB_PATH if not variant else LIB_PATH.replace("librift_hip.so", f"librift_hip_{variant}.so")
    if not variant and os.environ.get("RIFT_LIB"):       # diagnostic: an A/B build of the library (rift_amd.build.build_variant)
        path = os.environ["RIFT_LIB"]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). rift_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"librift_hip.so does not export {name}")
    lib.rift_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rift_ctx_create_ex.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.rift_ctx_operand_format.argtypes = [vp]
    lib.rift_ctx_destroy.argtypes = [vp]
    lib.rift_ctx_destroy.restype = None
    lib.rift_last_error.argtypes = [vp]
    lib.rift_last_error.restype = C.c_char_p
    lib.rift_model_load.argtypes = [vp, C.POINTER(RiftTensorDesc), C.c_int, vp]
    lib.rift_forward.argtypes = [vp, C.POINTER(RiftFeatureBatch), C.POINTER(RiftOutputs), C.c_int, C.c_uint32, vp]
    lib.rift_forward_head.argtypes = [vp, vp]
    lib.rift_forward_head_back.argtypes = [vp, C.c_int, vp]
    lib.rift_loss_backward.argtypes = [vp, C.c_int, C.POINTER(RiftLossIn), C.POINTER(RiftLossOut), vp]
    lib.rift_loss_finalize.argtypes = [vp, C.POINTER(RiftLossOut), C.c_int, vp]
    lib.rift_set_param_event.argtypes = [vp, vp]
    lib.rift_check_finite.argtypes = [vp, vp]
    lib.rift_set_dp.argtypes = [vp, C.POINTER(RiftDp)]
    lib.rift_set_prepare_stream.argtypes = [vp, vp]
    lib.rift_set_side_stream.argtypes = [vp, vp]
    lib.rift_comm_unique_id.argtypes = [vp, vp]
    lib.rift_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.rift_comm_all_reduce.argtypes = [vp, vp, C.c_int64, vp]
    lib.rift_group_advantage_tick.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(RiftTickCBV), C.c_int, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp]
    lib.rift_comm_destroy.argtypes = [vp]
    lib.rift_loss_finalize_clip.argtypes = [vp, C.POINTER(RiftLossOut), C.c_int, C.c_float, vp, vp]
    lib.rift_tap.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), vp]
    lib.rift_critic_forward.argtypes = [vp, C.POINTER(RiftCritic), vp, C.c_int, vp, vp]
    lib.rift_critic_loss_backward.argtypes = [vp, C.POINTER(RiftCritic), vp, vp, C.c_int, vp, vp, vp]
    lib.rift_critic_finalize.argtypes = [vp] * 14
    lib.rift_clip_grad_norm.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64), C.c_int, C.c_float, vp, vp]
    lib.rift_update_tail.argtypes = [vp, C.POINTER(RiftLossOut), C.c_int, C.c_float, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                     C.POINTER(vp), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double, vp]
    lib.rift_adamw_step.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                    C.c_double, C.c_double, vp]
    lib.rift_sft_teacher_mode.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.rift_other_vehicle_rollout.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp]
    lib.rift_collision_matrix.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    lib.rift_off_road_matrix.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int] + [C.c_double] * 7 + [vp, vp]
    lib.rift_op_linear.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.rift_gae.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, vp, vp]
    lib.rift_discounted_return.argtypes = [vp, vp, vp, C.c_double, C.c_int, vp, vp]
    lib.rift_normalize_advantage.argtypes = [vp, vp, C.c_int, vp]
    lib.rift_group_advantage.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.rift_rollout_return.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                        C.c_double, vp, vp]
    lib.rift_collate.argtypes = [vp, C.POINTER(RiftReplayArena), vp, C.c_int, C.c_int, C.POINTER(RiftFeatureBatch),
                                 vp, vp, vp, vp, vp]
    lib.rift_op_linear_bench.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int,
                                         C.POINTER(C.c_float), vp]
    lib.rift_ref_line_info.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.rift_rollout.argtypes = [vp, C.POINTER(RiftRolloutIO), vp]
    lib.rift_prof_enable.argtypes = [vp, C.c_int]
    lib.rift_prof_report.argtypes = [vp, C.c_char_p, C.c_int]
    for name in EXPORTS:
        if name not in ("rift_ctx_destroy", "rift_last_error"):
            getattr(lib, name).restype = C.c_int
    _libs[variant] = lib
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _KnownStreams(threading.local):        # per host thread: a policy driven from one thread must not hand its stream to another thread's calls
    def __init__(self):
        self.stack = []


_KNOWN_STREAM = _KnownStreams()


def _stream():
    """The stream handle the C-ABI calls are given: torch's current stream -- or the one a caller has declared current (known_stream)."""
    st = _KNOWN_STREAM.stack
    if st:
        return st[-1]
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_DEBUG_STREAMS = os.environ.get("RIFT_DEBUG_STREAMS") == "1"      # (read once: known_stream is entered two or three times per update step)


class known_stream:
    """`with known_stream(s):` -- the caller states that `s` IS torch's current stream for the duration (it entered `torch.cuda.stream(s)` or
    read `torch.cuda.current_stream()` itself), which saves the ~7 us lookup in every C-ABI call of the block: eight per update step, a
    quarter of the step's host time, and the host time is what bounds a small-batch step."""

    def __init__(self, s: torch.cuda.Stream):
        self.h = C.c_void_p(s.cuda_stream)

    def __enter__(self):
        if _DEBUG_STREAMS:                                    # the declaration must be true
            assert self.h.value == torch.cuda.current_stream().cuda_stream, "known_stream(s): s is not torch's current stream"
        _KNOWN_STREAM.stack.append(self.h)
        return self

    def __exit__(self, *exc):
        _KNOWN_STREAM.stack.pop()
        return False


def _dev(t: torch.Tensor, dtype, device) -> torch.Tensor:
    if t.dtype != dtype or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=dtype).contiguous()
    return t


def feature_batch(data: Dict, device) -> (RiftFeatureBatch, list):
    """Build the POD descriptor of a collated feature dict.
    Returns (struct, keepalive list of the tensors whose pointers it holds)."""
    f32, i8, u8 = torch.float32, torch.int8, torch.bool
    ag, mp, rl = data["agent"], data["map"], data["reference_line"]
    so = data.get("static_objects")
    keep = []

    def put(t, dt):
        t = _dev(t, dt, device)
        keep.append(t)
        return t

    fb = RiftFeatureBatch()
    pos = put(ag["position"], f32)
    bs, A, T = pos.shape[:3]
    Mp = mp["point_position"].shape[1]
    R = rl["position"].shape[1]
    S = so["position"].shape[1] if so is not None else 0
    fb.bs, fb.A, fb.Mp, fb.R, fb.S, fb.T = bs, A, Mp, R, S, T
    t = {
        "agent_position": pos, "agent_heading": put(ag["heading"], f32), "agent_velocity": put(ag["velocity"], f32),
        "agent_shape": put(ag["shape"], f32), "agent_category": put(ag["category"], i8),
        "agent_valid_mask": put(ag["valid_mask"], u8),
        "map_point_position": put(mp["point_position"], f32), "map_point_vector": put(mp["point_vector"], f32),
        "map_point_orientation": put(mp["point_orientation"], f32), "map_polygon_center": put(mp["polygon_center"], f32),
        "map_polygon_type": put(mp["polygon_type"], i8), "map_polygon_on_route": put(mp["polygon_on_route"], u8),
        "map_polygon_tl_status": put(mp["polygon_tl_status"], i8),
        "map_polygon_has_speed_limit": put(mp["polygon_has_speed_limit"], u8),
        "map_polygon_speed_limit": put(mp["polygon_speed_limit"], f32), "map_valid_mask": put(mp["valid_mask"], u8),
        "ref_position": put(rl["position"], f32), "ref_vector": put(rl["vector"], f32),
        "ref_orientation": put(rl["orientation"], f32), "ref_valid_mask": put(rl["valid_mask"], u8),
        "current_state": put(data["current_state"], f32),
    }
    if S > 0:
        t.update({"static_position": put(so["position"], f32), "static_heading": put(so["heading"], f32),
                  "static_shape": put(so["shape"], f32), "static_category": put(so["category"], i8),
                  "static_valid_mask": put(so["valid_mask"], u8)})
    for name in _FB_PTRS:
        setattr(fb, name, t[name].data_ptr() if name in t else None)
    fb.cs_ld = t["current_state"].shape[1]
    return fb, keep


class Engine:
    """One RiftCtx bound to one device + the torch tensors it borrows."""

    def __init__(self, device=None, operands: str = "bf16", variant: str = ""):
        """operands: "bf16" | "fp16" -- the MFMA operand format of the fused kernels (forward(fp32=True) ignores it).
        variant: "" | "stats" (the diagnostic library; bf16 operands only)."""
        if operands not in OPERANDS:
            raise ValueError(f"operands must be one of {sorted(OPERANDS)}, got {operands!r}")
        if not torch.cuda.is_available():
            raise RuntimeError("rift_amd.Engine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = load_library(variant)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.ctx = vp()
        self.operands = operands
        rc = self.lib.rift_ctx_create_ex(self.device.index or 0, OPERANDS[operands], C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"rift_ctx_create_ex failed ({rc})")
        self._params = None
        self._names = None
        self._keep = []
        self._stage_host = self._stage_dev = None
        self._stage_off = 0

    # ---- small host inputs (the rollout tick's per-CBV readings) ---------------------------------
    _STAGE_BYTES = 8 << 20
    _NP_OF = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8,
              torch.int8: np.int8, torch.bool: np.bool_}

    def stage_tree(self, data):
        """A nested dict of CPU tensors (a collated feature batch) -> the same dict on the device.  Round 6: the leaves are laid side by side
        in the pinned arena and go up in ONE asynchronous copy (a tick's batch is ~34 small tensors: 34 copies of ~8 us of host time each
        were a quarter of the tick, tools/tick_latency.py --profile); leaves the arena cannot take (other dtypes, > 2 MiB, empty, already on
        the device) go through stage() / .to() one by one."""
        leaves = []

        def walk(d):
            if isinstance(d, dict):
                return {k: walk(v) for k, v in d.items()}
            if torch.is_tensor(d):
                leaves.append(d)
                return len(leaves) - 1
            raise NotImplementedError(type(d))

        shape = walk(data)
        out = self.stage_many(leaves)

        def build(d):
            return {k: build(v) for k, v in d.items()} if isinstance(d, dict) else out[d]
        return build(shape)

    def stage_many(self, tensors):
        """[CPU tensor] -> [device tensor], the arena-sized ones through one span of the pinned arena and one asynchronous copy (see stage())."""
        if self._stage_host is None:
            self._stage_host = torch.empty(self._STAGE_BYTES, dtype=torch.uint8).pin_memory()
            self._stage_dev = torch.empty(self._STAGE_BYTES, dtype=torch.uint8, device=self.device)
        plan, total = [], 0
        for t in tensors:
            if t.is_cuda or t.dtype not in self._NP_OF or t.numel() == 0 or t.numel() * t.element_size() > self._STAGE_BYTES // 4:
                plan.append(None)
                continue
            a = np.ascontiguousarray(t.detach().numpy())
            plan.append((a, total))
            total = (total + a.nbytes + 255) & ~255
        if total > self._STAGE_BYTES // 2:               # (an unusually large batch: leaf by leaf, each with its own wrap check)
            return [self.stage(t, t.dtype) if p is not None else (t if t.is_cuda else t.to(self.device)) for t, p in zip(tensors, plan)]
        off = (self._stage_off + 255) & ~255
        if off + total > self._STAGE_BYTES:
            torch.cuda.synchronize(self.device)          # (see stage(): every consumer of the slots about to be rewritten is done)
            off = 0
        host = self._stage_host.numpy()
        for p in plan:
            if p is not None:
                a, o = p
                host[off + o:off + o + a.nbytes] = a.reshape(-1).view(np.uint8)
        if total:
            self._stage_dev[off:off + total].copy_(self._stage_host[off:off + total], non_blocking=True)
        self._stage_off = off + total
        out = []
        for t, p in zip(tensors, plan):
            if p is None:
                out.append(t if t.is_cuda and t.device == self.device else t.to(self.device))
            else:
                a, o = p
                out.append(self._stage_dev[off + o:off + o + a.nbytes].view(t.dtype).view(a.shape))
        return out

    def stage(self, a, dtype: torch.dtype) -> torch.Tensor:
        """Host array / CPU tensor -> device tensor of `dtype` through a pinned arena and an ASYNCHRONOUS copy on the current stream.
        `tensor.to(device)` from pageable memory waits for everything the stream holds; a tick's evaluation chain makes ~15 such uploads
        per CBV, each between kernel launches, and that wait -- not the copies -- was a third of the tick (tools/tick_latency.py).  The
        arena advances by the call and wraps behind a stream synchronisation, so a staged tensor is valid until 8 MiB of later uploads:
        it is for the inputs of the call at hand, not for keeping (a policy's `data` dict is invalid after its tick), and the arena is one per
        engine: not for concurrent callers.  Large (> 2 MiB) and empty inputs take the ordinary path."""
        if torch.is_tensor(a):
            if a.is_cuda:
                return _dev(a, dtype, self.device)
            a = a.detach().numpy()
        a = np.ascontiguousarray(a, dtype=self._NP_OF[dtype])
        n = a.nbytes
        if n == 0 or n > self._STAGE_BYTES // 4:
            return torch.from_numpy(a).to(self.device)
        if self._stage_host is None:
            self._stage_host = torch.empty(self._STAGE_BYTES, dtype=torch.uint8).pin_memory()
            self._stage_dev = torch.empty(self._STAGE_BYTES, dtype=torch.uint8, device=self.device)
        off = (self._stage_off + 255) & ~255
        if off + n > self._STAGE_BYTES:
            torch.cuda.synchronize(self.device)      # every consumer of the slots about to be rewritten is done -- on ANY stream (a staged batch may
                                                     # have been handed to a hook that reads it elsewhere); once per 8 MiB of uploads
            off = 0
        self._stage_host.numpy()[off:off + n] = a.reshape(-1).view(np.uint8)
        d = self._stage_dev[off:off + n]
        d.copy_(self._stage_host[off:off + n], non_blocking=True)
        self._stage_off = off + n
        return d.view(dtype).view(a.shape)

    def close(self):
        if self.ctx:
            self.lib.rift_ctx_destroy(self.ctx)
            self.ctx = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0 and getattr(self, "_dp_error", None) is not None:
            e, self._dp_error = self._dp_error, None
            raise RuntimeError(f"{what}: the data-parallel exchange raised") from e
        if rc != 0:
            msg = self.lib.rift_last_error(self.ctx)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ---- parameters -------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Bind views onto `sd`'s tensors (moved to the device if necessary; the engine keeps
        references, pi_head.* and BatchNorm running statistics are read / written in place)."""
        params = {}
        for k, v in sd.items():
            dt = torch.int64 if v.dtype == torch.int64 else torch.float32
            params[k] = _dev(v, dt, self.device)
        descs = (RiftTensorDesc * len(params))()
        names = []
        for i, (k, v) in enumerate(params.items()):
            nb = k.encode()
            names.append(nb)
            descs[i].name = nb
            descs[i].data = v.data_ptr()
            descs[i].numel = v.numel()
            descs[i].ndim = min(v.dim(), 4)
            shp = list(v.shape)[-4:] if v.dim() > 4 else list(v.shape)
            # collapse leading singleton dims of (1,1,12,128)-style parameters to at most 4 dims
            for d in range(4):
                descs[i].shape[d] = shp[d] if d < len(shp) else 1
        self._params, self._names = params, names
        self._check(self.lib.rift_model_load(self.ctx, descs, len(params), _stream()), "rift_model_load")
        return params

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, data: Dict, train=False, need_traj=False, fp32=False, no_drop=False, bn_update=True,
                seed: int = 0) -> Dict[str, torch.Tensor]:
        fb, keep = feature_batch(data, self.device)
        bs, A, R = fb.bs, fb.A, fb.R
        o = RiftOutputs()
        out = {"probability": torch.empty(bs, R, 12, device=self.device),
               "hidden": torch.empty(bs, 128, device=self.device)}
        if need_traj:
            out["trajectory"] = torch.empty(bs, R, 12, 80, 6, device=self.device)
            out["prediction"] = torch.empty(bs, max(A - 1, 0), 80, 6, device=self.device)
            out["ref_free_trajectory"] = torch.empty(bs, 80, 4, device=self.device)
        for k, v in out.items():
            setattr(o, k, v.data_ptr())
        flags = (F_TRAIN if train else 0) | (F_NEED_TRAJ if need_traj else 0) | (F_FP32 if fp32 else 0) | \
                (F_NO_DROP if no_drop else 0) | (0 if bn_update else F_NO_BN_UPDATE)
        self._check(self.lib.rift_forward(self.ctx, C.byref(fb), C.byref(o), flags, C.c_uint32(seed & 0xFFFFFFFF),
                                          _stream()), "rift_forward")
        self._keep = keep + list(out.values())
        self._bs = bs
        return out

    def forward_raw(self, fb: RiftFeatureBatch, out: RiftOutputs, flags: int, seed: int):
        """Hot-loop entry: prebuilt descriptors, no tensor conversion, no allocation."""
        rc = self.lib.rift_forward(self.ctx, C.byref(fb), C.byref(out), flags, C.c_uint32(seed & 0xFFFFFFFF), _stream())
        if rc != 0:
            self._check(rc, "rift_forward")
        self._bs = fb.bs

    def set_side_stream(self, stream: Optional["torch.cuda.Stream"]):
        """rift_set_side_stream: the forward's second chain runs on `stream` instead of a stream of the library's own (a host that places its
        streams on the hardware queues itself: RLFTTrainer); None returns to the library's."""
        self._side_stream = stream                     # keeps the torch stream object alive while the engine holds its handle
        rc = self.lib.rift_set_side_stream(self.ctx, C.c_void_p(stream.cuda_stream if stream is not None else 0))
        if rc != 0:
            self._check(rc, "rift_set_side_stream")

    def set_prepare_stream(self, stream: Optional["torch.cuda.Stream"]):
        """rift_set_prepare_stream: the input-only preparation of the following forwards runs on `stream` (behind the gather of the batch
        the caller queued there) instead of between two steps on the forward's stream; None switches it off."""
        self._prep_stream = stream                     # keeps the torch stream object alive while the engine holds its handle
        rc = self.lib.rift_set_prepare_stream(self.ctx, C.c_void_p(stream.cuda_stream if stream is not None else 0))
        if rc != 0:
            self._check(rc, "rift_set_prepare_stream")

    def forward_head(self, back: int = 0):
        """The policy head of the last F_DEFER_HEAD forward -- or of the one `back` forwards before it -- on the current stream (the caller
        has ordered it behind that forward)."""
        rc = self.lib.rift_forward_head_back(self.ctx, back, _stream())
        if rc != 0:
            self._check(rc, "rift_forward_head")

    def loss_backward_raw(self, kind: int, li: RiftLossIn, lo: RiftLossOut):
        rc = self.lib.rift_loss_backward(self.ctx, kind, C.byref(li), C.byref(lo), _stream())
        if rc != 0:
            self._check(rc, "rift_loss_backward")

    def loss_finalize_raw(self, lo: RiftLossOut, accumulate: int = 0):
        rc = self.lib.rift_loss_finalize(self.ctx, C.byref(lo), accumulate, _stream())
        if rc != 0:
            self._check(rc, "rift_loss_finalize")

    def check_finite(self):
        """Sync point of the reference's `assert torch.isfinite(q).all()` (planning_decoder.py:175): waits for the current stream and
        raises if any forward since the last check produced non-finite decoder queries."""
        self._check(self.lib.rift_check_finite(self.ctx, _stream()), "rift_forward (decoder finiteness check)")

    def set_dp(self, scene_offset: int, global_bs: int, xchg: torch.Tensor, exchange):
        """rift_set_dp: this rank's forwards process scenes [scene_offset, scene_offset + bs) of a `global_bs`-scene minibatch.
        `xchg`: caller-owned device f64 buffer; `exchange(t)`: in-place SUM all-reduce of a slice of it over the ranks, enqueued on the
        current stream.  The BatchNorm statistics and the r2r quirk masks of rift_forward then cover the global minibatch."""
        assert xchg.dtype == torch.float64 and xchg.is_cuda and xchg.is_contiguous()
        if getattr(self, "_dp_key", None) != (xchg.data_ptr(), id(exchange)):
            def _cb(user, offset, count, stream):
                try:
                    # the header's contract: the all-reduce is enqueued on the stream the engine names (it is torch's current stream
                    # unless the caller handed rift_forward another one)
                    if (stream or 0) != torch.cuda.current_stream(xchg.device).cuda_stream:
                        with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=xchg.device)):
                            exchange(xchg[offset:offset + count])
                    else:
                        exchange(xchg[offset:offset + count])
                    return 0
                except BaseException as e:          # an exception must not unwind through the C frames
                    self._dp_error = e
                    return -1
            self._dp_cb = EXCHANGE_FN(_cb)
            self._dp_key = (xchg.data_ptr(), id(exchange))
            self._dp_keep = (xchg, exchange)
        d = RiftDp()
        d.scene_offset, d.global_bs, d.xchg, d.xchg_len, d.exchange, d.user = scene_offset, global_bs, xchg.data_ptr(), xchg.numel(), self._dp_cb, None
        self._check(self.lib.rift_set_dp(self.ctx, C.byref(d)), "rift_set_dp")

    # ---- the library-owned communicator (rift_comm_*: RCCL through dlopen; hosts without torch.distributed) ---------------------------------
    def comm_unique_id(self) -> bytes:
        """128 bytes (an ncclUniqueId) from ONE rank; hand them to every rank out of band."""
        buf = C.create_string_buffer(128)
        self._check(self.lib.rift_comm_unique_id(self.ctx, C.cast(buf, vp)), "rift_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self.lib.rift_comm_init(self.ctx, C.cast(buf, vp), rank, world), "rift_comm_init")

    def comm_all_reduce(self, t: torch.Tensor):
        """In-place SUM all-reduce of a device f64 tensor over the library's communicator, on the current stream."""
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        self._check(self.lib.rift_comm_all_reduce(self.ctx, _ptr(t), t.numel(), _stream()), "rift_comm_all_reduce")

    def comm_destroy(self):
        self._check(self.lib.rift_comm_destroy(self.ctx), "rift_comm_destroy")

    def set_dp_library_comm(self, scene_offset: int, global_bs: int, xchg: torch.Tensor):
        """rift_set_dp with exchange = NULL: the forward's exchanges go over the communicator of comm_init()."""
        assert xchg.dtype == torch.float64 and xchg.is_cuda and xchg.is_contiguous()
        d = RiftDp()
        d.scene_offset, d.global_bs, d.xchg, d.xchg_len, d.exchange, d.user = scene_offset, global_bs, xchg.data_ptr(), xchg.numel(), EXCHANGE_FN(0), None
        self._dp_keep = (xchg, None)
        self._dp_key = None
        self._check(self.lib.rift_set_dp(self.ctx, C.byref(d)), "rift_set_dp")

    def clear_dp(self):
        self._check(self.lib.rift_set_dp(self.ctx, None), "rift_set_dp")
        self._dp_key = self._dp_cb = self._dp_keep = None

    def set_param_event(self, event: Optional["torch.cuda.Event"]):
        """rift_set_param_event: forwards wait for `event` (recorded after the optimizer step) right before they read pi_head."""
        self._param_event = event                                   # keep the handle alive
        self._check(self.lib.rift_set_param_event(self.ctx, C.c_void_p(event.cuda_event) if event is not None else None),
                    "rift_set_param_event")

    def loss_finalize_clip_raw(self, lo: RiftLossOut, accumulate: int, max_norm: float, total_norm: Optional[torch.Tensor]):
        """rift_loss_finalize + clip_grad_norm_ over the six pi_head gradients, one launch."""
        rc = self.lib.rift_loss_finalize_clip(self.ctx, C.byref(lo), accumulate, float(max_norm),
                                              _ptr(total_norm) if total_norm is not None else None, _stream())
        if rc != 0:
            self._check(rc, "rift_loss_finalize_clip")

    def make_adam_list(self, params, grads, exp_avg, exp_avg_sq, steps):
        """Host-side argument arrays of rift_adamw_step for a fixed list of tensors (built once, reused every step)."""
        n = len(params)
        arr = lambda ts: (vp * n)(*[t.data_ptr() for t in ts])
        return {"n": n, "p": arr(params), "g": arr(grads), "m": arr(exp_avg), "v": arr(exp_avg_sq), "s": arr(steps),
                "numel": (C.c_int64 * n)(*[t.numel() for t in params]), "keep": (params, grads, exp_avg, exp_avg_sq, steps)}

    def adamw_step_raw(self, al, lrs, wds, step_new: float, beta1: float, beta2: float, eps: float):
        n = al["n"]
        rc = self.lib.rift_adamw_step(self.ctx, n, al["p"], al["g"], al["m"], al["v"], al["s"], al["numel"],
                                      (C.c_double * n)(*lrs), (C.c_double * n)(*wds), float(step_new), float(beta1), float(beta2),
                                      float(eps), _stream())
        if rc != 0:
            self._check(rc, "rift_adamw_step")

    def update_tail_raw(self, lo: RiftLossOut, accumulate: int, max_norm: float, total_norm, al, lrs, wds, step_new: float, beta1: float,
                        beta2: float, eps: float):
        """rift_update_tail: finalize + gradient-norm clip + AdamW of the six pi_head tensors, one launch (al: make_adam_list of exactly those)."""
        n = al["n"]
        rc = self.lib.rift_update_tail(self.ctx, C.byref(lo), accumulate, float(max_norm), _ptr(total_norm) if total_norm is not None else None, n,
                                       al["p"], al["g"], al["m"], al["v"], al["s"], (C.c_double * n)(*lrs), (C.c_double * n)(*wds), float(step_new),
                                       float(beta1), float(beta2), float(eps), _stream())
        if rc != 0:
            self._check(rc, "rift_update_tail")

    def prof_enable(self, on: bool):
        self._check(self.lib.rift_prof_enable(self.ctx, 1 if on else 0), "rift_prof_enable")

    def prof_report(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 18)
        self._check(self.lib.rift_prof_report(self.ctx, buf, len(buf)), "rift_prof_report")
        return json.loads(buf.value.decode())

    def tap(self, name: str) -> torch.Tensor:
        n = C.c_int64(0)
        self._check(self.lib.rift_tap(self.ctx, name.encode(), None, C.byref(n), _stream()), "rift_tap")
        t = torch.empty(n.value, device=self.device)
        self._check(self.lib.rift_tap(self.ctx, name.encode(), _ptr(t), C.byref(n), _stream()), "rift_tap")
        return t

    # ---- loss + backward ---------------------------------------------------------------------
    def loss_backward(self, kind: str, batch: Dict[str, torch.Tensor], clip_epsilon=0.2, lambda_entropy=0.01):
        """Phase 1: per-rank sums.  Returns (stats[2] f64, flat_grad_sum[16897] f32, argmax_rm or None)."""
        dev = self.device
        li = RiftLossIn()
        keep = []

        def put(key, dt):
            if key not in batch or batch[key] is None:
                return None
            t = _dev(batch[key], dt, dev)
            keep.append(t)
            return t.data_ptr()

        li.old_group_logits = put("old_group_logits_torch", torch.float32)
        li.ref_group_logits = put("ref_group_logits_torch", torch.float32)
        li.group_advantage = put("group_advantage_torch", torch.float64)
        li.group_valid_mask = put("group_advantage_mask_torch", torch.bool)
        li.action_mode = put("action_mode_torch", torch.int64)
        li.advantage = put("advantage_torch", torch.float32)
        li.old_log_prob = put("old_log_prob_torch", torch.float32)
        li.returns = put("return_torch", torch.float32)
        li.clip_epsilon, li.lambda_entropy = clip_epsilon, lambda_entropy
        lo = RiftLossOut()
        stats = torch.zeros(2, dtype=torch.float64, device=dev)
        flat = torch.zeros(PI_NPARAM, dtype=torch.float32, device=dev)
        lo.stats, lo.flat_grad_sum = stats.data_ptr(), flat.data_ptr()
        argmax = None
        if kind in ("reinforce", "sft"):          # chosen (r, m): REINFORCE's argmax / SFT's target label
            argmax = torch.zeros(self._bs, 2, dtype=torch.int64, device=dev)
            lo.argmax_rm = argmax.data_ptr()
        self._check(self.lib.rift_loss_backward(self.ctx, LOSS_KINDS[kind], C.byref(li), C.byref(lo), _stream()),
                    "rift_loss_backward")
        self._keep_loss = keep
        return stats, flat, argmax

    def loss_finalize(self, stats, flat, grads: Dict[str, torch.Tensor], accumulate=False) -> torch.Tensor:
        """Phase 2: loss = -S/count and grads = -flat/count into the six pi_head .grad tensors
        (`grads` keyed by 'mlp.0.weight', 'mlp.0.bias', 'mlp.1.weight', 'mlp.1.bias', 'mlp.3.weight', 'mlp.3.bias')."""
        lo = RiftLossOut()
        loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        lo.loss, lo.stats, lo.flat_grad_sum = loss.data_ptr(), stats.data_ptr(), flat.data_ptr()
        lo.grad_w1, lo.grad_b1 = _ptr(grads.get("mlp.0.weight")), _ptr(grads.get("mlp.0.bias"))
        lo.grad_ln_w, lo.grad_ln_b = _ptr(grads.get("mlp.1.weight")), _ptr(grads.get("mlp.1.bias"))
        lo.grad_w2, lo.grad_b2 = _ptr(grads.get("mlp.3.weight")), _ptr(grads.get("mlp.3.bias"))
        self._check(self.lib.rift_loss_finalize(self.ctx, C.byref(lo), 1 if accumulate else 0, _stream()),
                    "rift_loss_finalize")
        return loss

    # ---- single ops ---------------------------------------------------------------------------
    def op_linear(self, x, w, b=None, ln_w=None, ln_b=None, act=0, fp32=False):
        x = _dev(x, torch.float32, self.device)
        w = _dev(w, torch.float32, self.device)
        b = None if b is None else _dev(b, torch.float32, self.device)
        ln_w = None if ln_w is None else _dev(ln_w, torch.float32, self.device)
        ln_b = None if ln_b is None else _dev(ln_b, torch.float32, self.device)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=self.device)
        self._check(self.lib.rift_op_linear(self.ctx, _ptr(x), M, K, _ptr(w), _ptr(b), N, _ptr(ln_w), _ptr(ln_b), act,
                                            1 if fp32 else 0, _ptr(y), _stream()), "rift_op_linear")
        return y

    def op_linear_bench(self, x, w, b=None, ln_w=None, ln_b=None, act=0, residual=None, reps=50):
        """Average microseconds per launch of the forward's GEMM kernel on this shape (diagnostic)."""
        dev = self.device
        x, w = _dev(x, torch.float32, dev), _dev(w, torch.float32, dev)
        opt = [None if t is None else _dev(t, torch.float32, dev) for t in (b, ln_w, ln_b, residual)]
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=dev)
        ms = C.c_float(0)
        self._check(self.lib.rift_op_linear_bench(self.ctx, _ptr(x), M, K, _ptr(w), _ptr(opt[0]), N, _ptr(opt[1]), _ptr(opt[2]),
                                                  act, _ptr(opt[3]), _ptr(y), reps, C.byref(ms), _stream()), "rift_op_linear_bench")
        return ms.value * 1e3

    def ref_line_info(self, trajectories, ref_pos_list, ref_angle_list, Ts=40):
        """TrajEvaluator.get_ref_line_info on device.  trajectories (R, M, Tfull, C>=6 or 4)."""
        dev = self.device
        traj = _dev(trajectories, torch.float32, dev)
        R, M, Tfull, Cc = traj.shape
        assert Cc == 6
        Pmax = max(p.shape[0] for p in ref_pos_list)
        if all(not (torch.is_tensor(p) and p.is_cuda) for p in ref_pos_list):
            # host-side lines (the rollout tick: straight from the observation): padded on the host, ONE upload -- [x, y, angle] planes + lengths
            host = np.zeros((R, Pmax, 3), dtype=np.float32)
            for r in range(R):
                n = ref_pos_list[r].shape[0]
                host[r, :n, :2] = np.asarray(ref_pos_list[r], dtype=np.float32)
                host[r, :n, 2] = np.asarray(ref_angle_list[r], dtype=np.float32)
            hp = np.concatenate([host[..., :2].reshape(-1), host[..., 2].reshape(-1)])      # [positions (R, Pmax, 2) | angles (R, Pmax)]
            up = self.stage(hp, torch.float32)
            rp, ra = up[:R * Pmax * 2].view(R, Pmax, 2), up[R * Pmax * 2:].view(R, Pmax)
        else:
            rp = torch.zeros(R, Pmax, 2, device=dev)
            ra = torch.zeros(R, Pmax, device=dev)
            for r in range(R):
                n = ref_pos_list[r].shape[0]
                rp[r, :n] = ref_pos_list[r].to(dev)
                ra[r, :n] = ref_angle_list[r].to(dev)
        rl = self.stage(np.array([p.shape[0] for p in ref_pos_list], dtype=np.int32), torch.int32)
        G = R * M
        dd = torch.empty(G, Ts, device=dev)
        da = torch.empty(G, Ts, device=dev)
        ci = torch.empty(G, Ts, dtype=torch.int32, device=dev)
        self._check(self.lib.rift_ref_line_info(self.ctx, _ptr(traj), G, Tfull, Ts, M, _ptr(rp), _ptr(ra), _ptr(rl), Pmax, _ptr(dd),
                                                _ptr(da), _ptr(ci), _stream()), "rift_ref_line_info")
        return dd, da, ci

    def new_pid_state(self, capacity: int):
        """Persistent state of the two BatchPIDTorch filters of TrackPropagate (zero-initialised, never reset)."""
        dev = self.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        return {"turn_buf": z(capacity, 20), "turn_ptr": z(capacity, dt=torch.int32), "turn_len": z(capacity, dt=torch.int32),
                "speed_buf": z(capacity, 20), "speed_ptr": z(capacity, dt=torch.int32), "speed_len": z(capacity, dt=torch.int32)}

    def rollout(self, trajectories, center_state, pid_state, g_per_group=None):
        """TrajEvaluator.get_center_rollout / TrackPropagate.propagate on device.
        trajectories (G, Tfull, 6); center_state (n_groups, 6) = x, y, heading, speed, width, length."""
        dev = self.device
        traj = _dev(trajectories, torch.float32, dev)
        G, Tfull, _ = traj.shape
        cs = self.stage(center_state, torch.float32).view(-1, 6)
        gper = g_per_group or G // cs.shape[0]
        io = RiftRolloutIO()
        io.trajectories, io.G, io.Tfull, io.G_per_group, io.center_state = traj.data_ptr(), G, Tfull, gper, cs.data_ptr()
        for k in ("turn_buf", "turn_ptr", "turn_len", "speed_buf", "speed_ptr", "speed_len"):
            assert pid_state[k].shape[0] >= G
            setattr(io, k, pid_state[k].data_ptr())
        out = {"center": torch.empty(G, 80, 2, device=dev), "vertices": torch.empty(G, 80, 4, 2, device=dev),
               "closest_index": torch.empty(G, 79, dtype=torch.int32, device=dev),
               "aim_idx": torch.empty(G, 79, dtype=torch.int32, device=dev)}
        for k in ("angle", "speed", "acc", "ang_vel", "ang_acc"):
            out[k] = torch.empty(G, 80, device=dev)
        for k, v in out.items():
            setattr(io, k, v.data_ptr())
        self._check(self.lib.rift_rollout(self.ctx, C.byref(io), _stream()), "rift_rollout")
        self._keep_ro = (traj, cs)
        return out

    def other_vehicle_rollout(self, steer, throttle, brake, speed, location, yaw_deg, extent, num_future_frames: int = 40,
                              near_lane_change: bool = True, bbox_inflation_ratio: float = 1.1):
        """get_other_vehicle_rollout on the device: per-actor arrays (see rift_hip.h) -> (N, T, 4, 2) float64 device tensor."""
        dev = self.device
        f64 = lambda a: np.asarray(a.cpu() if torch.is_tensor(a) else a, dtype=np.float64)    # noqa: E731
        act_h = np.stack([f64(steer), f64(throttle), f64(brake)], -1)
        N = act_h.shape[0]
        out = torch.empty(N, num_future_frames, 4, 2, dtype=torch.float64, device=dev)
        if N:
            # the five per-actor arrays in ONE upload: [controls (N,3) | speed (N) | location (N,3) | yaw (N) | extent (N,2)] as blocks of one buffer
            blocks = [act_h.reshape(-1), f64(speed).reshape(-1), f64(location).reshape(-1), f64(yaw_deg).reshape(-1), f64(extent).reshape(-1)]
            assert [b.size for b in blocks] == [3 * N, N, 3 * N, N, 2 * N], "per-actor arrays of different lengths"
            up = self.stage(np.concatenate(blocks), torch.float64)
            act, sp, loc, yaw, ext = up[:3 * N], up[3 * N:4 * N], up[4 * N:7 * N], up[7 * N:8 * N], up[8 * N:]
            self._check(self.lib.rift_other_vehicle_rollout(self.ctx, _ptr(act), _ptr(sp), _ptr(loc), _ptr(yaw), _ptr(ext), N, num_future_frames,
                                                            1 if near_lane_change else 0, float(bbox_inflation_ratio), _ptr(out), _stream()),
                        "rift_other_vehicle_rollout")
        return out

    def group_advantage_tick(self, trajectory, cbvs, pid_state, gamma: float = 0.98):
        """The group advantages of one tick's CBVs in one C-ABI call (rift_group_advantage_tick): `trajectory` (bs, Rb, 12, Tfull, 6) the tick's
        model output (device), `cbvs` a list of dicts in tick order --
            batch_index, center_state (6), ref_pos / ref_angle (lists of the valid lines' valid points, host), actors (None or the dict of
            nearby-actor readings of other_vehicle_rollout), off_road (None or (mask (H, W) uint8, (x, y, heading)))
        -- whose valid reference lines are a prefix of their rows.  Everything host-side goes up in ONE staged copy.  Returns the (K, Rb, 12) f64
        device tensor; rows r >= R of a CBV are zero."""
        dev = self.device
        traj = _dev(trajectory, torch.float32, dev)
        bs, Rb, M, Tfull, Cc = traj.shape
        assert M == 12 and Cc == 6
        K = len(cbvs)
        # one blob: float64 block | float32 block | int32 block | uint8 block, offsets in BYTES from the blob's start
        plan = []
        o64 = o32 = oi = o8 = 0
        b64, b32, bi, b8 = [], [], [], []
        for v in cbvs:
            R = len(v["ref_pos"])
            Pmax = max(int(p.shape[0]) for p in v["ref_pos"])
            rp = np.zeros((R, Pmax, 2), dtype=np.float32); ra = np.zeros((R, Pmax), dtype=np.float32)
            for r in range(R):
                n = v["ref_pos"][r].shape[0]
                rp[r, :n] = np.asarray(v["ref_pos"][r], dtype=np.float32); ra[r, :n] = np.asarray(v["ref_angle"][r], dtype=np.float32)
            rl = np.array([p.shape[0] for p in v["ref_pos"]], dtype=np.int32)
            cs = np.asarray(v["center_state"], dtype=np.float32).reshape(6)
            e = {"R": R, "Pmax": Pmax, "cs": o32, "rp": o32 + 6, "ra": o32 + 6 + rp.size, "rl": oi, "N": 0, "act": None, "mask": None}
            b32 += [cs, rp.reshape(-1), ra.reshape(-1)]; o32 += 6 + rp.size + ra.size
            pad = (-o32) % 4                     # keep every float32 piece 16-byte aligned
            if pad: b32.append(np.zeros(pad, dtype=np.float32)); o32 += pad
            bi.append(rl); oi += R
            a = v.get("actors")
            if a is not None:
                g = lambda x: np.asarray(x.cpu() if torch.is_tensor(x) else x, dtype=np.float64)      # noqa: E731
                act = np.stack([g(a["steer"]), g(a["throttle"]), g(a["brake"])], -1)
                N = act.shape[0]
                blocks = [act.reshape(-1), g(a["speed"]).reshape(-1), g(a["location"]).reshape(-1), g(a["yaw_deg"]).reshape(-1), g(a["extent"]).reshape(-1)]
                assert [b.size for b in blocks] == [3 * N, N, 3 * N, N, 2 * N], "per-actor arrays of different lengths"
                if N:
                    e["N"], e["act"] = N, o64
                    b64 += blocks; o64 += 10 * N
            m = v.get("off_road")
            if m is not None:
                mask = np.ascontiguousarray(m[0], dtype=np.uint8)
                e["mask"], e["H"], e["W"], e["pose"] = o8, mask.shape[0], mask.shape[1], [float(m[1][0]), float(m[1][1]), float(m[1][2])]
                b8.append(mask.reshape(-1)); o8 += mask.size
                pad = (-o8) % 16
                if pad: b8.append(np.zeros(pad, dtype=np.uint8)); o8 += pad
            plan.append(e)
        n64 = o64 * 8
        n32 = o32 * 4
        ni = oi * 4
        ni_pad = (-(n64 + n32 + ni)) % 16
        blob = np.concatenate([np.concatenate(b64).view(np.uint8) if b64 else np.zeros(0, np.uint8),
                               np.concatenate(b32).view(np.uint8), np.concatenate(bi).view(np.uint8), np.zeros(ni_pad, np.uint8)] + b8)
        up = self.stage(blob, torch.uint8)
        base = up.data_ptr()
        base32, basei, base8 = base + n64, base + n64 + n32, base + n64 + n32 + ni + ni_pad
        arr = (RiftTickCBV * K)()
        for k, (v, e) in enumerate(zip(cbvs, plan)):
            t = arr[k]
            t.batch_index, t.R, t.Pmax, t.n_actors = int(v["batch_index"]), e["R"], e["Pmax"], e["N"]
            t.center_state, t.ref_pos, t.ref_angle, t.ref_len = base32 + 4 * e["cs"], base32 + 4 * e["rp"], base32 + 4 * e["ra"], basei + 4 * e["rl"]
            t.actors = (base + 8 * e["act"]) if e["act"] is not None else None
            if e["mask"] is not None:
                t.off_road_mask, t.H, t.W = base8 + e["mask"], e["H"], e["W"]
                t.pose[0], t.pose[1], t.pose[2] = e["pose"]
        Gmax = 12 * max(e["R"] for e in plan)
        for k in ("turn_buf", "turn_ptr", "turn_len", "speed_buf", "speed_ptr", "speed_len"):
            assert pid_state[k].shape[0] >= Gmax
        out = torch.zeros(K, Rb, 12, dtype=torch.float64, device=dev)
        self._check(self.lib.rift_group_advantage_tick(self.ctx, _ptr(traj), Rb, Tfull, arr, K, _ptr(pid_state["turn_buf"]), _ptr(pid_state["turn_ptr"]),
                                                       _ptr(pid_state["turn_len"]), _ptr(pid_state["speed_buf"]), _ptr(pid_state["speed_ptr"]),
                                                       _ptr(pid_state["speed_len"]), float(gamma), _ptr(out), _stream()), "rift_group_advantage_tick")
        self._keep_tick = (traj, up)
        return out

    def sft_teacher_mode(self, trajectory, teacher_infos, frame_rate: int = 10):
        """generate_target_label's teacher side: trajectory (bs,R,M,T,6), teacher_infos (bs,5) -> (bs,2) int64 (r, m) of the candidate
        whose PID target speed is closest to the teacher's; feed it as `action_mode_torch` to loss_backward("sft")."""
        dev = self.device
        tr = _dev(trajectory, torch.float32, dev)
        ti = _dev(teacher_infos, torch.float32, dev)
        bs, R, M, T, _ = tr.shape
        out = torch.empty(bs, 2, dtype=torch.int64, device=dev)
        self._check(self.lib.rift_sft_teacher_mode(self.ctx, _ptr(tr), _ptr(ti), bs, R, M, T, frame_rate, _ptr(out), _stream()),
                    "rift_sft_teacher_mode")
        return out

    def collision_matrix(self, center_vertices, other_vertices, Ts: int = 40):
        """get_collision_matrix: envelope overlap of candidate footprints (G,Tc,4,2) with forecast neighbours (N,Ts,4,2) -> (G,Ts) bool."""
        dev = self.device
        cv = _dev(center_vertices, torch.float32, dev)
        G, Tc = cv.shape[:2]
        ov = _dev(torch.as_tensor(other_vertices), torch.float64, dev)
        N = ov.shape[0]
        assert N == 0 or (ov.shape[1] >= Ts and tuple(ov.shape[2:]) == (4, 2))
        if N and ov.shape[1] != Ts:
            ov = ov[:, :Ts].contiguous()
        out = torch.empty(G, Ts, dtype=torch.uint8, device=dev)
        self._check(self.lib.rift_collision_matrix(self.ctx, _ptr(cv), G, Tc, _ptr(ov) if N else None, N, Ts, _ptr(out), _stream()),
                    "rift_collision_matrix")
        return out.bool()

    def off_road_matrix(self, rollout_center, off_road_mask, origin, heading: float, resolution_hw=(0.5, -0.5), offset=(200.0, 200.0)):
        """get_off_road_matrix lookup: rollout_center (G,T,2), off_road_mask (H,W) uint8 with 1 = not drivable -> (G,T) bool."""
        dev = self.device
        rc = _dev(rollout_center, torch.float32, dev)
        G, T = rc.shape[:2]
        m = self.stage(off_road_mask, torch.uint8)
        H, W = m.shape
        out = torch.empty(G, T, dtype=torch.uint8, device=dev)
        self._check(self.lib.rift_off_road_matrix(self.ctx, _ptr(rc), G * T, _ptr(m), H, W, float(origin[0]), float(origin[1]), float(heading),
                                                  float(resolution_hw[0]), float(resolution_hw[1]), float(offset[0]), float(offset[1]),
                                                  _ptr(out), _stream()), "rift_off_road_matrix")
        return out.bool()

    def make_clip_list(self, grads):
        """Pre-build the (pointer, numel) arrays of a fixed list of .grad tensors for clip_grad_norm_raw."""
        n = len(grads)
        ptrs = (vp * n)(*[g.data_ptr() for g in grads])
        nums = (C.c_int64 * n)(*[g.numel() for g in grads])
        return (ptrs, nums, n, list(grads))

    def clip_grad_norm_raw(self, clip_list, max_norm: float, total_norm: Optional[torch.Tensor] = None):
        ptrs, nums, n, _ = clip_list
        rc = self.lib.rift_clip_grad_norm(self.ctx, ptrs, nums, n, max_norm, _ptr(total_norm), _stream())
        if rc != 0:
            self._check(rc, "rift_clip_grad_norm")

    # ---- PPO critic ----------------------------------------------------------------------------
    @staticmethod
    def critic_desc(sd: Dict[str, torch.Tensor]) -> "RiftCritic":
        """Views onto a CriticPPO state_dict (net.{0,2,4}.{weight,bias}, state_avg/std, value_avg/std; device fp32)."""
        w = RiftCritic()
        keep = []
        for f, k in (("w0", "net.0.weight"), ("b0", "net.0.bias"), ("w1", "net.2.weight"), ("b1", "net.2.bias"), ("w2", "net.4.weight"),
                     ("b2", "net.4.bias"), ("state_avg", "state_avg"), ("state_std", "state_std"), ("value_avg", "value_avg"),
                     ("value_std", "value_std")):
            t = sd[k]
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError(f"critic parameter {k} must be a contiguous fp32 device tensor")
            setattr(w, f, t.data_ptr())
            keep.append(t)
        w._keep = keep
        return w

    def critic_forward(self, sd: Dict[str, torch.Tensor], state: torch.Tensor) -> torch.Tensor:
        st = _dev(state, torch.float32, self.device)
        out = torch.empty(st.shape[0], dtype=torch.float32, device=self.device)
        w = self.critic_desc(sd)
        self._check(self.lib.rift_critic_forward(self.ctx, C.byref(w), _ptr(st), st.shape[0], _ptr(out), _stream()), "rift_critic_forward")
        return out

    def critic_loss_backward_raw(self, w: "RiftCritic", state: torch.Tensor, reward_sum: torch.Tensor, stats: torch.Tensor, flat: torch.Tensor):
        rc = self.lib.rift_critic_loss_backward(self.ctx, C.byref(w), _ptr(state), _ptr(reward_sum), state.shape[0], _ptr(stats), _ptr(flat), _stream())
        if rc != 0:
            self._check(rc, "rift_critic_loss_backward")

    def critic_finalize_raw(self, flat: torch.Tensor, stats: torch.Tensor, grads):
        rc = self.lib.rift_critic_finalize(self.ctx, _ptr(flat), _ptr(stats), *[_ptr(g) for g in grads], _stream())
        if rc != 0:
            self._check(rc, "rift_critic_finalize")

    def gae(self, rewards, undones, values, next_values, unterminated, gamma=0.98, lambda_=0.98):
        dev = self.device
        r = _dev(rewards, torch.float64, dev)
        a = [_dev(t, torch.float32, dev) for t in (undones, values, next_values, unterminated)]
        n = r.numel()
        out = torch.empty(n, dtype=torch.float32, device=dev)
        self._check(self.lib.rift_gae(self.ctx, _ptr(r), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), gamma, lambda_, n,
                                      _ptr(out), _stream()), "rift_gae")
        return out

    def discounted_return(self, rewards, dones, gamma=0.98):
        dev = self.device
        r = _dev(rewards, torch.float64, dev)
        d = _dev(dones, torch.float32, dev)
        out = torch.empty(r.numel(), dtype=torch.float64, device=dev)
        self._check(self.lib.rift_discounted_return(self.ctx, _ptr(r), _ptr(d), gamma, r.numel(), _ptr(out), _stream()),
                    "rift_discounted_return")
        return out

    def normalize_advantage_(self, x):
        assert x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
        self._check(self.lib.rift_normalize_advantage(self.ctx, _ptr(x), x.numel(), _stream()), "rift_normalize_advantage")
        return x

    def group_advantage(self, returns):
        r = _dev(returns, torch.float64, self.device)
        ng, G = r.shape
        out = torch.empty_like(r)
        self._check(self.lib.rift_group_advantage(self.ctx, _ptr(r), ng, G, _ptr(out), _stream()), "rift_group_advantage")
        return out

    def rollout_return(self, delta_dis, delta_angle, speed, acc, ang_vel, ang_acc, collision, off_road, gamma=0.98):
        dev = self.device
        f = [_dev(t, torch.float32, dev) for t in (delta_dis, delta_angle, speed, acc, ang_vel, ang_acc)]
        col, off = self.stage(collision, torch.bool), self.stage(off_road, torch.bool)
        G, Ts = f[1].shape
        out = torch.empty(G, dtype=torch.float64, device=dev)
        self._check(self.lib.rift_rollout_return(self.ctx, *[_ptr(t) for t in f], _ptr(col), col.shape[1], _ptr(off),
                                                 off.shape[1], G, Ts, gamma, _ptr(out), _stream()), "rift_rollout_return")
        return out
