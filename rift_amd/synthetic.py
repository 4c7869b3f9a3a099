"""Seeded synthetic replay scenes and policy weights.

There is no CARLA, no pretrained checkpoint and no dataset on the build / GPU
boxes, so parity tests, ``smoke()`` and ``bench.py`` all run on scenes and
weights produced here.  Everything is generated on the CPU with a
``torch.Generator`` seeded per scene (``SCENE_SEED0 + scene_idx``), so the same
tensors are reproduced bit-for-bit wherever the same torch build runs; the
golden fixtures carry SHA-256 digests to detect a drift.

Scene layout = the per-scene ``PlutoFeature.data`` dict of the reference
(pluto/feature_builder/pluto_feature.py:18-126, schema in SURVEY.md Appendix A),
i.e. what ``PlutoFeature.collate`` consumes.
"""
import hashlib
import math
from typing import Dict, List, Optional

import torch

SCENE_SEED0 = 20250515
WEIGHT_SEED0 = 7_2025_0515
NUM_MODES = 12
HIST_STEPS = 21
REF_POINTS = 120
MAP_POINTS = 20


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def make_scene(scene_idx: int, num_agents: int = 64, num_polygons: int = 20,
               r_min: int = 1, r_max: int = 6, max_static: int = 0) -> Dict:
    """One replay entry: feature dict + RIFT/GRPO/PPO/REINFORCE extras.
    max_static > 0: S ~ U{1..max_static} static objects (static_objects_encoder.py:17-40; CARLA produces none, nuPlan scenes do), drawn from
    a generator of their own so that every other tensor of the scene is the one the S = 0 scene has."""
    g = _gen(SCENE_SEED0 + scene_idx)
    A, Mp, T = num_agents, num_polygons, HIST_STEPS

    def randn(*s):
        return torch.randn(*s, generator=g)

    def rand(*s):
        return torch.rand(*s, generator=g)

    def randint(lo, hi, s):
        return torch.randint(lo, hi, s, generator=g)

    # ---- agents: constant-velocity tracks with a little jitter, ego (index 0) at the origin
    base = randn(A, 2) * 30.0
    head0 = (rand(A) * 2 - 1) * math.pi
    speed = randn(A).abs() * 5.0
    base[0] = 0.0
    head0[0] = 0.0
    t_rel = (torch.arange(T, dtype=torch.float32) - (T - 1)) * 0.1
    yaw_rate = randn(A) * 0.1
    heading = head0[:, None] + yaw_rate[:, None] * t_rel[None, :] + randn(A, T) * 0.01
    vel = torch.stack([heading.cos(), heading.sin()], -1) * speed[:, None, None] + randn(A, T, 2) * 0.05
    position = base[:, None, :] + torch.cumsum(vel * 0.1, dim=1)
    position = position - position[:, -1:, :] + base[:, None, :]
    shape = (1.0 + 3.0 * rand(A, 1, 2)).repeat(1, T, 1).contiguous()
    category = randint(0, 4, (A,)).to(torch.int8)
    valid = rand(A, T) < 0.9
    dead = rand(A) < 0.1
    valid[dead] = False
    valid[0] = True
    agent = {
        "position": position.float(), "heading": heading.float(), "velocity": vel.float(),
        "shape": shape.float(), "category": category, "valid_mask": valid,
        "target": torch.zeros(A, 0, 3),
    }

    # ---- map: short lane segments (centre line + two boundaries)
    c = randn(Mp, 2) * 40.0
    ang = (rand(Mp) * 2 - 1) * math.pi
    curv = randn(Mp) * 0.02
    s = torch.arange(MAP_POINTS, dtype=torch.float32) - (MAP_POINTS - 1) / 2
    th = ang[:, None] + curv[:, None] * s[None, :]
    step = torch.stack([th.cos(), th.sin()], -1) * (0.8 + 0.4 * rand(Mp, 1, 1))
    centre = c[:, None, :] + torch.cumsum(step, dim=1) - step.sum(1, keepdim=True) / 2
    normal = torch.stack([-th.sin(), th.cos()], -1)
    half_w = 1.5 + rand(Mp, 1, 1)
    pts = torch.stack([centre, centre + normal * half_w, centre - normal * half_w], dim=1)  # (Mp,3,20,2)
    vec = torch.cat([pts[:, :, 1:] - pts[:, :, :-1], pts[:, :, -1:] - pts[:, :, -2:-1]], dim=2)
    orient = th[:, None, :].repeat(1, 3, 1) + randn(Mp, 3, MAP_POINTS) * 0.01
    mvalid = rand(Mp, MAP_POINTS) < 0.95
    mvalid[:, 0] = True
    mid = MAP_POINTS // 2
    map_ = {
        "point_position": pts.float().contiguous(), "point_vector": vec.float().contiguous(),
        "point_orientation": orient.float().contiguous(),
        "point_side": torch.arange(3, dtype=torch.int8)[None].repeat(Mp, 1),
        "polygon_center": torch.cat([centre[:, mid], th[:, mid, None]], -1).float(),
        "polygon_position": centre[:, 0].float().contiguous(),
        "polygon_orientation": th[:, 0].float().contiguous(),
        "polygon_type": randint(0, 3, (Mp,)).to(torch.int8),
        "polygon_on_route": rand(Mp) < 0.5,
        "polygon_tl_status": randint(0, 4, (Mp,)).to(torch.int8),
        "polygon_has_speed_limit": rand(Mp) < 0.7,
        "polygon_speed_limit": (rand(Mp) * 15.0).float(),
        "polygon_road_block_id": randint(0, 1000, (Mp,)).to(torch.int32),
        "valid_mask": mvalid,
    }
    map_["polygon_speed_limit"][~map_["polygon_has_speed_limit"]] = 0.0

    # ---- reference lines: R ragged, valid prefix of the 120 points
    R = int(randint(r_min, r_max + 1, (1,)).item())
    plen = randint(30, REF_POINTS + 1, (R,))
    a0 = randn(R) * 0.3
    k0 = randn(R) * 0.01
    sidx = torch.arange(REF_POINTS, dtype=torch.float32)
    rth = a0[:, None] + k0[:, None] * sidx[None, :]
    rstep = torch.stack([rth.cos(), rth.sin()], -1) * 1.0
    rpos = randn(R, 1, 2) * 2.0 + torch.cumsum(rstep, dim=1)
    rvec = torch.cat([rpos[:, 1:] - rpos[:, :-1], rpos[:, -1:] - rpos[:, -2:-1]], dim=1)
    rvalid = sidx[None, :] < plen[:, None].float()
    rpos = torch.where(rvalid[..., None], rpos, torch.zeros_like(rpos))
    rvec = torch.where(rvalid[..., None], rvec, torch.zeros_like(rvec))
    rori = torch.where(rvalid, rth, torch.zeros_like(rth))
    ref = {
        "position": rpos.float().contiguous(), "vector": rvec.float().contiguous(),
        "orientation": rori.float().contiguous(), "valid_mask": rvalid,
        "future_projection": torch.zeros(R, 8, 2),
    }
    static = {
        "position": torch.zeros(0, 2), "heading": torch.zeros(0), "shape": torch.zeros(0, 2),
        "category": torch.zeros(0, dtype=torch.int8), "valid_mask": torch.zeros(0, dtype=torch.bool),
    }
    if max_static > 0:
        gs = _gen(SCENE_SEED0 + 1_000_003 + scene_idx)
        S = int(torch.randint(1, max_static + 1, (1,), generator=gs).item())
        svalid = torch.rand(S, generator=gs) < 0.8
        svalid[0] = True
        static = {
            "position": (torch.randn(S, 2, generator=gs) * 30.0).float(), "heading": ((torch.rand(S, generator=gs) * 4 - 2) * math.pi).float(),   # beyond +-pi: the encoder wraps it
            "shape": (0.5 + 2.5 * torch.rand(S, 2, generator=gs)).float(), "category": torch.randint(0, 4, (S,), generator=gs).to(torch.int8),
            "valid_mask": svalid,
        }
    cur = randn(7).float()
    cur[:3] = 0.0
    feature = {
        "agent": agent, "map": map_, "reference_line": ref, "static_objects": static,
        "current_state": cur, "origin": randn(2).float() * 100.0, "angle": (rand(()) * 2 - 1) * math.pi,
    }

    # ---- replay extras (rift_datamodule.py:33-49, ppo_datamodule.py:55-68, reinforce_datamodule.py)
    adv = torch.randn(R, NUM_MODES, generator=g, dtype=torch.float64)
    adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-5)
    rmask = torch.ones(R, NUM_MODES, dtype=torch.bool)
    extras = {
        "group_advantage": adv, "group_advantage_mask": rmask,
        "old_group_logits": randn(R, NUM_MODES).float(), "old_group_logits_mask": rmask.clone(),
        "ref_group_logits": randn(R, NUM_MODES).float(),
        "reward": float(randn(()).item()), "done": bool(rand(()) < 0.05), "terminated": bool(rand(()) < 0.02),
        "old_log_prob": float(-rand(()).item() * 4.0),
        "action_mode": torch.tensor([int(randint(0, R, (1,)).item()), int(randint(0, NUM_MODES, (1,)).item())]),
        "return": float(randn(()).item()),
    }
    return {"feature": feature, "extras": extras}


PAD_KEYS = ("agent", "map", "reference_line", "static_objects")
STACK_KEYS = ("current_state", "origin", "angle")


def collate_features(features: List[Dict]) -> Dict:
    """CPU collation with the semantics of PlutoFeature.collate
    (pluto_feature.py:83-94): zero-pad dim 0 of every tensor of the pad groups
    to the batch maximum, stack the rest."""
    from torch.nn.utils.rnn import pad_sequence
    out = {}
    for key in PAD_KEYS:
        if key in features[0]:
            out[key] = {k: pad_sequence([f[key][k] for f in features], batch_first=True)
                        for k in features[0][key].keys()}
    for key in STACK_KEYS:
        out[key] = torch.stack([f[key] for f in features], dim=0)
    return out


def collate_scenes(scenes: List[Dict]) -> Dict:
    """Batch dict in the layout RIFTCollate produces (rift_datamodule.py:20-51),
    plus the GRPO/PPO/REINFORCE extras."""
    from torch.nn.utils.rnn import pad_sequence
    ex = [s["extras"] for s in scenes]

    def pad(k):
        return pad_sequence([e[k] for e in ex], batch_first=True)

    return {
        "cur_pluto_feature_torch": collate_features([s["feature"] for s in scenes]),
        "group_advantage_torch": pad("group_advantage"),
        "group_advantage_mask_torch": pad("group_advantage_mask"),
        "old_group_logits_torch": pad("old_group_logits"),
        "old_group_logits_mask_torch": pad("old_group_logits_mask"),
        "ref_group_logits_torch": pad("ref_group_logits"),
        "action_mode_torch": torch.stack([e["action_mode"] for e in ex]),
        "old_log_prob_torch": torch.tensor([e["old_log_prob"] for e in ex], dtype=torch.float32),
        "return_torch": torch.tensor([e["return"] for e in ex], dtype=torch.float32),
    }


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def perturbed_state_dict(manifest: Dict[str, List[int]], seed: int = WEIGHT_SEED0) -> Dict[str, torch.Tensor]:
    """Deterministic non-trivial values for every tensor of the reference
    ``PlanningModel.state_dict()`` (names/shapes given by `manifest`).

    Follows the scale of ``PlanningModel._init_weights`` (pluto_model.py:108-120:
    xavier-uniform matrices, N(0, 0.02) embeddings) but makes biases, norm
    affine terms, BatchNorm running statistics and rpb non-trivial so that a
    parity test exercises every term."""
    sd = {}
    for i, (name, shape) in enumerate(manifest.items()):
        g = _gen(seed + i)
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.tensor(3, dtype=torch.long)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "rpb":
            t = torch.randn(shape, generator=g) * 0.2
        elif leaf == "bias" or leaf == "in_proj_bias":
            t = torch.randn(shape, generator=g) * 0.02
        elif len(shape) == 1:  # LayerNorm / BatchNorm weight
            t = 1.0 + torch.randn(shape, generator=g) * 0.05
        elif "emb" in name.split(".")[-2] and leaf == "weight" and len(shape) == 2 and shape[0] <= 4:
            t = torch.randn(shape, generator=g) * 0.02  # nn.Embedding tables
        elif leaf in ("m_emb", "m_pos", "pos_embed", "query"):
            t = torch.randn(shape, generator=g) * 0.02
        elif leaf == "weight" and name.endswith("freqs.weight"):
            t = torch.randn(shape, generator=g) * 0.02  # Fourier frequency table (nn.Embedding)
        else:  # Linear / Conv1d / in_proj matrices: xavier-uniform bound
            fan_out = shape[0] * (shape[2] if len(shape) == 3 else 1)
            fan_in = shape[1] * (shape[2] if len(shape) == 3 else 1)
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        sd[name] = t.float() if t.dtype != torch.long else t
    return sd


def digest(tensors: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for k in sorted(tensors):
        t = tensors[k]
        h.update(k.encode())
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def flatten_dict(d: Dict, prefix: str = "") -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten_dict(v, prefix + k + "."))
        elif isinstance(v, torch.Tensor):
            out[prefix + k] = v
    return out
