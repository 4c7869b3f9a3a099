"""Shared test helpers: regenerate the seeded inputs/weights of a golden case and
check their digests against the fixture."""
import json
import os

import numpy as np
import torch

from rift_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

CASES = {
    "small": (list(range(100, 106)), 12, 8, 1, 4),
    "full": ([7, 8], 64, 20, 1, 6),
    # round 5 (gen_golden.py `shapes`): reference-generated fixtures at the shapes of the other kernel variants, and with static objects
    "dense": ([300, 301, 302], 128, 40, 8, 16),
    "carla": ([310, 311, 312], 49, 60, 1, 6),
    "static": ([320, 321, 322, 323], 12, 8, 1, 4, 5),
}
SHAPE_CASES = ("dense", "carla", "static")


def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


_sd_cache = {}


def weights():
    if "sd" not in _sd_cache:
        _sd_cache["sd"] = syn.perturbed_state_dict(manifest())
    return _sd_cache["sd"]


def build_batch(case):
    idx, A, Mp, r0, r1, *rest = CASES[case]
    return syn.collate_scenes([syn.make_scene(i, A, Mp, r0, r1, *rest) for i in idx])


def token_padding(data):
    """(bs, A + Mp + S) key-padding mask of the scene tokens (pluto_model.py:133-147)."""
    parts = [~data["agent"]["valid_mask"].any(-1), ~data["map"]["valid_mask"].any(-1)]
    if data["static_objects"]["valid_mask"].shape[1] > 0:
        parts.append(~data["static_objects"]["valid_mask"])
    return torch.cat(parts, dim=-1)


def load_case(case):
    gold = dict(np.load(os.path.join(GOLDEN, f"pluto_{case}.npz")))
    batch = build_batch(case)
    sd = weights()
    assert syn.digest(sd) == str(gold["weight_digest"]), "regenerated weights differ from the fixture's"
    assert syn.digest(syn.flatten_dict(batch)) == str(gold["input_digest"]), "regenerated inputs differ"
    return gold, batch, sd


def carla_magnitude_batch(n=3, seed0=8800, A=49, Mp=40):
    """Scenes at the magnitudes a CARLA rollout produces BEFORE PlutoFeature.normalize's map crop (the range study of the fp16 operand
    format): agent and map coordinates out to +-500 m (the generator's N(0, 30^2) / N(0, 40^2) m stretched and clipped), speeds up to
    40 m/s, speed limits U(0, 40), reference lines running 0.5 - 4 m per point (up to ~480 m), shapes of rift_pluto.yaml:35-36 (49 agents)."""
    scenes = []
    for i in range(n):
        s = syn.make_scene(seed0 + i, A, Mp, 1, 6)
        f = s["feature"]
        g = torch.Generator().manual_seed(seed0 + 100 + i)
        ag, mp, rl = f["agent"], f["map"], f["reference_line"]
        ag["position"] = (ag["position"] * (500.0 / 90.0)).clamp(-500, 500)
        ag["position"][0] -= ag["position"][0, -1:].clone()                          # the CBV itself stays at the origin of its frame
        sp = ag["velocity"].norm(dim=-1, keepdim=True).clamp_min(1e-3)
        ag["velocity"] = ag["velocity"] / sp * (sp * (40.0 / 15.0)).clamp(max=40.0)
        shift = mp["polygon_center"][:, None, None, :2] * (500.0 / 120.0 - 1.0)
        mp["point_position"] = (mp["point_position"] + shift).clamp(-500, 500)
        mp["polygon_center"][:, :2] = (mp["polygon_center"][:, :2] * (500.0 / 120.0)).clamp(-500, 500)
        mp["polygon_speed_limit"] = torch.rand(mp["polygon_speed_limit"].shape, generator=g) * 40.0
        step = 0.5 + 3.5 * torch.rand(rl["position"].shape[0], 1, 1, generator=g)
        rl["position"] = rl["position"] * step
        rl["vector"] = rl["vector"] * step
        f["current_state"][3:5] = torch.tensor([38.0, -3.0])                         # longitudinal / lateral speed of the CBV
        scenes.append(s)
    batch = syn.collate_scenes(scenes)
    batch["advantage_torch"] = torch.randn(n, generator=torch.Generator().manual_seed(seed0))
    return batch


def clone_tree(d):
    return {k: clone_tree(v) if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v)
            for k, v in d.items()}


def max_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def advantage_inputs(n=4096, G=48, Ts=40, seed=424242):
    """Seeded inputs shared by gen_golden and the tests (regenerated, not stored)."""
    g = torch.Generator().manual_seed(seed)
    r = {}
    r["rewards"] = torch.randn(n, generator=g, dtype=torch.float64)
    done = torch.rand(n, generator=g) < 0.05
    term = done & (torch.rand(n, generator=g) < 0.4)
    r["dones"] = done.float()
    r["undones"] = 1.0 - done.float()
    r["unterminated"] = 1.0 - term.float()
    r["values"] = torch.randn(n, generator=g)
    r["next_values"] = torch.randn(n, generator=g)
    f = lambda *s: torch.randn(*s, generator=g)
    r["delta_dis"] = (f(G, Ts) * 1.5).numpy()
    r["delta_angle"] = (f(G, Ts) * 0.8).numpy()
    r["speed"] = (f(G, Ts) * 6.0 + 5.0).numpy()
    r["acc"] = (f(G, Ts) * 3.0).numpy()
    r["ang_vel"] = (f(G, Ts) * 0.5).numpy()
    r["ang_acc"] = (f(G, Ts) * 3.0).numpy()
    r["collision"] = (torch.rand(G, Ts, generator=g) < 0.02).numpy()
    r["off_road"] = (torch.rand(G, 2 * Ts, generator=g) < 0.05).numpy()
    return r




def rollout_inputs(seed=777, R=3, M=12, T=80):
    """Seeded candidate trajectories (R, M, T, 6) = (x, y, cos, sin, vx, vy), ragged reference lines and the
    centre-vehicle state used by the rollout fixtures (regenerated, not stored)."""
    import math
    g = torch.Generator().manual_seed(seed)
    v = 2.0 + 10.0 * torch.rand(R, M, generator=g)
    kappa = torch.randn(R, M, generator=g) * 0.02
    t = torch.arange(T, dtype=torch.float32) * 0.1
    s = v[..., None] * t
    th = kappa[..., None] * s + torch.randn(R, M, 1, generator=g) * 0.05
    step = torch.stack([th.cos(), th.sin()], -1) * (v[..., None, None] * 0.1)
    pos = torch.cumsum(step, dim=2) + torch.randn(R, M, 1, 2, generator=g) * 0.2
    traj = torch.cat([pos, th.cos()[..., None], th.sin()[..., None], step * 10.0], dim=-1).float().contiguous()
    ref_pos, ref_ang = [], []
    for r in range(R):
        n = int(torch.randint(60, 121, (1,), generator=g))
        a = float(torch.randn((), generator=g) * 0.1)
        k = float(torch.randn((), generator=g) * 0.005)
        ss = torch.arange(n, dtype=torch.float32)
        ang = a + k * ss
        p = torch.cumsum(torch.stack([ang.cos(), ang.sin()], -1), dim=0) + torch.randn(1, 2, generator=g) * 0.5
        ref_pos.append(p.float().contiguous())
        ref_ang.append(ang.float().contiguous())
    state = {"pos": (10.0, -5.0), "heading": 0.3, "speed": 6.0, "width": 2.0, "length": 4.6}
    return traj, ref_pos, ref_ang, state


# ---- PPO critic fixture inputs (tests/golden/ppo_critic.npz was generated from these by gen_golden.gen_critic) ----
def critic_inputs(n=48, R=3, M=12, seed=515):
    """Seeded PPO minibatch: logits (n,R,M) with the last reference line of odd scenes padded, critic state (n,128)."""
    g = torch.Generator().manual_seed(seed)
    prob = torch.randn(n, R, M, generator=g)
    r_pad = torch.zeros(n, R, dtype=torch.bool)
    r_pad[1::2, R - 1] = True
    state = torch.randn(n, 128, generator=g)
    return {"probability": prob, "r_pad": r_pad, "state": state,
            "advantage": torch.randn(n, generator=g), "reward_sum": torch.randn(n, generator=g) * 2.0,
            "old_log_prob": -3.0 + 0.3 * torch.randn(n, generator=g),
            "action_mode": torch.stack([torch.randint(0, R - 1, (n,), generator=g), torch.randint(0, M, (n,), generator=g)], 1)}


def critic_weights(seed=77):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, (o, k) in zip((0, 2, 4), ((256, 128), (256, 256), (1, 256))):
        sd[f"net.{i}.weight"] = torch.randn(o, k, generator=g) / k ** 0.5
        sd[f"net.{i}.bias"] = 0.1 * torch.randn(o, generator=g)
    sd["state_avg"] = 0.2 * torch.randn(128, generator=g)
    sd["state_std"] = 0.5 + torch.rand(128, generator=g)
    sd["value_avg"] = torch.tensor([0.3])
    sd["value_std"] = torch.tensor([1.7])
    return sd


def inference_inputs(seed=909, R=4, M=12, T=80):
    """Seeded candidate set of one CBV for the rollout-side helpers (tests/golden/inference.npz)."""
    g = np.random.default_rng(seed)
    t = np.arange(1, T + 1)[None, None, :] * 0.1
    speed = g.uniform(2.0, 9.0, size=(R, M, 1))
    curv = g.uniform(-0.05, 0.05, size=(R, M, 1))
    heading = curv * speed * t
    cand = np.stack([speed * t * np.cos(heading * 0.5), speed * t * np.sin(heading * 0.5), heading], axis=-1)
    return {"candidates": cand.astype(np.float64), "probability": g.normal(size=(R, M)).astype(np.float32),
            "ref_free": np.stack([5.0 * t[0, 0], 0.02 * t[0, 0] ** 2, 0.008 * t[0, 0]], axis=-1).astype(np.float64),
            "origin": np.array([12.5, -3.25]), "angle": 0.6}


def other_vehicle_inputs(seed=4242, N=7):
    """Seeded nearby-actor states for get_other_vehicle_rollout (tests/golden/other_vehicles.npz): steer / throttle / brake of the
    last control, speed (m/s), CARLA location (x, y, z; left-handed), yaw in degrees, bounding-box half extents (x = length/2, y)."""
    g = np.random.default_rng(seed)
    brake = (g.random(N) < 0.3).astype(np.float64)
    return {"steer": g.uniform(-0.6, 0.6, N), "throttle": g.uniform(0.0, 0.9, N) * (1 - brake), "brake": brake,
            "speed": np.concatenate([[0.2, 0.9], g.uniform(1.5, 14.0, N - 2)]),        # two actors below the 1 m/s extent threshold
            "location": np.stack([g.normal(20, 15, N), g.normal(-5, 15, N), g.uniform(0, 0.3, N)], -1),
            "yaw_deg": g.uniform(-180, 180, N), "extent": np.stack([g.uniform(1.8, 2.6, N), g.uniform(0.8, 1.1, N)], -1)}


def sft_inputs(seed=31337, bs=6, R=4, M=12, T=80):
    """Seeded inputs of the SFT teacher objective (tests/golden/sft.npz): policy logits with the model's -1e6 on padded reference lines,
    candidate trajectories (padded lines hold values too, as the model emits them), ragged valid lines, teacher_infos (bs, 5) =
    target speed, origin x, y, heading, speed."""
    g = torch.Generator().manual_seed(seed)
    rcount = torch.tensor([1, 4, 2, 3, 4, 1])[:bs]
    rvalid = torch.arange(R)[None, :] < rcount[:, None]
    prob = torch.randn(bs, R, M, generator=g)
    prob = prob.masked_fill(~rvalid[..., None], -1e6)
    v = 1.0 + 11.0 * torch.rand(bs, R, M, generator=g)
    kappa = torch.randn(bs, R, M, generator=g) * 0.02
    t = torch.arange(1, T + 1, dtype=torch.float32) * 0.1
    th = kappa[..., None] * v[..., None] * t + torch.randn(bs, R, M, 1, generator=g) * 0.1
    pos = torch.cumsum(torch.stack([th.cos(), th.sin()], -1) * (v[..., None, None] * 0.1), dim=3) + torch.randn(bs, 1, 1, 1, 2, generator=g) * 20
    traj = torch.cat([pos, th.cos()[..., None], th.sin()[..., None], torch.zeros(bs, R, M, T, 2)], -1).float().contiguous()
    teacher = torch.stack([1.0 + 10.0 * torch.rand(bs, generator=g), pos[:, 0, 0, 0, 0], pos[:, 0, 0, 0, 1],
                           torch.rand(bs, generator=g) * 2 - 1, 3.0 + 5.0 * torch.rand(bs, generator=g)], -1).float()
    return {"probability": prob, "trajectory": traj, "ref_valid_mask": rvalid[..., None].expand(bs, R, 120).contiguous(),
            "r_pad": ~rvalid, "teacher_infos": teacher}


def rtr_inputs(seed=27183):
    """Seeded inputs of the RTR objective (tests/golden/rtr.npz): the SFT inputs plus PPO's per-scene extras (rtr_trainer.py:173-196)."""
    inp = sft_inputs()
    g = torch.Generator().manual_seed(seed)
    bs, R, M = inp["probability"].shape
    rcount = (~inp["r_pad"]).sum(1)
    inp.update({"state": torch.randn(bs, 128, generator=g), "advantage": torch.randn(bs, generator=g),
                "reward_sum": torch.randn(bs, generator=g) * 2.0, "old_log_prob": -3.0 + 0.3 * torch.randn(bs, generator=g),
                "action_mode": torch.stack([(torch.rand(bs, generator=g) * rcount).long().clamp(max=R - 1),
                                            torch.randint(0, M, (bs,), generator=g)], 1)})
    return inp


def collate_scenes_ragged():
    """Seeded scenes with ragged agent / polygon / reference-line counts (tests/golden/collate.npz = the reference's RIFTCollate on them)."""
    dims = [(9, 5, 2), (16, 10, 4), (3, 7, 1), (12, 2, 3), (16, 9, 4), (5, 10, 2), (1, 1, 1)]
    scenes = []
    for i, (A, Mp, R) in enumerate(dims):
        s = syn.make_scene(5000 + i, A, Mp, R, R)
        scenes.append({"feature": s["feature"], "extras": s["extras"]})
    return scenes


def raw_feature_inputs(seed=2718, A=9, Mp=14, R=3, T=26, S=2):
    """A raw feature dict as PlutoFeatureBuilder hands it to PlutoFeature.normalize: numpy float64 in the GLOBAL frame, 21 history + 5
    future agent steps, polygons partly beyond the +-radius crop, static objects, a route (tests/golden/normalize.npz)."""
    g = np.random.default_rng(seed)
    cx, cy, th = 312.5, -48.25, 0.83
    state = np.array([cx, cy, th, 4.2, 0.1, -0.03, 0.002], dtype=np.float64)
    far = g.random(Mp) < 0.3
    centre = np.stack([cx + g.normal(0, 40, Mp) + far * 400.0, cy + g.normal(0, 40, Mp)], -1)
    pts = centre[:, None, None, :] + g.normal(0, 6, (Mp, 3, 20, 2))
    return {
        "current_state": state,
        "agent": {"position": np.stack([cx, cy]) + g.normal(0, 30, (A, T, 2)), "heading": g.uniform(-3.14, 3.14, (A, T)),
                  "velocity": g.normal(0, 5, (A, T, 2)), "shape": g.uniform(1, 4, (A, T, 2)), "category": g.integers(0, 4, A).astype(np.int8),
                  "valid_mask": g.random((A, T)) < 0.85},
        "map": {"point_position": pts, "point_vector": g.normal(0, 1, (Mp, 3, 20, 2)), "point_orientation": g.uniform(-3.14, 3.14, (Mp, 3, 20)),
                "point_side": np.tile(np.arange(3, dtype=np.int8), (Mp, 1)), "polygon_center": np.concatenate([centre, g.uniform(-3.14, 3.14, (Mp, 1))], -1),
                "polygon_position": centre + g.normal(0, 1, (Mp, 2)), "polygon_orientation": g.uniform(-3.14, 3.14, Mp),
                "polygon_type": g.integers(0, 3, Mp).astype(np.int8), "polygon_on_route": g.random(Mp) < 0.5,
                "polygon_tl_status": g.integers(0, 4, Mp).astype(np.int8), "polygon_has_speed_limit": g.random(Mp) < 0.7,
                "polygon_speed_limit": g.uniform(0, 15, Mp), "polygon_road_block_id": g.integers(0, 99, Mp).astype(np.int32)},
        "static_objects": {"position": np.stack([cx, cy]) + g.normal(0, 20, (S, 2)), "heading": g.uniform(-3.14, 3.14, S),
                           "shape": g.uniform(0.5, 2, (S, 2)), "category": g.integers(0, 4, S).astype(np.int8), "valid_mask": np.ones(S, dtype=bool)},
        "route": {"position": np.stack([cx, cy]) + g.normal(0, 50, (30, 2))},
        "reference_line": {"position": np.stack([cx, cy]) + g.normal(0, 25, (R, 120, 2)), "vector": g.normal(0, 1, (R, 120, 2)),
                           "orientation": g.uniform(-3.14, 3.14, (R, 120)), "valid_mask": g.random((R, 120)) < 0.9,
                           "future_projection": g.normal(0, 1, (R, 8, 2))},
    }


BUFFER_KEYS = ['CBVs_obs', 'CBVs_actions', 'CBVs_actions_old_group_logits', 'CBVs_group_advantage', 'CBVs_next_obs', 'CBVs_reward',
               'CBVs_terminated', 'CBVs_done']          # planning/config/rift_pluto.yaml:8-16


def buffer_store_sequences(n_seq=120, seed=60221):
    """Seeded store() call sequences for CBVRolloutBuffer (shared by gen_golden.gen_buffer and the tests; regenerated, not stored).
    Every sequence = (capacity, [data_dict per store call]).  A data_dict is one env's rollout chunk as carla_runner.py:455-463 hands it
    over: per key a list over the chunk's steps of {cbv_id: value}, plus 'CBV_ids'.  Values are integer codes (sequence, cbv, step of
    that CBV) so that the stored ORDER is readable from the buffer.  The sequences cover episodes shorter than six steps (dropped),
    chunks that end mid-episode (staged across calls), several CBVs finishing in one chunk, exact fills and overflows."""
    rng = np.random.default_rng(seed)
    seqs = []
    for q in range(n_seq):
        capacity = int(rng.choice([16, 24, 40, 64]))
        alive, age, calls, next_id = {}, {}, [], 1
        fill_target = capacity + int(rng.integers(0, 30)) if q % 3 else capacity          # q % 3 == 0: aim at an exact fill
        stored = 0
        while stored < fill_target and len(calls) < 200:
            steps = int(rng.integers(1, 9))
            chunk = {k: [] for k in BUFFER_KEYS + ['CBV_ids']}
            for _ in range(steps):
                while len(alive) < int(rng.integers(1, 4)):
                    alive[next_id] = int(rng.choice([2, 3, 5, 6, 7, 9, 14, 23]))         # episode length of the new CBV
                    age[next_id] = 0
                    next_id += 1
                ids = sorted(alive)
                if q % 5 == 0:
                    ids = ids[::-1]                                                      # iteration order = order of 'CBV_ids'
                chunk['CBV_ids'].append(list(ids))
                row = {k: {} for k in BUFFER_KEYS}
                for c in ids:
                    age[c] += 1
                    done = age[c] >= alive[c]
                    code = (q * 1000 + c) * 100 + age[c]
                    for k in BUFFER_KEYS:
                        row[k][c] = done if k == 'CBVs_done' else (bool(done and c % 2) if k == 'CBVs_terminated' else code)
                    if done:
                        if alive[c] > 5:
                            stored += alive[c]
                        del alive[c], age[c]
                for k in BUFFER_KEYS:
                    chunk[k].append(row[k])
            calls.append(chunk)
        seqs.append((capacity, calls))
    return seqs


def traj_flag_kat():
    """tests/golden/traj_flags_kat.json as arrays: (centre footprints (G,1,4,2) f32, list of neighbour arrays (N_g,1,4,2) f64, expected (G,)),
    (raster mask (H,W) u8, per case (point (1,1,2) f32, origin, heading, expected))."""
    import json
    doc = json.load(open(os.path.join(GOLDEN, "traj_flags_kat.json")))
    col = [(c["name"], np.asarray(c["center"], dtype=np.float32).reshape(1, 1, 4, 2),
            np.asarray(c["others"], dtype=np.float64).reshape(-1, 1, 4, 2), bool(c["expect"])) for c in doc["collision"]]
    r = doc["raster"]
    mask = np.zeros((r["height"], r["width"]), dtype=np.uint8)
    for row, c_ in r["ones_row_col"]:
        mask[row, c_] = 1
    off = [(c["name"], np.asarray(c["point"], dtype=np.float32).reshape(1, 1, 2), tuple(c["origin"]), float(c["heading"]), bool(c["expect"]))
           for c in doc["off_road"]]
    return col, mask, off


def off_road_cases():
    """tests/golden/off_road.npz (tests/golden/gen_golden.py off_road: the reference's own get_off_road_matrix / global_to_pixel run on a
    mask that a cv2.fillPoly stand-in filled with a seeded pattern): [(name, mask (H,W) u8, points (G,T,2) f32, (x, y, heading), expected (G,T) bool)]."""
    g = np.load(os.path.join(GOLDEN, "off_road.npz"))
    return [(str(g[f"{i}.name"]), g[f"{i}.mask"], g[f"{i}.points"], tuple(float(v) for v in g[f"{i}.pose"]), g[f"{i}.off_road"])
            for i in range(int(g["n_case"]))]


# ---- recorded CARLA readings for the feature builder (tests/golden/feature_builder.npz) ---------------------------------------------------
def _ns(**kw):
    import types
    return types.SimpleNamespace(**kw)


def _pose(x, y, heading):
    return _ns(x=float(x), y=float(y), heading=float(heading), array=np.array([x, y], dtype=np.float64))


def _vec(x, y):
    return _ns(x=float(x), y=float(y), array=np.array([x, y], dtype=np.float64))


def _ring(xy):
    """Duck-typed shapely polygon: `.exterior.coords.xy` = (xs, ys)."""
    xy = np.asarray(xy, dtype=np.float64)
    return _ns(exterior=_ns(coords=_ns(xy=(xy[:, 0], xy[:, 1]))))


def bbox_fill(mask, vertices, value):
    """The deterministic raster fill both sides of the cost-map fixture use in place of OpenCV (absent here): the vertices' bounding box."""
    v = np.asarray(vertices).reshape(-1, 2)
    x0, x1 = max(int(v[:, 0].min()), 0), min(int(v[:, 0].max()), mask.shape[1] - 1)
    y0, y1 = max(int(v[:, 1].min()), 0), min(int(v[:, 1].max()), mask.shape[0] - 1)
    if x0 <= x1 and y0 <= y1:
        mask[y0:y1 + 1, x0:x1 + 1] = value


def feature_builder_world(case: str, agent_type=lambda n: n, layer=lambda n: n):
    """Seeded recorded readings of one CBV tick, as the reference's PlutoFeatureBuilder asks CarlaDataProvider / the map API / the route
    planner for them.  `agent_type` / `layer` turn the names 'VEHICLE', 'LANE', ... into the objects a side wants (the reference's
    enums for gen_golden, plain strings for the mirror).  Cases:
      'busy'  : 4 neighbours = max_agent (one pedestrian, one with a 7-step history, distances not in list order), 5 lanes + 2 lane
                connectors + 2 crosswalks (one lane far outside the +-radius crop, one repeated on-route road), 3 reference lines of
                61 / 500 / 9 route points, mode train_cbv (cost maps);
      'alone' : no neighbours, 2 lanes, one reference line of 3 points (nothing valid after the every-4th subsampling), mode eval."""
    g = np.random.default_rng({"busy": 9001, "alone": 9002}[case])
    T, P = 21, 20
    cx, cy, th = (212.25, -31.5, 0.7) if case == "busy" else (-40.0, 18.75, -2.1)

    def agent_history(n, x0, y0, kind, width, length):
        out = []
        for t in range(n):
            x, y, h = x0 + 0.4 * t + g.normal(0, 0.02), y0 + 0.1 * t + g.normal(0, 0.02), 0.3 + 0.01 * t
            box = _ns(width=width, length=length, geometry=_ring([[x - 1, y - 2], [x + 1, y - 2], [x + 1, y + 2], [x - 1, y + 2], [x - 1, y - 2]]))
            ag = _ns(center=_pose(x, y, h), velocity=_vec(g.normal(4, 1), g.normal(0, 0.3)), box=box, tracked_object_type=agent_type(kind))
            out.append(_ns(agent_state=ag, center=_pose(x, y, h)))
        return out

    def center_history():
        out = []
        for t in range(T + 4):                         # longer than needed: the builder keeps the last 21
            x, y, h = cx - 0.5 * (T + 3 - t), cy - 0.1 * (T + 3 - t), th + 0.004 * (t - T)
            dyn = _ns(rear_axle_velocity_2d=_vec(g.normal(5, 0.5), g.normal(0, 0.2)), rear_axle_acceleration_2d=_vec(g.normal(0, 0.4), g.normal(0, 0.1)),
                      angular_velocity=float(g.normal(0, 0.05)), speed=5.0, center_velocity_2d=_vec(5.0, 0.0))
            out.append(_ns(rear_axle=_pose(x, y, h), center=_pose(x + 1.4 * np.cos(h), y + 1.4 * np.sin(h), h), dynamic_car_state=dyn,
                           tire_steering_angle=float(g.normal(0, 0.03))))
        return out

    center = _ns(id=100, bounding_box=_ns(extent=_ns(x=2.45, y=1.05, z=0.8)))
    hist = {100: center_history()}
    nearby = []
    if case == "busy":
        for i, (dx, dy, kind, n) in enumerate([(30.0, 5.0, "VEHICLE", T + 2), (-6.0, 2.0, "PEDESTRIAN", T), (12.0, -9.0, "VEHICLE", 7), (3.0, 40.0, "BICYCLE", T)]):
            a = _ns(id=200 + i, bounding_box=_ns(extent=_ns(x=2.0, y=0.9, z=0.7)))
            nearby.append(a)
            hist[a.id] = agent_history(n, cx + dx, cy + dy, kind, 1.8 + 0.1 * i, 4.2 + 0.2 * i)

    def lane(token, ox, oy, road_id):
        base = np.stack([ox + np.linspace(0, 38, P + 1), oy + 0.02 * np.linspace(0, 38, P + 1) ** 1.5], -1) + g.normal(0, 0.01, (P + 1, 2))
        edges = np.stack([base, base + np.array([0.0, 1.75]), base - np.array([0.0, 1.75])])         # (3, P + 1, 2): centre, left, right
        ring = np.concatenate([edges[1], edges[2][::-1], edges[1][:1]])
        return _ns(token_id=str(token), centerline=base, edges=edges, road_id=str(road_id), polygon=_ring(ring), speed_limit_mps=None if token % 2 else 8.33)

    def crosswalk(token, ox, oy):
        base = np.stack([ox + np.zeros(P + 1), oy + np.linspace(0, 6, P + 1)], -1)
        return _ns(token_id=str(token), edges=np.stack([base, base + np.array([1.5, 0.0]), base - np.array([1.5, 0.0])]))

    if case == "busy":
        objects = {layer("LANE"): [lane(11, cx - 20, cy - 3, 7), lane(12, cx + 10, cy + 4, 8), lane(13, cx + 500, cy, 9), lane(14, cx - 60, cy + 30, 7),
                                   lane(15, cx - 5, cy - 40, 3)],
                   layer("LANE_CONNECTOR"): [lane(21, cx + 15, cy - 12, 8), lane(22, cx - 90, cy - 70, 5)],
                   layer("CROSSWALK"): [crosswalk(31, cx + 8, cy - 2), crosswalk(32, cx - 30, cy + 12)]}
        road_ids = [7, 8]
        lines = []
        for n, lat in ((61, 0.0), (500, 3.5), (9, -3.5)):
            s_ = np.arange(n) * 0.25
            lines.append(np.stack([cx + s_ * np.cos(th) - lat * np.sin(th), cy + s_ * np.sin(th) + lat * np.cos(th), th + 0.002 * s_], -1))
        mode = "train_cbv"
    else:
        objects = {layer("LANE"): [lane(41, cx - 10, cy - 1, 2), lane(42, cx + 5, cy + 6, 4)], layer("LANE_CONNECTOR"): [], layer("CROSSWALK"): []}
        road_ids = [4]
        lines = [np.array([[cx, cy, th], [cx + 0.2, cy, th], [cx + 0.4, cy, th]])]
        mode = "eval"
    map_api = _ns(map_sample_points=P, speed_limit_mps=8.33, query_proximal_map_data=lambda point, radius: objects)
    provider = _ns(get_history_state=lambda actor: hist[actor.id], get_current_state=lambda actor: hist[actor.id][-1],
                   get_frame_rate=lambda: 10, get_map_api=lambda: map_api)
    planner = _ns(build_reference_line=lambda c, st, r: (lines, ["route-elements"], {"road_ids": road_ids, "lane_ids": [1, 2]}, ["interaction-wp"]))
    config = {"obs": {"max_agent": 4, "radius": 120, "history_horizon": 2}}
    return _ns(config=config, provider=provider, planner=planner, center=center, nearby=nearby, mode=mode)
