"""Generate the golden fixtures by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference):

    python -m tests.golden.gen_golden

Writes small ``.npz`` fixtures (inputs are regenerated from seeds by
``rift_amd.synthetic``; fixtures hold digests of the regenerated inputs /
weights plus the reference's outputs) and ``state_dict_manifest.json`` (names and
shapes of the reference ``PlanningModel.state_dict()``, a data description).
No reference source text is stored.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from rift_amd import synthetic as syn  # noqa: E402
from tests.golden import ref_loader  # noqa: E402

CASES = {
    # name: (scene indices, num_agents, num_polygons, r_min, r_max[, max_static])
    "small": (list(range(100, 106)), 12, 8, 1, 4),
    "full": ([7, 8], 64, 20, 1, 6),
    # round 5 (`python -m tests.golden.gen_golden shapes`): the shapes that run the OTHER kernel variants, pinned to the reference itself --
    # BASELINE configs[4] (168 token slots, R up to 16: enc_w_kernel / dec_w_kernel<., true>), the shapes train_cbv produces (rift_pluto.yaml:35-36:
    # 49 agent slots + 60 polygons = 109 token slots) and scenes WITH static objects (static_objects_encoder.py:17-40; S = 0 everywhere else)
    "dense": ([300, 301, 302], 128, 40, 8, 16),
    "carla": ([310, 311, 312], 49, 60, 1, 6),
    "static": ([320, 321, 322, 323], 12, 8, 1, 4, 5),
}
SHAPE_CASES = ("dense", "carla", "static")
# what the shape cases keep (the dense trajectory tensors alone would be megabytes): the policy's outputs the losses read, the hooks that localise
# a disagreement, the RIFT loss and its pi_head gradients, the train-mode BatchNorm forward
SHAPE_KEYS = ("eval.probability", "eval.hidden", "eval.ref_free_trajectory", "eval.tap.x_agent", "eval.tap.x_polygon", "eval.tap.enc_out", "eval.tap.q_final",
              "trainbn.probability", "trainbn.hidden", "weight_digest", "input_digest")


def build_batch(case):
    idx, A, Mp, r0, r1, *rest = CASES[case]
    scenes = [syn.make_scene(i, A, Mp, r0, r1, *rest) for i in idx]
    return syn.collate_scenes(scenes)


def clone_data(d):
    return {k: clone_data(v) if isinstance(v, dict) else v.clone() for k, v in d.items()}


def to_np(t):
    return t.detach().cpu().numpy()


def main(cases=("small", "full")):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_loader.install()
    model = ref_loader.planning_model()
    manifest = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    sd = syn.perturbed_state_dict(manifest)
    model.load_state_dict(sd, strict=True)
    weight_digest = syn.digest(sd)

    rt = ref_loader.rift_trainer_module()
    grpo_t = importlib.import_module("rift.cbv.planning.fine_tuner.rlft.grpo_pluto.grpo_trainer")
    reinf_t = importlib.import_module("rift.cbv.planning.fine_tuner.rlft.reinforce_pluto.reinforce_trainer")
    # ppo_trainer imports hydra-bound PPOPlutoModel; its loss needs only these attributes:
    import torch.nn as nn

    for case in cases:
        batch = build_batch(case)
        data = batch["cur_pluto_feature_torch"]
        out = {"weight_digest": weight_digest, "input_digest": syn.digest(syn.flatten_dict(batch))}

        # ---------------- eval-mode forward (reference) with module taps
        model.eval()
        taps = {}
        hooks = []

        def tap(name):
            def fn(mod, inp, res):
                taps[name] = res
            return fn

        hooks.append(model.agent_encoder.register_forward_hook(tap("x_agent")))
        hooks.append(model.map_encoder.register_forward_hook(tap("x_polygon")))
        hooks.append(model.norm.register_forward_hook(tap("enc_out")))
        hooks.append(model.planning_decoder.cat_x_proj.register_forward_hook(tap("q_final")))
        for i, blk in enumerate(model.encoder_blocks):
            hooks.append(blk.register_forward_hook(tap(f"enc{i}")))
        for i, blk in enumerate(model.planning_decoder.decoder_blocks):
            hooks.append(blk.register_forward_hook(tap(f"dec{i}")))
        with torch.no_grad():
            res = model(clone_data(data))
        for h in hooks:
            h.remove()
        for k in ("probability", "hidden", "trajectory", "prediction", "ref_free_trajectory",
                  "output_trajectory", "output_prediction", "candidate_trajectories",
                  "output_ref_free_trajectory"):
            out["eval." + k] = to_np(res[k])
        for k, v in taps.items():
            out["eval.tap." + k] = to_np(v)

        # ---------------- RIFT loss + pi_head grads through the reference trainer (eval-mode trunk)
        trainer = rt.LightningTrainer(model, lr=1e-4, cl_lr_decay=0.9, weight_decay=1e-5, epochs=16,
                                      warmup_epochs=3, frame_rate=10,
                                      trainable_layers=["planning_decoder.pi_head"],
                                      use_drivable_area_loss=False)
        trainer.eval()
        n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
        assert n_train == 16897, n_train
        for kind in ("rift", "grpo", "reinforce", "ppo"):
            model.zero_grad(set_to_none=True)
            b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            cur = clone_data(data)
            res = model(cur)
            prob = res["probability"]
            if kind == "rift":
                loss = rt.LightningTrainer.get_rift_loss(trainer, prob, cur, b)
            elif kind == "grpo":
                loss = grpo_t.LightningTrainer.get_grpo_loss(trainer, prob, cur, b)
            elif kind == "reinforce":
                r_pad = ~cur["reference_line"]["valid_mask"].any(-1)
                prob.masked_fill_(r_pad.unsqueeze(-1), -1e8)  # reinforce_trainer.py:126
                max_idx = torch.argmax(prob.view(prob.shape[0], -1), dim=1)
                r_idx, m_idx = max_idx // prob.size(-1), max_idx % prob.size(-1)
                loss = reinf_t.LightningTrainer.get_reinforce_loss(trainer, prob, r_idx, m_idx, b)
                out["reinforce.r_idx"], out["reinforce.m_idx"] = to_np(r_idx), to_np(m_idx)
            else:
                # ppo_trainer.py:161-183 with a zero value loss (value net handled separately)
                ns = types.SimpleNamespace(clip_epsilon=0.2, lambda_entropy=0.01,
                                           value_criterion=nn.SmoothL1Loss(),
                                           model=types.SimpleNamespace(value_net=lambda s: torch.zeros(s.shape[0])))
                r_pad = ~cur["reference_line"]["valid_mask"].any(-1)
                prob.masked_fill_(r_pad.unsqueeze(-1), -1e8)  # ppo_trainer.py:133
                bs = prob.shape[0]
                g = torch.Generator().manual_seed(99)
                b["state_torch"] = torch.zeros(bs, 128)
                b["advantage_torch"] = torch.randn(bs, generator=g)
                b["reward_sum_torch"] = torch.zeros(bs)
                out["ppo.advantage"] = to_np(b["advantage_torch"])
                ppo_mod = _load_ppo_loss_fn()
                loss = ppo_mod(ns, prob, b["action_mode_torch"], b)
            loss.backward()
            out[f"{kind}.loss"] = to_np(loss.detach().double())
            for n, p in model.planning_decoder.pi_head.named_parameters():
                out[f"{kind}.grad.{n}"] = to_np(p.grad)

        # ---------------- train mode with every drop probability 0 (BatchNorm batch statistics)
        m2 = ref_loader.planning_model(drop_path=0.0, dropout=0.0, state_dropout=0.0)
        m2.load_state_dict(sd, strict=True)
        m2.train()
        with torch.no_grad():
            res2 = m2(clone_data(data))
        out["trainbn.probability"] = to_np(res2["probability"])
        out["trainbn.hidden"] = to_np(res2["hidden"])
        for k, v in m2.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out["trainbn.stat." + k] = to_np(v)

        if case in SHAPE_CASES:
            out = {k: v for k, v in out.items() if k in SHAPE_KEYS or k.startswith("rift.") or k == "eval.trajectory" and case != "dense"}
        path = os.path.join(HERE, f"pluto_{case}.npz")
        np.savez_compressed(path, **out)
        print(case, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB",
              {k: float(out[k + '.loss']) for k in ('rift', 'grpo', 'reinforce', 'ppo') if k + '.loss' in out})


def _load_ppo_loss_fn():
    """Return the reference's `get_ppo_loss` as a function object.  Its module imports
    hydra/carla-bound policies at the top, so only that one FunctionDef is compiled,
    in memory, from the reference file where it lies (nothing is copied to disk)."""
    import ast
    import torch.nn.functional as F
    path = os.path.join(ref_loader.REF_ROOT, "rift/cbv/planning/fine_tuner/rlft/ppo_pluto/ppo_trainer.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "get_ppo_loss")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "F": F}
    exec(compile(mod, path, "exec"), ns)
    return ns["get_ppo_loss"]




# ------------------------------------------------------------------------------------------
# advantage / return fixtures (GAE, discounted return, dense-reward rollout return, z-score)
# ------------------------------------------------------------------------------------------
def _ref_function(rel_path, name, extra_ns=None):
    """Compile ONE function of a reference file in memory (its module top imports carla/hydra-bound code)."""
    import ast
    path = os.path.join(ref_loader.REF_ROOT, rel_path)
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch, "np": np}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def gen_advantage():
    ref_loader.install()
    import types as _t
    from tests.helpers import advantage_inputs
    inp = advantage_inputs()
    gae = _ref_function("rift/cbv/planning/fine_tuner/rlft/ppo_pluto/ppo_datamodule.py", "get_advantages_GAE")
    cret = _ref_function("rift/cbv/planning/fine_tuner/rlft/reinforce_pluto/reinforce_datamodule.py", "compute_return")
    rr = _ref_function("rift/cbv/planning/fine_tuner/rlft/traj_eval/traj_evaluator.py", "get_rollout_return")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_reward_model", os.path.join(ref_loader.REF_ROOT, "rift/gym_carla/reward/reward_model.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    out = {}
    adv = gae(inp["rewards"], inp["undones"], inp["values"], inp["next_values"], inp["unterminated"], 0.98, 0.98)
    out["gae"] = adv.numpy()
    out["gae_normalized"] = ((adv - adv.mean()) / (adv.std(dim=0) + 1e-5)).numpy()   # ppo_datamodule.py:166
    out["returns"] = cret(inp["rewards"], inp["dones"], 0.98).numpy()
    fake_self = _t.SimpleNamespace(reward_model=rm.DenseRewardModel())
    ret = rr(fake_self, inp["delta_dis"], inp["delta_angle"], inp["speed"], inp["acc"], inp["ang_vel"], inp["ang_acc"],
             inp["collision"], inp["off_road"])
    out["rollout_return"] = ret
    out["group_advantage"] = (ret - np.mean(ret)) / (np.std(ret) + 1e-5)            # traj_evaluator.py:467-470
    # WarmupCosLR table (warmup_cos_lr.py:39-54) for the RLFT config: lr 1e-4, min 0.9e-4, 3 warmup, 16 epochs
    wl = importlib.import_module("rift.cbv.planning.pluto.optim.warmup_cos_lr")
    # torch 2.10 dropped the `verbose` ctor argument the reference passes on, so its get_lr is
    # evaluated on a duck-typed scheduler state instead of a constructed instance
    lrs = []
    for e in range(16):
        fake = _t.SimpleNamespace(last_epoch=e, warmup_epochs=3, lr=1e-4, min_lr=0.9e-4, epochs=16,
                                  optimizer=_t.SimpleNamespace(param_groups=[{}]))
        lrs.append(wl.WarmupCosLR.get_lr(fake)[0])
    out["warmup_cos_lr"] = np.array(lrs)
    path = os.path.join(HERE, "advantage.npz")
    np.savez_compressed(path, **out)
    print("advantage ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB")




# ------------------------------------------------------------------------------------------
# candidate rollout fixtures (reference TrackPropagate + TrajEvaluator.get_ref_line_info)
# ------------------------------------------------------------------------------------------
def gen_rollout():
    ref_loader.install()
    import types as _t
    from tests.helpers import rollout_inputs
    sys.modules.setdefault("rift.cbv.planning.pluto.utils.nuplan_state_utils",
                           _t.ModuleType("rift.cbv.planning.pluto.utils.nuplan_state_utils")).CarlaAgentState = object
    tu = _t.ModuleType("rift.util.torch_util")
    tu.get_device_name = lambda: "cpu"
    sys.modules["rift.util.torch_util"] = tu
    tp = importlib.import_module("rift.cbv.planning.fine_tuner.rlft.traj_eval.track_propogate")
    ref_info = _ref_function("rift/cbv/planning/fine_tuner/rlft/traj_eval/traj_evaluator.py", "get_ref_line_info")
    out = {}
    prop = tp.TrackPropagate(virtual_time_step=0.1)
    for call, seed in enumerate((777, 778)):          # two consecutive calls: the PID filters keep their state
        traj, ref_pos, ref_ang, st = rollout_inputs(seed)
        t40 = traj[:, :, :40, :]
        dd, da = ref_info(None, t40, ref_pos, ref_ang)
        out[f"c{call}.delta_dis"], out[f"c{call}.delta_angle"] = dd, da
        # get_center_rollout preprocessing (traj_evaluator.py:115-153), reproduced with the reference's own ops
        heading = torch.atan2(t40[..., 3], t40[..., 2])
        o = torch.cat([t40[..., :2], heading[..., None]], axis=-1)
        R, M, T, C = o.shape
        o = o.reshape(-1, T, C)
        cpos = torch.tensor(st["pos"], dtype=torch.float32)
        ch = torch.tensor(st["heading"], dtype=torch.float32)
        first = o[:, 0, :2]
        o[:, :, :2] -= first.unsqueeze(1)
        cos_h, sin_h = torch.cos(ch), torch.sin(ch)
        rot = torch.stack((torch.stack([cos_h, sin_h], dim=-1), torch.stack([-sin_h, cos_h], dim=-1)), dim=-2)
        gpos = torch.matmul(o[..., :2], rot) + cpos
        ghead = o[..., 2] + ch
        ego = _t.SimpleNamespace(dynamic_car_state=_t.SimpleNamespace(speed=st["speed"]),
                                 car_footprint=_t.SimpleNamespace(width=st["width"], length=st["length"]))
        res = prop.propagate(gpos, ghead, [ego])
        for k, v in zip(("center", "angle", "speed", "acc", "ang_vel", "ang_acc", "vertices"), res):
            out[f"c{call}.{k}"] = v
    path = os.path.join(HERE, "rollout.npz")
    np.savez_compressed(path, **out)
    print("rollout ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB")


# ------------------------------------------------------------------------------------------
# PPO critic fixture: the reference's CriticPPO (gym_carla/utils/net.py:420-431) and its full get_ppo_loss
# (ppo_trainer.py:161-183: SmoothL1 value loss + clipped actor loss + entropy) with autograd gradients of value_net
# ------------------------------------------------------------------------------------------
def gen_critic():
    import types
    import torch.nn as nn
    from tests.helpers import critic_inputs, critic_weights
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_net", os.path.join(ref_loader.REF_ROOT, "rift/gym_carla/utils/net.py"))
    net = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(net)
    critic = net.CriticPPO(dims=[256, 256], state_dim=128, action_dim=3)    # ppo_pluto.yaml:43-45
    critic.load_state_dict(critic_weights(), strict=True)
    for p_ in critic.parameters():      # what freeze_parameters(["planning_decoder.pi_head", "value_net"]) does (ppo_trainer.py:84-96):
        p_.requires_grad = True         # EVERY parameter of value_net becomes trainable, state_avg/std and value_avg/std included
    inp = critic_inputs()
    out = {}
    with torch.no_grad():
        out["value"] = critic(inp["state"]).numpy()
    ns = types.SimpleNamespace(clip_epsilon=0.2, lambda_entropy=0.01, value_criterion=nn.SmoothL1Loss(),
                               model=types.SimpleNamespace(value_net=critic))
    prob = inp["probability"].clone().requires_grad_(True)
    pm = prob.masked_fill(inp["r_pad"].unsqueeze(-1), -1e8)                 # ppo_trainer.py:133
    b = {"state_torch": inp["state"], "advantage_torch": inp["advantage"], "reward_sum_torch": inp["reward_sum"],
         "old_log_prob_torch": inp["old_log_prob"]}
    loss = _load_ppo_loss_fn()(ns, pm, inp["action_mode"], b)
    loss.backward()
    out["loss"] = loss.detach().double().numpy()
    out["value_loss"] = nn.SmoothL1Loss()(critic(inp["state"]), inp["reward_sum"]).detach().double().numpy()
    out["dprobability"] = prob.grad.numpy()
    for n_, p_ in critic.named_parameters():
        if p_.grad is not None:
            out["grad." + n_] = p_.grad.numpy()
    path = os.path.join(HERE, "ppo_critic.npz")
    np.savez_compressed(path, **out)
    print("ppo critic ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", float(out["loss"]), float(out["value_loss"]))


# ------------------------------------------------------------------------------------------
# rollout-side inference helpers: _trim_candidates / _global_to_local (pluto.py:196-279) and PIDController.control_pid
# (controller/pid_controller.py:39-100), run on seeded inputs
# ------------------------------------------------------------------------------------------
def gen_inference():
    import types
    import importlib.util
    from scipy.special import softmax
    from tests.helpers import inference_inputs
    inp = inference_inputs()
    import numpy.typing as npt
    ns = {"softmax": softmax, "npt": npt, "CarlaAgentState": object, "Tuple": tuple}
    trim = _ref_function("rift/cbv/planning/pluto/pluto.py", "_trim_candidates", ns)
    g2l = _ref_function("rift/cbv/planning/pluto/pluto.py", "_global_to_local", ns)
    spec = importlib.util.spec_from_file_location("ref_pid", os.path.join(ref_loader.REF_ROOT, "rift/cbv/planning/pluto/controller/pid_controller.py"))
    pid = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pid)
    state = types.SimpleNamespace(rear_axle=types.SimpleNamespace(array=inp["origin"], heading=inp["angle"]))
    fake_self = types.SimpleNamespace(_topk=10)
    out = {}
    traj, score, orig, n_ref, n_mode = trim(fake_self, inp["candidates"].copy(), inp["probability"].copy(), state, inp["ref_free"].copy())
    out["trim.traj"], out["trim.score"], out["trim.orig"] = traj, score, orig.astype(np.int64)
    best = int(score.argmax())
    local = g2l(fake_self, traj[best, 1:], state)
    out["local"] = local
    ctrl = pid.PIDController()
    acts = []
    for k in range(6):      # consecutive ticks: the PID windows carry state
        acts.append([float(v) for v in ctrl.control_pid(local[:, :2] * (1.0 + 0.05 * k), 3.0 + 0.5 * k)])
    out["actions"] = np.array(acts)
    path = os.path.join(HERE, "inference.npz")
    np.savez_compressed(path, **out)
    print("inference ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", out["trim.orig"].tolist(), out["actions"][0])


def gen_other_vehicles():
    """TrajEvaluator.get_other_vehicle_rollout (traj_evaluator.py:160-239) run as the reference wrote it: the method and
    compute_agents_vertices are compiled in memory from the reference file, KinematicBicycleModel and GlobalConfig are imported from
    rift/ego/pdm_lite (numpy only; `carla` is a permissive stand-in because GlobalConfig's class body names CARLA enums)."""
    import importlib.util
    import types
    from tests.helpers import other_vehicle_inputs

    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    class Vector3D:
        def __init__(self, x=0.0, y=0.0, z=0.0):
            self.x, self.y, self.z = x, y, z

        def length(self):
            return float(np.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2))

    carla = types.ModuleType("carla")
    carla.__getattr__ = lambda k: _Any()
    carla.Vector3D = Vector3D
    sys.modules["carla"] = carla

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_loader.REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    cfg = load("ref_pdm_config", "rift/ego/pdm_lite/config.py").GlobalConfig()
    kbm = load("ref_kbm", "rift/ego/pdm_lite/kinematic_bicycle_model.py").KinematicBicycleModel(cfg)
    TE = "rift/cbv/planning/fine_tuner/rlft/traj_eval/traj_evaluator.py"
    cav = _ref_function(TE, "compute_agents_vertices")
    rollout = _ref_function(TE, "get_other_vehicle_rollout", {"carla": carla, "compute_agents_vertices": cav})
    inp = other_vehicle_inputs()
    N = len(inp["steer"])

    def actor(i):
        ctl = types.SimpleNamespace(steer=float(inp["steer"][i]), throttle=float(inp["throttle"][i]), brake=float(inp["brake"][i]))
        loc = types.SimpleNamespace(x=float(inp["location"][i, 0]), y=float(inp["location"][i, 1]), z=float(inp["location"][i, 2]))
        return types.SimpleNamespace(get_control=lambda: ctl, get_velocity=lambda: Vector3D(float(inp["speed"][i]), 0.0, 0.0),
                                     get_location=lambda: loc,
                                     get_transform=lambda: types.SimpleNamespace(rotation=types.SimpleNamespace(yaw=float(inp["yaw_deg"][i]))),
                                     bounding_box=types.SimpleNamespace(extent=Vector3D(float(inp["extent"][i, 0]), float(inp["extent"][i, 1]), 0.8)))

    fake_self = types.SimpleNamespace(config=cfg, other_vehicle_model=kbm, bbox_inflation_ratio=1.1)
    out = {"vertices": rollout(fake_self, [actor(i) for i in range(N)], num_future_frames=40),
           "vertices_empty_shape": np.array(rollout(fake_self, [], num_future_frames=40).shape),
           "config": np.array([cfg.time_step, cfg.front_wheel_base, cfg.rear_wheel_base, cfg.steering_gain, cfg.brake_acceleration,
                               cfg.throttle_acceleration, cfg.slow_speed_extent_factor_ego, cfg.extent_other_vehicles_bbs_speed_threshold,
                               cfg.high_speed_min_extent_y_other_vehicle, cfg.high_speed_extent_y_factor_other_vehicle,
                               cfg.high_speed_min_extent_x_other_vehicle, cfg.high_speed_min_extent_x_other_vehicle_lane_change])}
    path = os.path.join(HERE, "other_vehicles.npz")
    np.savez_compressed(path, **out)
    print("other vehicles ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", out["vertices"].shape, out["vertices"].dtype, out["config"])


def gen_off_road():
    """TrajEvaluator.get_off_road_matrix (traj_evaluator.py:277-317) with its callees global_to_pixel (:319-322) and fill_polygon (:323-325),
    compiled in memory from the reference file and run as written.  What is stood in for: the CARLA map (CarlaDataProvider.get_map_api()
    returns a few dummy drivable-area polygons) and cv2.fillPoly -- the stand-in does NOT rasterise, it writes a seeded pattern of `value`
    into the mask it is handed and keeps a reference to it.  Everything the reference does once the mask exists -- the pose's rotation
    matrix, global_to_pixel in float64 against the float32 resolution / offset vectors, np.round (half to even), the bounds test, the
    [y, x] lookup and the == 1 test -- is the reference's own numpy; the device kernel and the oracle take exactly that mask as their input.
    Cases: axis-aligned pose with exact half-pixel ties and raster-edge points, rotated / translated poses, points far outside the raster,
    and a non-square raster (the reference's offset is [map_height / 2, map_width / 2] applied to (x, y): only visible when H != W)."""
    import itertools
    import types
    TE = "rift/cbv/planning/fine_tuner/rlft/traj_eval/traj_evaluator.py"
    LANE, CONNECTOR, OTHER = "lane", "lane_connector", "crosswalk"

    class _Poly:
        def __init__(self, xy):
            self.exterior = types.SimpleNamespace(coords=types.SimpleNamespace(xy=(xy[:, 0].copy(), xy[:, 1].copy())))

    out, n_case = {}, 0
    cases = [  # (name, H, W, origin, heading, number of points (G, T), spread of the points in metres, seed)
        ("axis_ties", 400, 400, (12.0, -7.0), 0.0, (6, 40), 70.0, 11),
        ("rotated", 400, 400, (13.25, -7.5), 0.7, (16, 80), 60.0, 12),
        ("rotated_far", 400, 400, (-250.125, 991.5), -2.4, (12, 80), 90.0, 13),
        ("non_square", 200, 300, (3.5, 4.25), 1.1, (8, 40), 50.0, 14),
    ]
    for name, Hh, Ww, origin, heading, (G, T), spread, seed in cases:
        rng = np.random.default_rng(seed)
        held = {}

        def fill_poly(mask, pts, value, _rng=rng, _held=held):
            assert mask.dtype == np.uint8 and len(pts) == 1 and pts[0].dtype == np.int32 and pts[0].ndim == 2 and pts[0].shape[1] == 2
            _held["mask"] = mask
            _held["calls"] = _held.get("calls", 0) + 1
            mask[_rng.random(mask.shape) < 0.22] = value          # a pattern, not a rasterisation: the lookup does not care what drew the mask

        cv2 = types.SimpleNamespace(fillPoly=fill_poly)
        polys = {LANE: [types.SimpleNamespace(polygon=_Poly(rng.normal(0, 30, (5, 2)) + origin)) for _ in range(2)],
                 CONNECTOR: [types.SimpleNamespace(polygon=_Poly(rng.normal(0, 30, (4, 2)) + origin))],
                 OTHER: [types.SimpleNamespace(polygon=_Poly(rng.normal(0, 30, (4, 2)) + origin))]}      # not a drivable-area layer: must not be drawn
        seen = {}

        class _MapApi:
            def query_proximal_map_data(self, point, radius):
                seen["point"], seen["radius"] = point, radius
                return polys

        provider = types.SimpleNamespace(get_map_api=lambda: _MapApi())
        ns = {"itertools": itertools, "cv2": cv2, "Point": lambda *a: tuple(float(v) for v in a), "CarlaDataProvider": provider,
              "DA": [LANE, CONNECTOR], "CarlaAgentState": object, "CarlaMap": object}
        fake = types.SimpleNamespace(map_height=Hh, map_width=Ww, resolution=0.5,
                                     resolution_hw=np.array([0.5, -0.5], dtype=np.float32),                 # traj_evaluator.py:100
                                     offset=np.array([Hh / 2, Ww / 2], dtype=np.float32))                   # :101
        fake.global_to_pixel = types.MethodType(_ref_function(TE, "global_to_pixel", ns), fake)
        fake.fill_polygon = types.MethodType(_ref_function(TE, "fill_polygon", ns), fake)
        off_road = _ref_function(TE, "get_off_road_matrix", ns)
        pts = rng.normal(0, spread, (G, T, 2)).astype(np.float32) + np.asarray(origin, dtype=np.float32)
        if heading == 0.0:
            # exact half-pixel ties and raster edges (axis-aligned pose, dyadic coordinates: exact in float32 and in the float64 pipeline):
            # pixel x = lx / 0.5 + H/2, pixel y = -ly / 0.5 + W/2
            k = 0
            for px in (-0.5, 0.5, 1.5, 10.5, 11.5, Ww - 1.5, Ww - 0.5, -0.75, Ww - 0.25, 0.0, Ww - 1.0, float(Ww)):
                for py in (-0.5, 0.5, 2.5, Hh - 1.5, Hh - 0.5, 7.0):
                    lx, ly = (px - Hh / 2) * 0.5, -(py - Ww / 2) * 0.5
                    pts[k // T, k % T] = (origin[0] + lx, origin[1] + ly)
                    k += 1
        state = types.SimpleNamespace(center=types.SimpleNamespace(array=np.asarray(origin, dtype=np.float64), heading=float(heading)))
        want = off_road(fake, pts, state)
        assert held["calls"] == 3 and seen["radius"] == max(Hh, Ww) * 0.5 / 2 and want.shape == (G, T) and want.dtype == np.bool_
        mask = held["mask"]
        assert mask.shape == (Hh, Ww) and 0.3 < mask.mean() < 0.7
        out.update({f"{n_case}.name": np.array(name), f"{n_case}.mask": mask.copy(), f"{n_case}.points": pts,
                    f"{n_case}.pose": np.array([origin[0], origin[1], heading], dtype=np.float64), f"{n_case}.off_road": want})
        print(f"  {name}: mask {mask.shape} mean {mask.mean():.3f}, {G * T} points, off-road share {want.mean():.3f}")
        n_case += 1
    out["n_case"] = np.array(n_case)
    path = os.path.join(HERE, "off_road.npz")
    np.savez_compressed(path, **out)
    print("off_road ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", n_case, "cases")


def gen_sft():
    """SFT teacher objective as the reference computes it: LightningTrainer._compute_objectives / get_teacher_loss / generate_target_label
    (fine_tuner/sft/sft_trainer.py:123-199) compiled in memory, sft/utils.global_to_local and PIDController imported from the reference."""
    import importlib.util
    import types
    import torch.nn.functional as F
    from tests.helpers import sft_inputs

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_loader.REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    g2l = load("ref_sft_utils", "rift/cbv/planning/fine_tuner/sft/utils.py").global_to_local
    pid = load("ref_pid2", "rift/cbv/planning/pluto/controller/pid_controller.py").PIDController()
    T_ = "rift/cbv/planning/fine_tuner/sft/sft_trainer.py"
    ns = {"F": F, "global_to_local": g2l, "Dict": dict}
    objectives = _ref_function(T_, "_compute_objectives", ns)
    fake = types.SimpleNamespace(controller=pid, frame_rate=10)
    fake.generate_target_label = types.MethodType(_ref_function(T_, "generate_target_label", ns), fake)
    fake.get_teacher_loss = types.MethodType(_ref_function(T_, "get_teacher_loss", ns), fake)
    inp = sft_inputs()
    prob = inp["probability"].clone().requires_grad_(True)
    cur_res = {"trajectory": inp["trajectory"].clone(), "probability": prob * 1.0}
    cur_data = {"reference_line": {"valid_mask": inp["ref_valid_mask"]}}
    out = objectives(fake, cur_res, cur_data, {"teacher_infos": inp["teacher_infos"]})
    out["loss"].backward()
    target, _ = fake.generate_target_label(inp["trajectory"], cur_res["probability"].detach(), inp["teacher_infos"],
                                           torch.argmax(cur_res["probability"].detach().view(prob.shape[0], -1), 1) // 12,
                                           torch.argmax(cur_res["probability"].detach().view(prob.shape[0], -1), 1) % 12)
    tgt = target.view(prob.shape).nonzero()
    res = {"loss": np.array(float(out["loss"])), "dloss_dprob": prob.grad.numpy(), "target_r": tgt[:, 1].numpy(), "target_m": tgt[:, 2].numpy()}
    path = os.path.join(HERE, "sft.npz")
    np.savez_compressed(path, **res)
    print("sft ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", float(out["loss"]), res["target_r"].tolist(), res["target_m"].tolist())


def gen_collate():
    """Collation as the reference does it: PlutoFeature.collate (pluto/feature_builder/pluto_feature.py:25-96) inside RIFTCollate.__call__
    (fine_tuner/rlft/rift_pluto/rift_datamodule.py:20-51), imported from the reference and run on ragged seeded scenes (agent, polygon
    and reference-line counts all differ per scene).  The fixture holds every tensor of the padded batch."""
    import types
    from tests.helpers import collate_scenes_ragged
    ref_loader.install()
    for pkg in ("rift.gym_carla", "rift.gym_carla.buffer", "rift.util"):      # package shells: their __init__ import carla / pygame
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(ref_loader.REF_ROOT, *pkg.split("."))]
            sys.modules[pkg] = m
    dm = importlib.import_module("rift.cbv.planning.fine_tuner.rlft.rift_pluto.rift_datamodule")
    pf = importlib.import_module("rift.cbv.planning.pluto.feature_builder.pluto_feature")
    scenes = collate_scenes_ragged()
    batch_in = [{"CBVs_obs": {"raw_pluto_feature": pf.PlutoFeature(data=s["feature"])},
                 "CBVs_group_advantage": {"advantage": s["extras"]["group_advantage"].numpy(),
                                          "valid_mask": s["extras"]["group_advantage_mask"].numpy()},
                 "CBVs_actions_old_group_logits": {"logits": s["extras"]["old_group_logits"].numpy(),
                                                   "valid_mask": s["extras"]["old_group_logits_mask"].numpy()}} for s in scenes]
    res = dm.RIFTCollate()(batch_in)
    out = {}
    for grp, v in res["cur_pluto_feature_torch"].data.items():
        if isinstance(v, dict):
            for k, t in v.items():
                out[f"feature/{grp}/{k}"] = to_np(t)
        else:
            out[f"feature/{grp}"] = to_np(v)
    for k in ("group_advantage_torch", "group_advantage_mask_torch", "old_group_logits_torch", "old_group_logits_mask_torch"):
        out[k] = to_np(res[k])
    out["input_digest"] = syn.digest({f"{i}/{k}": v for i, s in enumerate(scenes) for k, v in syn.flatten_dict(s["feature"]).items()})
    path = os.path.join(HERE, "collate.npz")
    np.savez_compressed(path, **out)
    print("collate ->", path, f"{os.path.getsize(path) / 1e3:.1f} KB", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and k.startswith("feature/agent")})


def gen_normalize():
    """PlutoFeature.normalize / to_feature_tensor as the reference runs them (pluto_feature.py:98-126,166-263) on a seeded raw feature dict:
    first call (map crop, origin / angle) and a later call (first_time=False) on a fresh copy."""
    from tests.helpers import raw_feature_inputs
    ref_loader.install()
    pf = importlib.import_module("rift.cbv.planning.pluto.feature_builder.pluto_feature")
    out = {}
    for tag, first in (("first", True), ("later", False)):
        res = pf.PlutoFeature.normalize(raw_feature_inputs(), first_time=first, radius=120, hist_steps=21)
        ten = res.to_feature_tensor()
        for src, name in ((res.data, "np"), (ten.data, "tensor")):
            for grp, v in src.items():
                if isinstance(v, dict):
                    for k, t in v.items():
                        out[f"{tag}/{name}/{grp}/{k}"] = to_np(t) if torch.is_tensor(t) else np.asarray(t)
                else:
                    out[f"{tag}/{name}/{grp}"] = to_np(v) if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, "normalize.npz")
    np.savez_compressed(path, **out)
    print("normalize ->", path, f"{os.path.getsize(path) / 1e3:.1f} KB", len(out), "arrays")


def _flatten_feature(prefix, data, out):
    for k, v in data.items():
        if isinstance(v, dict):
            _flatten_feature(f"{prefix}/{k}", v, out)
        elif isinstance(v, (np.ndarray, np.generic, float, int)):
            out[f"{prefix}/{k}"] = np.asarray(v)
        elif isinstance(v, list) and all(isinstance(x, (str, int, np.integer)) for x in v):
            out[f"{prefix}/{k}"] = np.asarray([str(x) for x in v])


def gen_feature_builder():
    """PlutoFeatureBuilder as the reference runs it (pluto/feature_builder/pluto_feature_builder.py:30-401): its methods are compiled in
    memory (the module's top imports CARLA), bound to an object that carries the reference's own __init__ results, and run on the recorded
    readings of tests.helpers.feature_builder_world -- CarlaDataProvider, the map API and the route planner are the recorded objects,
    the enums are the reference's own (nuplan_plugin), PlutoFeature.normalize and CostMapManager are the reference's own classes.
    OpenCV is absent here: cv2.fillPoly / fillConvexPoly are tests.helpers.bbox_fill on BOTH sides (the fills are injected callables
    in the mirror), so the fixture pins the raster geometry, the distance transform and the float16 cast, not OpenCV's scan conversion."""
    import ast
    import importlib
    import types
    import warnings
    from tests.helpers import bbox_fill, feature_builder_world
    ref_loader.install()
    for name in ("geopandas", "numba", "shapely", "carla"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["geopandas"], "GeoDataFrame"):
        sys.modules["geopandas"].GeoDataFrame = object          # (maps_datatypes.py:15 only aliases the type)
    sh = sys.modules["shapely"]
    if not hasattr(sh, "Point"):
        class _P:                                   # query-point stand-in: the recorded map ignores it
            def __init__(self, *xy):
                self.xy = xy
        sh.Point, sh.Polygon, sh.LineString = _P, object, object
        sys.modules["shapely.geometry"] = types.ModuleType("shapely.geometry")
        sys.modules["shapely.geometry"].Point, sys.modules["shapely.geometry"].Polygon = _P, object
    cv2 = sys.modules["cv2"]
    cv2.fillPoly = lambda mask, pts, value: [bbox_fill(mask, p, value) for p in pts]
    cv2.fillConvexPoly = lambda mask, pts, value: bbox_fill(mask, pts, value)
    from nuplan_plugin.actor_state.tracked_objects_types import TrackedObjectType
    from nuplan_plugin.maps.maps_datatypes import SemanticMapLayer, TrafficLightStatusType
    # the map-utils module is CARLA / geopandas bound: the cost-map manager only names its CarlaMap type
    mu = types.ModuleType("rift.cbv.planning.pluto.utils.nuplan_map_utils")
    mu.CarlaMap = object
    for pkg in ("rift.cbv.planning.pluto", "rift.cbv.planning.pluto.utils"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(ref_loader.REF_ROOT, *pkg.split("."))]
            sys.modules[pkg] = m
    sys.modules["rift.cbv.planning.pluto.utils.nuplan_map_utils"] = mu
    cmm = importlib.import_module("rift.cbv.planning.pluto.utils.cost_map_manager")
    pf = importlib.import_module("rift.cbv.planning.pluto.feature_builder.pluto_feature")
    # (rotate_round_z_axis is @numba.njit in the reference; numba is absent here: the decorator becomes the identity, the body runs as numpy)
    rot = _ref_function("rift/cbv/planning/pluto/feature_builder/common.py", "rotate_round_z_axis",
                        {"numba": types.SimpleNamespace(njit=lambda f: f)})
    B_ = "rift/cbv/planning/pluto/feature_builder/pluto_feature_builder.py"
    out = {}
    for case in ("busy", "alone"):
        w = feature_builder_world(case, agent_type=lambda n: TrackedObjectType[n], layer=lambda n: SemanticMapLayer[n])
        ns = {"np": np, "TrackedObjectType": TrackedObjectType, "SemanticMapLayer": SemanticMapLayer, "TrafficLightStatusType": TrafficLightStatusType,
              "CarlaDataProvider": w.provider, "rotate_round_z_axis": rot, "Point": sh.Point, "LineString": sh.LineString, "warnings": warnings,
              "CostMapManager": cmm.CostMapManager, "PlutoFeature": pf.PlutoFeature, "List": list, "Set": set, "CarlaAgentState": object,
              "CarlaMap": object, "Point2D": object, "carla": types.SimpleNamespace(Vehicle=object), "CBVRoutePlanner": object}
        tree = ast.parse(open(os.path.join(ref_loader.REF_ROOT, B_)).read())
        cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PlutoFeatureBuilder")
        exec(compile(ast.Module(body=[cls], type_ignores=[]), B_, "exec"), ns)          # the class as written, names resolved from `ns`
        builder = ns["PlutoFeatureBuilder"](w.config, w.planner)
        feature, route_ids, lines, elements, wp = builder.build_feature(w.center, w.nearby, mode=w.mode)
        _flatten_feature(case, feature.data, out)
        out[f"{case}/route_road_ids"] = np.asarray(route_ids["road_ids"])
        out[f"{case}/n_lines"] = np.asarray(len(lines))
    path = os.path.join(HERE, "feature_builder.npz")
    np.savez_compressed(path, **out)
    print("feature_builder ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", len(out), "arrays;",
          {k: v.shape for k, v in out.items() if k.endswith(("agent/position", "map/point_position", "reference_line/position", "cost_maps"))})


def gen_rtr():
    """RTR objective as the reference computes it: LightningTrainer._compute_objectives / get_ppo_loss / get_teacher_loss /
    generate_target_label of fine_tuner/sft/rtr_pluto/rtr_trainer.py:130-255 compiled in memory, with the reference's CriticPPO
    (gym_carla/utils/net.py), sft/utils.global_to_local and PIDController imported from where they lie."""
    import importlib.util
    import types
    import torch.nn as nn
    import torch.nn.functional as F
    from tests.helpers import critic_weights, rtr_inputs

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_loader.REF_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    g2l = load("ref_sft_utils2", "rift/cbv/planning/fine_tuner/sft/utils.py").global_to_local
    pid = load("ref_pid3", "rift/cbv/planning/pluto/controller/pid_controller.py").PIDController()
    critic = load("ref_net2", "rift/gym_carla/utils/net.py").CriticPPO(dims=[256, 256], state_dim=128, action_dim=3)
    critic.load_state_dict(critic_weights(), strict=True)
    for p_ in critic.parameters():
        p_.requires_grad = True                       # freeze_parameters([pi_head, value_net]) (rtr_trainer.py:84-100)
    T_ = "rift/cbv/planning/fine_tuner/sft/rtr_pluto/rtr_trainer.py"
    ns = {"F": F, "global_to_local": g2l, "Dict": dict}
    fake = types.SimpleNamespace(controller=pid, frame_rate=10, clip_epsilon=0.2, lambda_entropy=0.01, value_criterion=nn.SmoothL1Loss(),
                                 model=types.SimpleNamespace(value_net=critic))
    for name in ("generate_target_label", "get_teacher_loss", "get_ppo_loss"):
        setattr(fake, name, types.MethodType(_ref_function(T_, name, ns), fake))
    objectives = _ref_function(T_, "_compute_objectives", ns)
    inp = rtr_inputs()
    prob = inp["probability"].clone().requires_grad_(True)
    cur_res = {"trajectory": inp["trajectory"].clone(), "probability": prob * 1.0}
    cur_data = {"reference_line": {"valid_mask": inp["ref_valid_mask"]}}
    batch = {"teacher_infos": inp["teacher_infos"], "action_mode_torch": inp["action_mode"], "state_torch": inp["state"],
             "advantage_torch": inp["advantage"], "reward_sum_torch": inp["reward_sum"], "old_log_prob_torch": inp["old_log_prob"]}
    out = objectives(fake, cur_res, cur_data, batch)
    out["loss"].backward()
    res = {"loss": np.array(float(out["loss"])), "ppo_loss": np.array(out["ppo_loss"]), "teacher_loss": np.array(out["teacher_loss"]),
           "dloss_dprob": prob.grad.numpy()}
    for n_, p_ in critic.named_parameters():
        if p_.grad is not None:
            res["grad." + n_] = p_.grad.numpy()
    path = os.path.join(HERE, "rtr.npz")
    np.savez_compressed(path, **res)
    print("rtr ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", float(out["loss"]), out["ppo_loss"], out["teacher_loss"])


def gen_buffer():
    """CBVRolloutBuffer (rift/gym_carla/buffer/cbv_rollout_buffer.py:16-138), the reference class itself: the seeded store() sequences
    of tests.helpers.buffer_store_sequences are replayed through it; after EVERY call the fixture records buffer_pos, buffer_full and
    the stored order (the integer codes of 'CBVs_obs'), and at the end what sample() / get_key_data() return."""
    import types
    from tests.helpers import BUFFER_KEYS, buffer_store_sequences
    ref_loader.install()
    for pkg in ("rift.gym_carla", "rift.gym_carla.buffer"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref_loader.REF_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    from rift.gym_carla.buffer.cbv_rollout_buffer import CBVRolloutBuffer as RefBuffer
    pos, full, order, order_off, done_col, sample_mid = [], [], [], [0], [], []
    seqs = buffer_store_sequences()
    for capacity, calls in seqs:
        buf = RefBuffer(1, 'train_cbv', {'buffer_capacity': capacity, 'data_keys': list(BUFFER_KEYS)})
        for chunk in calls:
            if buf.buffer_full:              # carla_runner.py:239-247 trains and resets as soon as the buffer is full; further stores would
                break                        # rotate the deques (maxlen) -- not a state the runner produces
            buf.store(chunk)
            pos.append(buf.buffer_pos)
            full.append(buf.buffer_full)
            order.extend(int(v) for v in buf.buffer_data['CBVs_obs'])
            order_off.append(len(order))
        if buf.buffer_full:
            done_col.extend(bool(v) for v in buf.get_key_data('CBVs_done'))
            sample_mid.append(int(buf.sample(capacity // 2)['CBVs_actions']))
        else:
            sample_mid.append(-1)
    out = {"buffer_pos": np.array(pos, dtype=np.int64), "buffer_full": np.array(full), "order": np.array(order, dtype=np.int64),
           "order_off": np.array(order_off, dtype=np.int64), "done_when_full": np.array(done_col), "sample_mid": np.array(sample_mid, dtype=np.int64),
           "n_seq": np.array(len(seqs))}
    path = os.path.join(HERE, "buffer.npz")
    np.savez_compressed(path, **out)
    print("buffer ->", path, f"{os.path.getsize(path) / 1e3:.1f} kB", len(seqs), "sequences,", len(pos), "store calls,",
          int(np.sum(full)), "calls ending full,", int(sum(1 for s in sample_mid if s >= 0)), "sequences filled")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "shapes":
        main(SHAPE_CASES)
    elif len(sys.argv) > 1 and sys.argv[1] == "buffer":
        gen_buffer()
    elif len(sys.argv) > 1 and sys.argv[1] == "feature_builder":
        gen_feature_builder()
    elif len(sys.argv) > 1 and sys.argv[1] == "rtr":
        gen_rtr()
    elif len(sys.argv) > 1 and sys.argv[1] == "normalize":
        gen_normalize()
    elif len(sys.argv) > 1 and sys.argv[1] == "collate":
        gen_collate()
    elif len(sys.argv) > 1 and sys.argv[1] == "sft":
        gen_sft()
    elif len(sys.argv) > 1 and sys.argv[1] == "other_vehicles":
        gen_other_vehicles()
    elif len(sys.argv) > 1 and sys.argv[1] == "off_road":
        gen_off_road()
    elif len(sys.argv) > 1 and sys.argv[1] == "inference":
        gen_inference()
    elif len(sys.argv) > 1 and sys.argv[1] == "rollout":
        gen_rollout()
    elif len(sys.argv) > 1 and sys.argv[1] == "critic":
        gen_critic()
    else:
        main()
        gen_advantage()
        gen_rollout()
        gen_critic()
        gen_inference()
        gen_other_vehicles()
        gen_off_road()
        gen_sft()
        gen_buffer()
        gen_rtr()
        gen_feature_builder()
