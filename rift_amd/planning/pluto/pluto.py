"""Rollout-side CBV policy: `PLUTO` (registry key 'pluto'; reference rift/cbv/planning/pluto/pluto.py:25-300) on the HIP engine.

Per tick and per environment: collate the CBVs' features -> ONE eval-mode forward with every output (rift_forward) -> per CBV: keep the
top-k candidates (+ the ref-free one), pick the best by score, move it to the CBV's frame, waypoint PID -> (throttle, steer, brake).

The reference reads the CBV's pose, speed and footprint, its neighbours and the HD map off the running CARLA server through the
`CarlaDataProvider` singleton.  Here that is ONE injected object, a *state source* (`CBVStateSource`): the CARLA runner passes an
adapter over its data provider (`CarlaStateSource`), tests and offline replays pass recorded states -- the policy itself never imports
CARLA, and nothing below the source differs between the two.
"""
import contextlib
import gc
from collections import defaultdict
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np
import torch

from rift_amd.planning.pluto.controller.pid_controller import PIDController
from rift_amd.planning.pluto.feature_builder.pluto_feature import PlutoFeature
from rift_amd.planning.pluto.inference import global_to_local, trim_candidates
from rift_amd.planning.pluto.model.pluto_model import PlanningModel


class CenterState(NamedTuple):
    """What the policy needs of a CBV right now: rear-axle pose (CarlaAgentState.rear_axle), the two speeds the reference reads --
    `speed` = dynamic_car_state.speed (rear axle; the closed-loop rollout's initial speed, track_propogate.py:628) and `center_speed` =
    dynamic_car_state.center_velocity_2d.magnitude() (the waypoint PID's input, pluto.py:252; None: same as `speed`) -- and the footprint
    (car_footprint.width / length) -- pluto.py:196-247,262-300, traj_evaluator.py:115-158."""
    x: float
    y: float
    heading: float
    speed: float
    width: float
    length: float
    center_speed: Optional[float] = None

    def rollout_tuple(self):
        """(x, y, heading, speed, width, length): TrajEvaluator.get_grpo_advantage's `center_state`."""
        return tuple(self[:6])

    def pid_speed(self) -> float:
        return float(self.speed if self.center_speed is None else self.center_speed)


class NoFlagSource:
    """Marker a state source returns (instead of None) to say "this deployment has no such input": `NoFlagSource.ALL_CLEAR`."""
    ALL_CLEAR = object()


class CBVStateSource:
    """Live-state interface of the rollout side.  `center_state` is needed by every policy; the other two only by the RIFT / GRPO
    group-advantage evaluation in train mode (TrajEvaluator.get_grpo_advantage, traj_evaluator.py:422-475), where the reference ALWAYS
    feeds the CBV's neighbours and the drivable-area raster.  A source that does not implement them therefore fails the train-mode tick
    loudly; returning `NoFlagSource.ALL_CLEAR` states explicitly that no candidate can collide / leave the road (offline replays
    without actors or map), returning None from nearby_actor_states means "no neighbours right now"."""

    def center_state(self, env_id, cbv_id) -> CenterState:
        raise NotImplementedError

    def nearby_actor_states(self, env_id, cbv_id):
        """Readings of the CBV's neighbours (steer, throttle, brake, speed, location (N,3), yaw_deg, extent (N,2): the inputs of
        get_other_vehicle_rollout, traj_evaluator.py:160-239); None when there are none right now."""
        raise NotImplementedError(f"{type(self).__name__}.nearby_actor_states: the group advantage needs the CBV's neighbours "
                                  "(rift_pluto.py:115); return NoFlagSource.ALL_CLEAR to evaluate without collisions on purpose")

    def off_road_raster(self, env_id, cbv_id):
        """(mask (H, W) uint8 with 1 = not drivable, (x, y, heading) of the raster origin): get_off_road_matrix's raster
        (traj_evaluator.py:273-322), drawn by the caller from its HD map."""
        raise NotImplementedError(f"{type(self).__name__}.off_road_raster: the group advantage needs the drivable-area raster "
                                  "(traj_evaluator.py:273-322); return NoFlagSource.ALL_CLEAR to evaluate without an off-road term on purpose")


def raster_drivable_area(polygons_xy, origin_xy, heading, map_height=400, map_width=400, resolution=0.5, fill_polygon=None):
    """The reference's drivable-area raster (traj_evaluator.py:273-292,324-331): ones = off road, every drivable polygon filled with 0 after
    global_to_pixel (rotate into the CBV's frame, scale by (res, -res), shift by (H/2, W/2)) and np.round.  `fill_polygon(mask, int32
    vertices (n, 2), value)` is cv2.fillPoly's job; it is injected so that this module does not import OpenCV."""
    origin = np.asarray(origin_xy, dtype=np.float64)
    rot = np.array([[np.cos(heading), -np.sin(heading)], [np.sin(heading), np.cos(heading)]], dtype=np.float64)
    res = np.array([resolution, -resolution], dtype=np.float32)
    off = np.array([map_height / 2, map_width / 2], dtype=np.float32)
    mask = np.ones((map_height, map_width), dtype=np.uint8)
    for poly in polygons_xy:
        px = np.matmul(np.asarray(poly, dtype=np.float64) - origin, rot) / res + off
        fill_polygon(mask, np.round(px).astype(np.int32), 0)
    return mask


class CarlaStateSource(CBVStateSource):
    """Adapter over the reference's CarlaDataProvider (rift/scenario/tools/carla_data_provider.py); importable only next to CARLA.
    Reads what rift_pluto.py:113-135 and traj_evaluator.py:160-239,273-292 read: the CBV's agent state, its nearby actors' controls /
    kinematics / extents, and the drivable-area polygons of the HD map around it (rastered with cv2.fillPoly as the reference does)."""
    DRIVABLE_LAYERS = ("LANE", "LANE_CONNECTOR")      # `DA` of traj_evaluator.py:31 (SemanticMapLayer names)

    def __init__(self, map_height=400, map_width=400, resolution=0.5):
        try:
            from rift.scenario.tools.carla_data_provider import CarlaDataProvider      # noqa: WPS433 (deployment-time import)
        except Exception as e:                                                          # pragma: no cover - needs CARLA
            raise RuntimeError("CarlaStateSource needs the reference's CarlaDataProvider (a running CARLA setup); pass a CBVStateSource "
                               "to the policy (config['state_source']) when running without it") from e
        self._cdp = CarlaDataProvider
        self.map_height, self.map_width, self.resolution = map_height, map_width, resolution

    def _agent_state(self, cbv_id):                                                     # pragma: no cover - needs CARLA
        return self._cdp.get_history_state(self._cdp.get_actor_by_id(cbv_id))[-1]

    def center_state(self, env_id, cbv_id) -> CenterState:                              # pragma: no cover - needs CARLA
        st = self._agent_state(cbv_id)
        return CenterState(float(st.rear_axle.x), float(st.rear_axle.y), float(st.rear_axle.heading), float(st.dynamic_car_state.speed),
                           float(st.car_footprint.width), float(st.car_footprint.length),
                           float(st.dynamic_car_state.center_velocity_2d.magnitude()))

    def nearby_actor_states(self, env_id, cbv_id):                                      # pragma: no cover - needs CARLA
        ego_id = self._cdp.get_ego_vehicle_by_env_id(env_id).id
        actors = self._cdp.get_CBV_nearby_agents(ego_id, cbv_id)
        if not actors:
            return None
        ctl = [a.get_control() for a in actors]
        loc = [a.get_location() for a in actors]
        return {"steer": np.array([c.steer for c in ctl], dtype=np.float64), "throttle": np.array([c.throttle for c in ctl], dtype=np.float64),
                "brake": np.array([c.brake for c in ctl], dtype=np.float64),
                "speed": np.array([a.get_velocity().length() for a in actors], dtype=np.float64),
                "location": np.array([[p.x, p.y, p.z] for p in loc], dtype=np.float64),
                "yaw_deg": np.array([a.get_transform().rotation.yaw for a in actors], dtype=np.float64),
                "extent": np.array([[a.bounding_box.extent.x, a.bounding_box.extent.y] for a in actors], dtype=np.float64)}

    def off_road_raster(self, env_id, cbv_id):                                          # pragma: no cover - needs CARLA
        import cv2                                                                       # noqa: WPS433 (deployment-time import)
        from shapely.geometry import Point                                               # noqa: WPS433
        st = self._agent_state(cbv_id)
        origin, heading = np.asarray(st.center.array, dtype=np.float64), float(st.center.heading)
        radius = max(self.map_height, self.map_width) * self.resolution / 2
        objects = self._cdp.get_map_api().query_proximal_map_data(Point(*origin), radius)
        polys = [np.stack(obj.polygon.exterior.coords.xy, axis=1) for layer, objs in objects.items()
                 if getattr(layer, "name", str(layer)) in self.DRIVABLE_LAYERS for obj in objs]
        mask = raster_drivable_area(polys, origin, heading, self.map_height, self.map_width, self.resolution,
                                    fill_polygon=lambda m, v, val: cv2.fillPoly(m, [v], val))
        return mask, (float(origin[0]), float(origin[1]), heading)


class CBVBasePolicy:   # rift/cbv/planning/base_policy.py:9-52
    name = 'base'
    type = 'unlearnable'

    def __init__(self, config, logger):
        self.config = config
        self.num_scenario = config['num_scenario']
        self._render_data = None
        self.route_planner = None

    def set_buffer(self, buffer, total_routes):
        self.buffer, self.total_routes = buffer, total_routes

    def set_route_planner(self, route_planner):
        self.route_planner = route_planner

    def train(self, e_i):
        raise NotImplementedError()

    def set_mode(self, mode):
        self.mode = mode

    def get_action(self, state, infos, deterministic):
        raise NotImplementedError()

    def get_render_data(self, env_id):
        return NotImplementedError()

    def log_episode_reward(self, episode_reward, episode):
        pass

    def load_model(self, resume=True):
        pass

    def save_model(self, episode):
        pass

    def clean_up(self):
        pass

    def finish(self):
        pass


class Candidates(NamedTuple):
    """One CBV's decision of a tick."""
    control: tuple                 # (throttle, steer, brake)
    trajectory: np.ndarray         # (79, 3) chosen global trajectory
    kept: np.ndarray               # (k [+1], 80, 3) kept candidates, global frame
    score: np.ndarray              # softmax over the kept logits (+ 0.25 for the ref-free candidate)
    flat_index: np.ndarray         # original r * M + m of every kept candidate (-1: ref-free), int64
    best: int
    n_mode: int
    probability: np.ndarray        # (R, M) raw logits of this CBV


@contextlib.contextmanager
def capped_host_threads(limit: int):
    """Cap torch's intra-op thread pool for the duration of an update or a rollout tick.  The update's host side is one thread issuing launches plus a few
    tiny CPU tensor ops per epoch (permutations, index slices); every one of those that enters the intra-op pool wakes ALL its workers,
    which then spin-wait between regions.  Measured on the GPU box (256 logical CPUs, torch default 128 threads, container quota 16
    cores): 129 busy threads, 8.9 CPU-seconds inside a 0.5 s update, the cgroup throttled in every 100 ms period -- the launching thread
    frozen for 60-70 ms five times per update (GPU idle, host stuck inside hipLaunchKernel).  With 4 threads: 0.55 CPU-seconds, no
    throttling, update 0.52 -> 0.26 s.  The reference runs with torch.set_num_threads(4) throughout (scripts/run.py:133,164); a
    process that already did the same is left alone."""
    cur = torch.get_num_threads()
    capped = bool(limit) and cur > limit
    if capped:
        torch.set_num_threads(int(limit))
    try:
        yield
    finally:
        if capped:
            torch.set_num_threads(cur)


@contextlib.contextmanager
def paused_cyclic_gc(enabled: bool = True):
    """Keep CPython's cyclic collector out of an update.  The rollout's store() calls allocate a few hundred thousand container objects
    per buffer generation, so full (generation 2) collections come round every few updates -- and one that lands inside train() walks the
    whole heap (the committed rows, the models' module trees) while the launch thread stands still: 106 ms measured inside a 187 ms update
    (tools/e2e_profile.py).  The update itself creates no cyclic garbage to speak of; reference counting frees everything it drops.  The
    collector is switched back on at exit, so a pending pass runs right behind the update (in the rollout phase, where the host is not the
    critical path) -- it is postponed, not skipped.  A process that runs with the collector off is left alone."""
    was = enabled and gc.isenabled()
    if was:
        gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class PLUTO(CBVBasePolicy):
    name = 'pluto'
    type = 'il'

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self.logger = logger
        self.radius = config.get('obs', {}).get('radius', config.get('radius', 120))
        self._use_prediction = config.get('use_prediction', False)
        self._topk = config.get('topk', 10)
        self._ckpt_path = config.get('ckpt_path')
        self._frame_rate = config.get('frame_rate', 10)
        self._step_interval = 1.0 / self._frame_rate
        self.device = torch.device(config.get('device', 'cuda:0'))
        self.pluto_model = PlanningModel(radius=self.radius).to(self.device)       # the inference model
        # MFMA operand format of the policy's engines ("fp16" | "bf16" | "fp32"; PlanningModel.compute_precision explains)
        self.compute_precision = config.get('compute_precision', self.pluto_model.compute_precision)
        self.pluto_model.compute_precision = self.compute_precision
        self.pluto_model.eval()
        self.controllers = defaultdict(lambda: defaultdict(lambda: PIDController(sample_interval=self._frame_rate)))
        self._state_source: Optional[CBVStateSource] = config.get('state_source')
        self._render = config.get('need_video_render', False)
        self._host_threads = config.get('host_threads', 4)     # torch intra-op threads during a tick (capped_host_threads; 0: leave them alone)
        self.mode = 'eval'
        if self._render:
            self.reset_render_data()

    # ---- state source ---------------------------------------------------------------------------------------------------
    @property
    def state_source(self) -> CBVStateSource:
        if self._state_source is None:
            self._state_source = CarlaStateSource()
        return self._state_source

    def set_state_source(self, source: CBVStateSource):
        self._state_source = source

    # ---- render buffers (filled for the CARLA video renderer; pluto.py:54-67) -----------------------------------------------
    RENDER_LISTS = ("route_ids_list", "reference_lines_list", "route_waypoints_list", "interaction_wp_list", "planning_trajectory_list",
                    "candidate_trajectories_list", "candidate_index_list", "predictions_list")

    def reset_render_data(self):
        self._render_data = {env_id: dict({"ego_states": {}, "nearby_agents_states": {}, "CBV_states": {}}, **{k: [] for k in self.RENDER_LISTS})
                             for env_id in range(self.num_scenario)}

    def get_render_data(self, env_id):
        return self._render_data[env_id] if self._render_data else {}

    def set_mode(self, mode):
        if mode == 'train':
            raise ValueError('Pluto policy not support training mode.')
        if mode != 'eval':
            raise ValueError(f'Unknown mode {mode}')
        self.mode = mode
        self.pluto_model.eval()

    # ---- checkpoints (pluto.py:130-141) ---------------------------------------------------------------------------------------
    @staticmethod
    def load_infer_checkpoint(checkpoint: str, device_name) -> Dict[str, torch.Tensor]:
        """Strip the 'model.' prefix; value_net.* (PPO) is not part of the inference model."""
        ckpt = torch.load(checkpoint, map_location=device_name, weights_only=False)
        sd = {k.replace("model.", "", 1) if k.startswith("model.") else k: v for k, v in ckpt["state_dict"].items()}
        return {k: v for k, v in sd.items() if not k.startswith("value_net")}

    def load_model(self, resume=True):
        if not self._ckpt_path:
            raise FileNotFoundError(f"{self.name}: config['ckpt_path'] is not set")
        self.pluto_model.load_state_dict(self.load_infer_checkpoint(self._ckpt_path, self.device))

    def save_model(self, episode):
        raise NotImplementedError()

    # ---- one tick -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _forward(self, CBVs_obs: Dict) -> (Dict, Dict[str, torch.Tensor]):
        """Collate the CBVs of one environment and run the inference model (eval mode, every output)."""
        model = self.pluto_model
        eng = model.engine()                     # bound first: the batch goes up through the engine's pinned staging arena (asynchronous copies
        data = eng.stage_tree(PlutoFeature.collate([o['raw_pluto_feature'] for o in CBVs_obs.values()]).data)      # instead of ~36 blocking ones)
        need, model.need_traj = model.need_traj, True
        try:
            out = model(data, engine=eng)
        finally:
            model.need_traj = need
        return data, out

    @staticmethod
    def _host(out: Dict[str, Any], key: str) -> np.ndarray:
        """The whole batch of one output on the host, read back ONCE per tick (the per-CBV `.cpu()` of the reference's loop is a device
        synchronisation each: three to four per CBV and tick)."""
        cache = out.setdefault("_host", {})
        if key not in cache:
            cache[key] = out[key].cpu().numpy()
        return cache[key]

    def _decide(self, out: Dict[str, torch.Tensor], index: int, env_id, cbv_id, state: CenterState) -> Candidates:
        """pluto.py:142-194: trim, choose by learned score, PID control."""
        cand = self._host(out, "candidate_trajectories")[index].astype(np.float64)
        prob = self._host(out, "probability")[index]
        rf = self._host(out, "output_ref_free_trajectory")[index].astype(np.float64) if "output_ref_free_trajectory" in out else None
        origin = np.array([state.x, state.y], dtype=np.float64)
        kept, score, flat, _, n_mode = trim_candidates(cand, prob, origin, float(state.heading), rf, self._topk)
        best = int(score.argmax())
        trajectory = kept[best, 1:]
        local = global_to_local(trajectory, origin, float(state.heading))
        control = self.controllers[env_id][cbv_id].control_pid(local[:, :2], state.pid_speed())      # centre speed, pluto.py:252
        return Candidates(control, trajectory, kept, score, flat, best, n_mode, prob)

    def _record_render(self, env_id, cbv_id, obs, state, decision: Candidates, out, index):
        rd = self._render_data[env_id]
        rd["CBV_states"][cbv_id] = state
        for key, src in (("route_ids_list", 'route_ids'), ("reference_lines_list", 'reference_lines'),
                         ("route_waypoints_list", 'route_waypoints'), ("interaction_wp_list", 'interaction_wp')):
            rd[key].append(obs.get(src))
        rd["planning_trajectory_list"].append(decision.trajectory)
        rd["candidate_trajectories_list"].append(decision.kept)
        rd["candidate_index_list"].append(decision.best)
        rd["predictions_list"].append(self._host(out, "output_prediction")[index].copy() if self._use_prediction else None)

    def _per_cbv(self, env_id, cbv_id, obs, data, out, index, state: CenterState, decision: Candidates) -> Dict[str, Any]:
        """Extra per-CBV outputs of a policy variant, keyed by the replay-buffer column they fill (none for plain PLUTO)."""
        return {}

    EXTRA_COLUMNS: tuple = ()

    def get_action(self, CBVs_obs_list, infos, deterministic=False) -> Dict[str, List[Dict[Any, Any]]]:
        # A tick is small CPU tensor work: torch's full intra-op pool spinning on it eats the container's CPU quota (capped_host_threads).
        # The cap is set ONCE, at the first tick, and stays (switching the pool size per tick costs more than it saves: measured 88 ms
        # stalls in one tick of ten); the reference runs under torch.set_num_threads(4) from its entry script (scripts/run.py:133,164).
        # PROCESS-WIDE, like the reference's own setting; said once in the log; config['host_threads'] = 0 leaves the pool alone.
        if self._host_threads and torch.get_num_threads() > self._host_threads:
            if self.logger is not None and hasattr(self.logger, "log"):
                self.logger.log(f">> {self.name}: torch intra-op threads {torch.get_num_threads()} -> {int(self._host_threads)} for the whole "
                                f"process (the reference's scripts/run.py does the same; config['host_threads'] = 0 keeps the pool)", 'yellow')
            torch.set_num_threads(int(self._host_threads))
        return self._get_action(CBVs_obs_list, infos, deterministic)

    def _get_action(self, CBVs_obs_list, infos, deterministic=False) -> Dict[str, List[Dict[Any, Any]]]:
        result = {key: [{} for _ in range(self.num_scenario)] for key in ('CBVs_actions',) + tuple(self.EXTRA_COLUMNS)}
        for info, CBVs_obs in zip(infos, CBVs_obs_list):
            if not CBVs_obs:
                continue
            env_id = info['env_id']
            data, out = self._forward(CBVs_obs)
            states = {cbv_id: self.state_source.center_state(env_id, cbv_id) for cbv_id in CBVs_obs}
            self._begin_env(env_id, CBVs_obs, data, out, states)
            for index, (cbv_id, obs) in enumerate(CBVs_obs.items()):
                state = states[cbv_id]
                decision = self._decide(out, index, env_id, cbv_id, state)
                result['CBVs_actions'][env_id][cbv_id] = decision.control
                for key, value in self._per_cbv(env_id, cbv_id, obs, data, out, index, state, decision).items():
                    result[key][env_id][cbv_id] = value
                if self._render:
                    self._record_render(env_id, cbv_id, obs, state, decision, out, index)
            self._finish_env(env_id, data, out)
        self._finish_columns(result)
        eng = self.pluto_model._engine                      # (bound by this tick's forward; engine() would walk the parameters once more)
        (eng if eng is not None else self.pluto_model.engine()).check_finite()      # the reference's isfinite assert on the decoder queries
        self._clean_CBVs(infos, CBVs_obs_list)
        return result

    def _begin_env(self, env_id, CBVs_obs, data, out, states):
        """Hook between an environment's forward and its per-CBV decisions (a policy variant that has device work of its own to issue for
        the whole environment: it then runs beside the host-side candidate trimming and PID of the loop below)."""

    def _finish_env(self, env_id, data, out):
        """Hook behind the last CBV of an environment (a policy variant that evaluates its per-CBV columns for the whole environment at once)."""

    def _finish_columns(self, result):
        """Per-CBV columns a policy variant left on the device (`_per_cbv`) become host arrays here, after the last CBV of the tick was
        issued: one wait for the whole tick instead of one per CBV."""
        for envs in result.values():
            for per_env in envs:
                for value in per_env.values():
                    if isinstance(value, dict):
                        for k, v in value.items():
                            if torch.is_tensor(v):
                                value[k] = v.cpu().numpy()

    def _clean_CBVs(self, infos, CBVs_obs_list):
        """Drop the PID state of CBVs that left the scene (pluto.py:112-123)."""
        for info, CBVs_obs in zip(infos, CBVs_obs_list):
            env_id = info['env_id']
            if env_id not in self.controllers:
                continue
            for cbv_id in [c for c in self.controllers[env_id] if c not in CBVs_obs]:
                del self.controllers[env_id][cbv_id]
            if not self.controllers[env_id]:
                del self.controllers[env_id]
