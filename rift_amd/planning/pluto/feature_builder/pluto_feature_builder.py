"""PlutoFeatureBuilder -- the tensorisation half of pluto/feature_builder/pluto_feature_builder.py:30-401.

The reference builds a CBV's observation from the running CARLA server: it asks the `CarlaDataProvider` singleton for the history
states of the CBV and its neighbours, the HD-map API for the lanes / crosswalks around it, and the route planner for its reference
lines, and turns those readings into the numpy arrays of the Appendix-A schema before `PlutoFeature.normalize` moves them into the CBV's
frame.  Here the readings come through ONE injected `provider` with the four CarlaDataProvider calls the reference makes
(`get_history_state(actor)`, `get_current_state(actor)`, `get_frame_rate()`, `get_map_api()`): a deployment passes the reference's
`CarlaDataProvider` class itself, tests and offline tools pass recorded readings.  Everything from the readings to the feature dict --
history assembly, nearest-neighbour selection, polygon / reference-line arrays, the drivable-area signed distance field -- is this
module; nothing here imports CARLA, shapely or OpenCV (the two raster fills are injected callables: cv2.fillPoly / cv2.fillConvexPoly
at deployment).  Pinned by tests/golden/feature_builder.npz = the reference's own methods run on the same recorded readings.

States and map objects are duck-typed exactly as the reference reads them (CarlaAgentState: `rear_axle`, `center`,
`dynamic_car_state`, `tire_steering_angle`, `agent_state`; lanes: `token_id`, `centerline`, `edges`, `road_id`, `polygon`,
`speed_limit_mps`; crosswalks: `token_id`, `edges`).  Enum-valued keys / fields may be the reference's enums or plain strings.
"""
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from rift_amd.planning.pluto.feature_builder.pluto_feature import PlutoFeature

# index = the category / type id the model embeds (pluto_feature_builder.py:46-62; agent_encoder.type_emb, map_encoder.type_emb)
AGENT_TYPES = ("EGO", "VEHICLE", "PEDESTRIAN", "BICYCLE")
POLYGON_TYPES = ("LANE", "LANE_CONNECTOR", "CROSSWALK")
TL_GREEN, TL_UNKNOWN = 0, 3          # nuplan_plugin/maps/maps_datatypes.py:98-106


def _name(x) -> str:
    """'LANE' from SemanticMapLayer.LANE, TrackedObjectType.VEHICLE -> 'VEHICLE', or the string itself."""
    return getattr(x, "name", str(x)).upper()


def _layer(objects: Dict[Any, list], name: str) -> list:
    for k, v in objects.items():
        if _name(k) == name:
            return list(v)
    return []


def _map_query_point(x: float, y: float):
    """What map_api.query_proximal_map_data takes as its centre: the reference's CarlaMap wants a shapely Point (it buffers it); without
    shapely (recorded maps in tests / offline tools) a plain (x, y) tuple."""
    try:
        from shapely import Point              # noqa: WPS433 (present next to CARLA, absent in the build image)
    except Exception:
        return (x, y)
    return Point(x, y)


def _to_local(vec: np.ndarray, heading: float) -> np.ndarray:
    """rotate_round_z_axis(vec, -heading) (feature_builder/common.py:122-128): world -> the state's own frame."""
    a = -heading
    return vec @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])


class PlutoFeatureBuilder:
    def __init__(self, config: dict, route_planner, provider, fill_polygon: Optional[Callable] = None,
                 fill_convex_polygon: Optional[Callable] = None) -> None:
        """config['obs']: max_agent, radius, history_horizon (planning/config/rift_pluto.yaml:33-37).  `route_planner`:
        build_reference_line(center, current_state, radius) -> (reference_lines, route_elements, route_ids, interaction_wp).
        fill_polygon(mask, int32 vertices (n, 2), value) / fill_convex_polygon(...): the raster fills of the cost map."""
        obs = config["obs"]
        self.config, self.obs_config = config, obs
        self.max_agent, self.radius, self.history_horizon = obs["max_agent"], obs["radius"], obs["history_horizon"]
        self.provider, self.route_planner = provider, route_planner
        self.frame_rate = provider.get_frame_rate()
        self.history_samples = int(self.history_horizon * self.frame_rate)
        self.map_api = provider.get_map_api()
        self.sample_points = self.map_api.map_sample_points
        self.lane_speed_limit_mps = self.map_api.speed_limit_mps
        self._fill_polygon, self._fill_convex_polygon = fill_polygon, fill_convex_polygon

    # ---- the observation of one CBV ----------------------------------------------------------------------------------------------------
    def build_feature(self, center, center_nearby_agents: Sequence, mode: str = None):
        """(PlutoFeature in the CBV's frame, route_ids, reference_lines, route_elements, interaction_wp), pluto_feature_builder.py:65-122."""
        steps = self.history_samples + 1
        history = list(self.provider.get_history_state(center))[-steps:]
        now = history[-1]
        data: Dict[str, Any] = {"current_state": self.process_current_agent_state(now)}
        own = self.get_center_agent_features(history, center)
        others, other_ids, other_polygons = self.get_agent_features(now.center, center_nearby_agents, steps)
        data["agent"] = {k: np.concatenate([own[k][None, ...], others[k]], axis=0) for k in others}
        data["agent_tokens"] = ["ego"] + other_ids
        data["static_objects"] = self.get_static_objects_features(now)
        reference_lines, route_elements, route_ids, interaction_wp = self.route_planner.build_reference_line(center, now, self.radius)
        data["map"], _ = self.get_map_features(self.map_api, _map_query_point(now.center.x, now.center.y), set(route_ids["road_ids"]), self.radius)
        data["reference_line"] = self.get_reference_line_features(own, reference_lines)
        if mode is not None and mode.startswith("train"):
            data["cost_maps"] = self.build_cost_maps(now.rear_axle.array, now.rear_axle.heading, others, other_polygons)
        return PlutoFeature.normalize(data, first_time=True, radius=self.radius), route_ids, reference_lines, route_elements, interaction_wp

    @staticmethod
    def process_current_agent_state(state) -> np.ndarray:
        """(x, y, heading) of the rear axle, longitudinal velocity / acceleration, steering angle, yaw rate (:124-138)."""
        d = state.dynamic_car_state
        return np.array([state.rear_axle.array[0], state.rear_axle.array[1], state.rear_axle.heading, d.rear_axle_velocity_2d.x,
                         d.rear_axle_acceleration_2d.x, state.tire_steering_angle, d.angular_velocity], dtype=np.float64)

    def get_center_agent_features(self, history_states: List, agent) -> Dict[str, np.ndarray]:
        """The CBV's own history row (:140-180): rear-axle pose, velocity / acceleration rotated into each state's own frame, footprint."""
        T = len(history_states)
        ext = agent.bounding_box.extent
        shape = np.empty((T, 2), dtype=np.float64)
        shape[:] = np.array([ext.y * 2., ext.x * 2.])                       # width, length
        return {
            "position": np.array([s.rear_axle.array for s in history_states], dtype=np.float64).reshape(T, 2),
            "heading": np.array([s.rear_axle.heading for s in history_states], dtype=np.float64),
            "velocity": np.array([_to_local(s.dynamic_car_state.rear_axle_velocity_2d.array, s.rear_axle.heading) for s in history_states],
                                 dtype=np.float64).reshape(T, 2),
            "acceleration": np.array([_to_local(s.dynamic_car_state.rear_axle_acceleration_2d.array, s.rear_axle.heading)
                                      for s in history_states], dtype=np.float64).reshape(T, 2),
            "shape": shape,
            "category": np.array(AGENT_TYPES.index("EGO"), dtype=np.int8),
            "valid_mask": np.ones(T, dtype=np.bool_),
        }

    def get_agent_features(self, query_xy, center_nearby_agents: Sequence, history_horizon_samples: int) -> Tuple[Dict, List, List]:
        """Neighbour rows ordered by their CURRENT distance to the CBV, at most max_agent of them (:182-251).  A neighbour's states fill
        the steps from 0 upwards (a younger actor leaves the LAST steps invalid, as in the reference); its category and footprint polygon
        are read off the state at the last step, so an actor with a short history keeps category 0 / polygon None.
        (With more than max_agent neighbours the reference's loop raises KeyError on the first one beyond the cut; here those are skipped.)"""
        N, T = min(len(center_nearby_agents), self.max_agent), history_horizon_samples
        feat = {"position": np.zeros((N, T, 2), dtype=np.float64), "heading": np.zeros((N, T), dtype=np.float64),
                "velocity": np.zeros((N, T, 2), dtype=np.float64), "shape": np.zeros((N, T, 2), dtype=np.float64),
                "category": np.zeros((N,), dtype=np.int8), "valid_mask": np.zeros((N, T), dtype=np.bool_)}
        if N == 0:
            return feat, [], []
        ids = np.array([a.id for a in center_nearby_agents])
        here = np.array([self.provider.get_current_state(a).center.array for a in center_nearby_agents])
        order = np.argsort(np.linalg.norm(here - query_xy.array[None, :], axis=1))[: self.max_agent]
        kept = ids[order]
        row_of = {i: r for r, i in enumerate(kept)}
        polygon = [None] * N
        for a in center_nearby_agents:
            r = row_of.get(a.id)
            if r is None:
                continue
            states = list(self.provider.get_history_state(a))[-T:]
            for t, st in enumerate(states):
                ag = st.agent_state
                feat["position"][r, t] = ag.center.array
                feat["heading"][r, t] = ag.center.heading
                feat["velocity"][r, t] = ag.velocity.array
                feat["shape"][r, t] = (ag.box.width, ag.box.length)
                feat["valid_mask"][r, t] = True
            if len(states) == T:
                last = states[-1].agent_state
                feat["category"][r] = AGENT_TYPES.index(_name(last.tracked_object_type))
                polygon[r] = last.box.geometry
        return feat, list(kept), polygon

    def get_static_objects_features(self, current_state) -> Dict[str, np.ndarray]:
        """No static objects in the CARLA setup (:253-263): empty arrays of the schema's shapes."""
        none = np.zeros((0, 6), dtype=np.float64)
        return {"position": none[:, :2], "heading": none[:, 2], "shape": none[:, 3:5], "category": none[:, -1],
                "valid_mask": np.zeros(0, dtype=np.bool_)}

    def get_map_features(self, map_api, query_xy, road_ids: set, radius: float) -> Tuple[Dict[str, np.ndarray], List[int]]:
        """Lanes, lane connectors and crosswalks within `radius` as three polylines each (centre line, left and right edge) of P points:
        point positions / segment vectors / orientations, a pose per polygon, type, on-route flag, traffic-light status (all green),
        speed limit (:265-363).  Row order = lanes, lane connectors, crosswalks, as the map API lists them."""
        objects = map_api.query_proximal_map_data(query_xy, radius)
        lanes, connectors, crosswalks = _layer(objects, "LANE"), _layer(objects, "LANE_CONNECTOR"), _layer(objects, "CROSSWALK")
        rows = lanes + connectors + crosswalks
        tokens = {int(o.token_id): i for i, o in enumerate(rows)}
        kinds = [0] * len(lanes) + [1] * len(connectors) + [2] * len(crosswalks)         # POLYGON_TYPES indices
        M, P, mid = len(rows), self.sample_points, int(self.sample_points / 2)
        f = {"point_position": np.zeros((M, 3, P, 2), dtype=np.float64), "point_vector": np.zeros((M, 3, P, 2), dtype=np.float64),
             "point_orientation": np.zeros((M, 3, P), dtype=np.float64), "point_side": np.zeros((M, 3), dtype=np.int8),
             "polygon_center": np.zeros((M, 3), dtype=np.float64), "polygon_position": np.zeros((M, 2), dtype=np.float64),
             "polygon_orientation": np.zeros(M, dtype=np.float64), "polygon_type": np.zeros(M, dtype=np.int8),
             "polygon_on_route": np.zeros(M, dtype=np.bool_), "polygon_tl_status": np.zeros(M, dtype=np.int8),
             "polygon_has_speed_limit": np.zeros(M, dtype=np.bool_), "polygon_speed_limit": np.zeros(M, dtype=np.float64),
             "polygon_road_block_id": np.zeros(M, dtype=np.int32)}
        for n, obj in enumerate(rows):
            i = tokens[int(obj.token_id)]                      # (a repeated token keeps the LAST row of that token, as the reference's dict does)
            edges = obj.edges
            f["point_vector"][i] = edges[:, 1:] - edges[:, :-1]
            f["point_position"][i] = edges[:, :-1]
            f["point_orientation"][i] = np.arctan2(f["point_vector"][i, :, :, 1], f["point_vector"][i, :, :, 0])
            f["point_side"][i] = np.arange(3)
            f["polygon_orientation"][i] = f["point_orientation"][i, 0, 0]
            f["polygon_type"][i] = kinds[i]
            if n < len(lanes) + len(connectors):
                line = obj.centerline
                f["polygon_center"][i] = np.concatenate([line[mid], [f["point_orientation"][i, 0, mid]]], axis=-1)
                f["polygon_position"][i] = line[0]
                f["polygon_on_route"][i] = int(obj.road_id) in road_ids
                f["polygon_tl_status"][i] = TL_GREEN                # every light is taken as green
                f["polygon_has_speed_limit"][i] = self.lane_speed_limit_mps is not None
                f["polygon_speed_limit"][i] = self.lane_speed_limit_mps
                f["polygon_road_block_id"][i] = int(obj.road_id)
            else:
                f["polygon_center"][i] = np.concatenate([edges[0, mid], [f["point_orientation"][i, 0, mid]]], axis=-1)
                f["polygon_position"][i] = edges[0, 0]
                f["polygon_tl_status"][i] = TL_UNKNOWN
        return f, list(tokens.keys())

    def get_reference_line_features(self, center_features: dict, reference_lines: List[np.ndarray]) -> Dict[str, np.ndarray]:
        """Every 4th route point, at most radius / 1 m of them (120): positions, segment vectors, orientations, valid prefix (:365-401).
        `future_projection` needs the CBV's FUTURE positions (open-loop training data); at rollout time there are none and it stays zero."""
        R, n_points = len(reference_lines), int(self.radius / 1.0)
        f = {"position": np.zeros((R, n_points, 2), dtype=np.float64), "vector": np.zeros((R, n_points, 2), dtype=np.float64),
             "orientation": np.zeros((R, n_points), dtype=np.float64), "valid_mask": np.zeros((R, n_points), dtype=np.bool_),
             "future_projection": np.zeros((R, 8, 2), dtype=np.float64)}
        if len(center_features["position"][self.history_samples + 1:]) > 0:
            raise NotImplementedError("future_projection of logged futures (shapely LineString.project / distance) is open-loop training data, "
                                      "not part of the rollout-side builder")
        for i, line in enumerate(reference_lines):
            sub = line[::4][: n_points + 1]
            n = len(sub) - 1
            f["position"][i, :n] = sub[:-1, :2]
            f["vector"][i, :n] = np.diff(sub[:, :2], axis=0)
            f["orientation"][i, :n] = sub[:-1, 2]
            f["valid_mask"][i, :n] = True
        return f

    # ---- drivable-area signed distance field (utils/cost_map_manager.py:16-117) -----------------------------------------------------------
    def build_cost_maps(self, origin, angle: float, agents: Optional[Dict[str, np.ndarray]], agents_polygon: Optional[List],
                        height: int = 200, width: int = 200, resolution: float = 0.2) -> np.ndarray:
        """(H, W, 1) float16: metres to the drivable-area boundary, positive inside, negative outside, in the CBV's rear-axle frame.
        Drivable = union of lane / lane-connector polygons minus the footprints of parked neighbours (valid >= 50 steps, moved < 1 m)."""
        from scipy import ndimage
        if self._fill_polygon is None or self._fill_convex_polygon is None:
            raise RuntimeError("cost maps need the two raster fills: pass fill_polygon=lambda m, v, val: cv2.fillPoly(m, [v], val) and "
                               "fill_convex_polygon=lambda m, v, val: cv2.fillConvexPoly(m, v, val) to PlutoFeatureBuilder")
        origin = np.asarray(origin)
        rot = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]], dtype=np.float64)
        res_hw = np.array([resolution, -resolution], dtype=np.float32)
        offset = np.array([height / 2, width / 2], dtype=np.float32)

        def pixels(polygon):
            xy = np.stack(polygon.exterior.coords.xy, axis=1)
            return np.round(np.matmul(xy - origin, rot) / res_hw + offset).astype(np.int32)

        drivable = np.zeros((height, width), dtype=np.uint8)
        speed = np.zeros((height, width), dtype=np.float32)          # (filled as the reference fills it; only the SDF is returned)
        reach = max(height, width) * resolution / 2
        objects = self.map_api.query_proximal_map_data(_map_query_point(float(origin[0]), float(origin[1])), reach)
        for obj in _layer(objects, "LANE") + _layer(objects, "LANE_CONNECTOR"):
            px = pixels(obj.polygon)
            self._fill_polygon(drivable, px, 1)
            self._fill_polygon(speed, px, obj.speed_limit_mps if obj.speed_limit_mps else 50)
        if agents is not None:
            for pos, mask, polygon in zip(agents["position"], agents["valid_mask"], agents_polygon):
                if mask.sum() < 50:
                    continue
                track = pos[mask]
                if np.linalg.norm(track[-1] - track[0]) < 1.0:
                    self._fill_convex_polygon(drivable, pixels(polygon), 0)
        sdf = ndimage.distance_transform_edt(drivable) - ndimage.distance_transform_edt(1 - drivable)
        sdf *= resolution
        return sdf[:, :, None].astype(np.float16)
