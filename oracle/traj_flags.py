"""CPU oracle (test infrastructure): the collision / off-road flags of the GRPO advantage pipeline, restated in numpy.

R/ = rift/cbv/planning/fine_tuner/rlft/traj_eval/traj_evaluator.py

PARITY UNPINNED: the reference computes these with third-party code that is not importable here -- Shapely==2.0.6
(requirements.txt:34; call sites R/:15-16,259-271) and opencv_python==4.10.0.84 (requirements.txt:21; call site R/:323-325) -- and it has
no tests or fixtures for them.  What is restated is the published behaviour of the calls the reference makes:
  * `STRtree.query(geometry)` with no predicate (shapely 2.0 API): "the integer indices of all geometries in the tree whose extents
    intersect the extent of the input geometry" -- an ENVELOPE test, closed intervals.  The reference then only asks whether the
    result is non-empty (R/:268-271), so collision == any envelope overlap.
  * the raster lookup after the mask has been drawn (R/:299-318) is plain numpy and is restated line by line; drawing the mask
    (cv2.fillPoly over HD-map polygons, R/:284-297,323-325) needs CARLA map data and stays with the caller.
"""
import numpy as np


def get_collision_matrix(center_rollout_vertices: np.ndarray, other_vehicle_rollout_vertices: np.ndarray) -> np.ndarray:
    """R/:241-275.  center: (G, Tc, 4, 2), other: (N, Ts, 4, 2) -> (G, Ts) bool (Ts is the OTHER array's step count, R/:248-249)."""
    G = center_rollout_vertices.shape[0]
    N, Ts = other_vehicle_rollout_vertices.shape[:2] if other_vehicle_rollout_vertices.size else (0, center_rollout_vertices.shape[1])
    out = np.zeros((G, Ts), dtype=np.bool_)
    if N == 0:
        return out
    for j in range(Ts):
        o = other_vehicle_rollout_vertices[:, j].astype(np.float64)              # (N, 4, 2)
        omin, omax = o.min(axis=1), o.max(axis=1)                                # envelopes of the tree's polygons
        for i in range(G):
            e = center_rollout_vertices[i, j].astype(np.float64)
            emin, emax = e.min(axis=0), e.max(axis=0)
            disjoint = (omin[:, 0] > emax[0]) | (omax[:, 0] < emin[0]) | (omin[:, 1] > emax[1]) | (omax[:, 1] < emin[1])
            out[i, j] = bool((~disjoint).any())
    return out


def global_to_pixel(coord, origin, rot_mat, resolution_hw, offset):
    """R/:319-322."""
    coord = np.matmul(coord - origin, rot_mat)
    return coord / resolution_hw + offset


def get_off_road_matrix(rollout_center: np.ndarray, off_road_mask: np.ndarray, origin, angle: float, map_width=400, map_height=400,
                        resolution=0.5) -> np.ndarray:
    """R/:277-318 from the point where the mask exists.  rollout_center (G, T, 2); off_road_mask (H, W) uint8, 1 = not drivable."""
    resolution_hw = np.array([resolution, -resolution], dtype=np.float32)         # R/:100
    offset = np.array([map_height / 2, map_width / 2], dtype=np.float32)          # R/:101
    origin = np.asarray(origin, dtype=np.float64)
    rot_mat = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]], dtype=np.float64)   # R/:285-288
    G, T, _ = rollout_center.shape
    all_points = rollout_center.reshape(-1, 2)
    pixel_coords = global_to_pixel(all_points, origin, rot_mat, resolution_hw, offset)
    pixel_indices = np.round(pixel_coords).astype(int)
    valid_x = (pixel_indices[:, 0] >= 0) & (pixel_indices[:, 0] < map_width)
    valid_y = (pixel_indices[:, 1] >= 0) & (pixel_indices[:, 1] < map_height)
    valid = valid_x & valid_y
    flags = np.zeros(all_points.shape[0], dtype=np.bool_)
    flags[valid] = off_road_mask[pixel_indices[valid, 1], pixel_indices[valid, 0]] == 1
    return flags.reshape(G, T)
