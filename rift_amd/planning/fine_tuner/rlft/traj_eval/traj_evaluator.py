"""TrajEvaluator -- device mirror of the GRPO group-advantage pipeline
(fine_tuner/rlft/traj_eval/traj_evaluator.py:82-475), candidate side.

    get_grpo_advantage = ref-line deviation (rift_ref_line_info) -> closed-loop PID + bicycle rollout
    (rift_rollout, persistent PID state as in the reference) -> discounted dense-reward return
    (rift_rollout_return) -> group z-score (rift_group_advantage, ddof 0, +1e-5).

The collision / off-road flags (traj_evaluator.py:160-331) come either from the caller (the reference's own shapely / raster code)
or from the device: `other_vehicle_vertices` (N, 40, 4, 2) -- the forecast footprints of get_other_vehicle_rollout, which needs CARLA
actors (or `nearby_actor_states`, the raw actor readings, forecast by rift_other_vehicle_rollout) -- gives the collision matrix
through rift_collision_matrix (the reference's STRtree query is an envelope test), and
`off_road_mask` + `center_pose` -- the raster the reference draws from the HD map with cv2.fillPoly -- gives the off-road matrix
through rift_off_road_matrix.
"""
from typing import Dict, List, Optional

import numpy as np
import torch


class TrajEvaluator:
    def __init__(self, engine, dt: float = 0.1, num_frames: int = 40, pid_capacity: int = 256):
        assert abs(dt - 0.1) < 1e-9 and num_frames == 40, "kernels are built for dt = 0.1 s, 40 evaluated frames"
        self.engine, self.dt, self.num_frames = engine, dt, num_frames
        self.pid_state = engine.new_pid_state(pid_capacity)     # BatchPIDTorch state: never reset (reference behaviour)
        self.last_rollout: Optional[Dict[str, torch.Tensor]] = None

    def get_grpo_advantage(self, center_state, trajectories: torch.Tensor, ref_line_pos: List[torch.Tensor],
                           ref_line_angle: List[torch.Tensor], collision_matrix=None, off_road_matrix=None, gamma: float = 0.98,
                           other_vehicle_vertices=None, off_road_mask=None, center_pose=None, nearby_actor_states=None, to_host: bool = True):
        """center_state: (x, y, heading, speed, width, length) of the CBV rear axle / footprint.
        trajectories: (R, M, 80, 6) raw model output of the valid reference lines.
        collision_matrix (G, >=40) / off_road_matrix (G, >=40): bool flags per candidate and frame, OR
        other_vehicle_vertices (N, >=40, 4, 2) float64 and off_road_mask (H, W) uint8 + center_pose (x, y, heading of the footprint
        centre, get_off_road_matrix's origin / angle) to have them computed on the device from this call's rollout; OR, instead of
        other_vehicle_vertices, nearby_actor_states = dict(steer, throttle, brake, speed, location (N,3), yaw_deg, extent (N,2)) read
        off the CARLA actors, forecast on the device (get_other_vehicle_rollout)."""
        eng = self.engine
        R, M = trajectories.shape[:2]
        G = R * M
        dd, da, _ = eng.ref_line_info(trajectories, ref_line_pos, ref_line_angle, Ts=self.num_frames)
        cs = torch.as_tensor(center_state, dtype=torch.float32).view(1, 6)
        ro = eng.rollout(trajectories.reshape(G, trajectories.shape[2], 6), cs, self.pid_state)
        self.last_rollout = ro
        T = self.num_frames
        if collision_matrix is None:
            if other_vehicle_vertices is None and nearby_actor_states is not None:
                other_vehicle_vertices = eng.other_vehicle_rollout(num_future_frames=T, **nearby_actor_states)
            if other_vehicle_vertices is None:
                raise ValueError("pass collision_matrix, other_vehicle_vertices or nearby_actor_states")
            collision_matrix = eng.collision_matrix(ro["vertices"], other_vehicle_vertices, Ts=T)
        if off_road_matrix is None:
            if off_road_mask is None or center_pose is None:
                raise ValueError("pass off_road_matrix or off_road_mask + center_pose")
            off_road_matrix = eng.off_road_matrix(ro["center"], off_road_mask, center_pose[:2], float(center_pose[2]))
        ret = eng.rollout_return(dd, da, ro["speed"][:, :T].contiguous(), ro["acc"][:, :T].contiguous(),
                                 ro["ang_vel"][:, :T].contiguous(), ro["ang_acc"][:, :T].contiguous(),
                                 torch.as_tensor(collision_matrix), torch.as_tensor(off_road_matrix), gamma)
        adv = eng.group_advantage(ret.view(1, G)).view(R, M)
        # (to_host=False: the advantage stays a device tensor -- a rollout tick reads all its CBVs' columns back at once)
        return {"advantage": adv.cpu().numpy() if to_host else adv, "valid_mask": np.ones((R, M), dtype=np.bool_)}
