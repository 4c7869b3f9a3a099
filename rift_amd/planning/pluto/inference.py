"""Rollout-side inference helpers: what RIFTPluto.get_action / _get_action does between the model
forward and the CARLA control call (rift_pluto.py:28-161, pluto.py:196-300), without CarlaDataProvider: the caller passes the CBV's
rear-axle pose and speed.  The model forward is the HIP engine in eval mode (all outputs); the per-CBV post-processing is a
few dozen candidates per tick and stays on the host in numpy / float64, as in the reference."""
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from scipy.special import softmax

from rift_amd.planning.pluto.controller.pid_controller import PIDController


def trim_candidates(candidate_trajectories: np.ndarray, probability: np.ndarray, origin: np.ndarray, angle: float,
                    ref_free_trajectory: Optional[np.ndarray] = None, topk: int = 10):
    """pluto.py:196-247: top-k candidates by logit (stable descending argsort), softmax over the kept logits, optional ref-free
    candidate with score 0.25 and index -1, rotation + translation to the global frame, first point duplicated in front.
    Returns (trajectories (k[+1], T+1, 3), scores, original flat indices int64, n_ref, n_mode)."""
    n_ref, n_mode, T, C = candidate_trajectories.shape
    cand = candidate_trajectories.reshape(-1, T, C)
    flat = probability.reshape(-1)
    order = np.argsort(-flat)
    traj = cand[order][:topk]
    score = softmax(flat[order][:topk])
    orig = np.arange(n_ref * n_mode)[order][:topk]
    if ref_free_trajectory is not None:
        traj = np.concatenate([traj, ref_free_trajectory[None, ...]], axis=0)
        score = np.concatenate([score, [0.25]], axis=0)
        orig = np.concatenate([orig, [-1]], axis=0)
    rot = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
    traj[..., :2] = np.matmul(traj[..., :2], rot) + origin
    traj[..., 2] += angle
    traj = np.concatenate([traj[..., 0:1, :], traj], axis=-2)
    return traj, score, orig, n_ref, n_mode


def global_to_local(global_trajectory: np.ndarray, origin: np.ndarray, angle: float) -> np.ndarray:
    """pluto.py:262-279: re-anchor the first point on the rear axle, rotate into the vehicle frame."""
    delta = origin - global_trajectory[0, :2]
    pos = global_trajectory[..., :2] + delta
    rot = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
    position = np.matmul(pos - origin, rot)
    heading = global_trajectory[..., 2] - angle
    return np.concatenate([position, heading[..., None]], axis=-1)


def action_mode_of(orig_index: int, n_mode: int) -> Tuple[int, int]:
    """(r_idx, m_idx) of a flat candidate index (rlft_pluto.py:174-176); integer, bit-exact."""
    return int(orig_index) // n_mode, int(orig_index) % n_mode


class PlutoInference:
    """Eval-mode policy step for a batch of CBVs: HIP forward with every output, then per-CBV candidate selection and PID control."""

    def __init__(self, model, topk: int = 10, use_ref_free: bool = True):
        self.model, self.topk, self.use_ref_free = model, topk, use_ref_free
        self.controllers: Dict = {}

    @torch.no_grad()
    def forward(self, feature_data: Dict) -> Dict[str, torch.Tensor]:
        was_training, need = self.model.training, self.model.need_traj
        self.model.eval()
        self.model.need_traj = True
        try:
            return self.model(feature_data)
        finally:
            self.model.need_traj = need
            self.model.train(was_training)

    def act(self, outputs: Dict[str, torch.Tensor], index: int, cbv_id, origin: np.ndarray, angle: float, speed: float):
        """-> ((throttle, steer, brake), chosen global trajectory (79, 3), candidates, scores, flat indices)."""
        cand = outputs["candidate_trajectories"][index].cpu().numpy().astype(np.float64)
        prob = outputs["probability"][index].cpu().numpy()
        rf = None
        if self.use_ref_free and "output_ref_free_trajectory" in outputs:
            rf = outputs["output_ref_free_trajectory"][index].cpu().numpy().astype(np.float64)
        cands, score, orig, _, n_mode = trim_candidates(cand, prob, np.asarray(origin, dtype=np.float64), float(angle), rf, self.topk)
        best = int(score.argmax())
        trajectory = cands[best, 1:]
        local = global_to_local(trajectory, np.asarray(origin, dtype=np.float64), float(angle))
        ctrl = self.controllers.setdefault(cbv_id, PIDController())
        return ctrl.control_pid(local[:, :2], float(speed)), trajectory, cands, score, orig
