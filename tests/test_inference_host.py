"""Rollout-side helpers (candidate trimming with integer indices, frame transforms, the waypoint PID with its error windows)
against the fixture produced by the reference's own _trim_candidates / _global_to_local / PIDController.control_pid."""
import os

import numpy as np

from rift_amd.planning.pluto import inference as inf
from rift_amd.planning.pluto.controller.pid_controller import PIDController
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden", "inference.npz")


def test_trim_transform_and_pid_match_reference_fixture():
    gold = np.load(GOLD)
    inp = H.inference_inputs()
    traj, score, orig, n_ref, n_mode = inf.trim_candidates(inp["candidates"].copy(), inp["probability"].copy(), inp["origin"], inp["angle"],
                                                           inp["ref_free"].copy(), topk=10)
    assert np.array_equal(orig.astype(np.int64), gold["trim.orig"])            # integer candidate indices: bit-exact
    assert np.abs(traj - gold["trim.traj"]).max() < 1e-12 and np.abs(score - gold["trim.score"]).max() < 1e-7
    assert (n_ref, n_mode) == (4, 12)
    best = int(score.argmax())
    local = inf.global_to_local(traj[best, 1:], inp["origin"], inp["angle"])
    assert np.abs(local - gold["local"]).max() < 1e-12
    r, m = inf.action_mode_of(orig[0], n_mode)
    assert (r, m) == (int(orig[0]) // 12, int(orig[0]) % 12)
    ctrl = PIDController()
    for k in range(6):
        got = ctrl.control_pid(local[:, :2] * (1.0 + 0.05 * k), 3.0 + 0.5 * k)
        assert np.abs(np.array([float(v) for v in got]) - gold["actions"][k]).max() < 1e-12, k
