"""Candidate rollout oracle against reference-generated golden vectors (CPU): bit-exact, since the oracle
runs the same PyTorch-CPU fp32 operations in the same order as the reference."""
import os

import numpy as np
import torch

from oracle import rollout as orl
from tests import helpers as H

GOLD = dict(np.load(os.path.join(H.GOLDEN, "rollout.npz")))


def test_ref_line_info_and_rollout_two_consecutive_calls():
    ro = orl.Rollout()
    for call, seed in enumerate((777, 778)):
        traj, ref_pos, ref_ang, st = H.rollout_inputs(seed)
        t40 = traj[:, :, :40, :]
        dd, da, ci = orl.ref_line_info(t40, ref_pos, ref_ang)
        assert np.array_equal(dd.numpy(), GOLD[f"c{call}.delta_dis"])
        assert np.array_equal(da.numpy(), GOLD[f"c{call}.delta_angle"])
        assert ci.min() >= 0 and all(int(ci[r * 12:(r + 1) * 12].max()) < len(ref_pos[r]) for r in range(3))
        gpos, ghead = orl.to_global(t40, torch.tensor(st["pos"]), torch.tensor(st["heading"]))
        res = ro.propagate(gpos, ghead, st["speed"], st["width"], st["length"])
        for k in ("center", "angle", "speed", "acc", "ang_vel", "ang_acc", "vertices"):
            assert np.array_equal(res[k].numpy(), GOLD[f"c{call}.{k}"]), (call, k)   # PID state carried over on call 1
        assert res["closest_index"].shape == (36, 79) and res["aim_idx"].shape == (36, 79)
