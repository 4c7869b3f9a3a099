"""Advantage / return oracle against reference-generated golden vectors (CPU)."""
import os

import numpy as np
import torch

from oracle import advantage as adv
from tests import helpers as H

GOLD = dict(np.load(os.path.join(H.GOLDEN, "advantage.npz")))


def test_gae_and_normalisation():
    i = H.advantage_inputs()
    a = adv.get_advantages_gae(i["rewards"], i["undones"], i["values"], i["next_values"], i["unterminated"])
    assert a.dtype == torch.float32
    assert np.array_equal(a.numpy(), GOLD["gae"])          # same sequential fp64 recurrence -> bit-exact
    n = adv.normalize_advantage(a)
    assert H.max_err(n, GOLD["gae_normalized"]) < 1e-6


def test_discounted_return():
    i = H.advantage_inputs()
    r = adv.compute_return(i["rewards"], i["dones"])
    assert r.dtype == torch.float64
    assert np.array_equal(r.numpy(), GOLD["returns"])


def test_rollout_return_and_group_advantage():
    i = H.advantage_inputs()
    ret = adv.rollout_return(i["delta_dis"], i["delta_angle"], i["speed"], i["acc"], i["ang_vel"], i["ang_acc"],
                             i["collision"], i["off_road"])
    # the golden was produced under numpy 2.x scalar promotion (float32-weak python floats); the oracle follows
    # the reference environment's numpy 1.24 promotion -> agreement to float32 rounding of single reward terms
    assert np.max(np.abs(ret - GOLD["rollout_return"]) / (1 + np.abs(GOLD["rollout_return"]))) < 2e-6
    z = adv.group_zscore(ret)
    assert np.max(np.abs(z - GOLD["group_advantage"])) < 1e-5
    assert abs(z.mean()) < 1e-12 and abs(z.std() - 1) < 1e-4


def test_warmup_cos_lr_table():
    lrs = [adv.warmup_cos_lr(e, 1e-4, 0.9e-4, 3, 16) for e in range(16)]
    assert np.allclose(lrs, GOLD["warmup_cos_lr"], rtol=0, atol=1e-18)
