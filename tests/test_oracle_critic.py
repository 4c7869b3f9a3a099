"""The PPO critic / full PPO objective restatement against the fixture generated from the reference's CriticPPO and
get_ppo_loss (tests/golden/ppo_critic.npz)."""
import os

import numpy as np
import torch

from oracle import critic as ocr
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ppo_critic.npz")


def test_critic_and_ppo_loss_match_reference_fixture():
    gold = np.load(GOLD)
    sd, inp = H.critic_weights(), H.critic_inputs()
    v = ocr.critic_forward(sd, inp["state"])
    assert np.abs(v.numpy() - gold["value"]).max() < 1e-6
    loss, vloss, dprob, grads, _ = ocr.ppo_loss_and_grads(sd, inp["probability"], inp["r_pad"], inp["state"], inp["action_mode"],
                                                          inp["advantage"], inp["old_log_prob"], inp["reward_sum"])
    assert abs(float(loss) - float(gold["loss"])) < 1e-6 and abs(float(vloss) - float(gold["value_loss"])) < 1e-6
    assert np.abs(dprob.numpy() - gold["dprobability"]).max() < 1e-7
    for k in ocr.CRITIC_KEYS:
        assert np.abs(grads[k].numpy() - gold["grad." + k]).max() < 1e-6, k
