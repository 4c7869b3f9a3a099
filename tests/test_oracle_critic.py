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


def test_rtr_objective_oracle_matches_reference_fixture():
    """oracle/critic.rtr_loss_and_grads against tests/golden/rtr.npz = the reference's own rtr_trainer._compute_objectives / get_ppo_loss /
    get_teacher_loss / generate_target_label + CriticPPO on H.rtr_inputs: total loss, its PPO and teacher parts, d loss / d logits and the
    critic gradients (lambda_rl = 5 on the PPO part, value loss included)."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "rtr.npz"))
    sd, inp = H.critic_weights(), H.rtr_inputs()
    loss, ppo, teacher, dprob, grads = ocr.rtr_loss_and_grads(sd, inp["probability"], inp["r_pad"], inp["trajectory"], inp["teacher_infos"],
                                                              inp["state"], inp["action_mode"], inp["advantage"], inp["old_log_prob"],
                                                              inp["reward_sum"])
    assert abs(float(loss) - float(gold["loss"])) < 1e-5 and abs(float(ppo) - float(gold["ppo_loss"])) < 1e-6
    assert abs(float(teacher) - float(gold["teacher_loss"])) < 1e-6
    assert np.abs(dprob.numpy() - gold["dloss_dprob"]).max() < 1e-6
    n = 0
    for k in ocr.CRITIC_KEYS:
        assert np.abs(grads[k].numpy() - gold["grad." + k]).max() < 1e-5, k
        n += 1
    assert n == 10


def test_traj_flag_oracle_envelope_and_raster_semantics():
    """oracle/traj_flags.py (unpinned restatement of the STRtree envelope query and the raster lookup): touching envelopes collide,
    an empty neighbour list gives zeros with the candidate's step count, pixels round half to even and points off the raster are
    not off-road."""
    from oracle import traj_flags as otf
    sq = lambda x0, y0, w: np.array([[x0 + w, y0 + w], [x0, y0 + w], [x0, y0], [x0 + w, y0]], dtype=np.float64)
    center = np.stack([sq(0, 0, 2), sq(10, 10, 2)])[:, None].repeat(3, 1).astype(np.float32)          # (2, 3, 4, 2)
    other = np.stack([np.stack([sq(2, 0, 1), sq(2.0001, 0, 1), sq(-5, -5, 1)])])                       # (1, 3, 4, 2)
    col = otf.get_collision_matrix(center, other)
    assert col.tolist() == [[True, False, False], [False, False, False]]
    assert otf.get_collision_matrix(center, np.zeros((0, 3, 4, 2))).shape == (2, 3)
    # a diagonal footprint whose envelope overlaps although the polygons do not: still a "collision" (envelope query)
    diag = np.array([[[3.0, 0.0], [0.0, 3.0], [-0.1, 2.9], [2.9, -0.1]]])[None]                          # thin diagonal sliver
    corner = sq(0, 0, 0.5)[None, None].astype(np.float32)
    assert otf.get_collision_matrix(corner, diag)[0, 0]
    mask = np.zeros((400, 400), dtype=np.uint8)
    mask[200, 10] = 1
    mask[200, 12] = 1
    pts = np.array([[[-94.75, 0.0], [-94.25, 0.0], [-94.5, 0.0], [1e4, 0.0]]], dtype=np.float32)   # pixels 10.5 -> 10, 11.5 -> 12, 11, outside
    off = otf.get_off_road_matrix(pts, mask, origin=(0.0, 0.0), angle=0.0)
    assert off.tolist() == [[True, True, False, False]]


def test_traj_flag_oracle_against_hand_derived_known_answers():
    """oracle/traj_flags.py against tests/golden/traj_flags_kat.json: 20 collision and 22 off-road cases derived by hand from the
    documented semantics of what the reference calls (STRtree.query without predicate = inclusive envelope intersection; np.round half
    to even; y axis flipped by the negative resolution; outside the raster = not off road) -- each case carries its reasoning."""
    from oracle import traj_flags as otf
    col, mask, off = H.traj_flag_kat()
    assert len(col) >= 20 and len(off) >= 20
    for name, center, others, want in col:
        assert bool(otf.get_collision_matrix(center, others)[0, 0]) == want, name
    for name, pt, origin, heading, want in off:
        assert bool(otf.get_off_road_matrix(pt, mask, origin, heading)[0, 0]) == want, name


def test_off_road_oracle_matches_reference_fixture():
    """oracle/traj_flags.get_off_road_matrix against tests/golden/off_road.npz -- the output of the reference's own
    TrajEvaluator.get_off_road_matrix + global_to_pixel (traj_evaluator.py:277-322) on the mask its fill loop left behind: bit-exact on
    every point, including exact half-pixel ties (np.round: half to even), raster-edge and far-outside points, rotated poses and a
    non-square raster (the reference's offset [map_height / 2, map_width / 2] is applied to (x, y))."""
    from oracle import traj_flags as otf
    cases = H.off_road_cases()
    assert [c[0] for c in cases] == ["axis_ties", "rotated", "rotated_far", "non_square"]
    for name, mask, pts, (x, y, heading), want in cases:
        got = otf.get_off_road_matrix(pts, mask, (x, y), heading, map_width=mask.shape[1], map_height=mask.shape[0])
        assert got.dtype == np.bool_ and np.array_equal(got, want), name
        assert 0.1 < want.mean() < 0.6, name
    # the tie rows of the axis-aligned case really are ties: pixel x = 10.5 reads column 10, 11.5 reads 12, 399.5 falls off the raster
    name, mask, pts, (x, y, heading), want = cases[0]
    px = (pts[..., 0].astype(np.float64) - x) / 0.5 + 200.0
    assert (np.abs(px - np.floor(px) - 0.5) < 1e-12).sum() >= 36


def test_other_vehicle_rollout_oracle_matches_reference_fixture():
    """oracle/traj_flags.get_other_vehicle_rollout against tests/golden/other_vehicles.npz, produced by running the reference's own
    TrajEvaluator.get_other_vehicle_rollout + KinematicBicycleModel + GlobalConfig on the seeded actors of H.other_vehicle_inputs."""
    from oracle import traj_flags as otf
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "other_vehicles.npz"))
    assert np.allclose(gold["config"], [otf.CFG[k] for k in (
        "time_step", "front_wheel_base", "rear_wheel_base", "steering_gain", "brake_acceleration", "throttle_acceleration",
        "slow_speed_extent_factor_ego", "extent_other_vehicles_bbs_speed_threshold", "high_speed_min_extent_y_other_vehicle",
        "high_speed_extent_y_factor_other_vehicle", "high_speed_min_extent_x_other_vehicle",
        "high_speed_min_extent_x_other_vehicle_lane_change")], rtol=0, atol=0)
    inp = H.other_vehicle_inputs()
    got = otf.get_other_vehicle_rollout(**inp)
    assert got.shape == gold["vertices"].shape == (7, 40, 4, 2) and got.dtype == np.float64
    assert np.array_equal(got, gold["vertices"])                                  # same numpy operations in the same order: bit-exact
    assert otf.get_other_vehicle_rollout(*[np.zeros((0,))] * 4, np.zeros((0, 3)), np.zeros((0,)), np.zeros((0, 2))).shape == tuple(gold["vertices_empty_shape"])


def test_sft_teacher_objective_oracle_matches_reference_fixture():
    """oracle/losses.sft_loss against tests/golden/sft.npz, the output of the reference's own _compute_objectives / get_teacher_loss /
    generate_target_label + sft/utils.global_to_local + PIDController.batch_control_pid on H.sft_inputs: loss, d loss / d logits and the
    integer target label (best reference line of the policy, mode of the teacher) bit-exact."""
    from oracle import losses
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "sft.npz"))
    inp = H.sft_inputs()
    prob = inp["probability"].clone().requires_grad_(True)
    loss, best_r, m_idx = losses.sft_loss(prob * 1.0, inp["r_pad"], inp["trajectory"], inp["teacher_infos"])
    loss.backward()
    assert np.array_equal(best_r.numpy(), gold["target_r"]) and np.array_equal(m_idx.numpy(), gold["target_m"])
    assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-6
    assert np.abs(prob.grad.numpy() - gold["dloss_dprob"]).max() < 1e-7
