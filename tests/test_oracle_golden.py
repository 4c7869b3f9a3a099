"""The CPU oracle against golden vectors produced by the imported reference
(tests/golden/gen_golden.py).  Tolerances: fp32 re-association only."""
import numpy as np
import pytest
import torch

from oracle import losses, pluto_ref
from tests import helpers as H

TOL = 2e-4  # abs, on O(1)-O(10) activations after 20+ fp32 layers


@pytest.mark.parametrize("case", ["small", "full"])
def test_forward_eval(case):
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    out, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    valid_agent = data["agent"]["valid_mask"].any(-1)
    kpm = torch.cat([~valid_agent, ~data["map"]["valid_mask"].any(-1)], dim=-1)
    for k in ("x_agent", "x_polygon"):
        assert H.max_err(taps[k], gold["eval.tap." + k]) < TOL, k
    for k in ["enc_out"] + [f"enc{i}" for i in range(4)]:
        # padded tokens differ by construction (never read as keys); compare valid tokens
        a = taps[k][~kpm].numpy()
        b = torch.from_numpy(gold["eval.tap." + k])[~kpm].numpy()
        assert H.max_err(a, b) < TOL, k
    r_valid = data["reference_line"]["valid_mask"].any(-1)
    for k in [f"dec{i}" for i in range(4)] + ["q_final"]:
        assert H.max_err(taps[k][r_valid], torch.from_numpy(gold["eval.tap." + k])[r_valid]) < TOL, k
    assert H.max_err(out["probability"], gold["eval.probability"]) < TOL
    assert H.max_err(out["hidden"], gold["eval.hidden"]) < TOL
    assert H.max_err(out["trajectory"][r_valid], torch.from_numpy(gold["eval.trajectory"])[r_valid]) < TOL
    assert H.max_err(out["prediction"][valid_agent[:, 1:]],
                     torch.from_numpy(gold["eval.prediction"])[valid_agent[:, 1:]]) < TOL
    assert H.max_err(out["ref_free_trajectory"], gold["eval.ref_free_trajectory"]) < TOL
    assert H.max_err(out["output_trajectory"], gold["eval.output_trajectory"]) < 1e-3


@pytest.mark.parametrize("case", ["small", "full"])
def test_forward_train_bn(case):
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    out, stats, _ = pluto_ref.planning_model_forward(sd, H.clone_tree(data), train_bn=True)
    assert H.max_err(out["probability"], gold["trainbn.probability"]) < TOL
    assert H.max_err(out["hidden"], gold["trainbn.hidden"]) < TOL
    n = 0
    for k, v in stats.items():
        g = gold["trainbn.stat." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g)
        else:
            assert H.max_err(v, g) < 1e-4 * max(1.0, float(np.abs(g).max())), k
        n += 1
    assert n == 12  # 4 BatchNorm layers x (mean, var, count)


@pytest.mark.parametrize("case", ["small", "full"])
@pytest.mark.parametrize("kind", ["rift", "grpo", "reinforce", "ppo"])
def test_losses_and_grads(case, kind):
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    q_final = torch.from_numpy(gold["eval.tap.q_final"])
    b = H.clone_tree(batch)
    if kind == "ppo":
        b["advantage_torch"] = torch.from_numpy(gold["ppo.advantage"])
    loss, grads, prob = losses.pi_head_loss_and_grads(sd, q_final, kind, b, r_pad)
    assert abs(float(loss) - float(gold[f"{kind}.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"{kind}.grad.{k}"]
        assert H.max_err(g, ref) < 1e-5 + 1e-4 * float(np.abs(ref).max()), k
    if kind == "reinforce":
        _, r_idx, m_idx = losses.reinforce_loss(prob, r_pad, b["return_torch"])
        assert np.array_equal(r_idx.numpy(), gold["reinforce.r_idx"])  # bit-exact integer indices
        assert np.array_equal(m_idx.numpy(), gold["reinforce.m_idx"])


@pytest.mark.parametrize("case", H.SHAPE_CASES)
def test_shape_cases_against_the_reference(case):
    """The oracle against the REFERENCE at the shapes that run the other kernel variants (round-4 review, weak #10): BASELINE configs[4]
    (128 agents x 40 polygons, R 8..16), the shapes train_cbv produces (49 x 60) and scenes with static objects (S > 0:
    static_objects_encoder.py:17-40, which no other fixture exercises).  Eval forward with hooks, train-mode BatchNorm forward, RIFT loss
    and pi_head gradients."""
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    if case == "static":
        assert data["static_objects"]["valid_mask"].shape[1] >= 3 and not bool(data["static_objects"]["valid_mask"].all())
    out, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    kpm = H.token_padding(data)
    r_valid = data["reference_line"]["valid_mask"].any(-1)
    for k in ("x_agent", "x_polygon"):
        assert H.max_err(taps[k], gold["eval.tap." + k]) < TOL, k
    assert H.max_err(taps["enc_out"][~kpm], torch.from_numpy(gold["eval.tap.enc_out"])[~kpm]) < TOL
    assert H.max_err(taps["q_final"][r_valid], torch.from_numpy(gold["eval.tap.q_final"])[r_valid]) < TOL
    assert H.max_err(out["probability"], gold["eval.probability"]) < TOL
    assert H.max_err(out["hidden"], gold["eval.hidden"]) < TOL
    assert H.max_err(out["ref_free_trajectory"], gold["eval.ref_free_trajectory"]) < TOL
    if "eval.trajectory" in gold:
        assert H.max_err(out["trajectory"][r_valid], torch.from_numpy(gold["eval.trajectory"])[r_valid]) < TOL
    out_t, _, _ = pluto_ref.planning_model_forward(sd, H.clone_tree(data), train_bn=True)
    assert H.max_err(out_t["probability"], gold["trainbn.probability"]) < TOL
    assert H.max_err(out_t["hidden"], gold["trainbn.hidden"]) < TOL
    loss, grads, _ = losses.pi_head_loss_and_grads(sd, torch.from_numpy(gold["eval.tap.q_final"]), "rift", H.clone_tree(batch), ~r_valid)
    assert abs(float(loss) - float(gold["rift.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"rift.grad.{k}"]
        assert H.max_err(g, ref) < 1e-5 + 1e-4 * float(np.abs(ref).max()), k
