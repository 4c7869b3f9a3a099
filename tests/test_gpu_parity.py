"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the reference-generated
golden fixtures.  Needs a real MI355X: `pytest -m gpu`.

Tolerances (written here, per the parity contract):
  fp32 mode  (v_mfma_f32_16x16x4_f32, exact fp32 fma chains): logits / activations 1e-4 abs, losses 1e-5,
             pi_head grads 1e-5 + 1e-4 rel; integer indices bit-exact.
  bf16 mode  (v_mfma_f32_16x16x32_bf16, fp32 accumulate; the precision BASELINE.json names and bench.py's headline runs; NOT the product
             default since round 4 -- that is fp16, below).  Every MFMA operand of ~50 chained contractions
             is rounded to 8 mantissa bits, which puts ~1.5e-2 on the logits whatever the kernel does
             (tests/diagnostics/precision_study.py reproduces it on the CPU from operand rounding alone: bf16 2.3e-2 / 1.8e-2 on
             small / full, fp16 2.7e-3, three-pass bf16 3e-5).  Bars are ~2.5x what MI355X measures:
               logits 4e-2 abs (measured 1.5e-2 small, 1.1e-2 full);
               losses of the 6-scene fixture: RIFT 2.5e-4 (8.9e-5), GRPO / REINFORCE / PPO 3.5e-3 (1.0e-3 / 4.4e-4 / 1.4e-3);
               RIFT loss of the 256-scene benchmark minibatch 1e-4, north_star's bar (5.8e-5: per-scene errors average out);
               pi_head gradient of that minibatch against the fp32 oracle: ||dg|| / ||g|| 0.16, cosine 0.988
               (test_benchmark_batch_bf16_gradients_against_the_fp32_oracle).
             The loss / backward kernels themselves are fp32 / fp64 and are held to 1e-5 against the oracle evaluated on the SAME
             (HIP) pi_head input.
  fp16 mode  (v_mfma_f32_16x16x32_f16: the same kernels built for fp16 operands, `Engine(operands="fp16")` / compute_precision="fp16";
             same step time as bf16).  11 significand bits instead of 8.  Measured on MI355X (tests/diagnostics/operand_report.py):
               logits 2.5e-3 / 1.8e-3 (small / full), 4.1e-3 on the 256-scene batch                      bar 8e-3
             Two arithmetics have been measured.  Rounds 2 - 4 (fp32 VALU neighbourhood attention, two-pass LayerNorm with its affine
             part, slot-ordered encoder keys; build define RIFT_F16_R4) / round 5 (the bf16 build's arithmetic: csrc/opfmt.h -- MFMA
             neighbourhood attention, folded one-pass LayerNorm, compacted encoder rows, K = 16 attention MFMAs; 5 % faster; the default):
               RIFT loss 4.5e-5 / 2.8e-5 / 8.2e-6 | 7.2e-5 / 1.7e-5 / 1.1e-5 (small / full / 256 scenes)  bar 1e-4 (north_star)
               GRPO / REINFORCE / PPO: 256 scenes 4.0e-5 / 2.3e-5 / 1.7e-5 | 3.1e-5 / 1.7e-5 / (see the test's print)   bar 1e-4
                                       6-scene fixture 1.4e-4 / 1.4e-4 / 9.3e-5 | 7.1e-5 / 4.9e-4 / 4.1e-6   bar 7.5e-4 (was 3.5e-4)
               over 8 seeded draws per fixture shape (FP16_DRAW_BARS below): RIFT 6.3e-5 | 5.4e-5 (6 scenes) / 1.4e-4 | 8.8e-5 (2 scenes),
               GRPO / REINFORCE / PPO up to 3.5e-4 / 2.8e-4 / 1.1e-4 | 1.4e-4 / 4.6e-4 / 1.6e-4 (6 scenes) and 3.9e-4 / 1.0e-3 / 8.7e-4 |
               3.1e-4 / 6.7e-4 / 5.9e-4 (2 scenes) -- a 2-6-scene loss answers 2e-3 of logit noise with 1e-5 .. 1e-3 whatever the kernel
               does: a change of the summation order re-rolls which draw and which objective carries the maximum (REINFORCE moved up, GRPO
               down), the envelope is the same (precision_study.py FP16=1 shows no region dominating); every bar >= 1.5x its measured maximum
               pi_head gradient ||dg|| / ||g||: fixtures <= 3.7e-2, 8 draws <= 0.143, 256 scenes 0.6e-2 .. 4.5e-2 | 3.9e-2 .. 0.14 (REINFORCE:
               piecewise objective, an argmax flip in a handful of scenes)
             A caller that needs 1e-4 on every objective of a 2-6-scene batch, or 1e-4-relative gradients, sets compute_precision = "fp32".
"""
import os

import numpy as np
import pytest
import torch

from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu

from oracle import advantage as oadv, losses, pluto_ref  # noqa: E402


@pytest.fixture(scope="module")
def ffi():
    from rift_amd import _ffi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    _ffi.load_library()
    return _ffi


def err(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    d = (a.reshape(-1) - b.reshape(-1)).abs()
    assert torch.isfinite(d).all()
    return float(d.max()) if d.numel() else 0.0


SHAPES = [(70, 32, 96), (129, 128, 384), (64, 128, 512), (200, 512, 128), (33, 6, 128), (100, 129, 128),
          (257, 256, 160), (64, 27, 32), (50, 96, 64), (1, 128, 128), (12, 128, 384), (300, 128, 1), (0 + 65, 10, 128)]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_mfma_gemm_kernel(ffi, M, K, N):
    """Asymmetric random operands (a transposed C/D mapping cannot pass)."""
    eng = ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(M * 131 + K * 17 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = torch.nn.functional.linear(x, w, b)
    assert err(eng.op_linear(x, w, b, fp32=True), ref) < 2e-5
    assert err(eng.op_linear(x, w, b, fp32=False), ref) < 4e-2
    lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
    ref2 = torch.relu(torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (K,), lw, lb), w, b))
    assert err(eng.op_linear(x, w, b, ln_w=lw, ln_b=lb, act=1, fp32=True), ref2) < 5e-5
    eng.close()


MODES = ["fp32", "bf16", "fp16"]          # compute precisions: exact fp32 layer by layer | fused kernels on bf16 / fp16 MFMA operands


def _engine(ffi, mode):
    return ffi.Engine("cuda:0", operands="fp16" if mode == "fp16" else "bf16")


def _run_case(ffi, case, mode):
    if isinstance(mode, bool):
        mode = "fp32" if mode else "bf16"
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    eng = _engine(ffi, mode)
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    out = eng.forward(data, need_traj=True, fp32=mode == "fp32")
    torch.cuda.synchronize()
    return gold, batch, sd, data, eng, out


@pytest.mark.parametrize("case", ["small", "full"])
@pytest.mark.parametrize("mode", MODES)
def test_forward_eval(ffi, case, mode):
    gold, batch, sd, data, eng, out = _run_case(ffi, case, mode)
    fp32 = mode == "fp32"
    ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    tol = {"fp32": 1e-4, "bf16": 4e-2, "fp16": 8e-3}[mode]
    print(f"forward_eval[{case}, {mode}]: max |logit - reference| = {err(out['probability'], gold['eval.probability']):.3e}")
    bs, A = data["agent"]["position"].shape[:2]
    va = data["agent"]["valid_mask"].any(-1)
    kpm = torch.cat([~va, ~data["map"]["valid_mask"].any(-1)], dim=-1)
    rv = data["reference_line"]["valid_mask"].any(-1)
    N, R = kpm.shape[1], rv.shape[1]
    eo = eng.tap("enc_out").view(bs, N, 128).cpu()
    assert err(eo[~kpm], taps["enc_out"][~kpm]) < tol
    qf = eng.tap("q_final").view(bs, R, 12, 128).cpu()
    assert err(qf[rv], taps["q_final"][rv]) < {"fp32": 2e-4, "bf16": 2e-1, "fp16": 4e-2}[mode]
    assert err(out["probability"], ref["probability"]) < tol
    assert err(out["probability"], gold["eval.probability"]) < tol          # reference itself
    assert err(out["hidden"], gold["eval.hidden"]) < tol
    assert err(out["trajectory"].cpu()[rv], torch.from_numpy(gold["eval.trajectory"])[rv]) < tol
    assert err(out["prediction"].cpu()[va[:, 1:]], torch.from_numpy(gold["eval.prediction"])[va[:, 1:]]) < tol
    assert err(out["ref_free_trajectory"], gold["eval.ref_free_trajectory"]) < tol
    eng.close()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_fused_scene_encoder_on_the_112_row_layout_at_the_carla_shapes(ffi, mode, monkeypatch):
    """Round 6: scenes of 97 .. 112 token slots (what train_cbv collates: 49 agents + 60 polygons = 109) run the fused scene encoder on its
    112-row LDS layout (enc_fused.h: EncLay<112>, enc112.hip) -- out_proj per two-head chunk, the decoder's K | V^T in the dense per-head
    image its eight-key-tile variant gathers from -- instead of the two-pass enc_w_kernel.  One batch that walks its three bodies: valid counts
    109 (all), 104, 97 (seven tiles), 96, 81 (six), 80, 40, 3 (five) scattered over the slots, and a scene whose ego slot is padded (slot
    order, seven tiles).  Encoder rows, decoder queries, logits, predictions and the RIFT loss against the oracle at the mode's bars, and
    against the enc_w_kernel path (RIFT_ENC112=0) within the same bars (another summation order, not bit-identical)."""
    sd = H.weights()
    scenes = [syn.make_scene(4300 + i, num_agents=49, num_polygons=60) for i in range(9)]
    g = torch.Generator().manual_seed(78)
    counts = (109, 104, 97, 96, 81, 80, 40, 3, 90)
    for i, want in enumerate(counts):
        f = scenes[i]["feature"]
        f["agent"]["valid_mask"][:] = True
        f["map"]["valid_mask"][:] = True
        slots = torch.randperm(108, generator=g)[: 109 - want] + 1        # slots 1..108 to pad (slot 0 = the ego stays)
        for sl in slots.tolist():
            if sl < 49:
                f["agent"]["valid_mask"][sl] = False
            else:
                f["map"]["valid_mask"][sl - 49] = False
    scenes[8]["feature"]["agent"]["valid_mask"][0] = False
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    kpm, rv = H.token_padding(data), data["reference_line"]["valid_mask"].any(-1)
    assert [int((~kpm[i]).sum()) for i in range(8)] == list(counts[:8]) and kpm.shape[1] == 109
    tol = {"bf16": 4e-2, "fp16": 8e-3}[mode]
    got = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("RIFT_ENC112", wide)
        eng = _engine(ffi, mode)
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.prof_enable(True)
        out = eng.forward(data, need_traj=True)
        torch.cuda.synchronize()
        rep = eng.prof_report()
        assert ("enc_fused112_kernel" in rep) == (wide == "1") and ("enc_w_kernel" in rep) == (wide == "0") and "dec_w_kernel" in rep
        eng.prof_enable(False)
        bs, N, R = kpm.shape[0], kpm.shape[1], rv.shape[1]
        eo = eng.tap("enc_out").view(bs, N, 128).cpu()
        assert torch.isfinite(eo).all()
        for i in range(bs):
            assert err(eo[i][~kpm[i]], taps["enc_out"][i][~kpm[i]]) < tol, (wide, i)
        qf = eng.tap("q_final").view(bs, R, 12, 128).cpu()
        assert err(qf[rv], taps["q_final"][rv]) < {"bf16": 2e-1, "fp16": 4e-2}[mode]
        assert err(out["probability"], ref["probability"]) < tol
        va = data["agent"]["valid_mask"].any(-1)[:, 1:]
        assert err(out["prediction"].cpu()[va], ref["prediction"][va]) < tol
        stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        loss = float(eng.loss_finalize(stats, flat, grads).item())
        want_loss, _, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], "rift", H.clone_tree(batch), ~rv)
        assert abs(loss - float(want_loss)) < {"bf16": 3.5e-3, "fp16": 3.5e-4}[mode], (wide, loss, float(want_loss))
        eng.check_finite()
        got[wide] = (eo, out["probability"].cpu().clone())
        eng.close()
    for i in range(kpm.shape[0]):
        assert err(got["1"][0][i][~kpm[i]], got["0"][0][i][~kpm[i]]) < tol, i
    assert err(got["1"][1], got["0"][1]) < tol


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_compacted_scene_encoder_over_every_tile_count_and_the_padded_ego_fallback(ffi, mode):
    """Round 5: in the bf16 build `enc_fused_kernel` moves a scene's valid tokens to the front, runs a 5-tile body when at most 80 are valid and
    writes the rows back to their slots; the decoder reads that kernel's compacted key padding.  One batch that walks the cases: valid
    counts of 84 (all), 81, 80, 65, 64, 33, 17 and 3 tokens (scattered over the slots: the compaction is a real permutation), and a scene
    whose EGO slot is padded (the kernel keeps the slot order there).  Encoder rows and decoder queries of valid tokens / lines, logits and
    the RIFT loss against the oracle (the fp16 build keeps the slot order: the same test is its regression)."""
    sd = H.weights()
    scenes = [syn.make_scene(4100 + i) for i in range(9)]
    g = torch.Generator().manual_seed(77)
    for i, want in enumerate((84, 81, 80, 65, 64, 33, 17, 3, 70)):
        f = scenes[i]["feature"]
        f["agent"]["valid_mask"][:] = True
        f["map"]["valid_mask"][:] = True
        slots = torch.randperm(83, generator=g)[: 84 - want] + 1          # slots 1..83 to pad (slot 0 = the ego stays)
        for sl in slots.tolist():
            if sl < 64:
                f["agent"]["valid_mask"][sl] = False
            else:
                f["map"]["valid_mask"][sl - 64] = False
    scenes[8]["feature"]["agent"]["valid_mask"][0] = False                  # a padded ego slot: slot order, all six tiles
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    kpm, rv = H.token_padding(data), data["reference_line"]["valid_mask"].any(-1)
    assert [int((~kpm[i]).sum()) for i in range(8)] == [84, 81, 80, 65, 64, 33, 17, 3]
    eng = _engine(ffi, mode)
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.prof_enable(True)
    out = eng.forward(data, need_traj=True)
    torch.cuda.synchronize()
    assert "enc_fused_kernel" in eng.prof_report() and "dec_w_kernel" in eng.prof_report()
    eng.prof_enable(False)
    tol = {"bf16": 4e-2, "fp16": 8e-3}[mode]
    bs, N, R = kpm.shape[0], kpm.shape[1], rv.shape[1]
    eo = eng.tap("enc_out").view(bs, N, 128).cpu()
    assert torch.isfinite(eo).all()                                          # padded slots: zero rows or computed rows, never stale memory
    for i in range(bs):
        assert err(eo[i][~kpm[i]], taps["enc_out"][i][~kpm[i]]) < tol, i
    qf = eng.tap("q_final").view(bs, R, 12, 128).cpu()
    assert err(qf[rv], taps["q_final"][rv]) < {"bf16": 2e-1, "fp16": 4e-2}[mode]
    assert err(out["probability"], ref["probability"]) < tol
    va = data["agent"]["valid_mask"].any(-1)[:, 1:]
    assert err(out["prediction"].cpu()[va], ref["prediction"][va]) < tol
    stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng.loss_finalize(stats, flat, grads).item())
    want_loss, _, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], "rift", H.clone_tree(batch), ~rv)
    assert abs(loss - float(want_loss)) < {"bf16": 3.5e-3, "fp16": 3.5e-4}[mode]
    eng.close()


@pytest.mark.parametrize("case", H.SHAPE_CASES)
@pytest.mark.parametrize("mode", MODES)
def test_forward_at_the_other_kernel_variants_shapes_against_the_reference(ffi, case, mode):
    """HIP against fixtures the REFERENCE produced (not only the oracle) at the shapes that leave the benchmark's kernels: `dense` = BASELINE
    configs[4] (168 token slots, R up to 16 -> enc_w_kernel, dec_w_kernel<., true>), `carla` = rift_pluto.yaml:35-36 (109 token slots),
    `static` = scenes with static objects (static_objects_encoder.py:17-40: the device path of engine.hip `static_token_kernel`, the
    two-dimensional Fourier embedding and the unfused token assembly -- no other test runs S > 0).  Eval forward, train-mode BatchNorm
    forward, RIFT loss + pi_head gradients (fp32: the contract's 1e-5 / 1e-4; 16-bit modes: the bars of test_forward_eval)."""
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    fp32 = mode == "fp32"
    tol = {"fp32": 1e-4, "bf16": 4e-2, "fp16": 8e-3}[mode]
    eng = _engine(ffi, mode)
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.prof_enable(True)
    out = eng.forward(data, need_traj=True, fp32=fp32)
    torch.cuda.synchronize()
    ran = set(eng.prof_report())
    eng.prof_enable(False)
    if not fp32:
        if case == "dense":
            assert "enc_w_kernel" in ran and "dec_w_kernel" in ran, sorted(ran)
        if case == "static":
            assert "static_token_kernel" in ran, sorted(ran)
    kpm, rv = H.token_padding(data), data["reference_line"]["valid_mask"].any(-1)
    bs, N, R = kpm.shape[0], kpm.shape[1], rv.shape[1]
    eo = eng.tap("enc_out").view(bs, N, 128).cpu()
    assert err(eo[~kpm], torch.from_numpy(gold["eval.tap.enc_out"])[~kpm]) < tol
    qf = eng.tap("q_final").view(bs, R, 12, 128).cpu()
    assert err(qf[rv], torch.from_numpy(gold["eval.tap.q_final"])[rv]) < {"fp32": 2e-4, "bf16": 2e-1, "fp16": 4e-2}[mode]
    print(f"{case}[{mode}]: max |logit - reference| = {err(out['probability'], gold['eval.probability']):.3e}")
    assert err(out["probability"], gold["eval.probability"]) < tol
    assert err(out["hidden"], gold["eval.hidden"]) < tol
    assert err(out["ref_free_trajectory"], gold["eval.ref_free_trajectory"]) < tol
    if "eval.trajectory" in gold:
        assert err(out["trajectory"].cpu()[rv], torch.from_numpy(gold["eval.trajectory"])[rv]) < tol
    stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng.loss_finalize(stats, flat, grads).item())
    print(f"{case}[{mode}]: |RIFT loss - reference| = {abs(loss - float(gold['rift.loss'])):.3e}")
    assert abs(loss - float(gold["rift.loss"])) < {"fp32": 1e-5, "bf16": 3.5e-3, "fp16": 3.5e-4}[mode]
    if fp32:
        for k in grads:
            ref = gold[f"rift.grad.{k}"]
            assert err(grads[k], ref) < 1e-5 + 1e-4 * float(np.abs(ref).max()), k
    out_t = eng.forward(data, train=True, no_drop=True, bn_update=False, fp32=fp32)
    assert err(out_t["probability"], gold["trainbn.probability"]) < tol
    assert err(out_t["hidden"], gold["trainbn.hidden"]) < tol
    eng.close()


@pytest.mark.parametrize("case", ["small", "full"])
@pytest.mark.parametrize("kind", ["rift", "grpo", "reinforce", "ppo"])
def test_losses_and_pi_head_grads(ffi, case, kind):
    gold, batch, sd, data, eng, out = _run_case(ffi, case, True)
    b = H.clone_tree(batch)
    if kind == "ppo":
        b["advantage_torch"] = torch.from_numpy(gold["ppo.advantage"])
    stats, flat, am = eng.loss_backward(kind, b)
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = eng.loss_finalize(stats, flat, grads)
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(gold[f"{kind}.loss"])) < 1e-5      # north_star: losses within 1e-4
    for k in grads:
        ref = gold[f"{kind}.grad.{k}"]
        assert err(grads[k], ref) < 1e-5 + 1e-4 * float(np.abs(ref).max()), k
    if kind == "reinforce":   # integer indices: bit-exact
        assert np.array_equal(am.cpu().numpy()[:, 0], gold["reinforce.r_idx"])
        assert np.array_equal(am.cpu().numpy()[:, 1], gold["reinforce.m_idx"])
    eng.close()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("kind", ["rift", "grpo", "reinforce", "ppo"])
def test_loss_kernels_bf16_trunk(ffi, kind, mode):
    """16-bit-operand trunk: end-to-end loss within the stated tolerance of the mode, and the loss / backward kernels
    exact (1e-5) against the oracle evaluated on the HIP pi_head input."""
    gold, batch, sd, data, eng, out = _run_case(ffi, "small", mode)
    b = H.clone_tree(batch)
    if kind == "ppo":
        b["advantage_torch"] = torch.from_numpy(gold["ppo.advantage"])
    stats, flat, _ = eng.loss_backward(kind, b)
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = eng.loss_finalize(stats, flat, grads)
    torch.cuda.synchronize()
    lerr = abs(float(loss.item()) - float(gold[f"{kind}.loss"]))
    gnum = sum(float(((grads[k].cpu() - torch.from_numpy(gold[f"{kind}.grad.{k}"])).double() ** 2).sum()) for k in grads) ** 0.5
    gden = sum(float((torch.from_numpy(gold[f"{kind}.grad.{k}"]).double() ** 2).sum()) for k in grads) ** 0.5
    print(f"{mode} trunk, small fixture, {kind}: |loss - reference| = {lerr:.3e}, ||dg|| / ||g|| = {gnum / gden:.3e}")
    # bf16: one bar for the four objectives -- the error is the response of a 6-scene loss to ~1.5e-2 of bf16 logit noise and scatters between
    # 1e-4 and 2e-3 across objectives AND across kernel variants with identical rounding points (measured on MI355X, old LDS-resident /
    # wave-private decoder kernel: rift 1.5e-4 / 4.3e-4, grpo 6.8e-4 / 4.8e-4, reinforce 8.8e-4 / 1.7e-3, ppo 1.9e-3 / 5.0e-4).
    # fp16: RIFT (the north-star objective) within north_star's 1e-4 (4.5e-5 measured); the other three 1.4e-4 / 1.4e-4 / 9.3e-5 on this
    # 6-scene fixture -- 2.5e-3 of logit noise -- and <= 4e-5 at the benchmark batch (test_benchmark_batch_objectives_in_16bit_modes).
    # fp32 mode meets 1e-5 (test_losses_and_pi_head_grads).
    if mode == "bf16":
        assert lerr < 3.5e-3
    else:
        assert lerr < (1e-4 if kind == "rift" else 7.5e-4)      # (round-5 arithmetic: 7.2e-5 | 7.1e-5 / 4.9e-4 / 4.1e-6; see the header)
        assert gnum / gden < 8e-2                      # measured 0.2e-2 .. 1.1e-2 (bf16: 6e-2 .. 3e-1)
    rv = data["reference_line"]["valid_mask"].any(-1)
    qf = eng.tap("q_final").view(rv.shape[0], rv.shape[1], 12, 128).cpu()
    ol, og, _ = losses.pi_head_loss_and_grads(sd, qf, kind, H.clone_tree(b), ~rv)
    assert abs(float(loss.item()) - float(ol)) < 1e-5
    for k in grads:
        assert err(grads[k], og[k]) < 1e-5 + 1e-4 * float(og[k].abs().max()), k
    eng.close()


def test_rift_loss_of_the_full_fixture_in_fp16(ffi):
    """north_star's 1e-4 on the RIFT loss of the 2-scene `full` fixture in fp16 mode (measured 2.8e-5; bf16: 5.4e-4)."""
    gold, batch, sd, data, eng, out = _run_case(ffi, "full", "fp16")
    stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng.loss_finalize(stats, flat, grads).item())
    print(f"fp16 trunk, full fixture, rift: |loss - reference| = {abs(loss - float(gold['rift.loss'])):.3e}")
    assert abs(loss - float(gold["rift.loss"])) < 1e-4
    eng.close()


# Bars of the four objectives at the BENCHMARKED batch, per 16-bit mode: (loss bar per objective, gradient bar).  fp16 holds north_star's
# 1e-4 on all four; bf16 -- the mode BASELINE.json names and bench.py's headline runs -- holds it on the RIFT loss only: GRPO / REINFORCE /
# PPO sit at 3.4e-4 / 1.9e-4 / 3.1e-4 (reference lines in two-line rounds) and 2.3e-4 / 4.0e-4 / 4.2e-4 (packed rounds, round 4: another
# summation order of the BatchNorm-2 statistics re-rolls them) -- OUTSIDE 1e-4 either way; their bars are >= 1.7x the larger figure, not
# the contract.  Gradients 0.12 .. 0.24.
BENCH_BATCH_BARS = {
    "fp16": ({"rift": 1e-4, "grpo": 1e-4, "reinforce": 1e-4, "ppo": 1e-4}, 0.22),      # (gradient bar: 8e-2 on the round-4 arithmetic; REINFORCE measures 0.14 on round 5's)
    "bf16": ({"rift": 1e-4, "grpo": 8.5e-4, "reinforce": 7e-4, "ppo": 8e-4}, 0.45),
}


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_benchmark_batch_objectives_in_16bit_modes(ffi, mode):
    """The four objectives at the BENCHMARKED batch (256 scenes, train-mode BatchNorm, drops disabled) against the CPU oracle end to end,
    and the pi_head gradient that drives AdamW against the fp32 oracle's, in both 16-bit operand modes.
    fp16 (the product default): all four losses within north_star's 1e-4 (measured on MI355X 8.2e-6 / 4.0e-5 / 2.3e-5 / 1.7e-5 for
    RIFT / GRPO / REINFORCE / PPO), gradients ||dg|| / ||g|| 4.5e-2 / 3.0e-2 / 4.5e-2 / 5.9e-3.
    bf16 (bench.py's headline, because BASELINE.json names it): RIFT 5.9e-5 .. 6.7e-5 -- inside 1e-4 --, GRPO / REINFORCE / PPO up to 3.4e-4 /
    4.0e-4 / 4.2e-4 over the two pass-B variants -- NOT inside 1e-4: this test holds them to >= 1.7x what is measured and says so --,
    gradients 0.12 .. 0.24."""
    sd = H.weights()
    batch = syn.collate_scenes([syn.make_scene(1000 + i) for i in range(256)])
    batch["advantage_torch"] = torch.randn(256, generator=torch.Generator().manual_seed(99))     # PPO's per-scene (normalised) GAE advantage
    data = batch["cur_pluto_feature_torch"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    _, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    eng = _engine(ffi, mode)
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward(data, train=True, no_drop=True, bn_update=False)
    lbar, gbar = BENCH_BATCH_BARS[mode]
    for kind in ("rift", "grpo", "reinforce", "ppo"):
        lo, go, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], kind, H.clone_tree(batch), r_pad)
        stats, flat, _ = eng.loss_backward(kind, H.clone_tree(batch))
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        lh = float(eng.loss_finalize(stats, flat, grads).item())
        num = sum(float(((grads[k].cpu() - go[k]).double() ** 2).sum()) for k in go) ** 0.5
        den = sum(float((go[k].double() ** 2).sum()) for k in go) ** 0.5
        print(f"{mode} vs fp32 oracle, 256 scenes, {kind}: loss err {abs(lh - float(lo)):.2e} (bar {lbar[kind]:.1e}), ||dg|| / ||g|| {num / den:.3e} (bar {gbar})")
        assert abs(lh - float(lo)) < lbar[kind], kind
        assert num / den < gbar, kind
    eng.close()


# ---- fp16 bars with air: the worst case over seeded draws and kernel variants ---------------------------------------------------------
# Round 3 held the fp16 bars on ONE draw of each fixture size, for ONE arithmetic order (30-step loss 8.7e-5 against 1e-4): any kernel
# change that is not bit-identical re-rolled the noise.  Here every bar is evaluated as the MAXIMUM over 8 seeded draws of the 6-scene
# and of the 2-scene fixture shapes and over the engine's variants with a different summation order (history encoder on all agent slots
# instead of the ranked ones, RIFT_NAT_COMPACT=0; one stream, RIFT_TWO_STREAMS=0), and the bar sits at >= 1.5x that maximum
# (tests/diagnostics/fp16_margin.py prints the table).  What the table says about north_star's 1e-4 on small batches is in FP16_DRAW_BARS.
FP16_DRAWS = {"6-scene": [(list(range(2000 + 10 * d, 2006 + 10 * d)), 12, 8, 1, 4) for d in range(8)],
              "2-scene": [([3000 + 10 * d, 3001 + 10 * d], 64, 20, 1, 6) for d in range(8)]}
FP16_VARIANTS = ({}, {"RIFT_NAT_COMPACT": "0"}, {"RIFT_TWO_STREAMS": "0"})
# Measured maxima on MI355X (fp16, 8 draws x 3 variants; the variants are bit-identical to the default in eval mode, so the spread is the
# draws'), round-4 arithmetic | round-5 arithmetic (the default since round 5), and the bars at >= 1.5x the larger:
#   6-scene: logits 3.1e-3 | 3.3e-3, RIFT 6.3e-5 | 5.4e-5, GRPO 3.5e-4 | 1.4e-4, REINFORCE 2.8e-4 | 4.6e-4, PPO 1.1e-4 | 1.6e-4, RIFT gradient ||dg|| / ||g|| 0.143 | 0.143
#   2-scene: logits 2.4e-3 | 2.7e-3, RIFT 1.4e-4 | 8.8e-5, GRPO 3.9e-4 | 3.1e-4, REINFORCE 1.0e-3 | 6.7e-4, PPO 8.7e-4 | 5.9e-4, gradient 0.092 | 0.096
# Round 6 (the fp16 build evaluates GELU in packed fp16 like the bf16 build: opfmt.h; LayerNorm cancellation guard), same draws and variants:
#   6-scene: logits 2.9e-3, RIFT 7.2e-5, GRPO 2.1e-4, REINFORCE 3.9e-4, PPO 1.4e-4, gradient 0.143
#   2-scene: logits 2.8e-3, RIFT 5.1e-5, GRPO 2.6e-4, REINFORCE 1.03e-3, PPO 7.3e-4, gradient 0.092
# -- every bar below unchanged (none loosened); the 6-scene RIFT bar IS north_star's 1e-4 and is asserted as such (no 1.5x tripwire on it).
# i.e. on 2 - 6-scene batches fp16 does NOT hold north_star's 1e-4 on every draw: the RIFT loss stays inside it on all eight 6-scene draws
# (worst 6.3e-5) and on seven of eight 2-scene draws (worst 1.4e-4); GRPO / REINFORCE / PPO scatter between 1e-5 and 1e-3 (one or two
# scenes' clip decisions carry the loss).  The 1e-4 claim is for the benchmark batch (test_benchmark_batch_objectives_in_16bit_modes: all
# four objectives <= 4e-5 at 256 scenes); compute_precision = "fp32" is the mode that holds it on any batch.
FP16_DRAW_BARS = {
    "6-scene": {"logit": 5.5e-3, "rift": 1.0e-4, "grpo": 5.5e-4, "reinforce": 7.5e-4, "ppo": 2.5e-4, "grad": 0.22},
    "2-scene": {"logit": 4.5e-3, "rift": 2.2e-4, "grpo": 6.0e-4, "reinforce": 1.6e-3, "ppo": 1.4e-3, "grad": 0.15},
}


def fp16_margin_table(ffi, mode="fp16", verbose=True):
    """{shape: {metric: max over draws x variants}} of |HIP - fp32 oracle|: logits on valid lines, the four losses, RIFT gradient ||dg|| / ||g||."""
    sd = H.weights()
    table = {}
    saved = {k: os.environ.get(k) for v in FP16_VARIANTS for k in v}
    try:
        for shape, draws in FP16_DRAWS.items():
            worst = {}
            for d, (idx, A, Mp, r0, r1) in enumerate(draws):
                batch = syn.collate_scenes([syn.make_scene(i, A, Mp, r0, r1) for i in idx])
                batch["advantage_torch"] = torch.randn(len(idx), generator=torch.Generator().manual_seed(700 + d))
                data = batch["cur_pluto_feature_torch"]
                ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
                rv = data["reference_line"]["valid_mask"].any(-1)
                want = {k: losses.pi_head_loss_and_grads(sd, taps["q_final"], k, H.clone_tree(batch), ~rv) for k in ("rift", "grpo", "reinforce", "ppo")}
                for var in FP16_VARIANTS:
                    for k in saved:
                        os.environ.pop(k, None)
                    os.environ.update(var)
                    eng = _engine(ffi, mode)
                    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
                    out = eng.forward(data, need_traj=False)
                    row = {"logit": err(out["probability"].cpu()[rv], ref["probability"][rv])}
                    for kind, (lo, go, _) in want.items():
                        stats, flat, _ = eng.loss_backward(kind, H.clone_tree(batch))
                        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
                        row[kind] = abs(float(eng.loss_finalize(stats, flat, grads).item()) - float(lo))
                        if kind == "rift":
                            num = sum(float(((grads[k].cpu() - go[k]).double() ** 2).sum()) for k in go) ** 0.5
                            row["grad"] = num / sum(float((go[k].double() ** 2).sum()) for k in go) ** 0.5
                    eng.close()
                    if verbose:
                        print(f"  {mode} {shape} draw {d} {var or 'default'}: " + "  ".join(f"{k} {v:.2e}" for k, v in row.items()))
                    for k, v in row.items():
                        worst[k] = max(worst.get(k, 0.0), v)
            table[shape] = worst
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return table


def test_fp16_bars_hold_with_margin_over_seeded_draws_and_kernel_variants(ffi):
    table = fp16_margin_table(ffi, "fp16", verbose=False)
    for shape, worst in table.items():
        print(f"fp16 worst case over 8 draws x 3 variants, {shape}: " + "  ".join(f"{k} {v:.2e} (bar {FP16_DRAW_BARS[shape][k]:.1e})" for k, v in worst.items()))
    for shape, worst in table.items():
        for k, v in worst.items():
            assert v < FP16_DRAW_BARS[shape][k], (shape, k, v)
            if FP16_DRAW_BARS[shape][k] == 1.0e-4:      # north_star's bar itself (6-scene RIFT): nothing to move, the contract is the assertion above
                continue
            assert v < FP16_DRAW_BARS[shape][k] / 1.5 + 1e-12, f"{shape} {k}: {v:.2e} is within 1.5x of its bar {FP16_DRAW_BARS[shape][k]:.1e} -- measure again and move the bar"


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_16bit_operands_stay_in_range_on_carla_magnitudes(ffi, mode):
    """Range study of the 16-bit operand formats on the magnitudes `train_cbv` really feeds (H.carla_magnitude_batch: coordinates to
    +-500 m before the map crop, 40 m/s, speed limits to 40, reference lines to ~480 m; 49 agents + 40 polygons = 89 tokens).
    (1) The oracle, with a recorder on BOTH operands of every contraction (tests/diagnostics/precision_study.operand_range: every
    F.linear / conv1d, attention q / k / v / probabilities, per region of the model): the largest value any 16-bit kernel ever converts is
    the raw coordinate itself (500: Fourier x column, PointsEncoder features ~410 m); everything behind the first layer is O(10).  Bar:
    0.25 x 65504 (the fp16 maximum) -- two binades of head room.  (2) The HIP engine in this mode on the same batch, eval and train
    (BatchNorm batch statistics, no drops): every output finite, the non-finite flag -- now raised by the encoder and decoder kernels
    themselves by exponent bit pattern as well as by the policy head -- stays down, logits within the mode's bar of the oracle."""
    from tests.diagnostics import precision_study as PS
    sd = H.weights()
    batch = H.carla_magnitude_batch()
    data = batch["cur_pluto_feature_torch"]
    try:
        rng, (prob_o, _) = PS.operand_range(sd, batch)
    finally:
        PS.uninstall()
    print(f"max |operand| per region on CARLA magnitudes: " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(rng.items())))
    assert max(rng.values()) < 0.25 * 65504 and all(np.isfinite(v) for v in rng.values())
    assert max(v for k, v in rng.items() if k not in ("fourier", "pe", "ego", "dec_misc")) < 64      # behind the input layers: normalised activations
    rv = data["reference_line"]["valid_mask"].any(-1)
    eng = _engine(ffi, mode)
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    out = eng.forward(data, need_traj=True)
    eng.check_finite()
    for k, v in out.items():
        sel = v.cpu()[rv] if k in ("probability", "trajectory") else v.cpu()
        assert torch.isfinite(sel).all(), k
    e = err(out["probability"].cpu()[rv], prob_o[rv])
    print(f"{mode} on CARLA magnitudes: max |logit - oracle| = {e:.3e}")
    assert e < {"fp16": 8e-3, "bf16": 4e-2}[mode]              # the modes' logit bars (measured 2.3e-3 / 1.4e-2: no worse than on the fixtures)
    out_t = eng.forward(data, train=True, no_drop=True, bn_update=False)
    eng.check_finite()
    assert torch.isfinite(out_t["probability"].cpu()[rv]).all()
    eng.close()


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
@pytest.mark.parametrize("site", ["encoder_blocks.3.mlp.fc2.bias", "planning_decoder.decoder_blocks.3.ffn.3.bias", "planning_decoder.decoder_blocks.0.norm1.bias"])
def test_trunk_kernels_raise_the_non_finite_flag_themselves(ffi, mode, site, monkeypatch):
    """The reference asserts torch.isfinite(q).all() behind every decoder block (planning_decoder.py:175).  Rounds 1-3 raised the device
    flag in the policy-head kernels only (a NaN / Inf in the residual stream reaches q_final); round 4 adds an exponent-bit-pattern test
    -- integer compares, immune to the -fno-honor-nans build of those translation units -- where the rows leave the scene encoder and the
    decoder.  Checked with the policy head NOT run (a deferred-head forward stops in front of it): the flag is up after the trunk alone.
    (RIFT_DEC_DEFER=0: at this batch size a deferred-head forward would leave the decoder to the head call as well.)"""
    monkeypatch.setenv("RIFT_DEC_DEFER", "0")
    gold, batch, sd = H.load_case("small")
    data = batch["cur_pluto_feature_torch"]
    eng = _engine(ffi, mode)
    bad = {k: v.clone() for k, v in sd.items()}
    bad[site][7] = float("inf") if "norm1" not in site else float("nan")
    eng.load_state_dict(bad)
    fb, keep = ffi.feature_batch(data, eng.device)
    o = ffi.RiftOutputs()
    prob = torch.empty(fb.bs, fb.R, 12, device=eng.device)
    o.probability = prob.data_ptr()
    eng.forward_raw(fb, o, ffi.F_TRAIN | ffi.F_NO_DROP | ffi.F_DEFER_HEAD | ffi.F_NO_BN_UPDATE, 1)      # trunk only: the policy head is not issued
    with pytest.raises(RuntimeError, match="non-finite"):
        eng.check_finite()
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward_raw(fb, o, ffi.F_TRAIN | ffi.F_NO_DROP | ffi.F_DEFER_HEAD | ffi.F_NO_BN_UPDATE, 1)
    eng.check_finite()
    eng.close()


def test_benchmark_batch_rift_loss_within_1e4_in_bf16(ffi):
    """north_star's bar on the BENCHMARKED precision at the BENCHMARKED batch: the RIFT loss of a 256-scene minibatch
    (train-mode BatchNorm, drops disabled), bf16 MFMA trunk + loss kernel through the C-ABI, within 1e-4 of the CPU oracle
    (measured 6e-5; the fp32 mode is at 1e-9; 8-scene fixtures sit at 2e-4 in bf16, hence their looser bound above)."""
    sd = H.weights()
    scenes = [syn.make_scene(1000 + i) for i in range(256)]
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out_o, _, _ = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    want = float(losses.rift_loss(out_o["probability"], r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                  batch["group_advantage_mask_torch"]))
    for mode, tol in (("bf16", 1e-4), ("fp16", 3e-5), ("fp32", 1e-6)):
        eng = _engine(ffi, mode)
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.forward(data, train=True, no_drop=True, fp32=mode == "fp32", bn_update=False)
        stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        loss = float(eng.loss_finalize(stats, flat, grads).item())
        print(f"256-scene RIFT loss [{mode}]: |loss - oracle| = {abs(loss - want):.2e}")
        assert abs(loss - want) < tol, (mode, loss, want)
        eng.close()


def test_benchmark_batch_bf16_gradients_against_the_fp32_oracle(ffi):
    """The gradient that drives AdamW in the benchmarked mode -- bf16 trunk, 256-scene minibatch, train-mode BatchNorm -- against the
    fp32 ORACLE's gradient end to end (not the oracle fed with HIP's q_final).

    Two comparisons, because the RIFT objective is only piecewise smooth: clamp(ratio, 0.8, 1.2), min / max and the dual clip at 3A
    switch an entry's gradient on or off at a boundary, so an entry within the trunk's logit error of a boundary contributes its FULL
    gradient to the difference (an fp32-vs-fp32 perturbation of q_final by 1e-5 already moves this gradient by 0.7 %, measured with
    the oracle alone).  (a) the raw gradients; (b) the same with every entry whose oracle ratio lies within 5 % of a boundary removed
    from the objective on both sides.  Measured on MI355X: (a) ||dg|| / ||g|| = 0.158, cosine 0.988; (b) 0.191, 0.982 -- the smooth
    part carries the error, i.e. it is the bf16 rounding of q_final (~5e-2), not boundary flips.  Bars ~2.5x the measurement.
    tests/diagnostics/precision_study.py derives the same numbers on the CPU from operand rounding alone (bf16 0.157) and shows that a
    1e-4 relative bar needs 16 mantissa bits in EVERY contraction of the trunk (three bf16 MFMAs per product): no tail-only or
    weight-only refinement gets below 5e-2.  compute_precision = "fp32" is the mode that meets it (test_losses_and_pi_head_grads)."""
    import torch.nn.functional as F
    sd = H.weights()
    batch = syn.collate_scenes([syn.make_scene(1000 + i) for i in range(256)])
    data = batch["cur_pluto_feature_torch"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out_o, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    bs = r_pad.shape[0]
    lp = F.log_softmax(out_o["probability"].masked_fill(r_pad.unsqueeze(-1), -1e8).view(bs, -1), 1)
    lpo = F.log_softmax(batch["old_group_logits_torch"].masked_fill(r_pad.unsqueeze(-1), -1e8).view(bs, -1), 1)
    ratio = (lp - lpo).exp().view_as(out_o["probability"])
    near = ((ratio - 0.8).abs() < 0.04) | ((ratio - 1.2).abs() < 0.06) | ((ratio - 3.0).abs() < 0.15)
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    res = {}
    for name, mask in (("raw", batch["group_advantage_mask_torch"]), ("interior", batch["group_advantage_mask_torch"] & ~near)):
        b = H.clone_tree(batch)
        b["group_advantage_mask_torch"] = mask
        lo, go, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], "rift", H.clone_tree(b), r_pad)
        eng.forward(data, train=True, no_drop=True, fp32=False, bn_update=False)
        stats, flat, _ = eng.loss_backward("rift", b)
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        lh = float(eng.loss_finalize(stats, flat, grads).item())
        gmax = max(float(v.abs().max()) for v in go.values())
        rel = max(float((grads[k].cpu() - go[k]).abs().max()) for k in go) / gmax
        num = sum(float(((grads[k].cpu() - go[k]).double() ** 2).sum()) for k in go) ** 0.5
        den = sum(float((go[k].double() ** 2).sum()) for k in go) ** 0.5
        cos = sum(float((grads[k].cpu().double() * go[k].double()).sum()) for k in go) / den / sum(float((grads[k].cpu().double() ** 2).sum()) for k in go) ** 0.5
        res[name] = (abs(lh - float(lo)), rel, num / den, cos)
        print(f"bf16 vs fp32 oracle, 256 scenes [{name}]: loss err {res[name][0]:.2e}, max grad err / max|grad| {rel:.3e}, ||dg||/||g|| {num / den:.3e}, cosine {cos:.5f}"
              f" ({int(mask.sum())} entries)")
    eng.close()
    assert res["raw"][0] < 1e-4 and res["interior"][0] < 1e-4                      # north_star's bar on the loss, at the benchmarked batch
    assert res["raw"][2] < 0.4 and res["raw"][3] > 0.95
    assert res["interior"][2] < 0.45 and res["interior"][3] > 0.95


@pytest.mark.parametrize("case", ["small", "full"])
def test_train_mode_batchnorm_statistics(ffi, case):
    """Train mode with every drop probability 0: BatchNorm batch statistics + running-stat update."""
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    eng = ffi.Engine("cuda:0")
    p = eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    out = eng.forward(data, train=True, no_drop=True, fp32=True)
    torch.cuda.synchronize()
    assert err(out["probability"], gold["trainbn.probability"]) < 1e-4
    assert err(out["hidden"], gold["trainbn.hidden"]) < 1e-4
    n = 0
    for k in gold:
        if k.startswith("trainbn.stat."):
            name = k[len("trainbn.stat."):]
            if name.endswith("num_batches_tracked"):
                assert int(p[name].item()) == int(gold[k])
            else:
                assert err(p[name], gold[k]) < 1e-5 + 1e-4 * float(np.abs(gold[k]).max()), name
            n += 1
    assert n == 12
    eng.close()


def test_train_mode_dropout_is_seeded_and_finite(ffi):
    gold, batch, sd = H.load_case("small")
    data = batch["cur_pluto_feature_torch"]
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    a = eng.forward(data, train=True, seed=5, bn_update=False)["probability"].clone()
    b = eng.forward(data, train=True, seed=5, bn_update=False)["probability"].clone()
    c = eng.forward(data, train=True, seed=6, bn_update=False)["probability"].clone()
    d = eng.forward(data, train=True, no_drop=True, bn_update=False)["probability"].clone()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)                       # same seed -> same masks
    assert not torch.equal(a, c)                   # different seed -> different masks
    rv = data["reference_line"]["valid_mask"].any(-1)
    assert (a.cpu()[rv] - d.cpu()[rv]).abs().max() > 1e-3   # dropout actually perturbs the logits
    eng.close()


def test_advantage_kernels(ffi):
    eng = ffi.Engine("cuda:0")
    i = H.advantage_inputs()
    gold = dict(np.load(os.path.join(H.GOLDEN, "advantage.npz")))
    a = eng.gae(i["rewards"], i["undones"], i["values"], i["next_values"], i["unterminated"])
    assert err(a, gold["gae"]) < 1e-5                                   # fp64 scan, rounded once to fp32
    assert err(eng.normalize_advantage_(a.clone()), gold["gae_normalized"]) < 1e-5
    assert err(eng.discounted_return(i["rewards"], i["dones"]), gold["returns"]) < 1e-10
    T = lambda k: torch.from_numpy(i[k])  # noqa: E731
    ret = eng.rollout_return(T("delta_dis"), T("delta_angle"), T("speed"), T("acc"), T("ang_vel"), T("ang_acc"),
                             T("collision"), T("off_road"))
    oret = oadv.rollout_return(i["delta_dis"], i["delta_angle"], i["speed"], i["acc"], i["ang_vel"], i["ang_acc"],
                               i["collision"], i["off_road"])
    assert err(ret, oret) < 1e-5
    assert err(ret, gold["rollout_return"]) < 1e-4                      # north_star: advantages within 1e-4
    assert err(eng.group_advantage(ret.view(1, -1)), gold["group_advantage"]) < 1e-4
    # segment property: a done flag cuts the recurrence (size-independent check at n = 100k)
    n = 100_000
    g = torch.Generator().manual_seed(3)
    r = torch.randn(n, generator=g, dtype=torch.float64)
    done = (torch.rand(n, generator=g) < 0.01).float()
    ret2 = eng.discounted_return(r, done).cpu()
    k = int(torch.nonzero(done)[5])
    assert abs(float(ret2[k]) - float(r[k])) < 1e-12
    assert abs(float(ret2[k - 1]) - float(r[k - 1] + (0.0 if done[k - 1] else 0.98 * ret2[k]))) < 1e-9
    eng.close()


def test_advantage_kernel_edge_cases(ffi):
    """Empty and single-element buffers, a buffer that is one unbroken trajectory and one where every step ends an episode, the
    largest group (R = 16 -> G = 192) and a degenerate group whose returns are all equal (z-score 0 through the +1e-5 guard)."""
    eng = ffi.Engine("cuda:0")
    z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt)  # noqa: E731
    assert eng.gae(z(0, torch.float64), z(0), z(0), z(0), z(0)).numel() == 0
    assert eng.discounted_return(z(0, torch.float64), z(0)).numel() == 0
    one = eng.gae(torch.tensor([2.0], dtype=torch.float64), torch.tensor([1.0]), torch.tensor([0.5]), torch.tensor([0.25]), torch.tensor([1.0]))
    assert abs(float(one[0]) - (2.0 + 0.98 * 0.25 - 0.5)) < 1e-6
    g = torch.Generator().manual_seed(11)
    n = 4096
    r = torch.randn(n, generator=g, dtype=torch.float64)
    v, nv = torch.randn(n, generator=g), torch.randn(n, generator=g)
    for undone in (torch.ones(n), torch.zeros(n)):          # never done / always done
        for unterm in (torch.ones(n), torch.zeros(n)):
            got = eng.gae(r, undone, v, nv, unterm)
            want = oadv.get_advantages_gae(r, undone, v, nv, unterm, 0.98, 0.98)
            assert err(got, want) < 1e-5
    done = torch.zeros(n)
    assert err(eng.discounted_return(r, done), oadv.compute_return(r, done, 0.98)) < 1e-9
    for G in (12, 192):
        ret = torch.randn(5, G, generator=g, dtype=torch.float64)
        want = (ret - ret.mean(1, keepdim=True)) / (ret.std(1, unbiased=False, keepdim=True) + 1e-5)
        assert err(eng.group_advantage(ret), want) < 1e-9
    flat = torch.full((2, 48), 3.25, dtype=torch.float64)
    assert float(eng.group_advantage(flat).abs().max()) == 0.0
    eng.close()


def test_device_collation_matches_pad_sequence(ffi):
    """rift_collate (HBM arena gather) == PlutoFeature.collate / RIFTCollate semantics, bit-exact."""
    from rift_amd import synthetic as syn
    from rift_amd.replay import DeviceReplay
    eng = ffi.Engine("cuda:0")
    scenes = [syn.make_scene(500 + i, 12, 8, 1, 4) for i in range(20)]
    rp = DeviceReplay(scenes, "cuda:0")
    pick = [3, 17, 0, 9, 9, 12]
    idx = torch.tensor(pick, dtype=torch.int32, device="cuda:0")
    R_out = int(rp.r_count_cpu[pick].max())
    fb, b = rp.collate(eng, idx, R_out)
    torch.cuda.synchronize()
    ref = syn.collate_scenes([scenes[i] for i in pick])
    d = rp.batch_dict(b)
    flat_ref = syn.flatten_dict(ref["cur_pluto_feature_torch"])
    flat_hip = syn.flatten_dict(d)
    n = 0
    for k, v in flat_ref.items():
        if k in flat_hip:
            assert torch.equal(flat_hip[k].cpu(), v.to(flat_hip[k].dtype)), k
            n += 1
    assert n >= 21
    assert torch.equal(b["old_group_logits"].cpu(), ref["old_group_logits_torch"])
    assert torch.equal(b["group_advantage"].cpu(), ref["group_advantage_torch"])
    assert torch.equal(b["group_valid_mask"].cpu(), ref["group_advantage_mask_torch"])
    eng.close()


@pytest.mark.parametrize("fp32", [False, True])
def test_non_finite_decoder_queries_are_reported(ffi, fp32):
    """The reference asserts torch.isfinite(q).all() on the decoder queries (planning_decoder.py:175).  Here: a device flag raised by the
    policy-head kernels, surfaced as RIFT_ERR_NONFINITE by rift_check_finite (the host's sync point) -- and cleared by it."""
    gold, batch, sd = H.load_case("small")
    data = batch["cur_pluto_feature_torch"]
    eng = ffi.Engine("cuda:0")
    bad = {k: v.clone() for k, v in sd.items()}
    bad["planning_decoder.decoder_blocks.2.ffn.3.bias"][5] = float("inf")          # one Inf in the residual stream of layer 2
    eng.load_state_dict(bad)
    eng.forward(data, fp32=fp32)
    with pytest.raises(RuntimeError, match="non-finite decoder queries"):
        eng.check_finite()
    eng.check_finite()                                                             # the flag was cleared by the failing check
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward(data, fp32=fp32)
    eng.check_finite()
    eng.close()


def test_device_collation_matches_the_reference_generated_fixture(ffi):
    """rift_collate against tests/golden/collate.npz -- the output of the REFERENCE's RIFTCollate.__call__ / PlutoFeature.collate
    (rift_datamodule.py:20-51, pluto_feature.py:25-96) on ragged seeded scenes (agent, polygon and reference-line counts differ per
    scene): every tensor the engine consumes, bit-exact, in the order and with a repeated index as a DataLoader batch could hold."""
    import os
    from rift_amd.replay import DeviceReplay
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "collate.npz")))
    scenes = H.collate_scenes_ragged()
    assert syn.digest({f"{i}/{k}": v for i, s in enumerate(scenes) for k, v in syn.flatten_dict(s["feature"]).items()}) == str(gold["input_digest"])
    eng = ffi.Engine("cuda:0")
    rp = DeviceReplay(scenes, "cuda:0")
    idx = torch.arange(len(scenes), dtype=torch.int32, device="cuda:0")
    fb, b = rp.collate(eng, idx, int(rp.r_count_cpu.max()))
    torch.cuda.synchronize()
    flat = syn.flatten_dict(rp.batch_dict(b))
    n = 0
    for k, v in flat.items():
        ref = gold["feature/" + k.replace(".", "/")]
        got = v.cpu().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, got.dtype, ref.shape, ref.dtype)
        assert np.array_equal(got, ref), k
        n += 1
    assert n >= 21
    for mine, theirs in (("old_group_logits", "old_group_logits_torch"), ("group_advantage", "group_advantage_torch"),
                         ("group_valid_mask", "group_advantage_mask_torch")):
        got = b[mine].cpu().numpy()
        assert got.dtype == gold[theirs].dtype and np.array_equal(got, gold[theirs]), mine
    # a permuted pick with a repeat: rows of the fixture in that order (reference-line padding is to the fixture's batch maximum)
    pick = [4, 0, 6, 4, 2]
    fb, b = rp.collate(eng, torch.tensor(pick, dtype=torch.int32, device="cuda:0"), int(rp.r_count_cpu.max()))
    torch.cuda.synchronize()
    for k, v in syn.flatten_dict(rp.batch_dict(b)).items():
        assert np.array_equal(v.cpu().numpy(), gold["feature/" + k.replace(".", "/")][pick]), k
    assert np.array_equal(b["group_advantage"].cpu().numpy(), gold["group_advantage_torch"][pick])
    eng.close()


def test_fused_nat_level_matches_layerwise_path(ffi, monkeypatch):
    """The fused NAT level kernels against the layer-by-layer GEMM path and against the exact-fp32 path, on the history-encoder output of
    every sequence that output is read of: valid agents other than the ego (agent_encoder.py:77-87; the fused path runs on exactly those, in
    compacted order, the layer-wise paths on all)."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    hist = data["agent"]["valid_mask"][:, :, :21].any(-1).clone()
    hist[:, 0] = False
    hist = hist.flatten()
    assert 0 < int(hist.sum()) < hist.numel()
    outs = {}
    for name, env, fp32 in (("fused", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_NAT_UNFUSED", env)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.forward(data, fp32=fp32)
        outs[name] = eng.tap("nat_out").view(-1, 128).cpu().clone()[hist]
        eng.close()
    scale = float(outs["fp32"].abs().max())
    assert err(outs["fused"], outs["fp32"]) < 3e-2 * max(1.0, scale)
    assert err(outs["fused"], outs["layerwise"]) < 3e-2 * max(1.0, scale)
    # the fused kernel must not be a no-op: its bf16 rounding points differ from the layer-wise path
    assert not torch.equal(outs["fused"], outs["layerwise"])


def test_compacted_history_encoder_equals_the_uncompacted_launch(ffi, monkeypatch):
    """The history encoder on the compacted sequences (valid agents other than the ego, ranked inside nat_l0w_kernel) against the same kernels
    over all bs * A sequences (RIFT_NAT_COMPACT=0): a sequence's result does not depend on which tile it rides in and the ranking keeps its
    position in level 2's three-agent tiles (common.h: SeqCount), so everything downstream is bit-identical -- on the fixture, on a 64-scene synthetic batch (partial last tiles at every level) and on a batch whose agents are all
    invalid but the egos (no sequence at all)."""
    gold, batch, sd = H.load_case("full")
    cases = [batch["cur_pluto_feature_torch"], syn.collate_scenes([syn.make_scene(5000 + i) for i in range(64)])["cur_pluto_feature_torch"]]
    lone = syn.collate_scenes([syn.make_scene(5100 + i) for i in range(3)])["cur_pluto_feature_torch"]
    lone["agent"]["valid_mask"][:, 1:] = False
    cases.append(lone)
    for ci, data in enumerate(cases):
        hist = data["agent"]["valid_mask"][:, :, :21].any(-1).clone()
        hist[:, 0] = False
        hist = hist.flatten()
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RIFT_NAT_COMPACT", mode)
            eng = ffi.Engine("cuda:0")
            eng.load_state_dict({k: v.clone() for k, v in sd.items()})
            out = eng.forward(data, need_traj=True)
            torch.cuda.synchronize()
            got[mode] = {"nat": eng.tap("nat_out").view(-1, 128).cpu().clone()[hist], "enc": eng.tap("enc_out").cpu().clone(),
                         "prob": out["probability"].cpu().clone(), "traj": out["trajectory"].cpu().clone()}
            eng.close()
        assert ci == 2 or got["1"]["nat"].shape[0] > 0
        for k in got["1"]:
            assert torch.equal(got["1"][k], got["0"][k]), (ci, k)
        assert not torch.isnan(got["1"]["prob"]).any()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_one_pass_layernorm_survives_rows_whose_mean_dwarfs_their_spread(ffi, monkeypatch, mode):
    """LayerNorm property test (round-5 advisor): the fused kernels take the variance of the pre-norm LayerNorms in ONE pass,
    E[x^2] - mean^2 in fp32, which cancels on a row with |mean| >> sigma where torch's centred two-pass form is exact.  Such rows are made
    here by adding a constant c to every channel of the ConvTokenizer's bias (embedding.py:57-60): the level-0 residual stream then carries
    mean ~ c at unchanged spread through both NATLayers (LayerNorm removes it from every branch input, the residual keeps it), i.e.
    mean^2 / var ~ c^2: 64 loses 12 of fp32's 24 bits, 4096 all of them.  The kernels detect such rows (common.h: ln_cancels, more than
    10 bits lost) and take that LayerNorm's variance from the centred values; without the guard the c = 4096 outputs are noise.  Held to
    the bar of test_fused_nat_level_matches_layerwise_path against the exact-fp32 layer-by-layer path (two-pass LayerNorm) at every c."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    hist = data["agent"]["valid_mask"][:, :, :21].any(-1).clone()
    hist[:, 0] = False
    hist = hist.flatten()
    key = "agent_encoder.history_encoder.embed.proj.bias"
    assert key in sd
    for c in (0.0, 64.0, 4096.0):
        sdc = {k: v.clone() for k, v in sd.items()}
        sdc[key] = sdc[key] + c
        outs = {}
        for name, env, fp32 in (("fused", "0", False), ("fp32", "1", True)):
            monkeypatch.setenv("RIFT_NAT_UNFUSED", env)
            eng = _engine(ffi, "fp32" if fp32 else mode)
            eng.load_state_dict({k: v.clone() for k, v in sdc.items()})
            eng.forward(data, fp32=fp32)
            outs[name] = eng.tap("nat_out").view(-1, 128).cpu().clone()[hist]
            eng.close()
        scale = float(outs["fp32"].abs().max())
        e = err(outs["fused"], outs["fp32"])
        print(f"{mode} tokenizer bias + {c:g}: |fused - fp32| {e:.3e} (output scale {scale:.2f})")
        assert e < 3e-2 * max(1.0, scale), (mode, c, e)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_hidden_layer_overflow_raises_the_non_finite_flag(ffi, mode):
    """Round-5 advisor: the packed-fp16 GELU converts the fc1 pre-activation to fp16 without a clamp and the hidden layer stays an fp16
    operand (both builds): a pre-activation beyond 65504 becomes inf.  That must not turn into silent garbage: the inf travels through fc2
    into the residual stream and the encoder / decoder / policy-head kernels raise the sticky non-finite flag (exponent bit pattern), which
    `check_finite` turns into an error at the next host read -- the same contract as every other fp16 operand (opfmt.h).  The overflow is
    made by adding 1e5 to one NATLayer's fc1 bias; the unmodified weights pass the check."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    key = "agent_encoder.history_encoder.levels.0.blocks.0.mlp.fc1.bias"
    assert key in sd
    for bump, bad in ((0.0, False), (1.0e5, True)):
        sdc = {k: v.clone() for k, v in sd.items()}
        sdc[key] = sdc[key] + bump
        eng = _engine(ffi, mode)
        eng.load_state_dict(sdc)
        out = eng.forward(data, need_traj=False)
        if bad:
            with pytest.raises(RuntimeError, match="finite"):
                eng.check_finite()
            eng.check_finite()            # (the flag is cleared by the read that reported it)
        else:
            eng.check_finite()
            assert torch.isfinite(out["probability"]).all()
        eng.close()


def test_ranking_inside_the_preparation_launch_equals_the_ranking_kernel(ffi, monkeypatch):
    """Round 6: the ranks of the compacted history-encoder launch written by the first bs blocks of prep_kernel (kernels.h: rank_scene_body,
    last-block scan over per-scene class counts) against nat_rank_kernel behind the preparation (RIFT_RANK_IN_PREP=0) and against numpy:
    aidx[3 i + c] = the i-th valid non-ego agent slot of class c = slot % 3 in ascending order, cnt = the class counts -- integer arrays,
    bit-exact; everything downstream bit-identical.  Shapes: the fixture (A = 64), a 200-scene synthetic batch (the scan's threads own one scene
    each), the CARLA shape (A = 49: slot classes rotate with the scene), the dense-traffic shape (A = 128: two ballot words per scene), and a
    batch without any sequence.  Twice per engine: the block counter must be back at zero after a launch."""
    gold, batch, sd = H.load_case("full")
    cases = [batch["cur_pluto_feature_torch"], syn.collate_scenes([syn.make_scene(7000 + i) for i in range(200)])["cur_pluto_feature_torch"]]
    for A, Mp, r0, r1 in ((49, 60, 1, 6), (128, 40, 8, 16)):
        cases.append(syn.collate_scenes([syn.make_scene(7300 + i, num_agents=A, num_polygons=Mp, r_min=r0, r_max=r1) for i in range(7)])["cur_pluto_feature_torch"])
    lone = syn.collate_scenes([syn.make_scene(5100 + i) for i in range(3)])["cur_pluto_feature_torch"]
    lone["agent"]["valid_mask"][:, 1:] = False
    cases.append(lone)
    for ci, data in enumerate(cases):
        va = data["agent"]["valid_mask"][:, :, :21].any(-1).clone()
        va[:, 0] = False
        slots = torch.nonzero(va.flatten()).flatten().numpy()
        want_cnt = [int((slots % 3 == c).sum()) for c in range(3)]
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RIFT_RANK_IN_PREP", mode)
            eng = ffi.Engine("cuda:0")
            eng.load_state_dict({k: v.clone() for k, v in sd.items()})
            for rep in range(2):
                out = eng.forward(data, need_traj=True)
                torch.cuda.synchronize()
                aidx = eng.tap("nat_aidx").view(torch.int32).cpu().numpy()
                cnt = eng.tap("nat_cnt").view(torch.int32).cpu().numpy().tolist()
                assert cnt == want_cnt, (ci, mode, rep, cnt, want_cnt)
                for c in range(3):
                    assert np.array_equal(aidx[c::3][:want_cnt[c]], slots[slots % 3 == c]), (ci, mode, rep, c)
            got[mode] = {"nat": eng.tap("nat_out").cpu().clone(), "enc": eng.tap("enc_out").cpu().clone(), "prob": out["probability"].cpu().clone()}
            eng.close()
        hist = va.flatten()
        assert torch.equal(got["1"]["nat"].view(-1, 128)[hist], got["0"]["nat"].view(-1, 128)[hist]), ci
        assert torch.equal(got["1"]["enc"], got["0"]["enc"]) and torch.equal(got["1"]["prob"], got["0"]["prob"]), ci


def test_ranking_look_back_gives_up_instead_of_hanging(ffi, monkeypatch):
    """The scene blocks of the in-launch ranking wait for their predecessors' published counts (kernels.h: rank_scene_body).  A predecessor
    that never publishes -- injected here: RIFT_RANK_FAULT=1 keeps block 0 silent -- must not hang the device: the wait is bounded (a couple
    of seconds), the launch finishes, and the sticky flag turns the forward into an error at the next host read."""
    import time
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    monkeypatch.setenv("RIFT_RANK_FAULT", "1")
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    t0 = time.perf_counter()
    eng.forward(data, need_traj=False)
    with pytest.raises(RuntimeError, match="ranking"):
        eng.check_finite()
    assert time.perf_counter() - t0 < 60.0
    eng.close()
    monkeypatch.delenv("RIFT_RANK_FAULT")
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward(data, need_traj=False)
    eng.check_finite()
    eng.close()


def test_fused_encoder_matches_layerwise_path(ffi, monkeypatch):
    """The fused scene-encoder kernel (4 layers + final LN, one workgroup per scene) against the layer-wise
    GEMM/attention path and the exact-fp32 path, on the encoder output of every valid token."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    bs, A = data["agent"]["position"].shape[:2]
    va = data["agent"]["valid_mask"].any(-1)
    kpm = torch.cat([~va, ~data["map"]["valid_mask"].any(-1)], dim=-1)
    rv = data["reference_line"]["valid_mask"].any(-1)
    outs, dec = {}, {}
    for name, env, fp32 in (("fused", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_ENC_UNFUSED", env)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.forward(data, fp32=fp32)
        outs[name] = eng.tap("enc_out").view(bs, -1, 128).cpu().clone()[~kpm]
        dec[name] = eng.tap("dec3").view(rv.shape[0], rv.shape[1], 12, 128).cpu().clone()[rv]
        eng.close()
    assert err(outs["fused"], outs["fp32"]) < 5e-2
    assert err(outs["fused"], outs["layerwise"]) < 5e-2
    assert not torch.equal(outs["fused"], outs["layerwise"])
    # downstream: the fused decoder fed by the encoder kernel's bf16 K | V^T images ("fused") and by the fp32 K|V
    # projection GEMM ("layerwise" encoder) must both match the exact-fp32 decoder output
    scale = max(1.0, float(dec["fp32"].abs().max()))
    assert err(dec["fused"], dec["fp32"]) < 4e-2 * scale
    assert err(dec["layerwise"], dec["fp32"]) < 4e-2 * scale


def test_fused_fourier_embedding_matches_layerwise_path(ffi, monkeypatch):
    """The one-launch FourierEmbedding (features, 3 x (Linear-LN-ReLU-Linear), sum, LN-ReLU-Linear in LDS; the 129th
    input as an fp32 rank-1 update) against the layer-wise GEMM path and the exact-fp32 path, on the token position
    embedding (3 dims incl. the wrapped heading), the reference-line embedding and the 1-dim speed-limit embedding
    (seen through the map tokens)."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    outs = {}
    for name, env, fp32 in (("fused", "0", False), ("lds", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_FOURIER_UNFUSED", env)
        monkeypatch.setenv("RIFT_KEEP_TOKENS", "1")                        # (the encoder's fused token assembly writes the rows out for the tap)
        monkeypatch.setenv("RIFT_FO_W", "0" if name == "lds" else "1")     # "lds": the LDS-resident fourier_fused_kernel instead of fo_w_kernel
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.prof_enable(True)
        eng.forward(data, fp32=fp32)
        rep = eng.prof_report()
        assert ("fo_w_kernel" in rep) == (name == "fused") and ("fourier_fused_kernel" in rep) == (name == "lds")
        eng.prof_enable(False)
        # x_tokens is only meaningful where the (fused) encoder leaves its input untouched, i.e. in the two bf16 runs
        outs[name] = (eng.tap("r_emb").cpu().clone(), eng.tap("x_tokens").cpu().clone())
        eng.close()
    scale = max(1.0, float(outs["fp32"][0].abs().max()))
    assert err(outs["fused"][0], outs["fp32"][0]) < 3e-2 * scale
    assert err(outs["fused"][0], outs["layerwise"][0]) < 3e-2 * scale
    assert err(outs["lds"][0], outs["fp32"][0]) < 3e-2 * scale
    for k in ("fused", "lds"):
        assert err(outs[k][1], outs["layerwise"][1]) < 3e-2 * max(1.0, float(outs["layerwise"][1].abs().max()))


def test_fused_trajectory_heads_match_layerwise_path(ffi, monkeypatch):
    """The one-launch three-head MLPLayer kernel (planning trajectory and agent prediction, interleaved (.., 80, 6) output)
    against the GEMM + interleave path and the exact-fp32 path."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    rv = data["reference_line"]["valid_mask"].any(-1)
    va = data["agent"]["valid_mask"].any(-1)[:, 1:]      # predictions of VALID agents: a padded agent's token is never a key and its encoder row is read
    outs = {}                                            # by nobody -- the compacted encoder (round 5) leaves zeros there, the slot-ordered paths compute something
    for name, env, fp32 in (("fused", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_HEADS_UNFUSED", env)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.prof_enable(True)
        o = eng.forward(data, need_traj=True, fp32=fp32)
        assert ("heads3_fused_kernel" in eng.prof_report()) == (name == "fused")
        eng.prof_enable(False)
        outs[name] = (o["trajectory"].cpu()[rv], o["prediction"].cpu()[va])
        eng.close()
    for i in (0, 1):
        scale = max(1.0, float(outs["fp32"][i].abs().max()))
        assert err(outs["fused"][i], outs["fp32"][i]) < 3e-2 * scale
        assert err(outs["fused"][i], outs["layerwise"][i]) < 3e-2 * scale


def test_fused_ego_token_matches_layerwise_path(ffi, monkeypatch):
    """The one-launch StateAttentionEncoder (tokens, K|V MFMA, 4-head attention of the learned query over 6 tokens, out_proj)
    against the five-launch path and the exact-fp32 path."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    outs = {}
    for name, env, fp32 in (("fused", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_EGO_UNFUSED", env)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.prof_enable(True)
        eng.forward(data, fp32=fp32)
        assert ("ego_fused_kernel" in eng.prof_report()) == (name == "fused")
        eng.prof_enable(False)
        outs[name] = eng.tap("x_ego").cpu().clone()
        eng.close()
    scale = max(1.0, float(outs["fp32"].abs().max()))
    assert err(outs["fused"], outs["fp32"]) < 2e-2 * scale
    assert err(outs["fused"], outs["layerwise"]) < 2e-2 * scale


@pytest.mark.parametrize("train", [False, True])
def test_fused_points_encoder_matches_layerwise_path(ffi, monkeypatch, train):
    """The three-pass fused PointsEncoder (map polygons, reference lines; BatchNorm batch statistics in train mode,
    running statistics in eval mode) against the layer-wise GEMM / bn / max-pool path and the exact-fp32 path,
    including the BatchNorm running-statistic update."""
    gold, batch, sd = H.load_case("full")
    data = batch["cur_pluto_feature_torch"]
    outs = {}
    bn_keys = [k for k in sd if ("polygon_encoder" in k or "r_encoder" in k) and "running_" in k]
    assert len(bn_keys) == 8
    for name, env, fp32 in (("fused", "0", False), ("mid", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_PE_UNFUSED", env)
        monkeypatch.setenv("RIFT_PE_W", "0" if name == "mid" else "1")     # "mid": pass B as the LDS-resident pe_mid_kernel instead of pe_w_kernel
        eng = ffi.Engine("cuda:0")
        live = {k: v.clone().cuda() for k, v in sd.items()}
        eng.load_state_dict(live)
        eng.prof_enable(True)
        eng.forward(data, train=train, no_drop=True, fp32=fp32)
        rep = eng.prof_report()
        assert ("pe_w_kernel" in rep) == (name == "fused") and ("pe_mid_kernel" in rep) == (name == "mid")
        eng.prof_enable(False)
        outs[name] = (eng.tap("poly_pe").cpu().clone(), eng.tap("r_pe").cpu().clone(),
                      {k: live[k].cpu().clone() for k in bn_keys})
        eng.close()
    for i in (0, 1):
        scale = max(1.0, float(outs["fp32"][i].abs().max()))
        assert err(outs["fused"][i], outs["fp32"][i]) < 3e-2 * scale
        assert err(outs["fused"][i], outs["layerwise"][i]) < 3e-2 * scale
        assert err(outs["mid"][i], outs["fp32"][i]) < 3e-2 * scale
        assert err(outs["fused"][i], outs["mid"][i]) < 3e-2 * scale
    for k in bn_keys:   # running statistics: unchanged in eval mode, momentum-0.1 update of the same batch statistics in train mode
        tol = 2e-2 * max(1.0, float(outs["fp32"][2][k].abs().max()))
        assert err(outs["fused"][2][k], outs["fp32"][2][k]) < tol, k
        assert err(outs["mid"][2][k], outs["fp32"][2][k]) < tol, k
        if not train:
            assert torch.equal(outs["fused"][2][k], sd[k]), k
        else:
            assert not torch.equal(outs["fused"][2][k], sd[k]), k


@pytest.mark.parametrize("case", ["small", "full"])
def test_fused_decoder_matches_layerwise_path(ffi, monkeypatch, case):
    """The fused planning-decoder kernel (4 layers, one workgroup per scene, incl. the r2r mask quirk on
    heterogeneous reference-line counts) against the layer-wise path and the exact-fp32 path."""
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    rv = data["reference_line"]["valid_mask"].any(-1)
    assert len(set(rv.sum(-1).tolist())) > 1, "fixture must mix reference-line counts (exercises the quirk)"
    outs = {}
    for name, env, fp32 in (("fused", "0", False), ("layerwise", "1", False), ("fp32", "1", True)):
        monkeypatch.setenv("RIFT_DEC_UNFUSED", env)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        eng.prof_enable(True)
        eng.forward(data, fp32=fp32)
        # the fused kernel rounds to bf16 at the same points as the layer-wise MFMA path (most rows agree bit for bit), so
        # "it ran" is checked on the launch record, not on a difference in the output
        rep = eng.prof_report()
        assert ("dec_w_kernel" in rep) == (name == "fused")
        eng.prof_enable(False)
        outs[name] = eng.tap("dec3").view(rv.shape[0], rv.shape[1], 12, 128).cpu().clone()[rv]
        eng.close()
    scale = max(1.0, float(outs["fp32"].abs().max()))
    assert err(outs["fused"], outs["fp32"]) < 4e-2 * scale
    assert err(outs["fused"], outs["layerwise"]) < 4e-2 * scale


def test_candidate_rollout_and_ref_line_info(ffi):
    """rift_ref_line_info / rift_rollout against the oracle (bit-exact with the reference on CPU) on two
    consecutive calls (persistent PID state).  Integer outputs (closest reference indices, PID aim indices): bit-exact.
    Floats: the device's sinf / cosf / atan2f and torch-CPU's (SLEEF, 1 ulp) differ in the last bit, and the 79-step closed loop
    integrates that: tolerances are ~3x the worst error measured over 8 seeded calls (tests/diagnostics/a11_errors.py on MI355X:
    delta_dis 1.9e-6, delta_angle 2.4e-7, center 3.1e-4 m after 8 s, angle 7.2e-6, speed 1.4e-4, acc 7.6e-4, ang_vel 6.6e-6,
    ang_acc 4.2e-5; discounted return 9.3e-5 on values of O(10), group advantage 3.2e-6 -- north_star's bar for advantages is 1e-4)."""
    from oracle import rollout as orl
    eng = ffi.Engine("cuda:0")
    ro = orl.Rollout()
    pid = eng.new_pid_state(64)
    for call, seed in enumerate((777, 778)):
        traj, ref_pos, ref_ang, st = H.rollout_inputs(seed)
        t40 = traj[:, :, :40, :]
        dd, da, ci = orl.ref_line_info(t40, ref_pos, ref_ang)
        hdd, hda, hci = eng.ref_line_info(traj, ref_pos, ref_ang, Ts=40)
        assert np.array_equal(hci.cpu().numpy(), ci.numpy().astype(np.int32))          # bit-exact integer indices
        assert err(hdd, dd) < 1e-5 and err(hda, da) < 2e-6
        gpos, ghead = orl.to_global(t40, torch.tensor(st["pos"]), torch.tensor(st["heading"]))
        ref = ro.propagate(gpos, ghead, st["speed"], st["width"], st["length"])
        cs = torch.tensor([[st["pos"][0], st["pos"][1], st["heading"], st["speed"], st["width"], st["length"]]])
        out = eng.rollout(traj.reshape(-1, 80, 6), cs, pid)
        torch.cuda.synchronize()
        assert np.array_equal(out["closest_index"].cpu().numpy(), ref["closest_index"].numpy().astype(np.int32)), call
        assert np.array_equal(out["aim_idx"].cpu().numpy(), ref["aim_idx"].numpy().astype(np.int32)), call
        for k, tol in (("center", 1e-3), ("angle", 2e-5), ("speed", 5e-4), ("acc", 2.5e-3), ("ang_vel", 2e-5), ("ang_acc", 1.5e-4),
                       ("vertices", 1e-3)):
            assert err(out[k], ref[k]) < tol, (call, k, err(out[k], ref[k]))
    eng.close()


def test_eight_lane_rollout_equals_the_one_lane_kernel_bit_for_bit(ffi, monkeypatch):
    """rollout8_kernel (eight lanes per candidate: the 40-point closest-point search split over the group, everything else redundant) against
    rollout_kernel (one lane per candidate, RIFT_RO8=0): every output and the PID state equal bit for bit over consecutive calls with
    different group sizes -- the search's (distance, index) reduction keeps the sequential loop's first-minimum rule."""
    keys = ("center", "angle", "speed", "acc", "ang_vel", "ang_acc", "vertices", "closest_index", "aim_idx")
    runs = {}
    for ro8 in ("1", "0"):
        monkeypatch.setenv("RIFT_RO8", ro8)
        eng = ffi.Engine("cuda:0")
        pid = eng.new_pid_state(256)
        got = []
        for call, seed in enumerate((777, 778, 779, 780)):
            traj, _, _, st = H.rollout_inputs(seed, R=6 if call % 2 == 0 else 5)       # 72 / 60 candidates: nine / eight waves of eight
            flat = traj.reshape(-1, 80, 6).clone()
            for c, (i0, i1) in enumerate(((4, 5), (9, 10), (14, 15), (20, 21), (39, 38), (0, 1))):      # repeated path points: ties in the search, within a
                flat[c, i1, :2] = flat[c, i0, :2]                                                  # lane's five points and across two lanes'
            cs = torch.tensor([[st["pos"][0], st["pos"][1], st["heading"], st["speed"] + 0.3 * call, st["width"], st["length"]]])
            out = eng.rollout(flat, cs, pid)
            torch.cuda.synchronize()
            got.append({k: out[k].cpu().clone() for k in keys})
        got.append({k: v.cpu().clone() for k, v in pid.items()})
        runs[ro8] = got
        eng.close()
    for a, b in zip(runs["1"], runs["0"]):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_ppo_critic_forward_and_value_loss_backward(ffi):
    """CriticPPO forward and the full PPO objective (SmoothL1 value loss + clipped actor loss + entropy) against the fixture
    generated from the reference's CriticPPO / get_ppo_loss: value, total loss, critic gradients (1e-5)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ppo_critic.npz"))
    sd = {k: v.cuda().contiguous() for k, v in H.critic_weights().items()}
    inp = H.critic_inputs()
    eng = ffi.Engine("cuda:0")
    v = eng.critic_forward(sd, inp["state"])
    assert err(v, gold["value"]) < 1e-5
    n = inp["state"].shape[0]
    # actor half through the loss kernel on an explicit logits tensor: emulate with stats = (sum objective, n) from the oracle
    from oracle import critic as ocr
    loss_o, vloss_o, _, grads_o, _ = ocr.ppo_loss_and_grads(H.critic_weights(), inp["probability"], inp["r_pad"], inp["state"],
                                                            inp["action_mode"], inp["advantage"], inp["old_log_prob"], inp["reward_sum"])
    actor = float(loss_o) - float(vloss_o)
    stats = torch.tensor([-actor * n, float(n)], dtype=torch.float64, device="cuda")      # as rift_loss_backward(kind=PPO) leaves them
    flat = torch.zeros(ffi.CRITIC_NPARAM, dtype=torch.float32, device="cuda")
    w = eng.critic_desc(sd)
    eng.critic_loss_backward_raw(w, inp["state"].cuda(), inp["reward_sum"].cuda(), stats, flat)
    loss = float(-stats[0] / stats[1])
    assert abs(loss - float(gold["loss"])) < 1e-5
    grads = [torch.zeros_like(sd[k]) for k in ffi.CRITIC_KEYS]
    eng.critic_finalize_raw(flat, stats, grads)
    for k, g in zip(ffi.CRITIC_KEYS, grads):
        assert err(g, gold["grad." + k]) < 1e-5, k
    eng.close()


@pytest.mark.parametrize("scale", [10.0, 1e-3])
def test_native_clip_grad_norm_matches_torch(ffi, scale):
    """rift_clip_grad_norm against torch.nn.utils.clip_grad_norm_(params, 0.5) (Lightning's gradient_clip_val): clipping and
    pass-through cases, total norm reported."""
    eng = ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 128), (128,), (128,), (128,), (1, 128), (1,), (256, 256)]
    params = [torch.nn.Parameter(torch.zeros(s, device="cuda")) for s in shapes]
    for p_ in params:
        p_.grad = (torch.randn(p_.shape, generator=g) * scale).cuda()
    mine = [p_.grad.clone() for p_ in params]
    want_norm = torch.nn.utils.clip_grad_norm_(params, 0.5)
    tn = torch.zeros(1, device="cuda")
    eng.clip_grad_norm_raw(eng.make_clip_list(mine), 0.5, tn)
    assert abs(float(tn) - float(want_norm)) < 1e-5 * max(1.0, float(want_norm))
    for a, p_ in zip(mine, params):
        assert err(a, p_.grad) < 1e-6 * max(1.0, float(p_.grad.abs().max()))
    eng.close()


@pytest.mark.parametrize("scale,accumulate,use_xchg", [(1e-3, 0, False), (40.0, 0, True), (40.0, 1, False)])
def test_fused_finalize_clip_matches_finalize_then_torch_clip(ffi, scale, accumulate, use_xchg):
    """rift_loss_finalize_clip (one launch) against rift_loss_finalize followed by torch.nn.utils.clip_grad_norm_(pi_head, 0.5):
    pass-through and clipping cases, .grad accumulation, sums read from the f64 exchange buffer; loss and total norm reported."""
    eng = ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(11)
    shapes = [(128, 128), (128,), (128,), (128,), (1, 128), (1,)]
    flat = (torch.randn(ffi.PI_NPARAM, generator=g) * scale).cuda()
    stats = torch.tensor([-37.25, 4242.0], dtype=torch.float64, device="cuda")
    xchg = torch.cat([flat.double() * 2.0, stats * 2.0]) if use_xchg else None     # as after a 2-rank all-reduce of equal shards
    prev = [torch.randn(s, generator=g).cuda() * 0.01 for s in shapes]

    def run(fused):
        grads = [p.clone() for p in prev]
        lo = ffi.RiftLossOut()
        loss = torch.zeros(1, dtype=torch.float64, device="cuda")
        st = stats.clone()
        lo.loss, lo.stats, lo.flat_grad_sum = ffi._ptr(loss), ffi._ptr(st), ffi._ptr(flat)
        lo.exchange = ffi._ptr(xchg)
        lo.grad_w1, lo.grad_b1, lo.grad_ln_w, lo.grad_ln_b, lo.grad_w2, lo.grad_b2 = (ffi._ptr(t) for t in grads)
        tn = torch.zeros(1, device="cuda")
        if fused:
            eng.loss_finalize_clip_raw(lo, accumulate, 0.5, tn)
        else:
            eng.loss_finalize_raw(lo, accumulate)
            params = [torch.nn.Parameter(torch.zeros_like(t)) for t in grads]
            for p_, t in zip(params, grads):
                p_.grad = t
            tn = torch.nn.utils.clip_grad_norm_(params, 0.5).reshape(1)
        torch.cuda.synchronize()
        return grads, float(loss), float(tn), st.cpu()

    g1, l1, n1, s1 = run(True)
    g0, l0, n0, s0 = run(False)
    assert l1 == l0 and torch.equal(s1, s0)
    assert abs(n1 - n0) < 1e-5 * max(1.0, n0)
    assert (n0 > 0.5) == (scale > 1.0)                      # the two regimes really are clip / pass-through
    for a, b in zip(g1, g0):
        assert err(a, b) < 1e-6 * max(1.0, float(b.abs().max()))
    eng.close()


def test_native_adamw_matches_torch_adamw(ffi):
    """rift_adamw_step (one launch over both parameter groups, on torch's own optimizer state) against torch.optim.AdamW for the
    pi_head shapes: weight-decay group (1e-5) and no-decay group, lr changing between steps (WarmupCosLR), step counters advanced."""
    eng = ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 128), (1, 128), (128,), (128,), (128,), (1,)]
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    wds = [1e-5, 1e-5, 0.0, 0.0, 0.0, 0.0]

    def make():
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        opt = torch.optim.AdamW([{"params": ps[:2], "weight_decay": 1e-5}, {"params": ps[2:], "weight_decay": 0.0}],
                                lr=1e-4, fused=True)
        return ps, opt

    ref_p, ref_opt = make()
    my_p, my_opt = make()
    grads = [[torch.randn(s, generator=g).cuda() * (0.01 if k % 2 else 3.0) for s in shapes] for k in range(6)]
    lrs = [1e-4 / 3, 2e-4 / 3, 1e-4, 9.7e-5, 9.1e-5, 9e-5]
    al = None
    for k in range(6):
        for grp in list(ref_opt.param_groups) + list(my_opt.param_groups):
            grp["lr"] = lrs[k]
        for p_, p2, gr in zip(ref_p, my_p, grads[k]):
            p_.grad = gr.clone(); p2.grad = gr.clone()
        ref_opt.step()
        if k == 0:
            my_opt.step()                       # torch creates the state; the native kernel takes over from step 2
            st = [my_opt.state[p_] for p_ in my_p]
            assert all(s_["step"].is_cuda and s_["step"].dtype == torch.float32 for s_ in st)
        else:
            st = [my_opt.state[p_] for p_ in my_p]
            al = eng.make_adam_list(my_p, [p_.grad for p_ in my_p], [s_["exp_avg"] for s_ in st], [s_["exp_avg_sq"] for s_ in st],
                                    [s_["step"] for s_ in st])
            with torch.no_grad():
                eng.adamw_step_raw(al, [lrs[k]] * 6, wds, float(k + 1), 0.9, 0.999, 1e-8)
    torch.cuda.synchronize()
    # tolerances: a few fp32 ulps of the largest entry (measured: <= 2 ulps; the two kernels order the same fp32 operations differently)
    for a, b in zip(my_p, ref_p):
        sa, sb = my_opt.state[a], ref_opt.state[b]
        assert err(a.detach(), b.detach()) < 1e-6 * max(1.0, float(b.detach().abs().max()))
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        assert err(sa["exp_avg"], sb["exp_avg"]) < 1e-6 * max(1.0, float(sb["exp_avg"].abs().max()))
        assert err(sa["exp_avg_sq"], sb["exp_avg_sq"]) < 1e-6 * max(1.0, float(sb["exp_avg_sq"].abs().max()))
    eng.close()


def test_collision_and_off_road_flags_match_oracle(ffi):
    """rift_collision_matrix / rift_off_road_matrix against oracle/traj_flags.py (the reference's STRtree envelope query and raster
    lookup): boolean outputs bit-exact, including footprints that only touch, no neighbours at all, points outside the raster and
    pixel coordinates exactly on a .5 tie (round half to even)."""
    from oracle import traj_flags as otf
    eng = ffi.Engine("cuda:0")
    rng = np.random.default_rng(7)
    G, Tc, Ts, N = 48, 80, 40, 9

    def boxes(n, t, spread):
        c = rng.normal(0, spread, (n, t, 1, 2))
        half = rng.uniform(0.8, 2.6, (n, t, 1, 2))
        ang = rng.uniform(-np.pi, np.pi, (n, t))
        corners = np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], dtype=np.float64)[None, None] * half
        rot = np.stack([np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang)], -1).reshape(n, t, 2, 2)
        return c + np.einsum("ntkj,ntij->ntki", corners, rot)

    center = boxes(G, Tc, 12.0).astype(np.float32)
    other = boxes(N, Ts, 12.0)
    # exact touch: neighbour 0's envelope starts where candidate 0's ends at step 3; and a hair apart at step 4
    e = center[0, 3].astype(np.float64)
    other[0, 3] = np.array([[e[:, 0].max() + 2.0, 0], [e[:, 0].max(), 0], [e[:, 0].max(), 1], [e[:, 0].max() + 2.0, 1]]) + [0, e[:, 1].min()]
    e = center[0, 4].astype(np.float64)
    other[:, 4] += 1e3
    other[0, 4] = np.array([[np.nextafter(e[:, 0].max(), np.inf) + 2.0, 0], [np.nextafter(e[:, 0].max(), np.inf), 0],
                            [np.nextafter(e[:, 0].max(), np.inf), 1], [np.nextafter(e[:, 0].max(), np.inf) + 2.0, 1]]) + [0, e[:, 1].min()]
    want = otf.get_collision_matrix(center, other)
    got = eng.collision_matrix(torch.from_numpy(center), torch.from_numpy(other), Ts=Ts).cpu().numpy()
    assert want[0, 3] and not want[0, 4]
    assert 0.02 < want.mean() < 0.9
    assert np.array_equal(got, want)
    none = eng.collision_matrix(torch.from_numpy(center), torch.zeros(0, Ts, 4, 2, dtype=torch.float64), Ts=Ts).cpu().numpy()
    assert not none.any() and none.shape == (G, Ts)

    mask = (rng.random((400, 400)) > 0.6).astype(np.uint8)
    origin, angle = np.array([13.25, -7.5]), 0.7
    pts = rng.normal(0, 60.0, (G, Tc, 2)).astype(np.float32)            # +-100 m raster: some points fall outside
    # a pixel tie: local x chosen so that x / 0.5 + 200 = 10.5 exactly -> rounds to 10 (even); and 11.5 -> 12
    c, s_ = np.cos(angle), np.sin(angle)
    for k, lx in enumerate((-94.75, -94.25)):
        loc = np.array([lx, 3.0])
        pts[1, k] = (np.array([[c, -s_], [s_, c]]) @ loc + origin).astype(np.float32)
    want_o = otf.get_off_road_matrix(pts, mask, origin, angle)
    got_o = eng.off_road_matrix(torch.from_numpy(pts), torch.from_numpy(mask), origin, angle).cpu().numpy()
    assert 0.05 < want_o.mean() < 0.6
    assert np.array_equal(got_o, want_o)
    eng.close()


def test_off_road_kernel_matches_reference_fixture(ffi):
    """rift_off_road_matrix against tests/golden/off_road.npz: the reference's own get_off_road_matrix / global_to_pixel
    (traj_evaluator.py:277-322) on four masks -- half-pixel ties, raster edges, rotated poses, a non-square raster -- bit-exact."""
    eng = ffi.Engine("cuda:0")
    for name, mask, pts, (x, y, heading), want in H.off_road_cases():
        Hh, Ww = mask.shape
        got = eng.off_road_matrix(torch.from_numpy(pts), torch.from_numpy(mask), (x, y), heading, offset=(Hh / 2, Ww / 2)).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got.astype(bool), want), name
    eng.close()


def test_device_flags_against_hand_derived_known_answers(ffi):
    """rift_collision_matrix / rift_off_road_matrix against tests/golden/traj_flags_kat.json (42 hand-derived cases: touching and
    merely-envelope-overlapping footprints, degenerate footprints, half-pixel ties, raster edges, the flipped y axis, rotation)."""
    col, mask, off = H.traj_flag_kat()
    eng = ffi.Engine("cuda:0")
    for name, center, others, want in col:
        got = eng.collision_matrix(torch.from_numpy(center), torch.from_numpy(others), Ts=1).cpu().numpy()
        assert got.shape == (1, 1) and bool(got[0, 0]) == want, name
    # all collision cases at once (G candidates against each case's neighbours would mix cases: run the centre footprints as one batch
    # against a single common neighbour to exercise G > 1 and Tc > Ts)
    centers = np.concatenate([c[1] for c in col], axis=0).repeat(3, axis=1)                       # (G, 3, 4, 2)
    nb = np.asarray([[[3.0, 3.0], [2.0, 3.0], [2.0, 2.0], [3.0, 2.0]]], dtype=np.float64)[:, None].repeat(2, axis=1)   # (1, 2, 4, 2): the box [2,3]^2
    got = eng.collision_matrix(torch.from_numpy(centers), torch.from_numpy(nb), Ts=2).cpu().numpy()
    from oracle import traj_flags as otf
    assert np.array_equal(got, otf.get_collision_matrix(centers, nb))
    for name, pt, origin, heading, want in off:
        got = eng.off_road_matrix(torch.from_numpy(pt), torch.from_numpy(mask), origin, heading).cpu().numpy()
        assert bool(got[0, 0]) == want, name
    eng.close()


def test_traj_evaluator_grpo_advantage_with_device_flags(ffi):
    """TrajEvaluator.get_grpo_advantage end to end on the device -- ref-line deviation, closed-loop rollout, collision flags from the
    neighbours' forecast footprints, off-road flags from the raster, discounted return, group z-score -- against the oracle chain.
    The flags are computed from each side's OWN rollout (device vs CPU libm differ in the last ulp after 79 closed-loop steps), so
    they are compared through the advantage: north_star's 1e-4 abs on the z-scored returns (measured 3.2e-6 worst over 8 seeded calls,
    tests/diagnostics/a11_errors.py; held to 2e-5 here) unless a flag differs, which the test rules out first."""
    from oracle import advantage as oadv, rollout as orl, traj_flags as otf
    from rift_amd.planning.fine_tuner.rlft.traj_eval.traj_evaluator import TrajEvaluator
    eng = ffi.Engine("cuda:0")
    traj, ref_pos, ref_ang, st = H.rollout_inputs(991)
    R, M = traj.shape[:2]
    rng = np.random.default_rng(3)
    # neighbours: boxes drifting along the candidates' corridor (N, 40, 4, 2) float64; raster: a drivable band around the start pose
    N = 5
    t = np.arange(40)[None, :, None]
    ctr = np.array(st["pos"])[None, None] + rng.normal(0, 6.0, (N, 1, 2)) + t * rng.normal(0.4, 0.2, (N, 1, 2))
    corners = np.array([[2.4, 1.1], [-2.4, 1.1], [-2.4, -1.1], [2.4, -1.1]])[None, None]
    other = ctr[:, :, None, :] + corners
    mask = np.ones((400, 400), dtype=np.uint8)
    mask[170:232, :260] = 0                                            # drivable: a +-15 m band, up to 30 m ahead of the start pose
    pose = (st["pos"][0], st["pos"][1], st["heading"])
    te = TrajEvaluator(eng)
    got = te.get_grpo_advantage((st["pos"][0], st["pos"][1], st["heading"], st["speed"], st["width"], st["length"]), traj, ref_pos, ref_ang,
                                other_vehicle_vertices=other, off_road_mask=mask, center_pose=pose)
    ro_dev = te.last_rollout
    # oracle chain
    t40 = traj[:, :, :40, :]
    dd, da, _ = orl.ref_line_info(t40, ref_pos, ref_ang)
    gpos, ghead = orl.to_global(t40, torch.tensor(st["pos"]), torch.tensor(st["heading"]))
    ref = orl.Rollout().propagate(gpos, ghead, st["speed"], st["width"], st["length"])
    col = otf.get_collision_matrix(ref["vertices"].numpy(), other)
    off = otf.get_off_road_matrix(ref["center"].numpy(), mask, pose[:2], pose[2])
    col_dev = eng.collision_matrix(ro_dev["vertices"], other, Ts=40).cpu().numpy()
    off_dev = eng.off_road_matrix(ro_dev["center"], mask, pose[:2], pose[2]).cpu().numpy()
    assert col.any() and not col.all() and off.any() and not off.all()          # both kinds of flags are exercised
    assert np.array_equal(col_dev, col) and np.array_equal(off_dev, off)        # no candidate sits within an ulp of a box / pixel edge
    ret = oadv.rollout_return(dd.numpy(), da.numpy(), ref["speed"][:, :40].numpy(), ref["acc"][:, :40].numpy(),
                              ref["ang_vel"][:, :40].numpy(), ref["ang_acc"][:, :40].numpy(), col, off)
    want = oadv.group_zscore(ret).reshape(R, M)
    assert got["valid_mask"].all() and got["advantage"].shape == (R, M)
    assert err(torch.from_numpy(got["advantage"]), torch.from_numpy(want)) < 2e-5
    eng.close()


def test_other_vehicle_rollout_matches_reference_fixture(ffi):
    """rift_other_vehicle_rollout against tests/golden/other_vehicles.npz (the reference's get_other_vehicle_rollout run on the seeded
    actors): fp64 corner coordinates within 1e-9 (device libm vs numpy's in the last ulps over 40 recursive steps); and the chain
    forecast -> collision matrix on the device equals the oracle's collision matrix on the reference vertices."""
    import os
    from oracle import traj_flags as otf
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "other_vehicles.npz"))
    inp = H.other_vehicle_inputs()
    eng = ffi.Engine("cuda:0")
    got = eng.other_vehicle_rollout(**inp)
    assert got.shape == (7, 40, 4, 2) and got.dtype == torch.float64
    assert err(got, torch.from_numpy(gold["vertices"])) < 1e-9
    assert eng.other_vehicle_rollout(*[np.zeros((0,))] * 4, np.zeros((0, 3)), np.zeros((0,)), np.zeros((0, 2))).shape == (0, 40, 4, 2)
    rng = np.random.default_rng(5)
    ctr = gold["vertices"][rng.integers(0, 7, 48)].mean(2, keepdims=True) + rng.normal(0, 3.0, (48, 40, 1, 2))     # candidates near the actors
    cand = (ctr + np.array([[2.3, 1.0], [-2.3, 1.0], [-2.3, -1.0], [2.3, -1.0]])[None, None]).astype(np.float32)
    cand80 = np.concatenate([cand, cand], 1)
    want = otf.get_collision_matrix(cand80, gold["vertices"])
    have = eng.collision_matrix(torch.from_numpy(cand80), got, Ts=40).cpu().numpy()
    assert 0.05 < want.mean() < 0.95 and (have != want).mean() < 2e-3         # an ulp-level edge case may flip a single flag
    eng.close()


def test_sft_teacher_loss_and_pi_head_grads(ffi):
    """§8(f) rank 3, the SFT objective on the device: rift_sft_teacher_mode (integer label, bit-exact against the reference fixture)
    and loss kind "sft" (cross entropy against (the policy's best reference line, the teacher's mode), analytic pi_head backward)
    against the oracle on the HIP forward's own fp32 outputs: loss 1e-5, gradients 1e-4 relative."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "sft.npz"))
    eng = ffi.Engine("cuda:0")
    inp = H.sft_inputs()
    rm = eng.sft_teacher_mode(inp["trajectory"], inp["teacher_infos"]).cpu().numpy()
    assert np.array_equal(rm[:, 1], gold["target_m"])                      # the mode index is what enters the label
    # end to end on the model: forward with the trajectory head, label from the device, loss + backward
    g_, batch, sd, data, eng2, out = _run_case(ffi, "small", True)
    bs = out["probability"].shape[0]
    gen = torch.Generator().manual_seed(17)
    traj = out["trajectory"].detach().cpu()
    teacher = torch.stack([2.0 + 8.0 * torch.rand(bs, generator=gen), torch.zeros(bs), torch.zeros(bs), torch.rand(bs, generator=gen) - 0.5,
                           4.0 + torch.rand(bs, generator=gen)], -1)
    ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), need_traj=True, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    b = H.clone_tree(batch)
    b["trajectory_torch"], b["teacher_infos_torch"] = ref["trajectory"], teacher
    loss_o, grads_o, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], "sft", b, r_pad)
    _, best_r_o, m_o = losses.sft_loss(ref["probability"], r_pad, ref["trajectory"], teacher)
    mode = eng2.sft_teacher_mode(out["trajectory"], teacher)
    assert np.array_equal(mode[:, 1].cpu().numpy(), m_o.numpy())
    b["action_mode_torch"] = mode
    stats, flat, chosen = eng2.loss_backward("sft", b)
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng2.loss_finalize(stats, flat, grads).item())
    assert np.array_equal(chosen[:, 0].cpu().numpy(), best_r_o.numpy()) and np.array_equal(chosen[:, 1].cpu().numpy(), m_o.numpy())
    assert abs(loss - float(loss_o)) < 1e-5
    for k in grads:
        refg = grads_o[k]
        assert err(grads[k], refg) < 1e-5 + 1e-4 * float(refg.abs().max()), k
    eng.close(); eng2.close()
