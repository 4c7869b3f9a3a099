"""Shape / mask edge cases of the HIP forward against the CPU oracle (fp32 mode 1e-4 abs on logits, bf16 5e-2):
the dense-traffic configuration of BASELINE.json (128 agents, 40 polygons, 8-16 reference lines: the round-of-eight-tiles variants of the
wave-private encoder / decoder kernels carry it), a single-scene batch with one reference line, and degenerate masks
(no valid neighbour agent, an all-invalid polygon, an all-invalid reference line between valid ones)."""
import pytest
import torch

from oracle import losses, pluto_ref
from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from rift_amd import _ffi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    _ffi.load_library()
    return _ffi


def _check(ffi, scenes, train):
    sd = H.weights()
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    want, _, _ = pluto_ref.planning_model_forward(sd, data, train_bn=train, need_traj=True, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    # exact fp32, then the fused kernels on bf16 and on fp16 MFMA operands (logit bars: 1e-4 / 5e-2 / 1e-2)
    for mode, tol in (("fp32", 1e-4), ("bf16", 5e-2), ("fp16", 1e-2)):
        fp32 = mode == "fp32"
        eng = ffi.Engine("cuda:0", operands="fp16" if mode == "fp16" else "bf16")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        out = eng.forward(data, train=train, no_drop=True, need_traj=True, fp32=fp32, bn_update=False)
        got = out["probability"].cpu()
        assert torch.isfinite(got).all()
        assert (got[r_pad] == -1e6).all()                                   # pluto_model.py:203
        assert float((got - want["probability"])[~r_pad].abs().max()) < tol, (mode, train)
        tw = want["trajectory"][~r_pad]
        tg = out["trajectory"].cpu()[~r_pad]
        assert float((tg - tw).abs().max()) < {"fp32": 2e-3, "bf16": 0.5, "fp16": 0.1}[mode] * max(1.0, float(tw.abs().max()))
        stats, flat, _ = eng.loss_backward("rift", H.clone_tree(batch))
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        loss = float(eng.loss_finalize(stats, flat, grads).item())
        ref = float(losses.rift_loss(want["probability"], r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                     batch["group_advantage_mask_torch"]))
        assert abs(loss - ref) < {"fp32": 1e-5, "bf16": 5e-3, "fp16": 1e-3}[mode], (mode, abs(loss - ref))
        eng.close()


@pytest.mark.parametrize("train", [False, True])
def test_dense_traffic_configuration(ffi, train):
    scenes = [syn.make_scene(300 + i, num_agents=128, num_polygons=40, r_min=8, r_max=16) for i in range(3)]
    assert max(s["feature"]["reference_line"]["position"].shape[0] for s in scenes) > 6
    _check(ffi, scenes, train)


def test_single_scene_single_reference_line(ffi):
    _check(ffi, [syn.make_scene(77, r_min=1, r_max=1)], train=False)


@pytest.mark.parametrize("train", [False, True])
def test_degenerate_masks(ffi, train):
    scenes = [syn.make_scene(500 + i, r_min=3, r_max=5) for i in range(4)]
    f0, f1, f2 = scenes[0]["feature"], scenes[1]["feature"], scenes[2]["feature"]
    f0["agent"]["valid_mask"][1:] = False                      # no valid neighbour agent (the ego stays)
    f1["map"]["valid_mask"][3] = False                         # an all-invalid polygon
    f1["map"]["valid_mask"][7, 5:] = False                     # a partially valid one
    f2["reference_line"]["valid_mask"][1] = False              # an all-invalid reference line between valid ones
    ex = scenes[2]["extras"]
    ex["group_advantage_mask"][1] = False
    _check(ffi, scenes, train)


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_forward_does_not_depend_on_scratch_contents(monkeypatch, train):
    """RIFT_POISON_ARENA fills the engine's scratch arena before every forward and RIFT_POISON_LDS every CU's LDS before every launch:
    with a NaN pattern (0xFF), with other bytes and with zeros the outputs must be bit-identical and finite -- no kernel may read
    scratch or LDS that it (or the same forward) did not write (ragged
    shapes: 12 agents, 8 polygons, 1-3 reference lines, a partial last PointsEncoder tile)."""
    from rift_amd import _ffi as ffi, synthetic as syn
    from tests import helpers as H
    sd = H.weights()
    scenes = [syn.make_scene(300 + i, num_agents=12, num_polygons=8, r_min=1, r_max=3) for i in range(5)]
    data = syn.collate_features([s["feature"] for s in scenes])
    data = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in data.items()}
    outs = []
    # (arena byte, LDS byte): the LDS switch refills every CU's 160 KB of LDS before each launch (LDS keeps the previous kernel's
    # contents otherwise), so a kernel reading LDS it did not write is caught the same way
    for byte, lds in (("0xFF", "0xFF"), ("0x3F", "0x00"), ("0x00", None)):
        monkeypatch.setenv("RIFT_POISON_ARENA", byte)
        if lds is None:
            monkeypatch.delenv("RIFT_POISON_LDS", raising=False)
        else:
            monkeypatch.setenv("RIFT_POISON_LDS", lds)
        eng = ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        for _ in range(2):                                   # second forward: the weights-only products are cached by then
            out = eng.forward(data, train=train, seed=5, bn_update=False, need_traj=True)
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().clone() for k, v in out.items() if torch.is_tensor(v)})
        eng.close()
    rv = data["reference_line"]["valid_mask"].any(-1).cpu()
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all() or k in ("probability",), k
        for o in outs[1:]:
            assert torch.equal(outs[0][k], o[k]), k
    assert torch.isfinite(outs[0]["probability"][rv]).all()


@pytest.mark.gpu
def test_config0_grpo_step_on_dumped_carla_shaped_scenes(tmp_path):
    """BASELINE configs[0]: one GRPO update step on 8 pre-dumped rollout scenes with CARLA-like shapes (A <= 49 agents, ~60 map
    polygons, R <= 6 reference lines, all ragged) -- scenes written with rift_amd.replay.save_scenes, read back, collated; the CPU
    oracle step (forward in train mode with the drops off, GRPO objective, pi_head gradients) against the HIP step in fp32:
    loss 1e-5, gradients 1e-4 relative."""
    from rift_amd import _ffi as ffi
    from rift_amd.replay import load_scenes, save_scenes
    sd = H.weights()
    scenes = [syn.make_scene(5000 + i, num_agents=30 + 3 * (i % 7), num_polygons=52 + 2 * (i % 5), r_min=1, r_max=6) for i in range(8)]
    path = str(tmp_path / "carla_rollout_dump.npz")
    save_scenes(path, scenes)
    batch = syn.collate_scenes(load_scenes(path))
    data = batch["cur_pluto_feature_torch"]
    assert data["agent"]["position"].shape[1] <= 49 and data["map"]["point_position"].shape[1] >= 52
    out_o, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    loss_o, grads_o, _ = losses.pi_head_loss_and_grads(sd, taps["q_final"], "grpo", batch, r_pad)
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward(data, train=True, no_drop=True, fp32=True, bn_update=False)
    stats, flat, _ = eng.loss_backward("grpo", H.clone_tree(batch))
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng.loss_finalize(stats, flat, grads).item())
    assert abs(loss - float(loss_o)) < 1e-5
    for k in grads:
        ref = grads_o[k]
        assert float((grads[k].cpu() - ref).abs().max()) < 1e-5 + 1e-4 * float(ref.abs().max()), k
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("agents,polygons,rmax", [(76, 20, 6), (77, 20, 6), (92, 20, 6), (93, 20, 6), (64, 20, 7)])
def test_fused_kernel_size_limits(ffi, agents, polygons, rmax):
    """At and just beyond what the one-scene-per-workgroup kernels hold: N = 96 tokens exactly fills the fused encoder's 96-row layout and the
    decoder's six key tiles (76 agents + 20 polygons); N = 97 .. 112 run the fused encoder on its 112-row layout (round 6: enc_fused112_kernel)
    and the decoder's eight-key-tile variant; at N = 113 the encoder is the two-pass dense-traffic kernel (enc_w_kernel); R = 7 runs the
    standard decoder (R <= 8) -- all against the oracle, bf16 and fp32, eval and the loss."""
    scenes = [syn.make_scene(6000 + i, num_agents=agents, num_polygons=polygons, r_min=rmax, r_max=rmax) for i in range(3)]
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
    eng.prof_enable(True)
    eng.forward(syn.collate_scenes(scenes)["cur_pluto_feature_torch"], fp32=False)
    rep = eng.prof_report()
    eng.prof_enable(False)
    eng.close()
    assert ("enc_fused_kernel" in rep) == (agents + polygons <= 96)
    assert "dec_w_kernel" in rep
    assert ("enc_fused112_kernel" in rep) == (96 < agents + polygons <= 112)
    assert ("enc_w_kernel" in rep) == (agents + polygons > 112)       # the dense-traffic encoder, which also writes the decoder's K | V^T operands
    _check(ffi, scenes, train=False)


@pytest.mark.parametrize("r_max", [1, 2, 3, 5, 8])
def test_decoder_on_few_and_many_reference_lines(ffi, r_max):
    """The wave-private decoder kernel gives every reference line its own wave (R <= 8): batches whose padded R is r_max -- idle waves at
    small R, all eight working at R = 8 (when all eight waves also share the operand stream) -- against the exact-fp32 layer-wise path
    (bf16 bar of test_forward_eval), and bit-reproducible from one launch to the next."""
    sd = H.weights()
    batch = syn.collate_scenes([syn.make_scene(8100 + i, 24, 10, 1, r_max) for i in range(5)] + [syn.make_scene(8200, 24, 10, r_max, r_max)])
    data = batch["cur_pluto_feature_torch"]
    assert data["reference_line"]["position"].shape[1] == r_max
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    exact = eng.forward(data, fp32=True)["probability"].clone()
    eng.prof_enable(True)
    fused = eng.forward(data, fp32=False)["probability"].clone()
    assert "dec_w_kernel" in eng.prof_report()
    eng.prof_enable(False)
    again = eng.forward(data, fp32=False)["probability"].clone()
    torch.cuda.synchronize()
    rv = data["reference_line"]["valid_mask"].any(-1)
    assert float((fused - exact).abs().cpu()[rv].max()) < 4e-2
    assert torch.equal(fused, again)
    eng.close()


@pytest.mark.parametrize("agents,nscenes", [(5, 1), (7, 3), (64, 2), (33, 5), (1, 2), (2, 3), (3, 4)])
def test_agent_counts_off_the_tile_grid(ffi, agents, nscenes):
    """The NAT kernels tile the ranked history sequences (valid agents other than the ego) in fours (levels 0, 1) and threes (level 2, 8 tiles
    per round): agent counts that leave partial tiles, a partial last round and idle waves, a residue class without a member, no sequence at
    all (one agent per scene: the ego) -- against the oracle, eval and BatchNorm batch statistics."""
    scenes = [syn.make_scene(9100 + i, num_agents=agents, num_polygons=7, r_min=1, r_max=3) for i in range(nscenes)]
    _check(ffi, scenes, train=False)
    _check(ffi, scenes, train=True)


@pytest.mark.parametrize("train", [False, True])
def test_history_encoder_with_nothing_or_everything_to_run_on(ffi, train):
    """Batches in which NO scene has a valid agent besides the ego (the ranked launch has zero sequences: every level kernel and the FPN tail
    find nothing to do) and in which EVERY agent slot is valid at every step (no hole in the ranking)."""
    none = [syn.make_scene(9200 + i, num_agents=12, num_polygons=9, r_min=2, r_max=4) for i in range(3)]
    for s in none:
        s["feature"]["agent"]["valid_mask"][1:] = False
    _check(ffi, none, train)
    full = [syn.make_scene(9210 + i, num_agents=13, num_polygons=9, r_min=2, r_max=4) for i in range(3)]
    for s in full:
        s["feature"]["agent"]["valid_mask"][:] = True
    _check(ffi, full, train)


def test_train_mode_forward_is_bit_reproducible(ffi):
    """Two launches with the same seed give the same bits (train mode: dropout, DropPath, BatchNorm batch statistics).  The wave-private
    kernels keep operands in registers between a conversion and the MFMA that reads it; a missing wait state there shows up as
    launch-to-launch differences of whole tiles (the VALU -> MFMA operand hazard)."""
    sd = H.weights()
    data = syn.collate_scenes([syn.make_scene(9300 + i) for i in range(48)])["cur_pluto_feature_torch"]
    eng = ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    first = None
    for _ in range(6):
        out = eng.forward(data, train=True, seed=11, bn_update=False)
        cur = (out["probability"].clone(), eng.tap("enc_out").clone(), eng.tap("dec3").clone())
        torch.cuda.synchronize()
        if first is None:
            first = cur
        else:
            for a, b in zip(first, cur):
                assert torch.equal(a, b)
    other = eng.forward(data, train=True, seed=12, bn_update=False)["probability"]
    assert not torch.equal(other, first[0])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_two_stream_forward_equals_the_serial_forward(monkeypatch, train):
    """The forward runs the agent-history chain (NAT levels + FPN tail) on a second stream beside the map / reference-line chain and joins
    them at the token assembly (RIFT_TWO_STREAMS, on by default): outputs bit-identical to the one-stream order, repeatedly (a missing
    dependency between the chains would show as run-to-run differences)."""
    from rift_amd import _ffi
    torch.cuda.set_device(0)
    _ffi.load_library()
    scenes = [syn.make_scene(700 + i, r_min=2, r_max=6) for i in range(64)]
    data = syn.collate_scenes(scenes)["cur_pluto_feature_torch"]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RIFT_TWO_STREAMS", mode)
        eng = _ffi.Engine("cuda:0")
        eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
        runs = []
        for _ in range(4):
            out = eng.forward(data, train=train, seed=11, need_traj=True, bn_update=False)
            runs.append((out["probability"].cpu().clone(), out["trajectory"].cpu().clone()))
        for p, t in runs[1:]:
            assert torch.equal(p, runs[0][0]) and torch.equal(t, runs[0][1])
        outs[mode] = runs[0]
        eng.close()
    assert torch.isfinite(outs["1"][0][outs["1"][0] > -1e5]).all()
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_eight_key_tile_decoder_matches_the_dense_variant(monkeypatch, mode):
    """Batches with R <= 8 reference lines and 96 < N <= 128 token slots (what train_cbv collates: 49 agents + 60 polygons = 109) run the standard
    decoder kernel with eight key tiles gathered out of the dense K | V^T image (dec_w.hip: dec_w_kernel<., false, 8>); RIFT_DEC_DBG=32 sends
    them to the dense-traffic variant instead, as until round 5 (tools/shape_step.py carla shows the switch in the decoder's time: 119 against
    161 us).  Same rounding points, same key order, an exact residual hand-over either way (LDS here, global memory there): the eval outputs are
    BIT-IDENTICAL (measured on MI355X in both operand formats)."""
    from rift_amd import _ffi as ffi, synthetic as syn
    from tests import helpers as H
    sd = H.weights()
    scenes = [syn.make_scene(900 + i, num_agents=49, num_polygons=60, r_min=1, r_max=6) for i in range(5)]
    data = syn.collate_scenes(scenes)["cur_pluto_feature_torch"]
    outs = []
    for dbg in (None, "32"):
        if dbg is None:
            monkeypatch.delenv("RIFT_DEC_DBG", raising=False)
        else:
            monkeypatch.setenv("RIFT_DEC_DBG", dbg)
        eng = ffi.Engine("cuda:0", operands="fp16" if mode == "fp16" else "bf16")
        eng.load_state_dict({k: v.clone() for k, v in sd.items()})
        out = eng.forward(data, train=False, need_traj=True, bn_update=False)
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().clone() for k, v in out.items() if torch.is_tensor(v)})
        outs[-1]["q_final"] = eng.tap("q_final").cpu().clone()
        eng.close()
    rv = data["reference_line"]["valid_mask"].any(-1).cpu()
    dq = float((outs[0]["q_final"] - outs[1]["q_final"]).abs().max())
    dp = float((outs[0]["probability"] - outs[1]["probability"])[rv].abs().max())
    print(f"{mode}: eight-key-tile kernel against the dense variant: max |dq_final| {dq:.2e}, max |dlogit| {dp:.2e}")
    assert dq == 0.0 and dp == 0.0
