"""Host mirror of the update-loop driver: replay buffer staging, PlutoFeature collation, policy registry (CPU);
a miniature RLFTPluto.train() on the HIP engine (GPU)."""
import os

import numpy as np
import pytest
import torch

from rift_amd import synthetic as syn
from rift_amd.gym_carla.buffer.cbv_rollout_buffer import CBVRolloutBuffer
from rift_amd.planning.pluto.feature_builder.pluto_feature import PlutoFeature
from tests import helpers as H

KEYS = ['CBVs_obs', 'CBVs_reward', 'CBVs_done', 'CBVs_actions_old_group_logits', 'CBVs_group_advantage',
        'CBVs_actions_ref_group_logits']


def _step(ids, done_ids, t):
    d = {'CBV_ids': [ids]}
    d['CBVs_obs'] = [{i: ('obs', i, t) for i in ids}]
    d['CBVs_reward'] = [{i: float(t) for i in ids}]
    d['CBVs_done'] = [{i: i in done_ids for i in ids}]
    d['CBVs_actions_old_group_logits'] = [{i: t for i in ids}]
    d['CBVs_group_advantage'] = [{i: t for i in ids}]
    d['CBVs_actions_ref_group_logits'] = [{i: t for i in ids}]
    return d


def test_buffer_staging_drop_short_and_fill():
    """cbv_rollout_buffer.py:44-95: per-CBV staging until done; <= 5-step trajectories dropped; full at capacity."""
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': 10, 'data_keys': KEYS})
    for t in range(4):                                      # CBV 7: 4 steps then done -> dropped (n <= 5)
        buf.store(_step([7], [7] if t == 3 else [], t))
    assert len(buf) == 0 and not buf.buffer_full
    for t in range(7):                                      # CBV 1 and 2 interleaved; 1 finishes after 7 steps
        buf.store(_step([1, 2], [1] if t == 6 else [], t))
    assert len(buf) == 7 and [o[1] for o in buf.buffer_data['CBVs_obs']] == [1] * 7
    for t in range(7, 9):
        buf.store(_step([2], [2] if t == 8 else [], t))     # CBV 2: 9 steps -> only 3 fit
    assert len(buf) == 10 and buf.buffer_full
    assert [o[1] for o in buf.get_key_data('CBVs_obs')] == [1] * 7 + [2] * 3
    assert [o[2] for o in buf.get_key_data('CBVs_obs')][7:] == [0, 1, 2]
    s = buf.sample(8)
    assert s['CBVs_reward'] == 1.0
    assert buf.get_all_np_data()['CBVs_reward'].shape == (10, 1)
    buf.add_extra_data({'extra': list(range(10))})
    assert buf.sample([0, 9])['extra'] == [0, 9]
    buf.reset_buffer()
    assert len(buf) == 0 and not buf.buffer_full and 'extra' not in buf.buffer_data
    with pytest.raises(AssertionError):
        buf.get_key_data('CBVs_obs')


def test_buffer_reset_can_be_undone():
    """Round 6: `reset_buffer_reversibly` empties the buffer like `reset_buffer` but keeps what it held in a token; `restore(token)` puts the
    committed rows, the open episodes, the full flag and the extra columns back (RLFTPluto.train resets in the device's shadow and restores
    when the update fails to commit).  Rows are the same objects, the columns read as before, a store() after the restore carries on; a
    restore onto a buffer that has stored again is refused."""
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': 10, 'data_keys': KEYS})
    for t in range(7):
        buf.store(_step([1, 2], [1] if t == 6 else [], t))      # CBV 1 committed (7 rows), CBV 2 still open (7 staged steps)
    rows = list(buf._rows)
    token = buf.reset_buffer_reversibly()
    assert len(buf) == 0 and not buf.buffer_full and not buf._open
    buf.restore(token)
    assert len(buf) == 7 and all(a is b for a, b in zip(buf._rows, rows)) and not buf.buffer_full
    assert [o[1] for o in buf.buffer_data['CBVs_obs']] == [1] * 7
    for t in range(7, 9):
        buf.store(_step([2], [2] if t == 8 else [], t))          # the open episode of CBV 2 survived the round trip: 9 steps, 3 fit
    assert len(buf) == 10 and buf.buffer_full and [o[1] for o in buf.get_key_data('CBVs_obs')] == [1] * 7 + [2] * 3
    buf.add_extra_data({'extra': list(range(10))})
    token = buf.reset_buffer_reversibly()
    assert len(buf) == 0 and not buf.buffer_full and 'extra' not in buf.buffer_data
    buf.restore(token)                                           # a full buffer with its extra columns
    assert buf.buffer_full and len(buf) == 10 and buf.sample([0, 9])['extra'] == [0, 9]
    token = buf.reset_buffer_reversibly()
    for t in range(7):
        buf.store(_step([3], [3] if t == 6 else [], t))          # a new generation has been stored: the old one cannot come back over it
    with pytest.raises(AssertionError):
        buf.restore(token)


def test_buffer_matches_the_reference_class_on_seeded_store_sequences():
    """a14 pinned: tests/golden/buffer.npz holds what the REFERENCE's CBVRolloutBuffer (cbv_rollout_buffer.py:16-138, imported by
    gen_golden.gen_buffer) did on 120 seeded multi-CBV store sequences -- short episodes (dropped), episodes staged across calls,
    exact fills, overflows: buffer_pos / buffer_full / stored order after EVERY store call, the done column and a sample once full."""
    import os
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "buffer.npz")))
    seqs = H.buffer_store_sequences()
    assert len(seqs) == int(gold["n_seq"])
    call, n_full = 0, 0
    for q, (capacity, calls) in enumerate(seqs):
        buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': capacity, 'data_keys': list(H.BUFFER_KEYS)})
        for chunk in calls:
            if buf.buffer_full:
                break
            buf.store(chunk)
            assert len(buf) == buf.buffer_pos == int(gold["buffer_pos"][call]), (q, call)
            assert buf.buffer_full == bool(gold["buffer_full"][call]), (q, call)
            want = gold["order"][gold["order_off"][call]:gold["order_off"][call + 1]]
            assert [int(v) for v in buf.buffer_data['CBVs_obs']] == want.tolist(), (q, call)
            call += 1
        if buf.buffer_full:
            assert [bool(v) for v in buf.get_key_data('CBVs_done')] == gold["done_when_full"][n_full:n_full + capacity].tolist(), q
            n_full += capacity
            assert int(buf.sample(capacity // 2)['CBVs_actions']) == int(gold["sample_mid"][q])
            assert buf.get_all_np_data()['CBVs_reward'].shape == (capacity, 1)
        else:
            assert int(gold["sample_mid"][q]) == -1
    assert call == len(gold["buffer_pos"]) and n_full == len(gold["done_when_full"])


def test_pluto_feature_collate_matches_reference_semantics():
    scenes = [syn.make_scene(i, num_agents=6 + i, num_polygons=4 + i) for i in range(3)]
    pf = PlutoFeature.collate([PlutoFeature(data=s["feature"]) for s in scenes])
    want = syn.collate_features([s["feature"] for s in scenes])
    flat_a, flat_b = syn.flatten_dict(pf.data), syn.flatten_dict(want)
    assert flat_a.keys() == flat_b.keys()
    for k in flat_a:
        assert torch.equal(flat_a[k], flat_b[k]), k
    assert pf.data["agent"]["position"].shape[:2] == (3, 8)
    rt = PlutoFeature.deserialize(pf.serialize())
    assert torch.equal(rt.data["map"]["point_position"], pf.data["map"]["point_position"])


def _collate_fixture():
    import os
    import numpy as np
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "collate.npz")))
    scenes = H.collate_scenes_ragged()
    assert syn.digest({f"{i}/{k}": v for i, s in enumerate(scenes) for k, v in syn.flatten_dict(s["feature"]).items()}) == str(gold["input_digest"])
    return gold, scenes


def test_collation_matches_the_reference_generated_fixture():
    """tests/golden/collate.npz = the reference's own RIFTCollate.__call__ / PlutoFeature.collate (rift_datamodule.py:20-51,
    pluto_feature.py:25-96) run on ragged seeded scenes.  The host mirror and the helper every other parity test builds its batches
    with (rift_amd.synthetic.collate_scenes) must reproduce it bit for bit: every key, shape, dtype and value."""
    import numpy as np
    gold, scenes = _collate_fixture()
    mirror = PlutoFeature.collate([PlutoFeature(data=s["feature"]) for s in scenes]).data
    helper = syn.collate_scenes(scenes)
    want_keys = {k[len("feature/"):].replace("/", ".") for k in gold if k.startswith("feature/")}
    for name, got in (("mirror", syn.flatten_dict(mirror)), ("helper", syn.flatten_dict(helper["cur_pluto_feature_torch"]))):
        assert set(got) == want_keys, (name, set(got) ^ want_keys)
        for k, v in got.items():
            ref = gold["feature/" + k.replace(".", "/")]
            assert tuple(v.shape) == ref.shape and v.numpy().dtype == ref.dtype, (name, k, v.shape, v.dtype, ref.shape, ref.dtype)
            assert np.array_equal(v.numpy(), ref), (name, k)
    for k in ("group_advantage_torch", "group_advantage_mask_torch", "old_group_logits_torch", "old_group_logits_mask_torch"):
        assert helper[k].numpy().dtype == gold[k].dtype and np.array_equal(helper[k].numpy(), gold[k]), k


def test_policy_registry_and_lr_schedule():
    from oracle import advantage as oadv
    from rift_amd.planning import CBV_POLICY_LIST
    from rift_amd.planning.fine_tuner.rlft.trainer import WarmupCosLR
    assert {'rift_pluto', 'grpo_pluto', 'ppo_pluto', 'reinforce_pluto'} <= set(CBV_POLICY_LIST)
    assert CBV_POLICY_LIST['rift_pluto'].kind == 'rift' and CBV_POLICY_LIST['ppo_pluto'].type == 'learnable'
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = WarmupCosLR(opt, lr=1e-4, min_lr=9e-5, warmup_epochs=3, epochs=16)
    for e in range(16):
        assert abs(opt.param_groups[0]["lr"] - oadv.warmup_cos_lr(e, 1e-4, 9e-5, 3, 16)) < 1e-12
        sch.step()


def test_optimizer_groups_are_the_reference_rule_on_the_trainable_sets():
    """configure_optimizer's explicit grouping (Linear weights decay; biases, LayerNorm weight and the critic's constants do not) names
    the same two sorted groups as the reference's module-type rule (rift_trainer.py:279-362) for pi_head and for pi_head + value_net;
    a trainable layer of another kind is refused instead of being silently mis-grouped."""
    import torch.nn as nn
    from rift_amd.planning.fine_tuner.rlft.ppo_pluto.ppo_pluto import PPOPlutoModel
    from rift_amd.planning.fine_tuner.rlft.trainer import configure_optimizer, freeze_parameters

    def reference_rule(model):           # the reference's loop, restated on names (decay set, no-decay set)
        white = (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.MultiheadAttention, nn.LSTM, nn.GRU)
        black = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.LayerNorm, nn.Embedding)
        decay, no_decay = set(), set()
        for mn, m in model.named_modules():
            for pn, p in m.named_parameters():
                if not p.requires_grad:
                    continue
                full = f"{mn}.{pn}" if mn else pn
                if "bias" in pn:
                    no_decay.add(full)
                elif "weight" in pn:
                    if isinstance(m, white):
                        decay.add(full)
                    elif isinstance(m, black):
                        no_decay.add(full)
                else:
                    no_decay.add(full)
        return sorted(decay), sorted(no_decay)

    model = PPOPlutoModel(radius=120)
    names = {id(p): n for n, p in model.named_parameters()}
    for layers, n_decay, n_no in ((["planning_decoder.pi_head"], 2, 4), (["planning_decoder.pi_head", "value_net"], 5, 11)):
        freeze_parameters(model, layers)
        opt = configure_optimizer(model, 1e-4, 1e-5)
        got = [[names[id(p)] for p in g["params"]] for g in opt.param_groups]
        want = reference_rule(model)
        assert got[0] == want[0] and got[1] == want[1], (got, want)
        assert (len(got[0]), len(got[1])) == (n_decay, n_no)
        assert opt.param_groups[0]["weight_decay"] == 1e-5 and opt.param_groups[1]["weight_decay"] == 0.0
    freeze_parameters(model, ["agent_encoder.type_emb"])          # an Embedding: not a layer this path trains
    with pytest.raises(NotImplementedError, match="only Linear / LayerNorm"):
        configure_optimizer(model, 1e-4, 1e-5)


PPO_KEYS = ['CBVs_obs', 'CBVs_next_obs', 'CBVs_reward', 'CBVs_done', 'CBVs_terminated', 'CBVs_actions_old_log_prob', 'CBVs_actions_mode']


def _filled_ppo_buffer(n):
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': n, 'data_keys': PPO_KEYS})
    t = 0
    while not buf.buffer_full:
        for k in range(8):
            s, s2 = (syn.make_scene(t + j, num_agents=12, num_polygons=8, r_min=1, r_max=3) for j in (0, 1))
            ex = s["extras"]
            d = {'CBV_ids': [[3]], 'CBVs_obs': [{3: {'raw_pluto_feature': PlutoFeature(data=s["feature"])}}],
                 'CBVs_next_obs': [{3: {'raw_pluto_feature': PlutoFeature(data=s2["feature"])}}],
                 'CBVs_reward': [{3: float(ex["return"])}], 'CBVs_done': [{3: k == 7}], 'CBVs_terminated': [{3: k == 7 and t % 16 == 7}],
                 'CBVs_actions_old_log_prob': [{3: np.float32(ex["old_log_prob"])}], 'CBVs_actions_mode': [{3: ex["action_mode"].numpy()}]}
            buf.store(d)
            t += 1
    return buf


def _filled_buffer(n, with_ref):
    keys = [k for k in KEYS if with_ref or k != 'CBVs_actions_ref_group_logits']
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': n, 'data_keys': keys})
    t = 0
    while not buf.buffer_full:
        for k in range(8):                                  # 8-step trajectories of one CBV
            s = syn.make_scene(t, num_agents=12, num_polygons=8, r_min=1, r_max=3)
            ex = s["extras"]
            d = {'CBV_ids': [[3]], 'CBVs_obs': [{3: {'raw_pluto_feature': PlutoFeature(data=s["feature"])}}],
                 'CBVs_reward': [{3: float(ex["return"])}], 'CBVs_done': [{3: k == 7}],
                 'CBVs_actions_old_group_logits': [{3: {'logits': ex["old_group_logits"].numpy(),
                                                        'valid_mask': ex["old_group_logits_mask"].numpy()}}],
                 'CBVs_group_advantage': [{3: {'advantage': ex["group_advantage"].numpy(),
                                               'valid_mask': ex["group_advantage_mask"].numpy()}}]}
            if with_ref:
                d['CBVs_actions_ref_group_logits'] = [{3: {'logits': ex["ref_group_logits"].numpy()}}]
            buf.store(d)
            t += 1
    return buf


def _ragged_rift_buffer(capacity, seed0, caps=None, dims=((9, 5, 2), (16, 10, 4), (3, 7, 1), (12, 2, 3), (16, 9, 4), (5, 10, 2), (1, 1, 1)), with_ref=False):
    """A rift_pluto buffer filled through store() with scenes of ragged agent / polygon / reference-line counts (7-step episodes of two CBVs)."""
    keys = [k for k in KEYS if with_ref or k != 'CBVs_actions_ref_group_logits']
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': capacity, 'data_keys': keys, 'host_caps': caps or {}})
    t = 0
    while not buf.buffer_full:
        for k in range(7):
            d = {'CBV_ids': [[3, 8]], 'CBVs_obs': [{}], 'CBVs_reward': [{}], 'CBVs_done': [{}], 'CBVs_actions_old_group_logits': [{}],
                 'CBVs_group_advantage': [{}]}
            if with_ref:
                d['CBVs_actions_ref_group_logits'] = [{}]
            for c in (3, 8):
                A, Mp, R = dims[(t + c) % len(dims)]
                s = syn.make_scene(seed0 + 2 * t + (c == 8), A, Mp, R, R)
                ex = s["extras"]
                d['CBVs_obs'][0][c] = {'raw_pluto_feature': PlutoFeature(data=s["feature"])}
                d['CBVs_reward'][0][c], d['CBVs_done'][0][c] = float(ex["return"]), k == 6
                d['CBVs_actions_old_group_logits'][0][c] = {'logits': ex["old_group_logits"].numpy(), 'valid_mask': ex["old_group_logits_mask"].numpy()}
                d['CBVs_group_advantage'][0][c] = {'advantage': ex["group_advantage"].numpy(), 'valid_mask': ex["group_advantage_mask"].numpy()}
                if with_ref:
                    d['CBVs_actions_ref_group_logits'][0][c] = {'logits': ex["ref_group_logits"].numpy()}
            buf.store(d)
            t += 1
    return buf


def _assert_host_arena_is_the_packed_arena(buf):
    """HostReplay (filled row by row at store() time) == what DeviceReplay(buffer_to_scenes(buffer)) packs in one pass, bit for bit; the
    capacity padding beyond the largest scene is zero."""
    from rift_amd import replay as rp
    from rift_amd.planning.fine_tuner.rlft.rlft_pluto import buffer_to_scenes
    host = buf.host_replay()
    assert host is not None and host.T == 21
    scenes = buffer_to_scenes(buf)
    feats, ex = [s["feature"] for s in scenes], [s["extras"] for s in scenes]
    dims = {"agent": max(f["agent"]["position"].shape[0] for f in feats), "map": max(f["map"]["point_position"].shape[0] for f in feats),
            "reference_line": max(f["reference_line"]["position"].shape[0] for f in feats), "static_objects": 0}
    assert (host.dims["A"], host.dims["Mp"], host.dims["R"]) == (dims["agent"], dims["map"], dims["reference_line"])
    n = 0
    for name, grp, key, dt, _ in rp._FIELDS:
        if grp == "static_objects":
            continue
        want = rp._pad_stack([f[grp][key] for f in feats], dims[grp], dt)
        got = host.t[name]
        assert got.dtype == want.dtype and torch.equal(got[:, :dims[grp]], want), name
        assert not got[:, dims[grp]:].any(), name
        n += 1
    assert n == 20
    assert torch.equal(host.t["current_state"], torch.stack([f["current_state"] for f in feats]).float())
    R = dims["reference_line"]
    for name, key, dt in (("old_group_logits", "old_group_logits", torch.float32), ("group_advantage", "group_advantage", torch.float64),
                          ("group_valid_mask", "group_advantage_mask", torch.bool)) + ((("ref_group_logits", "ref_group_logits", torch.float32),)
                                                                                         if "ref_group_logits" in ex[0] else ()):
        want = rp._pad_stack([e[key] for e in ex], R, dt)
        assert host.t[name].dtype == dt and torch.equal(host.t[name][:, :R], want) and not host.t[name][:, R:].any(), name
    assert host.r_count.tolist() == [f["reference_line"]["position"].shape[0] for f in feats]
    return host


def test_store_lays_committed_rows_into_the_host_arena():
    """Round 4: CBVRolloutBuffer.store() writes every committed transition into the pinned structure-of-arrays mirror of the HBM arena
    (rift_amd.replay.HostReplay) -- the layout DeviceReplay used to build in a 4096 x 25 Python pass inside RLFTPluto.train.  Checked
    against that pass on ragged scenes, with capacities that have to double on the way, after a reset + refill with other (smaller)
    scenes (stale rows and stale padding must be gone), and for the GRPO reference logits."""
    buf = _ragged_rift_buffer(45, 7000, caps={"A": 4, "Mp": 4, "R": 2})
    host = _assert_host_arena_is_the_packed_arena(buf)
    assert host.grown >= 3 and host.caps["A"] >= 16 and host.caps["Mp"] >= 10 and host.caps["R"] >= 4
    buf.reset_buffer()
    assert len(buf) == 0 and buf.host_replay() is host and host.dims["A"] == 0
    buf2 = buf
    t = 0
    while not buf2.buffer_full:                      # refill the SAME buffer with smaller scenes
        for k in range(6):
            s = syn.make_scene(9000 + t, 5, 3, 1, 2)
            ex = s["extras"]
            buf2.store({'CBV_ids': [[1]], 'CBVs_obs': [{1: {'raw_pluto_feature': PlutoFeature(data=s["feature"])}}], 'CBVs_reward': [{1: 0.0}],
                        'CBVs_done': [{1: k == 5}],
                        'CBVs_actions_old_group_logits': [{1: {'logits': ex["old_group_logits"].numpy(), 'valid_mask': ex["old_group_logits_mask"].numpy()}}],
                        'CBVs_group_advantage': [{1: {'advantage': ex["group_advantage"].numpy(), 'valid_mask': ex["group_advantage_mask"].numpy()}}]})
            t += 1
    h2 = _assert_host_arena_is_the_packed_arena(buf2)
    assert h2 is host and h2.dims["A"] == 5 and h2.dims["Mp"] == 3
    _assert_host_arena_is_the_packed_arena(_ragged_rift_buffer(30, 7100, with_ref=True))
    # rows that are not PlutoFeature observations (the staging tests above store integers) are simply not mirrored
    plain = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': 10, 'data_keys': KEYS})
    for t in range(8):
        plain.store(_step([1], [1] if t == 7 else [], t))
    assert len(plain) == 8 and plain.host_replay() is None
    # numpy float64 features (what the CARLA-side builder hands over before to_feature_tensor) are cast like PlutoFeature.to_feature_tensor
    from rift_amd.replay import HostReplay
    s = syn.make_scene(1, 6, 4, 2, 2)
    f64 = {g: ({k: (v.double().numpy() if v.dtype == torch.float32 else v.numpy()) for k, v in d.items()} if isinstance(d, dict) else d.double().numpy())
           for g, d in s["feature"].items()}
    h = HostReplay(2)
    h.put(0, f64)
    h.put(1, s["feature"])
    for name in ("agent_position", "map_point_orientation", "ref_vector", "current_state", "agent_valid_mask"):
        assert torch.equal(h.t[name][0], h.t[name][1]), name


@pytest.mark.gpu
def test_streamed_arena_collates_like_the_packed_one_and_the_reference_fixture():
    """The arena uploaded from the host mirror (capacities above the batch dimensions: rift_collate crops every ragged dimension by a
    prefix copy) gathers the SAME batch as the packed arena and as tests/golden/collate.npz = the reference's RIFTCollate output."""
    from rift_amd import _ffi
    from rift_amd.replay import DeviceReplay, HostReplay
    torch.cuda.set_device(0)
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "collate.npz")))
    scenes = H.collate_scenes_ragged()
    host = HostReplay(len(scenes), caps={"A": 20, "Mp": 12, "R": 6, "S": 2})
    for i, s in enumerate(scenes):
        host.put(i, s["feature"], s["extras"])
    eng = _ffi.Engine("cuda:0")
    rp, packed = DeviceReplay.from_host(host, "cuda:0"), DeviceReplay(scenes, "cuda:0")
    assert (rp.A, rp.Mp, rp.Rcap) == (16, 10, 6) and rp.arena.A == 20 and rp.arena.Mp == 12
    for pick in (list(range(len(scenes))), [4, 0, 6, 4, 2]):
        idx = torch.tensor(pick, dtype=torch.int32, device="cuda:0")
        R_out = int(rp.r_count_cpu.max())
        _, b = rp.collate(eng, idx, R_out)
        _, bp = packed.collate(eng, idx, R_out)
        torch.cuda.synchronize()
        for k, v in syn.flatten_dict(rp.batch_dict(b)).items():
            ref = gold["feature/" + k.replace(".", "/")][pick]
            got = v.cpu().numpy()
            assert got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref), k
        for k in b:
            if b[k] is not None:
                assert torch.equal(b[k], bp[k]), k
        assert np.array_equal(b["group_advantage"].cpu().numpy(), gold["group_advantage_torch"][pick])
        assert np.array_equal(b["old_group_logits"].cpu().numpy(), gold["old_group_logits_torch"][pick])
    # a second replay generation in the same slots: the device arena is re-used, the batch buffers follow the new batch dimensions
    host.reset()
    small = [syn.make_scene(300 + i, 6, 4, 1, 2) for i in range(len(scenes))]
    for i, s in enumerate(small):
        host.put(i, s["feature"], s["extras"])
    keep = rp.t["agent_position"].data_ptr()
    rp.upload(host)
    assert rp.t["agent_position"].data_ptr() == keep and (rp.A, rp.Mp) == (6, 4)
    idx = torch.arange(len(small), dtype=torch.int32, device="cuda:0")
    _, b = rp.collate(eng, idx, int(rp.r_count_cpu.max()))
    _, bp = DeviceReplay(small, "cuda:0").collate(eng, idx, int(rp.r_count_cpu.max()))
    torch.cuda.synchronize()
    for k in b:
        if b[k] is not None:
            assert torch.equal(b[k], bp[k]), k
    eng.close()


@pytest.mark.gpu
def test_update_from_the_streamed_arena_equals_the_update_from_the_packed_one(tmp_path):
    """RLFTPluto.train() on the arena the buffer streamed at store() time writes the SAME checkpoint (every tensor bit for bit) as on the
    arena packed scene by scene inside train() (`stream_to_host: False`, the round-3 path)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    out = {}
    for mode in ("streamed", "packed"):
        cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path / mode), 'model_path': 'ckpt', 'device': 'cuda:0',
               'rlft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
        torch.manual_seed(0)                   # (ahead of the policy: Conv1d biases of a fresh PlanningModel come from the global generator)
        pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
        with torch.no_grad():
            for p in pol.pluto_model.parameters():
                if p.dim() > 1:
                    p.normal_(0, 0.05)
        pol.load_model(resume=True)
        pol.set_mode('train')
        buf = _ragged_rift_buffer(45, 7000, caps={"A": 4})
        if mode == "packed":
            buf._host = {'CBVs_obs': False}
        pol.set_buffer(buf)
        assert (buf.host_replay() is not None) == (mode == "streamed")
        fit = pol.train(3)
        assert ("_arenas" in pol.__dict__) == (mode == "streamed")
        out[mode] = (torch.load(fit["checkpoint"], weights_only=False)["state_dict"], fit["history"])
    a, b = out["streamed"], out["packed"]
    assert [h["val_loss"] for h in a[1]] == [h["val_loss"] for h in b[1]] and [h["train_loss"] for h in a[1]] == [h["train_loss"] for h in b[1]]
    assert a[0].keys() == b[0].keys() and all(torch.equal(a[0][k], b[0][k]) for k in a[0])


@pytest.mark.gpu
def test_ppo_train_updates_pi_head_and_critic(tmp_path):
    """PPOPluto.train(e_i): two buffer sweeps + GAE + normalisation on the device, then epochs of actor + critic steps;
    the checkpoint carries value_net.*, the inference model ignores it (pluto.py:130-133)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0',
           'rlft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
    pol = CBV_POLICY_LIST['ppo_pluto'](cfg, None)
    pol.load_model(resume=True)
    pol.set_mode('train')
    torch.manual_seed(0)
    with torch.no_grad():
        for p in pol.train_model.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    pol.pluto_model.load_state_dict({k: v for k, v in pol.train_model.state_dict().items() if not k.startswith("value_net")})
    before = {k: v.detach().clone() for k, v in pol.train_model.state_dict().items()}
    pol.set_buffer(_filled_ppo_buffer(48))
    fit = pol.train(3)
    assert len(fit["history"]) == 2 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["val_loss"]) for h in fit["history"])
    sd = torch.load(fit["checkpoint"], weights_only=False)["state_dict"]
    assert any(k.startswith("model.value_net.net.0") for k in sd) and "model.value_net.state_avg" in sd
    after = pol.train_model.state_dict()
    changed = {k for k in before if not torch.equal(before[k].cpu(), after[k].cpu()) and "num_batches_tracked" not in k}
    assert any(k.startswith("planning_decoder.pi_head") for k in changed)
    for k in ("value_net.net.0.weight", "value_net.net.4.bias", "value_net.state_std", "value_net.value_avg"):
        assert k in changed, k           # every value_net parameter trains (the reference's freeze_parameters quirk)
    for k in changed:
        assert k.startswith("planning_decoder.pi_head") or k.startswith("value_net.") or "running_" in k, k
    assert not any(k.startswith("value_net") for k in pol.pluto_model.state_dict())


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["rift_pluto", "grpo_pluto", "reinforce_pluto"])
def test_rlft_train_updates_only_pi_head_and_checkpoints(policy, tmp_path):
    """RLFTPluto.train(e_i): 90/10 split, epochs of HIP steps, top-1 checkpoint with 'model.'-prefixed keys,
    inference model reloaded, buffer reset (rlft_pluto.py:206-247)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0',
           'rlft': {'epochs': 3, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
    pol = CBV_POLICY_LIST[policy](cfg, None)
    torch.manual_seed(0)
    with torch.no_grad():
        for p in pol.pluto_model.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    before = {k: v.detach().clone() for k, v in pol.pluto_model.state_dict().items()}
    pol.load_model(resume=True)
    assert pol.continue_episode == 0 and pol.current_epoch == 0
    pol.set_mode('train')
    buf = _filled_buffer(48, with_ref=policy == 'grpo_pluto')
    pol.set_buffer(buf)
    fit = pol.train(7)
    hist = fit["history"]
    assert len(hist) == 3 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["val_loss"]) for h in hist)
    assert abs(hist[0]["lr"] - 1e-3 / 1) < 1e-12 or hist[0]["lr"] > 0
    ck = list((tmp_path / 'ckpt' / pol.load_agent_info).glob("*.ckpt"))
    assert len(ck) == 1 and ck[0].name.startswith("carla_episode=7-epoch=") and "-val_loss=" in ck[0].name
    sd = torch.load(ck[0], weights_only=False)["state_dict"]
    assert all(k.startswith("model.") for k in sd)
    after = pol.pluto_model.state_dict()
    changed = [k for k in before if not torch.equal(before[k].cpu(), after[k].cpu()) and "num_batches_tracked" not in k]
    assert changed, "pi_head must have moved"
    for k in changed:
        assert k.startswith("planning_decoder.pi_head") or "running_" in k, k   # frozen trunk; BN running stats move in train mode
    assert any(k.startswith("planning_decoder.pi_head") for k in changed)
    assert len(buf) == 0 and not buf.buffer_full
    assert pol.current_epoch == 1 and pol.checkpoint == ck[0].as_posix()
    # the next update decays the base lr: lr * 0.9 ** current_epoch (rlft_pluto.py:212)
    pol.set_buffer(_filled_buffer(48, with_ref=policy == 'grpo_pluto'))
    fit2 = pol.train(8)
    assert abs(fit2["lr"] - 1e-3 * 0.9) < 1e-12
    pol2 = CBV_POLICY_LIST[policy](cfg, None)
    pol2.load_model(resume=True)
    assert pol2.continue_episode == 8 and pol2.current_epoch == 2


@pytest.mark.gpu
def test_deferred_top1_decision_writes_the_checkpoint_of_the_per_epoch_decision(tmp_path):
    """Round 5: the update keeps every epoch's losses and moving tensors on the device and decides top-1 ONCE, behind the last epoch (one host
    read per update instead of two per epoch).  It must choose what Lightning's per-epoch `ModelCheckpoint(save_top_k=1, monitor=loss/val_loss)`
    chooses (training_builder.py:131-140; here: `checkpoint_every_improvement: True`, the per-epoch path): the same file name (epoch and
    val_loss), bit-identical tensors, the same history -- fp32 mode, drops disabled by seeding both runs alike (same e_i -> same seeds)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    results = {}
    for mode, every in (("deferred", False), ("per_epoch", True)):
        root = tmp_path / mode
        cfg = {'num_scenario': 1, 'ROOT_DIR': str(root), 'model_path': 'ckpt', 'device': 'cuda:0', 'compute_precision': 'fp32',
               'rlft': {'epochs': 6, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 3e-3, 'checkpoint_every_improvement': every}}
        torch.manual_seed(0)            # ahead of the constructor: the Conv1d biases keep their default (random) initialisation
        pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
        torch.manual_seed(0)
        with torch.no_grad():
            for p in pol.pluto_model.parameters():
                if p.dim() > 1:
                    p.normal_(0, 0.05)
        pol.load_model(resume=True)
        pol.set_mode('train')
        pol.set_buffer(_filled_buffer(48, with_ref=False))
        fit = pol.train(5)
        ck = list((root / 'ckpt' / pol.load_agent_info).glob("*.ckpt"))
        assert len(ck) == 1
        results[mode] = (fit, ck[0].name, torch.load(ck[0], weights_only=False)["state_dict"])
    (fa, na, sa), (fb, nb, sb) = results["deferred"], results["per_epoch"]
    for h1, h2 in zip(fa["history"], fb["history"]):
        print(f"epoch {h1['epoch']}: deferred train {h1['train_loss']:.9f} val {h1['val_loss']:.9f} | per-epoch train {h2['train_loss']:.9f} val {h2['val_loss']:.9f}")
    assert na == nb, (na, nb)
    assert [h["val_loss"] for h in fa["history"]] == [h["val_loss"] for h in fb["history"]]
    assert [h["train_loss"] for h in fa["history"]] == pytest.approx([h["train_loss"] for h in fb["history"]], rel=1e-12, abs=1e-12)
    assert [h["lr"] for h in fa["history"]] == [h["lr"] for h in fb["history"]]
    assert fa["best_val_loss"] == fb["best_val_loss"]
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k


@pytest.mark.gpu
def test_failed_update_keeps_the_rollout_buffer(tmp_path, monkeypatch):
    """Round-5 advisor: the deferred top-1 path resets the buffer in the device's shadow, BEFORE the update's one host read, the finiteness
    check and the checkpoint write.  The reset is reversible now (CBVRolloutBuffer.reset_buffer_reversibly): an update that raises at any
    of those points leaves the buffer as it found it -- full, same rows, host mirror usable -- as the reference does (its reset follows a
    successful fit, rlft_pluto.py:244-250); the next, successful update trains on that data and only then empties the buffer."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0', 'compute_precision': 'fp32',
           'rlft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
    torch.manual_seed(0)
    pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
    pol.load_model(resume=True)
    pol.set_mode('train')
    buf = _filled_buffer(48, with_ref=False)
    pol.set_buffer(buf)
    n0, rows0 = buf.buffer_pos, list(buf._rows)
    real_write = type(pol)._write_checkpoint

    def failing_write(self, *a, **k):
        raise OSError("disk full")

    monkeypatch.setattr(type(pol), "_write_checkpoint", failing_write)
    with pytest.raises(OSError):
        pol.train(1)
    assert buf.buffer_full and buf.buffer_pos == n0 and all(a is b for a, b in zip(buf._rows, rows0))
    assert not list((tmp_path / 'ckpt').rglob("*.ckpt"))
    # the finiteness check raising at the one host read (the sticky flag): same outcome
    from rift_amd.planning.fine_tuner.rlft import trainer as T
    monkeypatch.setattr(type(pol), "_write_checkpoint", real_write)
    real_check = T.RLFTTrainer.check_finite
    monkeypatch.setattr(T.RLFTTrainer, "check_finite", lambda self: (_ for _ in ()).throw(FloatingPointError("non-finite decoder queries")))
    with pytest.raises(FloatingPointError):
        pol.train(2)
    assert buf.buffer_full and buf.buffer_pos == n0 and all(a is b for a, b in zip(buf._rows, rows0))
    monkeypatch.setattr(T.RLFTTrainer, "check_finite", real_check)
    fit = pol.train(3)                     # the same data, now committed
    assert len(fit["history"]) == 2 and buf.buffer_pos == 0 and not buf.buffer_full
    assert len(list((tmp_path / 'ckpt').rglob("*.ckpt"))) == 1


@pytest.mark.gpu
def test_rollout_side_inference_step():
    """PlutoInference: eval-mode HIP forward with every output, then top-k candidate trimming (integer flat indices identical to
    those from the oracle's logits in exact-fp32 mode), frame transforms and the waypoint PID."""
    from oracle import pluto_ref
    from rift_amd.planning.pluto.inference import PlutoInference, trim_candidates
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel, finish_outputs
    from tests import helpers as H
    torch.cuda.set_device(0)
    sd = H.weights()
    model = PlanningModel(radius=120)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    model.compute_precision = "fp32"
    scenes = [syn.make_scene(40 + i, num_agents=16, num_polygons=10, r_min=2, r_max=4) for i in range(3)]
    data = syn.collate_features([s["feature"] for s in scenes])
    pi = PlutoInference(model)
    out = pi.forward({k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in data.items()})
    want_raw, _, _ = pluto_ref.planning_model_forward(sd, data, train_bn=False, need_traj=True, want_taps=True)
    want = finish_outputs({k: v for k, v in want_raw.items()}, data, 21, True)
    origin, angle = np.array([3.0, -7.0]), 0.35
    for i in range(3):
        (thr, steer, brake), traj, cands, score, orig = pi.act(out, i, cbv_id=100 + i, origin=origin, angle=angle, speed=4.0)
        assert traj.shape[1] == 3 and np.isfinite(traj).all() and -1.0 <= steer <= 1.0 and 0.0 <= thr <= 1.0
        w_c, w_s, w_o, _, _ = trim_candidates(want["candidate_trajectories"][i].numpy().astype(np.float64), want["probability"][i].numpy(),
                                              origin, angle, want["output_ref_free_trajectory"][i].numpy().astype(np.float64), 10)
        assert np.array_equal(orig, w_o)
        assert np.abs(cands - w_c).max() < 1e-3 and np.abs(score - w_s).max() < 1e-4


@pytest.mark.gpu
def test_update_on_second_stream_equals_serial_update(tmp_path, monkeypatch):
    """RLFTTrainer.overlap_update (exchange + finalize + clip + AdamW of step k on a second stream while the frozen trunk of step k+1
    runs; the engine waits for the update's event before reading pi_head) and the deferred tail on top of it (RIFT_PIPELINE: policy
    head, loss and backward move to that stream too -- two activation arenas, two batch-buffer sets) give bit-identical parameters,
    history and checkpoint losses to the serial order (RIFT_NO_OVERLAP=1): same kernels, same seeds, only the stream placement differs.
    Likewise the input prefetch (RLFTTrainer.gather: the next batch's gather and the forward's input preparation on a third stream,
    rift_set_prepare_stream; four activation arenas / batch-buffer sets), with the next step's history and map encoders beside the current
    step's encoder / decoder (the default) or gated behind them (RIFT_SIDE_GATE=1)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    results = {}
    monkeypatch.setenv("RIFT_OVERLAP", "1")
    for mode, (no_overlap, pipeline, prefetch, gate) in {"0": ("0", "0", "0", "0"), "1": ("1", "0", "0", "0"), "2": ("0", "1", "1", "0"),
                                                         "3": ("0", "1", "0", "0"), "4": ("0", "1", "1", "1")}.items():
        monkeypatch.setenv("RIFT_NO_OVERLAP", no_overlap)
        monkeypatch.setenv("RIFT_PIPELINE", pipeline)
        monkeypatch.setenv("RIFT_PREFETCH", prefetch)
        monkeypatch.setenv("RIFT_SIDE_GATE", gate)
        root = tmp_path / mode
        cfg = {'num_scenario': 1, 'ROOT_DIR': str(root), 'model_path': 'ckpt', 'device': 'cuda:0',
               'rlft': {'epochs': 3, 'warmup_epochs': 1, 'train_batch_size': 8, 'val_batch_size': 8, 'lr': 1e-3}}
        torch.manual_seed(0)                       # before the policy is built: its constructor draws the 1-D parameters
        pol = CBV_POLICY_LIST["rift_pluto"](cfg, None)
        with torch.no_grad():
            for p in pol.pluto_model.parameters():
                if p.dim() > 1:
                    p.normal_(0, 0.05)
        pol.load_model(resume=True)
        pol.set_mode('train')
        pol.set_buffer(_filled_buffer(48, with_ref=False))
        fit = pol.train(1)
        torch.cuda.synchronize()
        results[mode] = (fit["history"], {k: v.detach().cpu().clone() for k, v in pol.pluto_model.state_dict().items()
                                          if k.startswith("planning_decoder.pi_head")})
    h0, p0 = results["0"]
    for other in ("1", "2", "3", "4"):
        h1, p1 = results[other]
        assert [(h["train_loss"], h["val_loss"]) for h in h0] == [(h["train_loss"], h["val_loss"]) for h in h1], other
        for k in p0:
            assert torch.equal(p0[k], p1[k]), (other, k)


@pytest.mark.gpu
def test_sft_kind_trains_pi_head_towards_the_teacher_mode():
    """RLFTTrainer(kind="sft"): the teacher cross entropy with the label built on the device falls over a few
    steps on a fixed minibatch and only pi_head moves."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    from tests import helpers as H
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(700 + i, num_agents=12, num_polygons=8, r_min=1, r_max=3) for i in range(16)]
    replay = DeviceReplay(scenes, dev, rcap=3)
    model = PlanningModel(radius=120)
    model.load_state_dict(H.weights())
    model = model.to(dev)
    model.train()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = RLFTTrainer(model, kind="sft", lr=3e-3, gradient_clip_val=0.5)
    idx = torch.arange(16, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(2)
    teacher = torch.stack([2.0 + 6.0 * torch.rand(16, generator=g), torch.zeros(16), torch.zeros(16), torch.zeros(16), 5.0 * torch.ones(16)], -1).to(dev)
    losses_ = []
    model._no_drop = True
    for _ in range(12):
        fb, b = replay.collate(tr.engine, idx)
        b = dict(b)
        b["teacher_infos"] = teacher
        tr.training_step(fb, b)
        tr.wait_update()
        losses_.append(float(tr.loss.item()))
    assert all(np.isfinite(losses_)) and losses_[-1] < losses_[0] - 0.05, losses_
    after = model.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k]) and "num_batches_tracked" not in k and "running_" not in k]
    assert moved and all(k.startswith("planning_decoder.pi_head") for k in moved), moved


@pytest.mark.gpu
def test_rtr_objective_is_five_ppo_plus_teacher():
    """RLFTTrainer(kind="rtr") (rtr_trainer.py:131-171: loss = 5 * PPO objective + teacher cross entropy, pi_head and value_net trainable):
    on one fixed minibatch in fp32 with the drops off, its loss and gradients equal 5 x those of kind="ppo" plus those of kind="sft"
    (each of which is checked against the oracle on its own), the critic's gradients 5 x PPO's -- and loss, pi_head and value_net
    gradients equal the oracle chain whose objective is pinned to the reference's rtr_trainer._compute_objectives (tests/golden/rtr.npz)."""
    from rift_amd.planning.fine_tuner.rlft.ppo_pluto.ppo_pluto import PPOPlutoModel
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.replay import DeviceReplay
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(800 + i, num_agents=12, num_polygons=8, r_min=1, r_max=3) for i in range(8)]
    replay = DeviceReplay(scenes, dev, rcap=3)
    idx = torch.arange(8, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(4)
    extras = {"teacher_infos": torch.stack([2.0 + 6.0 * torch.rand(8, generator=g), torch.zeros(8), torch.zeros(8), torch.zeros(8), 5.0 * torch.ones(8)], -1).to(dev),
              "action_mode": torch.stack([s["extras"]["action_mode"] for s in scenes]).to(dev),
              "advantage": torch.randn(8, generator=g).to(dev), "old_log_prob": (-2.0 - torch.rand(8, generator=g)).to(dev),
              "state": torch.randn(8, 128, generator=g).to(dev), "reward_sum": torch.randn(8, generator=g).to(dev)}
    torch.manual_seed(11)
    base = PPOPlutoModel(radius=120)
    sd0 = {k: v.clone() for k, v in base.state_dict().items()}
    res = {}
    for kind in ("ppo", "sft", "rtr"):
        model = PPOPlutoModel(radius=120)
        model.load_state_dict(sd0)
        model = model.to(dev)
        model.train()
        model.compute_precision, model._no_drop = "fp32", True
        layers = ("planning_decoder.pi_head",) if kind == "sft" else ("planning_decoder.pi_head", "value_net")
        tr = RLFTTrainer(model, kind=kind, trainable_layers=layers)
        fb, b = replay.collate(tr.engine, idx)
        b = dict(b)
        b.update(extras)
        loss = tr.forward_loss(fb, b, train=True)
        torch.cuda.synchronize()
        res[kind] = (float(loss.item()), {k: p.grad.detach().cpu().clone() for k, p in tr.params.items()},
                     {k: p.grad.detach().cpu().clone() for k, p in tr.critic.items()} if tr.critic else None)
        tr.engine.close()
    lp, gp, cp = res["ppo"]
    ls, gs, _ = res["sft"]
    lr_, gr, cr = res["rtr"]
    assert abs(lr_ - (5.0 * lp + ls)) < 1e-5 * max(1.0, abs(lr_))
    for k in gr:
        want = 5.0 * gp[k] + gs[k]
        assert float((gr[k] - want).abs().max()) < 1e-6 + 1e-5 * float(want.abs().max()), k
    for k in cr:
        assert float((cr[k] - 5.0 * cp[k]).abs().max()) < 1e-6 + 1e-5 * float(cp[k].abs().max()) * 5.0, k
    # ---- and against the ORACLE chain end to end: oracle forward (train-mode BatchNorm) -> pi_head / value_net with autograd ->
    # oracle/critic.rtr objective, which tests/test_oracle_critic.py pins to the reference's own rtr_trainer._compute_objectives (rtr.npz)
    from oracle import critic as ocr, losses, pluto_ref
    import torch.nn.functional as F
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    trunk = {k: v for k, v in sd0.items() if not k.startswith("value_net.")}
    out_o, _, taps = pluto_ref.planning_model_forward(trunk, data, train_bn=True, need_traj=True, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    PI = "planning_decoder.pi_head."
    pp = {k: sd0[PI + k].clone().requires_grad_(True) for k in losses.PI_KEYS}
    cpar = {k: sd0["value_net." + k].clone().requires_grad_(True) for k in ocr.CRITIC_KEYS}
    prob = pluto_ref.mlp_layer(taps["q_final"], pluto_ref.SD({PI + k: v for k, v in pp.items()}, PI)).squeeze(-1).masked_fill(r_pad.unsqueeze(-1), -1e6)
    ex = {k: v.cpu() for k, v in extras.items()}
    ppo = F.smooth_l1_loss(ocr.critic_forward(cpar, ex["state"]), ex["reward_sum"]) + \
        losses.ppo_actor_loss(prob, r_pad, ex["action_mode"], ex["advantage"], ex["old_log_prob"])
    teacher, _, _ = losses.sft_loss(prob, r_pad, out_o["trajectory"], ex["teacher_infos"])
    want = 5.0 * ppo + teacher
    want.backward()
    print(f"rtr vs oracle chain: loss {lr_:.6f} vs {float(want):.6f}")
    assert abs(lr_ - float(want)) < 2e-5 * max(1.0, abs(float(want)))
    for k in gr:
        assert float((gr[k] - pp[k].grad).abs().max()) < 1e-5 + 2e-4 * float(pp[k].grad.abs().max()), k
    for k in cr:
        assert float((cr[k] - cpar[k].grad).abs().max()) < 1e-5 + 2e-4 * float(cpar[k].grad.abs().max()), k


@pytest.mark.gpu
def test_training_steps_are_bit_deterministic():
    """The same seeded update steps (train mode: dropout, DropPath, BatchNorm batch statistics, ragged batches of 8, 3 and 5 scenes) on two
    fresh engines give bit-identical losses and parameters: no atomics, no uninitialised reads, no order-dependent reductions."""
    from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    from rift_amd.replay import DeviceReplay
    from tests import helpers as H
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scenes = [syn.make_scene(i, num_agents=12, num_polygons=8, r_min=1, r_max=3) for i in range(24)]
    runs = []
    for _ in range(2):
        replay = DeviceReplay(scenes, dev, rcap=3)
        model = PlanningModel(radius=120)
        model.load_state_dict(H.weights())
        model = model.to(dev)
        model.train()
        tr = RLFTTrainer(model, kind="rift")
        g = torch.Generator().manual_seed(3)
        ls = []
        for bs in (8, 3, 5, 8):
            idx = torch.randperm(24, generator=g)[:bs].to(torch.int32).to(dev)
            fb, b = replay.collate(tr.engine, idx, int(replay.r_count_cpu[idx.cpu().long()].max()))
            tr.training_step(fb, b)
            tr.wait_update()
            ls.append(float(tr.loss.item()))
        torch.cuda.synchronize()
        runs.append((ls, {k: v.detach().cpu().clone() for k, v in tr.params.items()}))
        tr.engine.close()
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


class _RecordedStates:
    """A CBVStateSource with recorded values (what CarlaStateSource reads off the simulator)."""

    def __init__(self, with_neighbours=True):
        from rift_amd.planning.pluto.pluto import CBVStateSource, CenterState
        self.base = CBVStateSource
        self.CenterState = CenterState
        self.with_neighbours = with_neighbours

    def center_state(self, env_id, cbv_id):
        return self.CenterState(10.0 + cbv_id, -5.0, 0.3, 6.0 + 0.1 * cbv_id, 2.0, 4.6)

    def nearby_actor_states(self, env_id, cbv_id):
        return H.other_vehicle_inputs(seed=100 + cbv_id, N=4) if self.with_neighbours else None

    def off_road_raster(self, env_id, cbv_id):
        mask = np.ones((400, 400), dtype=np.uint8)
        mask[150:250, :300] = 0
        return mask, (10.0 + cbv_id, -5.0, 0.3)


@pytest.mark.gpu
def test_fused_tick_equals_the_per_cbv_evaluator_chain(tmp_path):
    """rift_group_advantage_tick (one C-ABI call and one staged upload per environment and tick) against the per-CBV chain of
    TrajEvaluator.get_grpo_advantage (config['fused_tick'] = False): the same kernels in the same order on the same inputs, so every
    column of every tick is equal bit for bit -- over ticks with one to five CBVs, ragged reference-line counts, the shared never-reset
    PID state carried from tick to tick, and a source that reports no neighbours for some CBVs and no raster for others."""
    from rift_amd.planning import CBV_POLICY_LIST
    from rift_amd.planning.pluto.pluto import NoFlagSource
    torch.cuda.set_device(0)

    class Mixed(_RecordedStates):
        def nearby_actor_states(self, env_id, cbv_id):
            if cbv_id % 3 == 0:
                return None
            return NoFlagSource.ALL_CLEAR if cbv_id % 5 == 4 else super().nearby_actor_states(env_id, cbv_id)

        def off_road_raster(self, env_id, cbv_id):
            return NoFlagSource.ALL_CLEAR if cbv_id % 4 == 1 else super().off_road_raster(env_id, cbv_id)

    sd = H.weights()
    runs = {}
    for fused in (True, False):
        cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0', 'state_source': Mixed(), 'fused_tick': fused}
        pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
        pol.pluto_model.load_state_dict(sd)
        pol.set_mode('train')
        cols = []
        for t, ids in enumerate([[1], [2, 3, 4], [5, 6, 7, 8, 9], [1, 4], [3, 5, 9, 2]]):
            feats = {c: syn.make_scene(5000 + 16 * t + c, num_agents=12, num_polygons=8, r_min=1, r_max=5)["feature"] for c in ids}
            obs = {c: {'raw_pluto_feature': PlutoFeature(data=feats[c])} for c in ids}
            act = pol.get_action([obs], [{'env_id': 0}], deterministic=False)
            for c in ids:
                R = int(np.asarray(feats[c]["reference_line"]["valid_mask"]).any(-1).sum())
                adv = act['CBVs_group_advantage'][0][c]
                assert adv['advantage'].shape == (R, 12) and adv['advantage'].dtype == np.float64 and adv['valid_mask'].shape == (R, 12)
                assert np.isfinite(adv['advantage']).all()
                cols.append((adv['advantage'], act['CBVs_actions_old_group_logits'][0][c]['logits'], np.asarray(act['CBVs_actions'][0][c])))
        runs[fused] = cols
        pol.pluto_model.release_engine()
    for (a0, l0, c0), (a1, l1, c1) in zip(runs[True], runs[False]):
        assert np.array_equal(a0, a1) and np.array_equal(l0, l1) and np.array_equal(c0, c1)


@pytest.mark.gpu
def test_staging_arena_wraps_and_large_inputs_take_the_ordinary_path():
    """Engine.stage (pinned arena + asynchronous copies for a tick's small host inputs): contents and dtypes survive the trip, the arena
    wraps behind a stream synchronisation once its 8 MiB are used up (40 x 300 KB here), inputs above a quarter of it and empty ones take
    the ordinary upload, device tensors pass through."""
    from rift_amd import _ffi
    torch.cuda.set_device(0)
    eng = _ffi.Engine("cuda:0")
    g = np.random.default_rng(3)
    for k in range(40):
        a = g.normal(size=(75, 1000)).astype(np.float32)             # 300 KB
        b = (g.random(1000) < 0.5)
        c = g.integers(-5, 5, size=(17, 3)).astype(np.int32)
        da, db, dc = eng.stage(a, torch.float32), eng.stage(b, torch.bool), eng.stage(torch.from_numpy(c), torch.int32)
        assert da.dtype == torch.float32 and db.dtype == torch.bool and dc.dtype == torch.int32 and da.is_cuda
        assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(db.cpu().numpy(), b) and np.array_equal(dc.cpu().numpy(), c), k
    assert eng._stage_off < eng._STAGE_BYTES
    big = g.normal(size=(3 << 20,)).astype(np.float32)               # 12 MB: not through the arena
    off = eng._stage_off
    assert np.array_equal(eng.stage(big, torch.float32).cpu().numpy(), big) and eng._stage_off == off
    assert eng.stage(np.zeros((0, 4), np.float64), torch.float64).shape == (0, 4)
    d64 = eng.stage(np.arange(7, dtype=np.float32), torch.float64)   # conversion on the host
    assert d64.dtype == torch.float64 and np.array_equal(d64.cpu().numpy(), np.arange(7, dtype=np.float64))
    t = torch.arange(5, device="cuda:0", dtype=torch.float32)
    assert eng.stage(t, torch.float32).data_ptr() == t.data_ptr()
    eng.close()


@pytest.mark.gpu
def test_tick_with_a_gap_in_the_valid_lines_falls_back_to_the_chain(tmp_path):
    """The fused tick call takes CBVs whose valid reference lines are a prefix of their rows (what PlutoFeature produces).  A CBV with an
    all-invalid line BETWEEN valid ones sends the whole environment through the per-CBV chain, in tick order; both policies agree bit for
    bit, and a 12-CBV tick (its rasters exceed the staged-upload limit) goes through the fused call on the ordinary upload path."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    sd = H.weights()
    runs = {}
    for fused in (True, False):
        cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0', 'state_source': _RecordedStates(), 'fused_tick': fused}
        pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
        pol.pluto_model.load_state_dict(sd)
        pol.set_mode('train')
        calls = []
        eng = pol.traj_evaluator.engine
        orig = eng.group_advantage_tick
        eng.group_advantage_tick = lambda *a, **k: (calls.append(len(a[1])), orig(*a, **k))[1]
        cols = []
        for t, ids in enumerate([[1, 2, 3], list(range(1, 13))]):
            feats = {c: syn.make_scene(6100 + 16 * t + c, num_agents=12, num_polygons=8, r_min=3, r_max=5)["feature"] for c in ids}
            if t == 0:
                feats[2]["reference_line"]["valid_mask"][1] = False          # a gap: lines 0 and 2.. stay valid
            obs = {c: {'raw_pluto_feature': PlutoFeature(data=feats[c])} for c in ids}
            act = pol.get_action([obs], [{'env_id': 0}], deterministic=False)
            for c in ids:
                R = int(np.asarray(feats[c]["reference_line"]["valid_mask"]).any(-1).sum())
                adv = act['CBVs_group_advantage'][0][c]['advantage']
                assert adv.shape == (R, 12) and np.isfinite(adv).all()
                cols.append(adv)
        assert calls == ([12] if fused else []), calls                    # the gap tick never reached the fused call
        runs[fused] = cols
        pol.pluto_model.release_engine()
    for a0, a1 in zip(runs[True], runs[False]):
        assert np.array_equal(a0, a1)


def test_state_source_contract():
    """The group advantage never runs on silently missing inputs: a source that does not implement the neighbour / raster readings fails
    the train-mode tick (the reference always feeds both, rift_pluto.py:113-135); the PID gets the CENTRE speed (pluto.py:252), the
    rollout the rear-axle speed (track_propogate.py:628); the raster helper places polygon vertices where the oracle's global_to_pixel does."""
    from oracle import traj_flags as otf
    from rift_amd.planning.pluto.pluto import CBVStateSource, CenterState, NoFlagSource, raster_drivable_area
    src = CBVStateSource()
    for call in (src.nearby_actor_states, src.off_road_raster):
        with pytest.raises(NotImplementedError, match="group advantage needs"):
            call(0, 1)
    st = CenterState(1.0, 2.0, 0.3, 6.0, 2.0, 4.6, center_speed=6.4)
    assert st.rollout_tuple() == (1.0, 2.0, 0.3, 6.0, 2.0, 4.6) and st.pid_speed() == 6.4
    assert CenterState(1.0, 2.0, 0.3, 6.0, 2.0, 4.6).pid_speed() == 6.0
    assert NoFlagSource.ALL_CLEAR is not None
    rng = np.random.default_rng(5)
    polys = [rng.normal(0, 40, size=(5, 2)) + np.array([12.0, -7.0]) for _ in range(3)]
    seen = []
    mask = raster_drivable_area(polys, (12.0, -7.0), 0.4, fill_polygon=lambda m, v, val: seen.append((v.copy(), val)))
    assert mask.shape == (400, 400) and mask.dtype == np.uint8 and mask.all() and len(seen) == 3
    rot = np.array([[np.cos(0.4), -np.sin(0.4)], [np.sin(0.4), np.cos(0.4)]])
    for poly, (v, val) in zip(polys, seen):
        want = np.round(otf.global_to_pixel(poly, np.array([12.0, -7.0]), rot, np.array([0.5, -0.5], dtype=np.float32),
                                            np.array([200.0, 200.0], dtype=np.float32))).astype(np.int32)
        assert val == 0 and v.dtype == np.int32 and np.array_equal(v, want)


@pytest.mark.gpu
def test_rollout_fills_the_buffer_and_the_update_trains_on_it(tmp_path):
    """The closed loop of train_cbv without the reference classes (carla_runner.py:207-235): RIFTPluto.get_action in train mode ->
    buffer.store -> RIFTPluto.train.  The columns get_action emits are checked against the oracle chain: old group logits = the
    model's logits of the valid reference lines, chosen control = PlutoInference on the same outputs, group advantage = oracle
    rollout + reward + z-score on the oracle-side flags (1e-4; exact-fp32 mode)."""
    from oracle import advantage as oadv, pluto_ref, rollout as orl, traj_flags as otf
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    src = _RecordedStates()
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0', 'state_source': src,
           'rlft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
    pol = CBV_POLICY_LIST['rift_pluto'](cfg, None)
    sd = H.weights()
    pol.pluto_model.load_state_dict(sd)
    pol.pluto_model.compute_precision = "fp32"
    pol.load_model(resume=True)
    pol.set_mode('train')
    keys = ['CBVs_obs', 'CBVs_actions', 'CBVs_reward', 'CBVs_done', 'CBVs_actions_old_group_logits', 'CBVs_group_advantage']
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': 36, 'data_keys': keys})
    pol.set_buffer(buf)
    t, checked = 0, 0
    while not buf.buffer_full:
        ids = [1, 2]
        feats = {c: syn.make_scene(3000 + 2 * t + j, num_agents=12, num_polygons=8, r_min=1, r_max=3)["feature"] for j, c in enumerate(ids)}
        obs = {c: {'raw_pluto_feature': PlutoFeature(data=feats[c])} for c in ids}
        act = pol.get_action([obs], [{'env_id': 0}], deterministic=False)
        assert set(act) == {'CBVs_actions', 'CBVs_actions_old_group_logits', 'CBVs_group_advantage'}
        for c in ids:
            thr, steer, brake = act['CBVs_actions'][0][c]
            R = feats[c]["reference_line"]["position"].shape[0]
            lg, adv = act['CBVs_actions_old_group_logits'][0][c], act['CBVs_group_advantage'][0][c]
            assert lg['logits'].shape == (R, 12) and lg['valid_mask'].all() and adv['advantage'].shape == (R, 12) and adv['advantage'].dtype == np.float64
            assert 0.0 <= thr <= 1.0 and -1.0 <= steer <= 1.0
        if t < 2:                                             # oracle chain for the first ticks (first calls of the persistent PID state per lane)
            data = syn.collate_features([feats[c] for c in ids])
            want, _, _ = pluto_ref.planning_model_forward(sd, data, train_bn=False, need_traj=True)
            for j, c in enumerate(ids):
                R = feats[c]["reference_line"]["position"].shape[0]
                lg = act['CBVs_actions_old_group_logits'][0][c]['logits']
                assert np.abs(lg - want["probability"][j, :R].numpy()).max() < 1e-4
            checked += 1
        buf.store({'CBV_ids': [ids], 'CBVs_obs': [obs], 'CBVs_actions': [act['CBVs_actions'][0]], 'CBVs_reward': [{c: 0.1 for c in ids}],
                   'CBVs_done': [{c: t % 6 == 5 for c in ids}], 'CBVs_actions_old_group_logits': [act['CBVs_actions_old_group_logits'][0]],
                   'CBVs_group_advantage': [act['CBVs_group_advantage'][0]]})
        t += 1
    assert checked == 2 and buf.buffer_full
    # group advantage of one fresh CBV against the oracle chain (fresh policy -> fresh PID state on both sides)
    pol2 = CBV_POLICY_LIST['rift_pluto'](dict(cfg, state_source=_RecordedStates(with_neighbours=True)), None)
    pol2.pluto_model.load_state_dict(sd)
    pol2.pluto_model.compute_precision = "fp32"
    pol2.set_mode('eval'); pol2.mode = 'train'
    feat = syn.make_scene(3333, num_agents=12, num_polygons=8, r_min=2, r_max=3)["feature"]
    act = pol2.get_action([{7: {'raw_pluto_feature': PlutoFeature(data=feat)}}], [{'env_id': 0}])
    got = act['CBVs_group_advantage'][0][7]['advantage']
    data = syn.collate_features([feat])
    want, _, _ = pluto_ref.planning_model_forward(sd, data, train_bn=False, need_traj=True)
    R = feat["reference_line"]["position"].shape[0]
    traj = want["trajectory"][0, :R]
    valid = feat["reference_line"]["valid_mask"]
    ref_pos = [feat["reference_line"]["position"][r][valid[r]] for r in range(R)]
    ref_ang = [feat["reference_line"]["orientation"][r][valid[r]] for r in range(R)]
    st = src.center_state(0, 7)
    t40 = traj[:, :, :40]
    dd, da, _ = orl.ref_line_info(t40, ref_pos, ref_ang)
    gpos, ghead = orl.to_global(t40, torch.tensor([st.x, st.y]), torch.tensor(st.heading))
    ro = orl.Rollout().propagate(gpos, ghead, st.speed, st.width, st.length)
    other = otf.get_other_vehicle_rollout(**src.nearby_actor_states(0, 7))
    col = otf.get_collision_matrix(ro["vertices"].numpy(), other)
    mask, pose = src.off_road_raster(0, 7)
    off = otf.get_off_road_matrix(ro["center"].numpy(), mask, pose[:2], pose[2])
    ret = oadv.rollout_return(dd.numpy(), da.numpy(), ro["speed"][:, :40].numpy(), ro["acc"][:, :40].numpy(), ro["ang_vel"][:, :40].numpy(),
                              ro["ang_acc"][:, :40].numpy(), col, off)
    assert np.abs(got - oadv.group_zscore(ret).reshape(R, 12)).max() < 1e-4
    # and the update trains on what the rollout stored
    fit = pol.train(1)
    assert len(fit["history"]) == 2 and all(np.isfinite(h["train_loss"]) for h in fit["history"]) and len(buf) == 0


@pytest.mark.gpu
def test_pluto_policy_and_ppo_columns(tmp_path):
    """Registry key 'pluto' (inference only: refuses train mode, needs a checkpoint) and the base RLFT columns (old log-prob, chosen
    (r, m) -- integers recomputed here from the emitted logits)."""
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    assert set(CBV_POLICY_LIST) == {'pluto', 'rift_pluto', 'grpo_pluto', 'reinforce_pluto', 'ppo_pluto', 'sft_pluto', 'rs_pluto', 'rtr_pluto'}
    sd = H.weights()
    ck = tmp_path / "pluto.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, ck)
    src = _RecordedStates()
    cfg = {'num_scenario': 2, 'device': 'cuda:0', 'state_source': src, 'ckpt_path': str(ck), 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt'}
    plain = CBV_POLICY_LIST['pluto'](cfg, None)
    with pytest.raises(ValueError):
        plain.set_mode('train')
    with pytest.raises(FileNotFoundError):
        CBV_POLICY_LIST['pluto'](dict(cfg, ckpt_path=None), None).load_model()
    plain.load_model()
    plain.set_mode('eval')
    feats = {c: syn.make_scene(4100 + c, num_agents=12, num_polygons=8, r_min=2, r_max=4)["feature"] for c in (5, 6)}
    obs = [{}, {c: {'raw_pluto_feature': PlutoFeature(data=f)} for c, f in feats.items()}]          # environment 0 has no CBV this tick
    infos = [{'env_id': 0}, {'env_id': 1}]
    a = plain.get_action(obs, infos)
    assert set(a) == {'CBVs_actions'} and a['CBVs_actions'][0] == {} and set(a['CBVs_actions'][1]) == {5, 6}
    ppo = CBV_POLICY_LIST['ppo_pluto'](cfg, None)
    ppo.load_model(resume=True)
    b = ppo.get_action(obs, infos)
    assert set(b) == {'CBVs_actions', 'CBVs_actions_old_log_prob', 'CBVs_actions_mode'}
    for c in (5, 6):
        assert b['CBVs_actions'][1][c] == a['CBVs_actions'][1][c]              # same model, same fresh PID state
        r, m = b['CBVs_actions_mode'][1][c]
        R = feats[c]["reference_line"]["position"].shape[0]
        # (r, m) = divmod(flat index, 12); the ref-free candidate carries flat index -1 -> (-1, 11), as in the reference (rlft_pluto.py:174-176)
        assert -1 <= r < R and 0 <= m < 12 and (r >= 0 or m == 11) and b['CBVs_actions_old_log_prob'][1][c] <= 0.0
    # a CBV that disappears frees its PID state
    plain.get_action([{}, {5: obs[1][5]}], infos)
    assert set(plain.controllers[1]) == {5}


def test_normalize_and_tensorisation_match_the_reference_generated_fixture():
    """tests/golden/normalize.npz = the reference's PlutoFeature.normalize + to_feature_tensor (pluto_feature.py:98-126,166-263) on a seeded
    raw global-frame feature dict -- first call (map crop to +-radius, origin / angle recorded) and a later call.  The mirror must give the
    same keys, shapes, dtypes (float64 -> float32 only in the tensor form) and values, bit for bit."""
    import os
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "normalize.npz")))
    for tag, first in (("first", True), ("later", False)):
        res = PlutoFeature.normalize(H.raw_feature_inputs(), first_time=first, radius=120, hist_steps=21)
        assert res.is_valid
        for name, src in (("np", res.data), ("tensor", res.to_feature_tensor().data)):
            flat = {}
            for grp, v in src.items():
                for k, t in (v.items() if isinstance(v, dict) else [(None, v)]):
                    flat[f"{tag}/{name}/{grp}" + (f"/{k}" if k else "")] = t.numpy() if torch.is_tensor(t) else np.asarray(t)
            want = {k for k in gold if k.startswith(f"{tag}/{name}/")}
            assert set(flat) == want, set(flat) ^ want
            for k, v in flat.items():
                assert v.shape == gold[k].shape and v.dtype == gold[k].dtype, (k, v.shape, v.dtype, gold[k].shape, gold[k].dtype)
                assert np.array_equal(v, gold[k]), k
    assert gold["first/np/map/point_position"].shape[0] < H.raw_feature_inputs()["map"]["point_position"].shape[0]     # the crop removed polygons
    # round trip through numpy
    back = PlutoFeature.normalize(H.raw_feature_inputs(), first_time=True, radius=120).to_feature_tensor().to_numpy()
    assert isinstance(back.data["agent"]["position"], np.ndarray) and back.data["agent"]["position"].dtype == np.float32


# ---- the SFT family's update side (fine_tuner/sft/{sft_pluto, rs_pluto/rs_pluto, rtr_pluto/rtr_pluto}.py + their datamodules) ----------------
SFT_KEYS = {'sft_pluto': ['CBVs_actions', 'CBVs_teacher_infos', 'CBVs_obs', 'CBVs_next_obs', 'CBVs_reward', 'CBVs_terminated', 'CBVs_done'],
            'rs_pluto': ['CBVs_actions', 'CBVs_teacher_rewards', 'CBVs_obs', 'CBVs_next_obs', 'CBVs_reward', 'CBVs_terminated', 'CBVs_done'],
            'rtr_pluto': ['CBVs_actions', 'CBVs_actions_old_log_prob', 'CBVs_actions_mode', 'CBVs_teacher_infos', 'CBVs_obs', 'CBVs_next_obs',
                          'CBVs_reward', 'CBVs_terminated', 'CBVs_done']}        # planning/config/{sft,rs,rtr}_pluto.yaml: data_keys


def _filled_sft_buffer(policy, n):
    """8-step episodes of one CBV with the teacher columns the reference's SFT-family rollouts store (sft_pluto.py:229-241: five floats, the
    teacher's target speed first; rs_pluto.py:134-136: -|teacher speed - desired speed|)."""
    buf = CBVRolloutBuffer(1, 'train_cbv', {'buffer_capacity': n, 'data_keys': SFT_KEYS[policy]})
    g = torch.Generator().manual_seed(5)
    t = 0
    while not buf.buffer_full:
        for k in range(8):
            s, s2 = (syn.make_scene(t + j, num_agents=12, num_polygons=8, r_min=1, r_max=3) for j in (0, 1))
            ex = s["extras"]
            speed = float(2.0 + 6.0 * torch.rand((), generator=g))
            d = {'CBV_ids': [[3]], 'CBVs_obs': [{3: {'raw_pluto_feature': PlutoFeature(data=s["feature"])}}],
                 'CBVs_next_obs': [{3: {'raw_pluto_feature': PlutoFeature(data=s2["feature"])}}],
                 'CBVs_actions': [{3: np.zeros(3, np.float32)}], 'CBVs_reward': [{3: float(ex["return"])}], 'CBVs_done': [{3: k == 7}],
                 'CBVs_terminated': [{3: k == 7 and t % 16 == 7}],
                 'CBVs_teacher_infos': [{3: torch.tensor([speed, 0.0, 0.0, 0.0, 5.0])}], 'CBVs_teacher_rewards': [{3: -abs(speed - 5.0)}],
                 'CBVs_actions_old_log_prob': [{3: np.float32(ex["old_log_prob"])}], 'CBVs_actions_mode': [{3: ex["action_mode"].numpy()}]}
            buf.store(d)
            t += 1
    return buf


def test_sft_family_registry_and_reward_shaping_columns():
    """The three SFT-family policies are in the registry under the reference's names with its types; RewardShapingPluto's reward column is
    CBVs_reward + 0.2 * CBVs_teacher_rewards (rs_datamodule.py:109-114, datamodule/rs_datamodule.yaml) and its return the oracle's scan;
    SFTPluto's teacher column is the (n, 5) stack SFTCollate yields."""
    from oracle import advantage as oadv
    from rift_amd.planning import CBV_POLICY_LIST
    from rift_amd.planning.fine_tuner.sft.sft_pluto import teacher_column
    assert {'sft_pluto', 'rs_pluto', 'rtr_pluto'} <= set(CBV_POLICY_LIST)
    assert [CBV_POLICY_LIST[k].kind for k in ('sft_pluto', 'rs_pluto', 'rtr_pluto')] == ['sft', 'rs', 'rtr']
    assert all(CBV_POLICY_LIST[k].type == 'learnable' for k in ('sft_pluto', 'rs_pluto', 'rtr_pluto'))
    buf = _filled_sft_buffer('rs_pluto', 24)
    pol = CBV_POLICY_LIST['rs_pluto']({'num_scenario': 1, 'device': 'cpu'}, None)
    pol.set_buffer(buf)
    r = pol.shaped_rewards()
    want = np.asarray(buf.get_key_data('CBVs_reward'), np.float64) + 0.2 * np.asarray(buf.get_key_data('CBVs_teacher_rewards'), np.float64)
    assert r.dtype == torch.float64 and np.array_equal(r.numpy(), want)
    ret = oadv.compute_return(torch.as_tensor(want), torch.as_tensor(np.asarray(buf.get_key_data('CBVs_done'), np.float64)), 0.98).numpy()
    assert abs(ret[7] - want[7]) < 1e-12 and abs(ret[6] - (want[6] + 0.98 * want[7])) < 1e-12          # episodes of 8: the scan restarts at a done
    col = teacher_column(_filled_sft_buffer('sft_pluto', 16), 'cpu')
    assert col.shape == (16, 5) and col.dtype == torch.float32 and torch.all(col[:, 4] == 5.0)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["sft_pluto", "rs_pluto", "rtr_pluto"])
def test_sft_family_train_updates_the_configured_layers(policy, tmp_path):
    """{SFT, RewardShaping, RTR}Pluto.train(e_i) on a buffer with the reference's data keys: the same update loop as the RLFT family (16-epoch
    schedule shortened here), the teacher rows / shaped returns / PPO columns prepared on the device once per update; pi_head moves, for RTR
    value_net too, nothing else; RS's returns are the oracle's scan of the shaped rewards."""
    from oracle import advantage as oadv
    from rift_amd.planning import CBV_POLICY_LIST
    torch.cuda.set_device(0)
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(tmp_path), 'model_path': 'ckpt', 'device': 'cuda:0',
           'sft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 16, 'val_batch_size': 16, 'lr': 1e-3}}
    pol = CBV_POLICY_LIST[policy](cfg, None)
    pol.load_model(resume=True)
    pol.set_mode('train')
    torch.manual_seed(0)
    with torch.no_grad():
        for p in pol.train_model.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    pol.pluto_model.load_state_dict({k: v for k, v in pol.train_model.state_dict().items() if not k.startswith("value_net")})
    before = {k: v.detach().clone() for k, v in pol.train_model.state_dict().items()}
    buf = _filled_sft_buffer(policy, 48)
    pol.set_buffer(buf)
    if policy == 'rs_pluto':
        from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
        want = oadv.compute_return(pol.shaped_rewards(), torch.as_tensor(np.asarray(buf.get_key_data('CBVs_done'), np.float64)), 0.98).numpy()
        tr = RLFTTrainer(pol.train_model, kind='rs')
        got = pol.preprocess_buffer(tr, None)["returns"]
        tr.close()
        assert got.dtype == torch.float32 and np.allclose(got.cpu().numpy(), want, rtol=0, atol=1e-5 * max(1.0, np.abs(want).max()))
    fit = pol.train(5)
    assert len(fit["history"]) == 2 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["val_loss"]) for h in fit["history"])
    after = pol.train_model.state_dict()
    changed = {k for k in before if not torch.equal(before[k].cpu(), after[k].cpu()) and "num_batches_tracked" not in k}
    assert any(k.startswith("planning_decoder.pi_head") for k in changed)
    for k in changed:
        assert k.startswith("planning_decoder.pi_head") or "running_" in k or (policy == 'rtr_pluto' and k.startswith("value_net.")), k
    if policy == 'rtr_pluto':
        assert "value_net.net.0.weight" in changed
    assert len(buf) == 0 and pol.current_epoch == 1
