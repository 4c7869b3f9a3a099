"""Host-side ``PlanningModel``: the reference's module tree as a *parameter container*
(identical ``state_dict()`` names, shapes, order and default initialisation --
pluto/model/pluto_model.py:22-120 and SURVEY.md Appendix B), whose ``forward`` runs
entirely in ``librift_hip.so``.

torch.nn modules are used only to own parameters / buffers, so that checkpoints,
``named_modules()``-based freezing (rift_trainer.py:78-90) and the AdamW parameter
grouping by module type (rift_trainer.py:279-362) behave exactly as in the reference.
None of their ``forward`` methods is ever called; there is no CPU fallback.
"""
import math
import weakref
from typing import Dict

import torch
import torch.nn as nn

from rift_amd.nuplan_plugin.modeling.torch_module_wrapper import TorchModuleWrapper
from rift_amd.nuplan_plugin.modeling.types import FeaturesType, TargetsType
from rift_amd.nuplan_plugin.trajectory.trajectory_sampling import TrajectorySampling

trajectory_sampling = TrajectorySampling(num_poses=8, time_horizon=8, interval_length=1)


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the computation runs in librift_hip.so")


def _mlp(cin, hidden, cout):  # MLPLayer.mlp / Fourier mlps layout: Linear, LayerNorm, ReLU, Linear
    return nn.Sequential(nn.Linear(cin, hidden), nn.LayerNorm(hidden), nn.ReLU(inplace=True), nn.Linear(hidden, cout))


class MLPLayer(_Container):
    def __init__(self, cin, hidden, cout):
        super().__init__()
        self.mlp = _mlp(cin, hidden, cout)


class FourierEmbedding(_Container):
    def __init__(self, input_dim, hidden_dim, num_freq_bands):
        super().__init__()
        self.freqs = nn.Embedding(input_dim, num_freq_bands)
        self.mlps = nn.ModuleList([_mlp(num_freq_bands * 2 + 1, hidden_dim, hidden_dim) for _ in range(input_dim)])
        self.to_out = nn.Sequential(nn.LayerNorm(hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, hidden_dim))


class PointsEncoder(_Container):
    def __init__(self, feat_channel, encoder_channel):
        super().__init__()
        self.first_mlp = nn.Sequential(nn.Linear(feat_channel, 128), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                       nn.Linear(128, 256))
        self.second_mlp = nn.Sequential(nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                        nn.Linear(256, encoder_channel))


class _NAttn(_Container):   # natten NeighborhoodAttention1D parameter layout: rpb, qkv, proj
    def __init__(self, dim, heads, ksz):
        super().__init__()
        self.rpb = nn.Parameter(torch.zeros(heads, 2 * ksz - 1))
        nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Container):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _NATLayer(_Container):
    def __init__(self, dim, heads, ksz, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _NAttn(dim, heads, ksz)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _Down(_Container):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Conv1d(dim, 2 * dim, kernel_size=3, stride=2, padding=1, bias=False)
        self.norm = nn.LayerNorm(2 * dim)


class _NATBlock(_Container):
    def __init__(self, dim, depth, heads, ksz, mlp_ratio, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([_NATLayer(dim, heads, ksz, mlp_ratio) for _ in range(depth)])
        self.downsample = _Down(dim) if downsample else None


class _Tok(_Container):
    def __init__(self, cin, dim):
        super().__init__()
        self.proj = nn.Conv1d(cin, dim, kernel_size=3, stride=1, padding=1)


class NATSequenceEncoder(_Container):
    def __init__(self, in_chans=9, embed_dim=32):
        super().__init__()
        ks, heads = [3, 3, 5], [2, 4, 8]
        self.embed = _Tok(in_chans, embed_dim)
        self.levels = nn.ModuleList([_NATBlock(embed_dim * 2 ** i, 2, heads[i], ks[i], 3, i < 2) for i in range(3)])
        for i in range(3):
            self.add_module(f"norm{i}", nn.LayerNorm(embed_dim * 2 ** i))
        n = embed_dim * 4
        self.lateral_convs = nn.ModuleList([nn.Conv1d(embed_dim * 2 ** i, n, 3, padding=1) for i in range(3)])
        self.fpn_conv = nn.Conv1d(n, n, 3, padding=1)


class StateAttentionEncoder(_Container):
    def __init__(self, state_channel, dim):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(1, dim) for _ in range(state_channel)])
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=4, batch_first=True)
        self.pos_embed = nn.Parameter(torch.Tensor(1, state_channel, dim))
        self.query = nn.Parameter(torch.Tensor(1, 1, dim))
        nn.init.normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.query, std=0.02)


class AgentEncoder(_Container):
    def __init__(self, state_channel, history_channel, dim):
        super().__init__()
        self.history_encoder = NATSequenceEncoder(history_channel, dim // 4)
        self.ego_state_emb = StateAttentionEncoder(state_channel, dim)
        self.type_emb = nn.Embedding(4, dim)


class MapEncoder(_Container):
    def __init__(self, polygon_channel, dim):
        super().__init__()
        self.polygon_encoder = PointsEncoder(polygon_channel + 4, dim)
        self.speed_limit_emb = FourierEmbedding(1, dim, 64)
        self.type_emb = nn.Embedding(3, dim)
        self.on_route_emb = nn.Embedding(2, dim)
        self.traffic_light_emb = nn.Embedding(4, dim)
        self.unknown_speed_emb = nn.Embedding(1, dim)


class StaticObjectsEncoder(_Container):
    def __init__(self, dim):
        super().__init__()
        self.obj_encoder = FourierEmbedding(2, dim, 64)
        self.type_emb = nn.Embedding(4, dim)
        nn.init.normal_(self.type_emb.weight, mean=0.0, std=0.01)


class TransformerEncoderLayer(_Container):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(dim, num_heads=num_heads, batch_first=True)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, dim * 4)


class AgentPredictor(_Container):
    def __init__(self, dim, future_steps):
        super().__init__()
        self.loc_predictor = MLPLayer(dim, 2 * dim, future_steps * 2)
        self.yaw_predictor = MLPLayer(dim, 2 * dim, future_steps * 2)
        self.vel_predictor = MLPLayer(dim, 2 * dim, future_steps * 2)


class DecoderLayer(_Container):
    def __init__(self, dim, num_heads, mlp_ratio, dropout):
        super().__init__()
        self.r2r_attn = nn.MultiheadAttention(dim, num_heads, dropout=dropout, batch_first=True)
        self.m2m_attn = nn.MultiheadAttention(dim, num_heads, dropout=dropout, batch_first=True)
        self.cross_attn = nn.MultiheadAttention(dim, num_heads, dropout=dropout, batch_first=True)
        self.ffn = nn.Sequential(nn.Linear(dim, dim * mlp_ratio), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                 nn.Linear(dim * mlp_ratio, dim))
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.norm3, self.norm4 = nn.LayerNorm(dim), nn.LayerNorm(dim)


class PlanningDecoder(_Container):
    def __init__(self, num_mode, decoder_depth, dim, num_heads, mlp_ratio, dropout, future_steps):
        super().__init__()
        self.decoder_blocks = nn.ModuleList([DecoderLayer(dim, num_heads, mlp_ratio, dropout)
                                             for _ in range(decoder_depth)])
        self.r_pos_emb = FourierEmbedding(3, dim, 64)
        self.r_encoder = PointsEncoder(6, dim)
        self.q_proj = nn.Linear(2 * dim, dim)
        self.m_emb = nn.Parameter(torch.Tensor(1, 1, num_mode, dim))
        self.m_pos = nn.Parameter(torch.Tensor(1, num_mode, dim))
        self.cat_x_proj = nn.Linear(2 * dim, dim)
        self.loc_head = MLPLayer(dim, 2 * dim, future_steps * 2)
        self.yaw_head = MLPLayer(dim, 2 * dim, future_steps * 2)
        self.vel_head = MLPLayer(dim, 2 * dim, future_steps * 2)
        self.pi_head = MLPLayer(dim, dim, 1)
        nn.init.normal_(self.m_emb, mean=0.0, std=0.01)
        nn.init.normal_(self.m_pos, mean=0.0, std=0.01)


class PlanningModel(TorchModuleWrapper):
    """Drop-in for rift.cbv.planning.pluto.model.pluto_model.PlanningModel (pluto_model.py:22-225).

    Only the default architecture of the CBV checkpoints is supported (dim 128, 4+4 layers,
    12 modes, state-attention ego encoder, cat_x, ref-free head): the HIP kernels are shaped for it.
    """

    def __init__(self, radius, dim=128, state_channel=6, polygon_channel=6, history_channel=9, history_steps=21,
                 future_steps=80, encoder_depth=4, decoder_depth=4, drop_path=0.2, dropout=0.1, num_heads=4,
                 num_modes=12, use_ego_history=False, state_attn_encoder=True, state_dropout=0.75,
                 use_hidden_proj=True, cat_x=True, ref_free_traj=True) -> None:
        super().__init__(feature_builders=[None], target_builders=[None], future_trajectory_sampling=trajectory_sampling)
        fixed = dict(dim=128, state_channel=6, polygon_channel=6, history_channel=9, history_steps=21, future_steps=80,
                     encoder_depth=4, decoder_depth=4, num_heads=4, num_modes=12, use_ego_history=False,
                     state_attn_encoder=True, use_hidden_proj=True, cat_x=True, ref_free_traj=True)
        given = dict(dim=dim, state_channel=state_channel, polygon_channel=polygon_channel,
                     history_channel=history_channel, history_steps=history_steps, future_steps=future_steps,
                     encoder_depth=encoder_depth, decoder_depth=decoder_depth, num_heads=num_heads, num_modes=num_modes,
                     use_ego_history=use_ego_history, state_attn_encoder=state_attn_encoder,
                     use_hidden_proj=use_hidden_proj, cat_x=cat_x, ref_free_traj=ref_free_traj)
        bad = {k: v for k, v in given.items() if fixed[k] != v}
        if bad:
            raise NotImplementedError(f"librift_hip.so is built for the default CBV architecture; unsupported: {bad}")
        if (drop_path, dropout, state_dropout) not in ((0.2, 0.1, 0.75), (0.0, 0.0, 0.0)):
            raise NotImplementedError("drop rates are compiled in: (0.2, 0.1, 0.75) or all zero")
        self.dim, self.history_steps, self.future_steps = dim, history_steps, future_steps
        self.use_hidden_proj, self.num_modes, self.radius, self.ref_free_traj = True, num_modes, radius, True
        self._no_drop = drop_path == 0.0

        self.pos_emb = FourierEmbedding(3, dim, 64)
        self.agent_encoder = AgentEncoder(state_channel, history_channel, dim)
        self.map_encoder = MapEncoder(polygon_channel, dim)
        self.static_objects_encoder = StaticObjectsEncoder(dim)
        self.encoder_blocks = nn.ModuleList(TransformerEncoderLayer(dim, num_heads) for _ in range(encoder_depth))
        self.norm = nn.LayerNorm(dim)
        self.agent_predictor = AgentPredictor(dim, future_steps)
        self.planning_decoder = PlanningDecoder(num_modes, decoder_depth, dim, num_heads, 4, dropout, future_steps)
        self.hidden_proj = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Linear(dim, dim))
        self.ref_free_decoder = MLPLayer(dim, 2 * dim, future_steps * 4)
        self.apply(self._init_weights)

        self._engine = None
        self._bound_version = None
        self._engine_users = weakref.WeakSet()     # live RLFTTrainers that hold self._engine (they register / unregister themselves)
        # "fp16" (fp16 MFMA operands, fp32 accumulate: the default -- the 16-bit mode that holds north_star's 1e-4 on the losses, at the
        # bf16 step time) | "bf16" (bf16 operands: what BASELINE.json names and bench.py's headline runs; RIFT loss within 1e-4 at the
        # benchmark batch only, tests/test_gpu_parity.py header) | "fp32" (exact fp32 MFMA, layer by layer: the reference's `precision: 32`)
        self.compute_precision = "fp16"
        self.need_traj = True                # trajectory heads are dead work for the RLFT losses; trainers switch it off
        self._seed = 0

    def _init_weights(self, m):   # pluto_model.py:108-120
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)

    # ---- engine binding -----------------------------------------------------------------------
    def _tensor_version(self):
        """What the engine's packed weight images depend on: storage and content version of every tensor it PACKS.  pi_head, the critic and
        the BatchNorm running statistics are read / written in place through their pointers at launch time (Engine.load_state_dict), so
        an in-place update of those (an optimizer step, RLFTPluto.train's refresh of the inference model) needs no re-bind: storage only."""
        every, packed = self._packed_tensors()
        return tuple([t.data_ptr() for t in every] + [t._version for t in packed])

    def _packed_tensors(self):
        """(every parameter / buffer, those of them the engine packs), cached: the module walk of named_parameters() / named_buffers() was
        1.8 ms per call, twice per rollout tick (tools/tick_latency.py).  The cache is re-validated on every call against the module tree
        itself -- each module still holds the same number of parameters / buffers / children, every child and every tensor is still the
        same object in its owner's dict -- so a replaced Parameter (load_state_dict(assign=True), setattr) or sub-module rebuilds it."""
        c = self.__dict__.get("_tensor_cache")
        if c is not None:
            dicts, links, owners, every, packed, lens = c
            if [len(d) for d in dicts] == lens and all([d.get(k) is ch for d, k, ch in links]) and all([d.get(k) is t for (d, k), t in zip(owners, every)]):
                return every, packed

        def live(n):
            return n.startswith("planning_decoder.pi_head.") or n.startswith("value_net.") or \
                n.endswith(".running_mean") or n.endswith(".running_var") or n.endswith(".num_batches_tracked")
        dicts, links, owners, every, packed = [], [], [], [], []
        for prefix, m in self.named_modules():
            dicts += [m._parameters, m._buffers, m._modules]
            links += [(m._modules, k, ch) for k, ch in m._modules.items()]
            for d in (m._parameters, m._buffers):
                for k, t in d.items():
                    if t is not None:
                        owners.append((d, k)); every.append(t)
                        if not live((prefix + "." if prefix else "") + k):
                            packed.append(t)
        self.__dict__["_tensor_cache"] = (dicts, links, owners, every, packed, [len(d) for d in dicts])
        return every, packed

    def engine(self):
        """Bind (or re-bind after load_state_dict / .to()) the parameter storage to the HIP context.
        The frozen trunk is re-packed only when one of its tensors changed; pi_head is read live."""
        from rift_amd import _ffi
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("PlanningModel runs on a HIP device only: call .to('cuda') first (no CPU fallback)")
        if self.compute_precision not in ("bf16", "fp16", "fp32"):
            raise ValueError(f"compute_precision must be 'bf16', 'fp16' or 'fp32', got {self.compute_precision!r}")
        operands = "fp16" if self.compute_precision == "fp16" else "bf16"
        if self._engine is None or self._engine.device != dev or (self._engine.operands != operands and self.compute_precision != "fp32"):
            if self._engine is not None:
                if len(self._engine_users):
                    raise RuntimeError("compute_precision / device changed while an RLFTTrainer holds this model's engine: close() the trainer "
                                       "first (its context, data-parallel hooks and parameter event would be left on a destroyed context)")
                self._engine.close()
            self._engine = _ffi.Engine(dev, operands=operands)
            self._bound_version = None
        ver = self._tensor_version()
        if ver != self._bound_version:
            sd = {k: v for k, v in self.state_dict(keep_vars=True).items() if not k.startswith("value_net.")}
            self._engine.load_state_dict({k: v.data for k, v in sd.items()})
            self._bound_version = self._tensor_version()
        return self._engine

    def release_engine(self):
        """Destroy the HIP context (its scratch arenas and side stream); the next engine() call builds a new one."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
            self._bound_version = None

    def forward(self, data: FeaturesType, engine=None) -> TargetsType:
        """`engine`: the result of a self.engine() call the caller made for THIS forward already (the rollout tick binds first, to stage
        its inputs through the engine's pinned arena): skips the second walk over the parameters."""
        eng = engine if engine is not None else self.engine()
        self._seed += 1
        out = eng.forward(data, train=self.training, need_traj=self.need_traj, fp32=self.compute_precision == "fp32",
                          no_drop=self._no_drop, seed=self._seed)
        return finish_outputs(out, data, self.history_steps, self.need_traj)


def finish_outputs(out: Dict[str, torch.Tensor], data, history_steps=21, need_traj=True) -> Dict[str, torch.Tensor]:
    """Derived outputs of pluto_model.py:182-223 (angles, best trajectory).  Rollout-side
    post-processing of SURVEY.md section 8(f) row 1 -- elementwise torch ops on the raw HIP outputs."""
    res = {"trajectory": out.get("trajectory"), "probability": out["probability"], "prediction": out.get("prediction"),
           "hidden": out["hidden"]}
    if not need_traj:
        return res
    rf = out["ref_free_trajectory"]
    res["ref_free_trajectory"] = rf
    res["output_ref_free_trajectory"] = torch.cat(
        [rf[..., :2], torch.arctan2(rf[..., 3], rf[..., 2]).unsqueeze(-1)], dim=-1)
    pred, dev = out["prediction"], out["probability"].device
    agent_pos = data["agent"]["position"][:, :, history_steps - 1].to(dev)
    agent_heading = data["agent"]["heading"][:, :, history_steps - 1].to(dev)
    A = agent_pos.shape[1]
    res["output_prediction"] = torch.cat(
        [pred[..., :2] + agent_pos[:, 1:A, None],
         torch.atan2(pred[..., 3], pred[..., 2]).unsqueeze(-1) + agent_heading[:, 1:A, None, None],
         pred[..., 4:6]], dim=-1)
    traj, prob = out["trajectory"], out["probability"]
    cand = torch.cat([traj[..., :2], torch.atan2(traj[..., 3], traj[..., 2]).unsqueeze(-1)], dim=-1)
    bs, R, M, T, _ = cand.shape
    best = prob.reshape(bs, R * M).argmax(-1)
    res["output_trajectory"] = cand.reshape(bs, R * M, T, -1)[torch.arange(bs, device=dev), best]
    res["candidate_trajectories"] = cand
    return res
