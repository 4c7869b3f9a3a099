"""CPU oracle (test infrastructure, never on the product path): functional
PyTorch-CPU fp32 restatement of the reference Pluto ``PlanningModel.forward``.

It operates directly on a reference-layout ``state_dict`` (Appendix B of
SURVEY.md) and a collated feature dict (Appendix A), so it needs none of the
reference's module classes.  Citations are relative to the upstream checkout:
``P/`` = ``rift/cbv/planning/pluto/model/``.

``neighborhood_attention_1d`` restates the published algorithm of the
third-party ``natten==0.14.6`` op (requirements.txt:15) -- *parity unpinned* at
that boundary (see oracle/__init__.py).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
LN_EPS = 1e-5


class SD:
    """Prefix view onto a flat state_dict."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str = ""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k: str) -> Tensor:
        return self.sd[self.prefix + k]

    def __contains__(self, k: str) -> bool:
        return (self.prefix + k) in self.sd

    def sub(self, p: str) -> "SD":
        return SD(self.sd, self.prefix + p + ".")


def linear(x, sd: SD, name: str):
    b = sd[name + ".bias"] if (name + ".bias") in sd else None
    return F.linear(x, sd[name + ".weight"], b)


def layer_norm(x, sd: SD, name: str):
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[name + ".bias"], LN_EPS)


def mlp_layer(x, sd: SD):
    """P/layers/mlp_layer.py:8-16: Linear -> LayerNorm -> ReLU -> Linear."""
    h = linear(x, sd, "mlp.0")
    h = F.relu(layer_norm(h, sd, "mlp.1"))
    return linear(h, sd, "mlp.3")


def mha(query, key, value, sd: SD, num_heads: int, key_padding_mask: Optional[Tensor] = None):
    """torch.nn.MultiheadAttention(batch_first=True) forward, no dropout.
    key_padding_mask: (B, Lk) bool, True = ignore."""
    w, b = sd["in_proj_weight"], sd["in_proj_bias"]
    E = w.shape[1]
    q = F.linear(query, w[:E], b[:E])
    k = F.linear(key, w[E:2 * E], b[E:2 * E])
    v = F.linear(value, w[2 * E:], b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    hd = E // num_heads
    q = q.view(B, Lq, num_heads, hd).transpose(1, 2)
    k = k.view(B, Lk, num_heads, hd).transpose(1, 2)
    v = v.view(B, Lk, num_heads, hd).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, sd["out_proj.weight"], sd["out_proj.bias"])


# --------------------------------------------------------------------------
# natten 0.14.6 NeighborhoodAttention1D (third-party; parity unpinned)
# call site: P/layers/embedding.py:169-178,192
# --------------------------------------------------------------------------
def na1d_window_start(i: int, L: int, k: int) -> int:
    return min(max(i - k // 2, 0), L - k)


def neighborhood_attention_1d(x, sd: SD, num_heads: int, kernel_size: int):
    """x: (B, L, C). qkv Linear -> q*head_dim^-0.5 -> scores over the clamped
    window of `kernel_size` neighbours + rpb[h, (j-i)+k-1] -> softmax -> AV -> proj."""
    B, L, C = x.shape
    hd = C // num_heads
    k = kernel_size
    assert L >= k
    qkv = linear(x, sd, "qkv").reshape(B, L, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, kk, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]  # (B, H, L, hd)
    rpb = sd["rpb"]  # (H, 2k-1)
    starts = torch.tensor([na1d_window_start(i, L, k) for i in range(L)])
    nbr = starts[:, None] + torch.arange(k)[None, :]  # (L, k) absolute neighbour index
    rel = nbr - torch.arange(L)[:, None] + (k - 1)  # (L, k) rpb index
    kn = kk[:, :, nbr]  # (B, H, L, k, hd)
    vn = v[:, :, nbr]
    s = torch.einsum("bhld,bhlkd->bhlk", q, kn) + rpb[:, rel][None]
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhlk,bhlkd->bhld", p, vn)
    o = o.permute(0, 2, 1, 3).reshape(B, L, C)
    return linear(o, sd, "proj")


# --------------------------------------------------------------------------
# NAT-FPN history encoder: P/layers/embedding.py:8-251
# --------------------------------------------------------------------------
NAT_HEADS = (2, 4, 8)
NAT_KERNEL = (3, 3, 5)


def nat_layer(x, sd: SD, heads: int, ksz: int):
    """P/layers/embedding.py:196-202 (drop_path = identity)."""
    x = x + neighborhood_attention_1d(layer_norm(x, sd, "norm1"), sd.sub("attn"), heads, ksz)
    h = linear(layer_norm(x, sd, "norm2"), sd, "mlp.fc1")
    h = linear(F.gelu(h), sd, "mlp.fc2")
    return x + h


def nat_sequence_encoder(x, sd: SD, taps=None):
    """x: (B, 9, 20) -> (B, 128).  P/layers/embedding.py:62-87."""
    x = F.conv1d(x, sd["embed.proj.weight"], sd["embed.proj.bias"], padding=1).permute(0, 2, 1)
    outs = []
    for lv in range(3):
        lsd = sd.sub(f"levels.{lv}")
        for blk in range(2):
            x = nat_layer(x, lsd.sub(f"blocks.{blk}"), NAT_HEADS[lv], NAT_KERNEL[lv])
        xo = x
        if lv < 2:  # ConvDownsampler, embedding.py:100-117
            x = F.conv1d(x.permute(0, 2, 1), lsd["downsample.reduction.weight"], None,
                         stride=2, padding=1).permute(0, 2, 1)
            x = layer_norm(x, lsd, "downsample.norm")
        outs.append(layer_norm(xo, sd, f"norm{lv}").permute(0, 2, 1).contiguous())
        if taps is not None:
            taps[f"nat_level{lv}"] = xo
    lat = [F.conv1d(outs[i], sd[f"lateral_convs.{i}.weight"], sd[f"lateral_convs.{i}.bias"], padding=1)
           for i in range(3)]
    for i in (2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(
            lat[i], scale_factor=lat[i - 1].shape[-1] / lat[i].shape[-1], mode="linear", align_corners=False)
    out = F.conv1d(lat[0], sd["fpn_conv.weight"], sd["fpn_conv.bias"], padding=1)
    return out[:, :, -1]


def state_attention_encoder(x, sd: SD, drop_mask: Optional[Tensor] = None):
    """P/modules/agent_encoder.py:99-140.  x: (bs, 6). drop_mask: (bs, 6) bool or None."""
    emb = torch.stack([F.linear(x[:, i, None], sd[f"linears.{i}.weight"], sd[f"linears.{i}.bias"])
                       for i in range(x.shape[1])], dim=1)
    emb = emb + sd["pos_embed"]
    query = sd["query"].repeat(emb.shape[0], 1, 1)
    return mha(query, emb, emb, sd.sub("attn"), 4, drop_mask)[:, 0]


def to_vector(feat, valid_mask):
    """P/modules/agent_encoder.py:41-52."""
    vm = valid_mask[..., :-1] & valid_mask[..., 1:]
    while vm.dim() < feat.dim():
        vm = vm.unsqueeze(-1)
    return torch.where(vm, feat[:, :, 1:] - feat[:, :, :-1], torch.zeros_like(feat[:, :, 1:]))


def agent_features(data, T=21):
    """9-channel diff features, P/modules/agent_encoder.py:54-75. -> (bs, A, 20, 9)."""
    ag = data["agent"]
    position, heading = ag["position"][:, :, :T], ag["heading"][:, :, :T]
    velocity, shape = ag["velocity"][:, :, :T], ag["shape"][:, :, :T]
    valid = ag["valid_mask"][:, :, :T]
    hv = to_vector(heading, valid)
    vmv = valid[..., 1:] & valid[..., :-1]
    return torch.cat([to_vector(position, valid), to_vector(velocity, valid),
                      torch.stack([hv.cos(), hv.sin()], dim=-1), shape[:, :, 1:],
                      vmv.float().unsqueeze(-1)], dim=-1)


def agent_encoder(data, sd: SD, taps=None, state_drop_mask=None):
    """P/modules/agent_encoder.py:54-96."""
    feat = agent_features(data)
    bs, A, T, C = feat.shape
    valid_agent = data["agent"]["valid_mask"][:, :, :21].any(-1).flatten()
    feat = feat.view(bs * A, T, C)
    x_tmp = nat_sequence_encoder(feat[valid_agent].permute(0, 2, 1).contiguous(),
                                 sd.sub("history_encoder"), taps)
    x_agent = torch.zeros(bs * A, 128)
    x_agent[valid_agent] = x_tmp
    x_agent = x_agent.view(bs, A, 128)
    x_ego = state_attention_encoder(data["current_state"][:, :6], sd.sub("ego_state_emb"), state_drop_mask)
    x_agent[:, 0] = x_ego
    return x_agent + sd["type_emb.weight"][data["agent"]["category"].long()]


# --------------------------------------------------------------------------
# PointsEncoder / Fourier / map encoder
# --------------------------------------------------------------------------
DP = None   # data-parallel restatement (tests only): {"bn_sync": f(sum, sumsq, n) -> reduced triple, "quirk_kpm": (global_bs, R) bool, "offset": int}


def batch_norm(x, sd: SD, name: str, train: bool, new_stats: Optional[dict] = None):
    """nn.BatchNorm1d on (rows, C). train=True: batch statistics (biased var) and
    running-stat update with unbiased var, momentum 0.1 (embedding.py:260,266).
    With the module-level DP hook set, the statistics are those of the GLOBAL minibatch: per-channel (sum, sum of squares, count) go
    through DP["bn_sync"] (an all-reduce) first -- SURVEY.md 8(e); the reference itself is single-device."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if train:
        n = x.shape[0]
        if DP is not None:
            s, q, n = DP["bn_sync"](x.double().sum(0), (x.double() ** 2).sum(0), float(n))
            mean = (s / n).float()
            var = (q / n - (s / n) ** 2).clamp_min(0).float()
        else:
            mean = x.mean(0)
            var = x.var(0, unbiased=False)
        if new_stats is not None:
            key = sd.prefix + name
            new_stats[key + ".running_mean"] = 0.9 * sd[name + ".running_mean"] + 0.1 * mean
            new_stats[key + ".running_var"] = 0.9 * sd[name + ".running_var"] + 0.1 * var * n / max(n - 1, 1)
            new_stats[key + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[name + ".running_mean"], sd[name + ".running_var"]
    return (x - mean) / torch.sqrt(var + BN_EPS) * w + b


def points_encoder(x, mask, sd: SD, train_bn=False, new_stats=None):
    """P/layers/embedding.py:271-296.  x: (B, n, C), mask: (B, n) -> (B, 128).
    Invalid points contribute all-zero rows to both max-pools."""
    B, n, _ = x.shape
    h = linear(x[mask], sd, "first_mlp.0")
    h = F.relu(batch_norm(h, sd, "first_mlp.1", train_bn, new_stats))
    h = linear(h, sd, "first_mlp.3")
    feat = torch.zeros(B, n, 256)
    feat[mask] = h
    pooled = feat.max(dim=1)[0]
    feat = torch.cat([feat, pooled.unsqueeze(1).repeat(1, n, 1)], dim=-1)
    h = linear(feat[mask], sd, "second_mlp.0")
    h = F.relu(batch_norm(h, sd, "second_mlp.1", train_bn, new_stats))
    h = linear(h, sd, "second_mlp.3")
    res = torch.zeros(B, n, h.shape[-1])
    res[mask] = h
    return res.max(dim=1)[0]


def fourier_embedding(x, sd: SD):
    """P/layers/fourier_embedding.py:45-55.  x: (..., D_in) -> (..., 128)."""
    d_in = x.shape[-1]
    f = x.unsqueeze(-1) * sd["freqs.weight"] * 2 * math.pi
    f = torch.cat([f.cos(), f.sin(), x.unsqueeze(-1)], dim=-1)
    acc = None
    for i in range(d_in):
        m = sd.sub(f"mlps.{i}")
        h = F.relu(layer_norm(linear(f[..., i, :], m, "0"), m, "1"))
        h = linear(h, m, "3")
        acc = h if acc is None else acc + h
    h = F.relu(layer_norm(acc, sd, "to_out.0"))
    return linear(h, sd, "to_out.2")


def map_features(data):
    """10-channel point features, P/modules/map_encoder.py:43-59. -> (bs, M, 20, 10)."""
    mp = data["map"]
    pp, pv, po, c = mp["point_position"], mp["point_vector"], mp["point_orientation"], mp["polygon_center"]
    return torch.cat([pp[:, :, 0] - c[..., None, :2], pv[:, :, 0],
                      torch.stack([po[:, :, 0].cos(), po[:, :, 0].sin()], dim=-1),
                      pp[:, :, 1] - pp[:, :, 0], pp[:, :, 2] - pp[:, :, 0]], dim=-1)


def map_encoder(data, sd: SD, train_bn=False, new_stats=None):
    """P/modules/map_encoder.py:31-93."""
    mp = data["map"]
    feat = map_features(data)
    bs, M, P, C = feat.shape
    x = points_encoder(feat.reshape(bs * M, P, C), mp["valid_mask"].view(bs * M, P),
                       sd.sub("polygon_encoder"), train_bn, new_stats).view(bs, M, -1)
    has = mp["polygon_has_speed_limit"]
    x_speed = torch.zeros(bs, M, 128)
    if has.any():
        x_speed[has] = fourier_embedding(mp["polygon_speed_limit"][has].unsqueeze(-1), sd.sub("speed_limit_emb"))
    x_speed[~has] = sd["unknown_speed_emb.weight"]
    return (x + sd["type_emb.weight"][mp["polygon_type"].long()]
            + sd["on_route_emb.weight"][mp["polygon_on_route"].long()]
            + sd["traffic_light_emb.weight"][mp["polygon_tl_status"].long()] + x_speed)


def static_objects_encoder(data, sd: SD):
    """P/modules/static_objects_encoder.py:17-31."""
    so = data["static_objects"]
    emb = fourier_embedding(so["shape"], sd.sub("obj_encoder")) + sd["type_emb.weight"][so["category"].long()]
    out = torch.zeros_like(emb)
    vm = so["valid_mask"]
    out[vm] = emb[vm]
    heading = (so["heading"] + math.pi) % (2 * math.pi) - math.pi
    return out, torch.cat([so["position"], heading.unsqueeze(-1)], dim=-1), ~vm


# --------------------------------------------------------------------------
# encoder block / decoder
# --------------------------------------------------------------------------
def encoder_block(x, sd: SD, key_padding_mask):
    """P/layers/transformer.py:73-94 (drop_path = identity)."""
    h = layer_norm(x, sd, "norm1")
    x = x + mha(h, h, h, sd.sub("attn"), 4, key_padding_mask)
    h = linear(layer_norm(x, sd, "norm2"), sd, "mlp.fc1")
    return x + linear(F.gelu(h), sd, "mlp.fc2")


def decoder_layer(tgt, memory, sd: SD, tgt_kpm, mem_kpm, m_pos):
    """P/modules/planning_decoder.py:42-86 (dropout = identity), incl. the r2r
    mask quirk (`tgt_key_padding_mask.repeat(M, 1)`, :56-60)."""
    bs, R, M, D = tgt.shape
    tgt = tgt.transpose(1, 2).reshape(bs * M, R, D)
    h = layer_norm(tgt, sd, "norm1")
    quirk = tgt_kpm.repeat(M, 1)          # row b*M + m gets the padding row of scene (b*M + m) % bs
    if DP is not None:                    # sharded minibatch: of the GLOBAL minibatch (rows of this shard start at offset * M)
        g = DP["quirk_kpm"]
        quirk = g[(DP["offset"] * M + torch.arange(bs * M)) % g.shape[0]]
    tgt = tgt + mha(h, h, h, sd.sub("r2r_attn"), 4, quirk)
    tmp = tgt.reshape(bs, M, R, D).transpose(1, 2).reshape(bs * R, M, D)
    valid = ~tgt_kpm.reshape(-1)
    tv = tmp[valid]
    h = layer_norm(tv, sd, "norm2")
    tv = tv + mha(h + m_pos, h + m_pos, h, sd.sub("m2m_attn"), 4)
    tgt = torch.zeros_like(tmp)
    tgt[valid] = tv
    tgt = tgt.reshape(bs, R * M, D)
    h = layer_norm(tgt, sd, "norm3")
    tgt = tgt + mha(h, memory, memory, sd.sub("cross_attn"), 4, mem_kpm)
    h = layer_norm(tgt, sd, "norm4")
    h = linear(F.relu(linear(h, sd, "ffn.0")), sd, "ffn.3")
    return (tgt + h).reshape(bs, R, M, D)


def ref_line_features(data):
    """P/modules/planning_decoder.py:139-153 -> (bs, R, 120, 6)."""
    rl = data["reference_line"]
    rp, rv, ro = rl["position"], rl["vector"], rl["orientation"]
    return torch.cat([rp - rp[..., 0:1, :2], rv, torch.stack([ro.cos(), ro.sin()], dim=-1)], dim=-1)


def planning_decoder(data, enc_emb, enc_kpm, sd: SD, train_bn=False, new_stats=None, taps=None, need_traj=True):
    """P/modules/planning_decoder.py:135-188."""
    rl = data["reference_line"]
    r_valid = rl["valid_mask"]
    r_kpm = ~r_valid.any(-1)
    feat = ref_line_features(data)
    bs, R, P, C = feat.shape
    r_emb = points_encoder(feat.reshape(bs * R, P, C), r_valid.view(bs * R, P), sd.sub("r_encoder"),
                           train_bn, new_stats).view(bs, R, -1)
    r_pos = torch.cat([rl["position"][:, :, 0], rl["orientation"][:, :, 0, None]], dim=-1)
    r_emb = r_emb + fourier_embedding(r_pos, sd.sub("r_pos_emb"))
    if taps is not None:
        taps["r_emb"] = r_emb
    M = sd["m_emb"].shape[2]
    r_emb = r_emb.unsqueeze(2).repeat(1, 1, M, 1)
    m_emb = sd["m_emb"].repeat(bs, R, 1, 1)
    q = linear(torch.cat([r_emb, m_emb], dim=-1), sd, "q_proj")
    if taps is not None:
        taps["q0"] = q
    for i in range(4):
        q = decoder_layer(q, enc_emb, sd.sub(f"decoder_blocks.{i}"), r_kpm, enc_kpm, sd["m_pos"])
        assert torch.isfinite(q).all()
        if taps is not None:
            taps[f"dec{i}"] = q
    x0 = enc_emb[:, 0].unsqueeze(1).unsqueeze(2).repeat(1, R, M, 1)
    q = linear(torch.cat([q, x0], dim=-1), sd, "cat_x_proj")
    if taps is not None:
        taps["q_final"] = q
    pi = mlp_layer(q, sd.sub("pi_head")).squeeze(-1)
    traj = None
    if need_traj:
        loc = mlp_layer(q, sd.sub("loc_head")).view(bs, R, M, 80, 2)
        yaw = mlp_layer(q, sd.sub("yaw_head")).view(bs, R, M, 80, 2)
        vel = mlp_layer(q, sd.sub("vel_head")).view(bs, R, M, 80, 2)
        traj = torch.cat([loc, yaw, vel], dim=-1)
    return traj, pi


def agent_predictor(x, sd: SD):
    """P/modules/agent_predictor.py:17-29."""
    bs, N, _ = x.shape
    loc = mlp_layer(x, sd.sub("loc_predictor")).view(bs, N, 80, 2)
    yaw = mlp_layer(x, sd.sub("yaw_predictor")).view(bs, N, 80, 2)
    vel = mlp_layer(x, sd.sub("vel_predictor")).view(bs, N, 80, 2)
    return torch.cat([loc, yaw, vel], dim=-1)


@torch.no_grad()
def planning_model_forward(sd_flat: Dict[str, Tensor], data, train_bn: bool = False,
                           need_traj: bool = True, state_drop_mask=None, want_taps: bool = False):
    """P/pluto_model.py:122-225, eval mode (train_bn=False) or train mode with every
    drop probability 0 but BatchNorm batch statistics (train_bn=True).

    Returns (out, new_bn_stats, taps)."""
    sd = SD(sd_flat)
    taps = {} if want_taps else None
    new_stats = {} if train_bn else None
    ag, mp = data["agent"], data["map"]
    agent_pos = ag["position"][:, :, 20]
    agent_heading = ag["heading"][:, :, 20]
    agent_mask = ag["valid_mask"][:, :, :21]
    center = mp["polygon_center"]
    bs, A = agent_pos.shape[:2]

    position = torch.cat([agent_pos, center[..., :2]], dim=1)
    angle = torch.cat([agent_heading, center[..., 2]], dim=1)
    angle = (angle + math.pi) % (2 * math.pi) - math.pi
    pos = torch.cat([position, angle.unsqueeze(-1)], dim=-1)
    kpm = torch.cat([~agent_mask.any(-1), ~mp["valid_mask"].any(-1)], dim=-1)

    x_agent = agent_encoder(data, sd.sub("agent_encoder"), taps, state_drop_mask)
    x_polygon = map_encoder(data, sd.sub("map_encoder"), train_bn, new_stats)
    x_static, static_pos, static_kpm = static_objects_encoder(data, sd.sub("static_objects_encoder"))
    x = torch.cat([x_agent, x_polygon, x_static], dim=1)
    pos = torch.cat([pos, static_pos], dim=1)
    kpm = torch.cat([kpm, static_kpm], dim=-1)
    if taps is not None:
        taps["x_agent"], taps["x_polygon"] = x_agent, x_polygon
    x = x + fourier_embedding(pos, sd.sub("pos_emb"))
    if taps is not None:
        taps["x_tokens"] = x
    for i in range(4):
        x = encoder_block(x, sd.sub(f"encoder_blocks.{i}"), kpm)
        if taps is not None:
            taps[f"enc{i}"] = x
    x = layer_norm(x, sd, "norm")
    if taps is not None:
        taps["enc_out"] = x

    out = {}
    if need_traj:
        out["prediction"] = agent_predictor(x[:, 1:A], sd.sub("agent_predictor"))
    traj, prob = planning_decoder(data, x, kpm, sd.sub("planning_decoder"), train_bn, new_stats, taps, need_traj)
    out["hidden"] = linear(F.relu(linear(x[:, 0], sd, "hidden_proj.0")), sd, "hidden_proj.2")
    if need_traj:
        rf = mlp_layer(x[:, 0], sd.sub("ref_free_decoder")).reshape(bs, 80, 4)
        out["ref_free_trajectory"] = rf
        out["output_ref_free_trajectory"] = torch.cat(
            [rf[..., :2], torch.arctan2(rf[..., 3], rf[..., 2]).unsqueeze(-1)], dim=-1)
        p = out["prediction"]
        out["output_prediction"] = torch.cat(
            [p[..., :2] + agent_pos[:, 1:A, None],
             torch.atan2(p[..., 3], p[..., 2]).unsqueeze(-1) + agent_heading[:, 1:A, None, None],
             p[..., 4:6]], dim=-1)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    prob = prob.masked_fill(r_pad.unsqueeze(-1), -1e6)
    out["probability"] = prob
    if need_traj:
        out["trajectory"] = traj
        ang = torch.atan2(traj[..., 3], traj[..., 2])
        cand = torch.cat([traj[..., :2], ang.unsqueeze(-1)], dim=-1)
        R, M, T = cand.shape[1:4]
        best = prob.reshape(bs, R * M).argmax(-1)
        out["output_trajectory"] = cand.reshape(bs, R * M, T, -1)[torch.arange(bs), best]
        out["candidate_trajectories"] = cand
    return out, new_stats, taps
