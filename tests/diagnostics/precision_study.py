"""CPU emulation study (test-side diagnostic, uses the oracle): which MFMA operand format does each region of the trunk need so that
the RIFT loss / pi_head gradients stay within north_star's 1e-4 of the fp32 oracle?

Every contraction of oracle/pluto_ref.py (F.linear, F.conv1d, the attention score / PV products) has its two operands rounded to a
chosen format -- bf16, fp16, or a bf16 hi/lo pair ("bf16x2" = 16 mantissa bits, what three bf16 MFMAs hi*hi + hi*lo + lo*hi compute) --
with fp32 accumulation, per region of the model.  Everything else (LayerNorm, softmax, GELU, residual stream) stays fp32, as in the
HIP kernels.

    python tests/diagnostics/precision_study.py [n_scenes]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import torch.nn.functional as F

from oracle import losses, pluto_ref
from rift_amd import synthetic as syn
from tests import helpers as H


RANGE = None        # {region: max |operand|} when set to a dict: every contraction operand that passes rnd() is recorded (operand_range())


def rnd(x, fmt):
    if RANGE is not None and x.numel():
        RANGE[Ctl.region] = max(RANGE.get(Ctl.region, 0.0), float(x.detach().abs().max()))
    if "/" in fmt:
        fmt = fmt.split("/")[0]
    if fmt == "fp32":
        return x
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "fp16":
        return x.half().float()
    if fmt == "bf16x2":
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if fmt == "fp16x2":
        hi = x.half().float()
        return hi + (x - hi).half().float()
    raise ValueError(fmt)


class Ctl:
    region = "other"
    fmt = {}          # region -> format
    default = "fp32"

    @classmethod
    def f(cls):
        return cls.fmt.get(cls.region, cls.default)

    @classmethod
    def fw(cls):     # weight-side format: "act/weight" selects them separately
        f = cls.f()
        return f.split("/")[1] if "/" in f else f


class FProxy:
    """torch.nn.functional with operand rounding on linear / conv1d."""

    def __getattr__(self, k):
        return getattr(F, k)

    @staticmethod
    def linear(x, w, b=None):
        f = Ctl.f()
        return F.linear(rnd(x, f), rnd(w, Ctl.fw()), b)

    @staticmethod
    def conv1d(x, w, b=None, **kw):
        f = Ctl.f()
        return F.conv1d(rnd(x, f), rnd(w, Ctl.fw()), b, **kw)


def mha_r(query, key, value, sd, num_heads, key_padding_mask=None):
    import math
    f = Ctl.f()
    w, b = sd["in_proj_weight"], sd["in_proj_bias"]
    E = w.shape[1]
    fw = Ctl.fw()
    q = F.linear(rnd(query, f), rnd(w[:E], fw), b[:E])
    k = F.linear(rnd(key, f), rnd(w[E:2 * E], fw), b[E:2 * E])
    v = F.linear(rnd(value, f), rnd(w[2 * E:], fw), b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    hd = E // num_heads
    q = q.view(B, Lq, num_heads, hd).transpose(1, 2)
    k = k.view(B, Lk, num_heads, hd).transpose(1, 2)
    v = v.view(B, Lk, num_heads, hd).transpose(1, 2)
    s = rnd(q * (1.0 / math.sqrt(hd)), f) @ rnd(k, f).transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (rnd(p, f) @ rnd(v, f)).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(rnd(o, f), rnd(sd["out_proj.weight"], fw), sd["out_proj.bias"])


def na1d_r(x, sd, num_heads, kernel_size):
    f = Ctl.f()
    B, L, C = x.shape
    hd = C // num_heads
    k = kernel_size
    qkv = pluto_ref.linear(x, sd, "qkv").reshape(B, L, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, kk, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    rpb = sd["rpb"]
    starts = torch.tensor([pluto_ref.na1d_window_start(i, L, k) for i in range(L)])
    nbr = starts[:, None] + torch.arange(k)[None, :]
    rel = nbr - torch.arange(L)[:, None] + (k - 1)
    kn = kk[:, :, nbr]
    vn = v[:, :, nbr]
    s = torch.einsum("bhld,bhlkd->bhlk", rnd(q, f), rnd(kn, f)) + rpb[:, rel][None]
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhlk,bhlkd->bhld", p, rnd(vn, f))
    o = o.permute(0, 2, 1, 3).reshape(B, L, C)
    return pluto_ref.linear(o, sd, "proj")


def region(name, fn):
    def wrapped(*a, **k):
        prev = Ctl.region
        Ctl.region = name
        try:
            return fn(*a, **k)
        finally:
            Ctl.region = prev
    return wrapped


_orig = {}


def install():
    _orig["mha"], _orig["na1d"], _orig["planning_decoder"] = pluto_ref.mha, pluto_ref.neighborhood_attention_1d, pluto_ref.planning_decoder
    pluto_ref.F = FProxy()
    pluto_ref.mha = mha_r
    pluto_ref.neighborhood_attention_1d = na1d_r
    for name, reg in (("nat_sequence_encoder", "nat"), ("state_attention_encoder", "ego"), ("points_encoder", "pe"),
                      ("fourier_embedding", "fourier"), ("encoder_block", "enc"), ("agent_predictor", "heads")):
        _orig[name] = getattr(pluto_ref, name)
        setattr(pluto_ref, name, region(reg, _orig[name]))
    dl = pluto_ref.decoder_layer
    _orig["decoder_layer"] = dl
    cnt = {"i": 0}

    def dec(*a, **k):
        i = cnt["i"] % 4
        cnt["i"] += 1
        return region(f"dec{i}", dl)(*a, **k)
    pluto_ref.decoder_layer = dec
    pd = pluto_ref.planning_decoder
    pluto_ref.planning_decoder = region("dec_misc", pd)     # q_proj, cat_x_proj, pi_head first linear (pi_head is fp32 in the kernel)


def run(sd, batch, fmt, default="fp32"):
    Ctl.fmt, Ctl.default = dict(fmt), default
    data = batch["cur_pluto_feature_torch"]
    out, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), train_bn=True, need_traj=False, want_taps=True)
    return out["probability"], taps["q_final"]


def operand_range(sd, batch, train_bn=False):
    """max |operand| of every contraction of the oracle forward (both operands of every F.linear / F.conv1d, the attention q / k / v and
    probabilities), per region of the model -- what the 16-bit kernels convert to their MFMA operand format.  Installs the hooks on first use."""
    global RANGE
    if "decoder_layer" not in _orig:
        install()
    RANGE = {}
    try:
        out = run(sd, batch, {}, "fp32") if train_bn else None
        if out is None:
            Ctl.fmt, Ctl.default = {}, "fp32"
            out_d, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(batch["cur_pluto_feature_torch"]), need_traj=True, want_taps=True)
            out = (out_d["probability"], taps["q_final"])
        return dict(RANGE), out
    finally:
        RANGE = None


def uninstall():
    """Put the oracle's own functions back (tests that share a process with other oracle users)."""
    if "decoder_layer" not in _orig:
        return
    import torch.nn.functional as F_
    pluto_ref.F = F_
    pluto_ref.mha, pluto_ref.neighborhood_attention_1d = _orig.pop("mha"), _orig.pop("na1d")
    pluto_ref.planning_decoder = _orig.pop("planning_decoder")
    for name, fn in list(_orig.items()):
        setattr(pluto_ref, name, fn)
    _orig.clear()


def loss_grads(sd, qf, batch, r_pad):
    Ctl.fmt, Ctl.default = {}, "fp32"
    l, g, _ = losses.pi_head_loss_and_grads(sd, qf, "rift", H.clone_tree(batch), r_pad)
    return float(l), g


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(8)
    sd = H.weights()
    install()
    REG = ["nat", "ego", "pe", "fourier", "enc", "dec0", "dec1", "dec2", "dec3", "dec_misc"]
    cases = {}
    if n == 0:
        for c in ("small", "full"):
            cases[c] = H.build_batch(c)
    else:
        cases[f"n{n}"] = syn.collate_scenes([syn.make_scene(1000 + i) for i in range(n)])
    configs = [("all bf16", {}, "bf16"), ("all fp16", {}, "fp16"), ("all bf16x2", {}, "bf16x2"),
               ("act bf16 / w bf16x2", {}, "bf16/bf16x2"), ("act fp16 / w fp16x2", {}, "fp16/fp16x2"), ("act bf16x2 / w bf16", {}, "bf16x2/bf16")]
    if os.environ.get("BRIEF"):
        REG_LOOP = []
    else:
        REG_LOOP = REG
    for r in REG_LOOP:
        configs.append((f"bf16 except {r}=fp32", {r: "fp32"}, "bf16"))
    if not os.environ.get("BRIEF"):
      configs += [
        ("bf16; dec*+misc fp32", {k: "fp32" for k in REG if k.startswith("dec")}, "bf16"),
        ("bf16; enc+dec*+misc fp32", {k: "fp32" for k in REG if k.startswith("dec") or k == "enc"}, "bf16"),
        ("bf16; nat fp32, pe fp32", {"nat": "fp32", "pe": "fp32"}, "bf16"),
        ("fp16; dec*+misc bf16x2", {k: "bf16x2" for k in REG if k.startswith("dec")}, "fp16"),
        ("fp16; enc+dec* bf16x2", {k: "bf16x2" for k in REG if k.startswith("dec") or k == "enc"}, "fp16"),
    ]
    if os.environ.get("FP16"):      # which regions carry the fp16 error: one region at a time exact, then cumulative sets
        configs = [("all fp16", {}, "fp16")] + [(f"fp16 except {r}=fp32", {r: "fp32"}, "fp16") for r in REG]
        configs += [("fp16; dec_misc+dec3 fp32", {"dec_misc": "fp32", "dec3": "fp32"}, "fp16"),
                    ("fp16; dec* + misc fp32", {k: "fp32" for k in REG if k.startswith("dec")}, "fp16"),
                    ("fp16; enc + dec* + misc fp32", {k: "fp32" for k in REG if k.startswith("dec") or k == "enc"}, "fp16"),
                    ("fp16; nat + pe fp32", {"nat": "fp32", "pe": "fp32"}, "fp16"),
                    ("fp16 act / fp32 weights", {}, "fp16/fp32"), ("fp32 act / fp16 weights", {}, "fp32/fp16")]
    for cname, batch in cases.items():
        data = batch["cur_pluto_feature_torch"]
        r_pad = ~data["reference_line"]["valid_mask"].any(-1)
        p0, q0 = run(sd, batch, {}, "fp32")
        l0, g0 = loss_grads(sd, q0, batch, r_pad)
        print(f"== {cname}: oracle loss {l0:.7f}")
        for label, fmt, default in configs:
            p, q = run(sd, batch, fmt, default)
            l, g = loss_grads(sd, q, batch, r_pad)
            gmax = max(float(g0[k].abs().max()) for k in g0)
            gerr = max(float((g[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-2 * gmax) for k in g0)
            print(f"  {label:34s} loss err {abs(l - l0):.2e}  max logit err {float((p - p0)[~r_pad].abs().max()):.2e}  "
                  f"qf err {float((q - q0)[~r_pad].abs().max()):.2e}  grad rel err {gerr:.2e}", flush=True)


if __name__ == "__main__":
    main()
