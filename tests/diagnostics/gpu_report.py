"""Run every HIP path against the CPU oracle / golden fixtures and print a table of max errors.
Diagnostic companion of tests/test_gpu_*.py (never asserts; writes gpurun_out/report.json)."""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from oracle import advantage as oadv, losses, pluto_ref  # noqa: E402
from rift_amd import _ffi  # noqa: E402
from tests import helpers as H  # noqa: E402

REPORT = {}


def rec(name, val):
    REPORT[name] = val
    print(f"{name:60s} {val}", flush=True)


def err(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    if a.numel() == 0:
        return 0.0
    d = (a.reshape(-1) - b.reshape(-1)).abs()
    return float(d.max()) if torch.isfinite(d).all() else float("nan")


def section(fn):
    try:
        t = time.time()
        fn()
        print(f"-- {fn.__name__} done in {time.time() - t:.1f}s", flush=True)
    except Exception:
        traceback.print_exc()
        REPORT[fn.__name__ + ".exception"] = traceback.format_exc()[-2000:]


def check_linear():
    eng = _ffi.Engine("cuda:0")
    g = torch.Generator().manual_seed(1)
    for (M, K, N) in [(70, 32, 96), (129, 128, 384), (64, 128, 512), (200, 512, 128), (33, 6, 128), (100, 129, 128),
                      (257, 256, 160), (64, 27, 32), (50, 96, 64), (1, 128, 128), (12, 128, 384), (300, 128, 1)]:
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ref = torch.nn.functional.linear(x, w, b)
        for fp32 in (True, False):
            y = eng.op_linear(x, w, b, fp32=fp32)
            rec(f"linear M{M} K{K} N{N} {'fp32' if fp32 else 'bf16'}", err(y, ref))
        lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
        ref2 = torch.nn.functional.gelu(torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (K,), lw, lb), w, b))
        y = eng.op_linear(x, w, b, ln_w=lw, ln_b=lb, act=2, fp32=True)
        rec(f"linear+LN+gelu M{M} K{K} N{N} fp32", err(y, ref2))
    eng.close()


def check_forward(case, fp32):
    tag = f"fwd[{case},{'fp32' if fp32 else 'bf16'}]"
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    eng = _ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    out = eng.forward(data, need_traj=True, fp32=fp32)
    torch.cuda.synchronize()
    ref, _, taps = pluto_ref.planning_model_forward(sd, H.clone_tree(data), want_taps=True)
    bs, A = data["agent"]["position"].shape[:2]
    va = data["agent"]["valid_mask"].any(-1)
    kpm = torch.cat([~va, ~data["map"]["valid_mask"].any(-1)], dim=-1)
    rv = data["reference_line"]["valid_mask"].any(-1)
    N = kpm.shape[1]
    R = rv.shape[1]
    # nat_out vs oracle x_agent is indirect; compare token taps
    xt = eng.tap("x_tokens").view(bs, N, 128).cpu()
    rec(tag + " x_tokens(valid)", err(xt[~kpm], taps["x_tokens"][~kpm]))
    xn = eng.tap("x_tokens_nopos").view(bs, N, 128).cpu()
    xa_ref = torch.cat([taps["x_agent"], taps["x_polygon"]], 1)
    rec(tag + " x_agent(valid)", err(xn[:, :A][va], taps["x_agent"][va]))
    rec(tag + " x_polygon", err(xn[:, A:], taps["x_polygon"]))
    eo = eng.tap("enc_out").view(bs, N, 128).cpu()
    rec(tag + " enc_out(valid)", err(eo[~kpm], taps["enc_out"][~kpm]))
    re = eng.tap("r_emb").view(bs, R, 128).cpu()
    rec(tag + " r_emb(valid)", err(re[rv], taps["r_emb"][rv]))
    d3 = eng.tap("dec3").view(bs, R, 12, 128).cpu()
    rec(tag + " dec3(valid)", err(d3[rv], taps["dec3"][rv]))
    qf = eng.tap("q_final").view(bs, R, 12, 128).cpu()
    rec(tag + " q_final(valid)", err(qf[rv], taps["q_final"][rv]))
    rec(tag + " probability vs oracle", err(out["probability"], ref["probability"]))
    rec(tag + " probability vs golden", err(out["probability"], gold["eval.probability"]))
    rec(tag + " hidden", err(out["hidden"], ref["hidden"]))
    rec(tag + " trajectory(valid)", err(out["trajectory"].cpu()[rv], ref["trajectory"][rv]))
    rec(tag + " prediction(valid)", err(out["prediction"].cpu()[va[:, 1:]], ref["prediction"][va[:, 1:]]))
    rec(tag + " ref_free", err(out["ref_free_trajectory"], ref["ref_free_trajectory"]))
    # losses (HIP forward activations -> HIP loss/backward) vs golden (reference end to end)
    r_pad = ~rv
    for kind in ("rift", "grpo", "reinforce", "ppo"):
        b = H.clone_tree(batch)
        if kind == "ppo":
            b["advantage_torch"] = torch.from_numpy(gold["ppo.advantage"])
        stats, flat, am = eng.loss_backward(kind, b)
        grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
        loss = eng.loss_finalize(stats, flat, grads)
        torch.cuda.synchronize()
        rec(tag + f" loss[{kind}] vs golden", abs(float(loss.item()) - float(gold[f"{kind}.loss"])))
        rec(tag + f" grad[{kind}] vs golden", max(err(grads[k], gold[f"{kind}.grad.{k}"]) for k in grads))
        # isolate the loss kernel: oracle on the HIP q_final
        ol, og, _ = losses.pi_head_loss_and_grads(sd, qf, kind, H.clone_tree(b), r_pad)
        rec(tag + f" loss[{kind}] vs oracle@hip_q", abs(float(loss.item()) - float(ol)))
        rec(tag + f" grad[{kind}] vs oracle@hip_q", max(err(grads[k], og[k]) for k in grads))
        if kind == "reinforce":
            rec(tag + " reinforce argmax exact", bool(np.array_equal(am.cpu().numpy()[:, 0], gold["reinforce.r_idx"])
                                                    and np.array_equal(am.cpu().numpy()[:, 1], gold["reinforce.m_idx"])))
    # train mode, no drop: BatchNorm batch statistics
    eng2 = _ffi.Engine("cuda:0")
    p2 = eng2.load_state_dict({k: v.clone() for k, v in sd.items()})
    out2 = eng2.forward(data, train=True, no_drop=True, fp32=fp32)
    torch.cuda.synchronize()
    rec(tag + " trainbn probability vs golden", err(out2["probability"], gold["trainbn.probability"]))
    rec(tag + " trainbn hidden vs golden", err(out2["hidden"], gold["trainbn.hidden"]))
    worst = 0.0
    for k in gold:
        if k.startswith("trainbn.stat."):
            name = k[len("trainbn.stat."):]
            worst = max(worst, err(p2[name].float(), gold[k].astype(np.float64)))
    rec(tag + " trainbn running stats", worst)
    # train mode with dropout: finite + deterministic in the seed
    o3 = eng2.forward(data, train=True, fp32=fp32, seed=5, bn_update=False)["probability"].clone()
    o4 = eng2.forward(data, train=True, fp32=fp32, seed=5, bn_update=False)["probability"].clone()
    o5 = eng2.forward(data, train=True, fp32=fp32, seed=6, bn_update=False)["probability"].clone()
    torch.cuda.synchronize()
    rec(tag + " dropout finite", bool(torch.isfinite(o3).all()))
    rec(tag + " dropout same seed diff", err(o3, o4))
    rec(tag + " dropout other seed diff", err(o3, o5))
    eng.close()
    eng2.close()


def check_advantage():
    eng = _ffi.Engine("cuda:0")
    i = H.advantage_inputs()
    gold = dict(np.load(os.path.join(H.GOLDEN, "advantage.npz")))
    a = eng.gae(i["rewards"], i["undones"], i["values"], i["next_values"], i["unterminated"])
    rec("gae vs golden", err(a, gold["gae"]))
    rec("gae normalized vs golden", err(eng.normalize_advantage_(a.clone()), gold["gae_normalized"]))
    r = eng.discounted_return(i["rewards"], i["dones"])
    rec("returns vs golden", err(r, gold["returns"]))
    T = lambda k: torch.from_numpy(i[k])
    ret = eng.rollout_return(T("delta_dis"), T("delta_angle"), T("speed"), T("acc"), T("ang_vel"), T("ang_acc"),
                             T("collision"), T("off_road"))
    oret = oadv.rollout_return(i["delta_dis"], i["delta_angle"], i["speed"], i["acc"], i["ang_vel"], i["ang_acc"],
                               i["collision"], i["off_road"])
    rec("rollout_return vs oracle", err(ret, oret))
    rec("rollout_return vs golden", err(ret, gold["rollout_return"]))
    z = eng.group_advantage(ret.view(1, -1))
    rec("group_advantage vs golden", err(z, gold["group_advantage"]))
    eng.close()


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(torch.cuda.get_device_name(0), flush=True)
    section(check_linear)
    for case in ("small", "full"):
        for fp32 in (True, False):
            section(lambda c=case, f=fp32: check_forward(c, f))
    section(check_advantage)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)
