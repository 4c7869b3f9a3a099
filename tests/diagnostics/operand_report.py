"""GPU diagnostic: what each compute precision (bf16 / fp16 MFMA operands, exact fp32) costs against the fp32 CPU oracle on the fixtures the
parity tests use -- logits, the four objectives on the 6- and 2-scene fixtures, RIFT loss + pi_head gradient on the 256-scene benchmark batch.
    python tests/diagnostics/operand_report.py > gpurun_out/operand_report.txt"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import losses, pluto_ref  # noqa: E402
from rift_amd import _ffi, synthetic as syn  # noqa: E402
from tests import helpers as H  # noqa: E402

torch.set_num_threads(min(32, os.cpu_count() or 1))


def engine(mode, sd):
    eng = _ffi.Engine("cuda:0", operands="fp16" if mode == "fp16" else "bf16")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    return eng


def grads_of(eng, sd, kind, b):
    stats, flat, _ = eng.loss_backward(kind, b)
    grads = {k: torch.zeros_like(sd["planning_decoder.pi_head." + k]).cuda() for k in losses.PI_KEYS}
    loss = float(eng.loss_finalize(stats, flat, grads).item())
    return loss, {k: v.cpu() for k, v in grads.items()}


def gnorm(g, go):
    num = sum(float(((g[k] - go[k]).double() ** 2).sum()) for k in go) ** 0.5
    den = sum(float((go[k].double() ** 2).sum()) for k in go) ** 0.5
    return num / den


for case in ("small", "full"):
    gold, batch, sd = H.load_case(case)
    data = batch["cur_pluto_feature_torch"]
    for mode in ("bf16", "fp16", "fp32"):
        eng = engine(mode, sd)
        out = eng.forward(data, need_traj=True, fp32=mode == "fp32")
        torch.cuda.synchronize()
        e = {k: float((out[k].cpu() - torch.from_numpy(gold["eval." + k])).abs().max()) for k in ("probability", "hidden", "ref_free_trajectory")}
        line = f"{case:5s} {mode}: logits {e['probability']:.2e} hidden {e['hidden']:.2e} ref_free {e['ref_free_trajectory']:.2e} |"
        for kind in ("rift", "grpo", "reinforce", "ppo"):
            b = H.clone_tree(batch)
            if kind == "ppo":
                b["advantage_torch"] = torch.from_numpy(gold["ppo.advantage"])
            loss, g = grads_of(eng, sd, kind, b)
            go = {k: torch.from_numpy(gold[f"{kind}.grad.{k}"]) for k in losses.PI_KEYS}
            line += f" {kind} loss {abs(loss - float(gold[kind + '.loss'])):.2e} dg/g {gnorm(g, go):.2e}"
        print(line, flush=True)
        eng.close()

sd = H.weights()
batch = syn.collate_scenes([syn.make_scene(1000 + i) for i in range(256)])
data = batch["cur_pluto_feature_torch"]
out_o, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
r_pad = ~data["reference_line"]["valid_mask"].any(-1)
g0 = torch.Generator().manual_seed(99)
batch["advantage_torch"] = torch.randn(256, generator=g0)            # PPO's per-scene GAE advantage (normalised in the reference)
ref = {}
for kind in ("rift", "grpo", "reinforce", "ppo"):
    ref[kind] = losses.pi_head_loss_and_grads(sd, taps["q_final"], kind, H.clone_tree(batch), r_pad)[:2]
for mode in ("bf16", "fp16", "fp32"):
    eng = engine(mode, sd)
    o = eng.forward(data, train=True, no_drop=True, fp32=mode == "fp32", bn_update=False)
    le = float((o["probability"].cpu() - out_o["probability"])[~r_pad].abs().max())
    line = f"256 scenes {mode}: logits {le:.2e} |"
    for kind in ("rift", "grpo", "reinforce", "ppo"):
        loss, g = grads_of(eng, sd, kind, H.clone_tree(batch))
        line += f" {kind} loss {abs(loss - float(ref[kind][0])):.2e} dg/g {gnorm(g, ref[kind][1]):.2e}"
    print(line, flush=True)
    eng.close()
