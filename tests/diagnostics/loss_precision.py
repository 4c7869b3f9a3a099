"""RIFT loss of the HIP path (bf16 and fp32 modes, train-mode BatchNorm, drops disabled) against the CPU oracle on batches of
growing size: how far the bf16 trunk moves the loss (north_star bar: 1e-4)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from oracle import losses, pluto_ref
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H

sd = H.weights()
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
for n in (8, 32, 64, 256):
    scenes = [syn.make_scene(1000 + i) for i in range(n)]
    batch = syn.collate_scenes(scenes)
    data = batch["cur_pluto_feature_torch"]
    t0 = time.time()
    torch.set_num_threads(32)
    out_o, _, taps = pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=False, want_taps=True)
    r_pad = ~data["reference_line"]["valid_mask"].any(-1)
    lo = float(losses.rift_loss(out_o["probability"], r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                batch["group_advantage_mask_torch"]))
    res = {}
    for mode in ("bf16", "fp32"):
        o = eng.forward(data, train=True, no_drop=True, fp32=(mode == "fp32"), bn_update=False)
        prob = o["probability"].cpu()
        res[mode] = (float(losses.rift_loss(prob, r_pad, batch["old_group_logits_torch"], batch["group_advantage_torch"],
                                            batch["group_advantage_mask_torch"])), float((prob - out_o["probability"])[~r_pad].abs().max()))
    print(f"n={n:4d} oracle loss {lo:.7f} | bf16 loss err {abs(res['bf16'][0]-lo):.2e} (max logit err {res['bf16'][1]:.2e}) | "
          f"fp32 loss err {abs(res['fp32'][0]-lo):.2e} (max logit err {res['fp32'][1]:.2e}) | oracle {time.time()-t0:.1f}s", flush=True)
