"""Measured device-vs-oracle error of every float output of the a11 chain (rift_ref_line_info, rift_rollout, return, group z-score),
two consecutive calls (persistent PID state), several seeds.  Run on a GPU box: python tests/diagnostics/a11_errors.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import torch
from oracle import advantage as oadv, rollout as orl
from rift_amd import _ffi
from tests import helpers as H

eng = _ffi.Engine("cuda:0")
worst = {}
for base in (777, 991, 1234, 4242):
    ro = orl.Rollout()
    pid = eng.new_pid_state(64)
    for call, seed in enumerate((base, base + 1)):
        traj, ref_pos, ref_ang, st = H.rollout_inputs(seed)
        t40 = traj[:, :, :40, :]
        dd, da, ci = orl.ref_line_info(t40, ref_pos, ref_ang)
        hdd, hda, hci = eng.ref_line_info(traj, ref_pos, ref_ang, Ts=40)
        gpos, ghead = orl.to_global(t40, torch.tensor(st["pos"]), torch.tensor(st["heading"]))
        ref = ro.propagate(gpos, ghead, st["speed"], st["width"], st["length"])
        cs = torch.tensor([[st["pos"][0], st["pos"][1], st["heading"], st["speed"], st["width"], st["length"]]])
        out = eng.rollout(traj.reshape(-1, 80, 6), cs, pid)
        torch.cuda.synchronize()
        e = {"delta_dis": float((hdd.cpu() - dd).abs().max()), "delta_angle": float((hda.cpu() - da).abs().max())}
        for k in ("center", "angle", "speed", "acc", "ang_vel", "ang_acc", "vertices"):
            e[k] = float((out[k].cpu().double() - ref[k].double()).abs().max())
        G = traj.shape[0] * traj.shape[1]
        col = np.zeros((G, 40), dtype=bool); off = np.zeros((G, 80), dtype=bool)
        ret_o = oadv.rollout_return(dd.numpy(), da.numpy(), ref["speed"][:, :40].numpy(), ref["acc"][:, :40].numpy(),
                                    ref["ang_vel"][:, :40].numpy(), ref["ang_acc"][:, :40].numpy(), col, off)
        ret_d = eng.rollout_return(hdd, hda, out["speed"][:, :40].contiguous(), out["acc"][:, :40].contiguous(), out["ang_vel"][:, :40].contiguous(),
                                   out["ang_acc"][:, :40].contiguous(), torch.from_numpy(col), torch.from_numpy(off)).cpu().numpy()
        adv_o = oadv.group_zscore(ret_o)
        adv_d = eng.group_advantage(torch.from_numpy(ret_d).view(1, -1)).cpu().numpy().reshape(-1)
        e["return"] = float(np.abs(ret_d - ret_o).max()); e["advantage"] = float(np.abs(adv_d - adv_o).max())
        e["int_equal"] = bool(np.array_equal(out["closest_index"].cpu().numpy(), ref["closest_index"].numpy().astype(np.int32)) and
                              np.array_equal(out["aim_idx"].cpu().numpy(), ref["aim_idx"].numpy().astype(np.int32)) and
                              np.array_equal(hci.cpu().numpy(), ci.numpy().astype(np.int32)))
        print(f"seed {seed} call {call}: " + "  ".join(f"{k} {v:.2e}" if not isinstance(v, bool) else f"{k} {v}" for k, v in e.items()), flush=True)
        for k, v in e.items():
            if not isinstance(v, bool):
                worst[k] = max(worst.get(k, 0.0), v)
print("WORST " + "  ".join(f"{k} {v:.2e}" for k, v in worst.items()))
