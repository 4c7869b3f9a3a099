"""Group-boundary timestamps (clock64 of wave 0 of workgroup 0) of the wave-private planning-decoder kernel (dec_w.h): RIFT_DEC_TS=1.
An interval = one group's compute on that wave + its wait at the next group barrier."""
import os, sys
os.environ["RIFT_DEC_TS"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
sd = H.weights()
RMAX = int(os.environ.get('DEC_TS_R', '6'))          # (reference-line slots of the batch: how the group times move with the tiles of a workgroup)
scenes = [syn.make_scene(i, r_max=RMAX) for i in range(256)]
batch = syn.collate_scenes(scenes)
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
for _ in range(3):
    eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
torch.cuda.synchronize()
ts = eng.tap("dec_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
print("n stamps", len(ts), "total", ts[-1] - ts[0], "dbg", os.environ.get("RIFT_DEC_DBG", "0"))
if os.environ.get("DEC_TS_BRIEF"):
    sys.exit(0)
per = ["r2r q (+LN1, xs read)", "r2r k", "r2r v + attn", "r2r out + res + xs write", "m2m q (+LN2, xs read)", "m2m k", "m2m v + attn",
       "m2m out + res + LN3", "cross q", "cross heads 0,1", "cross heads 2,3", "cross out + res + LN4",
       "fc1_0", "fc2_0", "fc1_1", "fc2_1", "fc1_2", "fc2_2", "fc1_3", "fc2_3 + res"]
assert len(d) == 4 * len(per), len(d)
L = np.array(d).reshape(4, len(per))
print("layer totals", L.sum(1).tolist())
for i, n in enumerate(per):
    print(f"  {n:28s} {L[:, i].tolist()}  mean {L[:, i].mean():.0f}")
