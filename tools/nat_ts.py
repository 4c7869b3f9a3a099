"""Phase timestamps (clock64 of workgroup 0, thread 0) of one fused NAT level kernel: RIFT_NAT_TS=<level+1>."""
import os, sys
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ["RIFT_NAT_TS"] = str(lv + 1)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
sd = H.weights()
scenes = [syn.make_scene(i) for i in range(256)]
batch = syn.collate_scenes(scenes)
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
for _ in range(3):
    eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
torch.cuda.synchronize()
ts = eng.tap("nat_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
print("level", lv, "n stamps", len(ts), "total", ts[-1] - ts[0])
nch = 2 if lv == 2 else 1
names = ["load"]
for b in range(2):
    names += ["LN1"]
    for ch in range(nch):
        names += [f"qkv{ch}", f"attn{ch}"]
    names += ["proj", "LN2"]
    for ch in range(nch):
        names += [f"fc1_{ch}", f"fc2_{ch}"]
    names += ["fc2epi"]
names += ["tail"]
per = len(names)
ntile = len(d) // per
import numpy as np
T = np.array(d[:ntile * per]).reshape(ntile, per)
print("tiles recorded", ntile, "per-tile totals", T.sum(1).tolist())
for i, n in enumerate(names):
    print(f"{n:8s} {T[:, i].tolist()}")
