"""Phase timestamps (clock64 of workgroup 0) of the fused planning-decoder kernel: RIFT_DEC_TS=1."""
import os, sys
os.environ["RIFT_DEC_TS"] = "1"
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
ts = eng.tap("dec_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
print("n stamps", len(ts), "total", ts[-1] - ts[0], "R of scene 0:", int(batch["cur_pluto_feature_torch"]["reference_line"]["valid_mask"][0].any(-1).sum()))
per = ["par_commit", "LN1", "r2r qkv0", "r2r attn0", "r2r qkv1", "r2r attn1", "r2r proj", "LN2", "m2m qkv0", "m2m attn0", "m2m qkv1",
       "m2m attn1", "m2m proj", "LN3", "cross q(+sync)", "kv0 stream", "x attn0(+sync)", "kv1 stream", "x attn1", "x proj", "LN4",
       "fc1_0", "epi0", "fc2_0|fc1_1", "epi1", "fc2_1|fc1_2", "epi2", "fc2_2|fc1_3", "epi3", "fc2_3+res"]
assert len(d) == 1 + 4 * len(per), len(d)
import numpy as np
L = np.array(d[:-1]).reshape(4, len(per))   # layer 0's par_commit includes the initial loads
print("writeback", d[-1], "layer totals", L.sum(1).tolist())
for i, n in enumerate(per):
    print(f"  {n:18s} {L[:, i].tolist()}  mean {L[:, i].mean():.0f}")
