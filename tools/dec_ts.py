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
names = ["load+commit"]
per = ["LN1", "r2r qkv0", "r2r attn0", "r2r qkv1", "r2r attn1", "r2r proj", "LN2", "m2m qkv0", "m2m attn0", "m2m qkv1", "m2m attn1", "m2m proj",
       "LN3", "cross q(+sync)", "kv0 stream", "x attn0(next sync)", "kv1 stream", "x attn1+sync", "x proj", "LN4",
       "fc1_0", "fc2_0|fc1_1", "sync", "fc2_1|fc1_2", "sync", "fc2_2|fc1_3", "sync", "fc2_3+epi"]
print(len(d), "intervals")
n_per = (len(d) - 2) // 4
for li in range(4):
    seg = d[1 + li * n_per: 1 + (li + 1) * n_per] if li else d[1:1 + n_per]
    if li == 2:
        print("layer 2 intervals:")
        for i, v in enumerate(seg):
            print(f"  {i:2d} {v}")
    print(f"layer {li} total {int(seg.sum())}")
print("first", d[0], "last", d[-1])
