import os, sys
os.environ["RIFT_ENC_TS"] = "1"
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
ts = eng.tap("enc_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = (ts[1:] - ts[:-1])
print("n stamps", len(ts), "total cycles", ts[-1] - ts[0])
names = ["load"]
for bi in range(4):
    names += ["LN1", "qkv0", "attn0", "qkv1", "attn1", "oproj", "LN2"]
    for hc in range(4):
        names += [f"fc1_{hc}", f"fc2_{hc}"]
    names += ["fc2epi"]
for n, v in zip(names, d):
    print(f"{n:8s} {v}")
