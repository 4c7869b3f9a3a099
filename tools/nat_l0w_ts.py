"""Section timestamps (clock64, wave 0 of workgroup 0) of the wave-private level-0 NAT kernel: RIFT_NAT_TS=1."""
import os, sys
os.environ["RIFT_NAT_TS"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
sd = H.weights()
batch = syn.collate_scenes([syn.make_scene(i) for i in range(256)])
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
for _ in range(3):
    eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
torch.cuda.synchronize()
ts = eng.tap("nat_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
names = ["prologue (weights -> LDS)"] + ["tile start", "tokenizer", "b0 attention half", "b0 MLP half", "b1 attention half", "b1 MLP half", "norm0 + staging + ds conv", "ds LayerNorm + store"] * 4
print("total cycles", ts[-1] - ts[0])
for n, v in zip(names, d):
    print(f"  {n:28s} {v}")
