"""Phase timestamps of pe_mid_kernel (tile 0), the LDS-resident pass B kept behind RIFT_PE_W=0: RIFT_PE_TS=<20|120>.
(The default pass B is pe_w_kernel: tools/pew_ts.py.)"""
import os, sys
n = sys.argv[1] if len(sys.argv) > 1 else "20"
os.environ["RIFT_PE_TS"] = n
os.environ["RIFT_PE_W"] = "0"
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
ts = eng.tap("pe_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
print("npts", n, "stamps", len(ts), "total", ts[-1] - ts[0])
for nm, v in zip(["stage", "h1", "f", "store+pool", "gp", "g+stats"], d):
    print(f"{nm:12s} {v}")
