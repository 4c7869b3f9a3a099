"""Group-boundary timestamps (wave 0 of workgroup 0) of the wave-private level-2 NAT kernel: RIFT_NAT_TS=3."""
import os, sys
os.environ["RIFT_NAT_TS"] = "3"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
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
print("n stamps", len(ts), "total", ts[-1] - ts[0])
names = ["q (+LN1)", "k", "v + attention", "proj + res + LN2", "fc1_0 + GELU", "fc2_0", "fc1_1 + GELU", "fc2_1", "fc1_2 + GELU", "fc2_2 (+res, next LN1 / out)"]
n = len(d) // 10
L = np.array(d[: n * 10]).reshape(n, 10)
for i, nm in enumerate(names):
    print(f"  {nm:30s} {L[:, i].tolist()}")
