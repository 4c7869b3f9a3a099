"""Timestamps of pe_w_kernel (workgroup 0, wave 0): RIFT_PEW_TS=1.  Per round: 7 group boundaries (W1 | W2 x 2 | W3a x 4) and, inside the
active path, marks after the held stores, the statistics write-out, h1, and around each g group's GEMM / epilogue.  Prints the cycle
differences between consecutive stamps of the first rounds."""
import os, sys
os.environ["RIFT_PEW_TS"] = "1"
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
ts = eng.tap("pew_ts").view(torch.int64).cpu().numpy()
ts = ts[ts != 0]
d = ts[1:] - ts[:-1]
print("stamps", len(ts), "total", ts[-1] - ts[0])
names = ["b0", "held", "stats", "h1", "b1", "b2", "b3"] + [x for q in range(4) for x in (f"g{q}pre", f"g{q}mm", f"g{q}epi", f"b{4+q}")][:-1]
per = len(names)
print(names)
for r in range(0, len(d), per):
    print("round", r // per, list(d[r:r + per]))
