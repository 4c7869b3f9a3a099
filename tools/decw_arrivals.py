"""When each wave of workgroup 0 ARRIVES at the group boundaries of one decoder layer (RIFT_DEC_TS=1: dec_w.hip DTS_ARR), relative to the
previous boundary's opening: which wave a group waits for.  Needs the diagnostic build of the decoder unit:
    HIPCC_EXTRA=-DRIFT_DEC_ARR=1 python -c "from rift_amd import build; build.build(force=True)"; python tools/decw_arrivals.py [layer]
(RIFT_DEC_DBG as for decw_ts.py; rebuild without the define afterwards)"""
import os, sys
os.environ["RIFT_DEC_TS"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
LAYER = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = H.weights()
batch = syn.collate_scenes([syn.make_scene(i) for i in range(256)])
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
for _ in range(3):
    eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
torch.cuda.synchronize()
ts = eng.tap("dec_ts").view(torch.int64).cpu().numpy()
opens = ts[:128]; opens = opens[opens != 0]                       # wave 0: the moment behind each barrier
arr = ts[128:128 + 8 * 112].reshape(8, 112)
names = ["r2r q", "r2r k", "r2r v+attn", "r2r out", "m2m q", "m2m k", "m2m v+attn", "m2m out+LN3", "cross q", "cross h01", "cross h23", "cross out+LN4",
         "fc1_0", "fc2_0", "fc1_1", "fc2_1", "fc1_2", "fc2_2", "fc1_3", "fc2_3+res"]
print("dbg", os.environ.get("RIFT_DEC_DBG", "0"), "layer", LAYER, " (ticks from the group's opening to each wave's arrival at its closing barrier; * = last)")
for g in range(20):
    k = LAYER * 20 + g                     # boundary k opens group k; arrival index k + 1 closes it
    t0 = opens[k]
    a = arr[:, k + 1] - t0
    last = int(np.argmax(a))
    print(f"  {names[g]:14s}" + "".join(f"{int(v):6d}{'*' if w == last else ' '}" for w, v in enumerate(a)) + f"   opens next after {int(opens[k + 1] - t0)}")
