"""Per-wave timeline of the FFN chunks of layer 1 in dec_w_kernel (RIFT_DEC_DBG=16 + RIFT_DEC_TS=1): stamps per chunk =
[after opening boundary of ffn.0 chunk (non-skewed waves), after its GEMM, (boundary of skewed waves), after the ReLU/dropout epilogue,
 after the ffn.3 boundary (non-skewed), after its GEMM, after the skewed waves' next boundary]."""
import os, sys
os.environ["RIFT_DEC_TS"] = "1"
os.environ["RIFT_DEC_DBG"] = os.environ.get("RIFT_DEC_DBG", "16")
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
ts = eng.tap("dec_ts").view(torch.int64).cpu().numpy().reshape(8, 32)
t0 = ts[ts > 0].min()
for w in range(8):
    row = ts[w]
    row = row[row > 0] - t0
    print(f"wave {w}:", " ".join(f"{int(v):6d}" for v in row))
