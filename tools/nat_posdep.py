"""Is the agent encoder's output of a scene independent of where the scene sits in the batch (bitwise)?  (diagnostic)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
sd = H.weights()
scenes = [syn.make_scene(300 + i) for i in range(12)]
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in sd.items()})
def run(idx):
    b = syn.collate_scenes([scenes[i] for i in idx])
    eng.forward(b["cur_pluto_feature_torch"], train=True, no_drop=True, bn_update=False)
    return eng.tap("nat_out").view(len(idx), 64, 128).clone()
full = run(range(12))
again = run(range(12))
print("deterministic:", torch.equal(full, again))
for lo, hi in ((0, 5), (5, 9), (9, 12), (0, 6), (6, 12), (3, 4)):
    part = run(range(lo, hi))
    d = (part - full[lo:hi]).abs()
    print(f"scenes [{lo},{hi}): max diff {float(d.max()):.3e}, differing agents {int((d.amax(-1) > 0).sum())} of {d.shape[0] * 64}",
          "first:", (d.amax(-1) > 0).nonzero()[:6].tolist())
