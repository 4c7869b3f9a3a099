import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
sd = H.weights()
data = syn.collate_scenes([syn.make_scene(5000 + i) for i in range(64)])["cur_pluto_feature_torch"]
hist = data["agent"]["valid_mask"][:, :, :21].any(-1).clone(); hist[:, 0] = False; hist = hist.flatten()
slots = hist.nonzero().flatten()
rank = torch.arange(len(slots))
got = {}
for mode in ("1", "0"):
    os.environ["RIFT_NAT_COMPACT"] = mode
    eng = _ffi.Engine("cuda:0"); eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    eng.forward(data); torch.cuda.synchronize()
    got[mode] = eng.tap("nat_out").view(-1, 128).cpu().clone()[hist]
    eng.close()
d = (got["1"] - got["0"]).abs().amax(-1)
diff = d > 0
print("hist sequences", len(slots), "differing", int(diff.sum()), "max", float(d.max()), "scale", float(got["0"].abs().max()))
for m in (3, 4, 12):
    same = (slots % m) == (rank % m)
    print(f"mod {m}: same position {int(same.sum())}; differing among same-position {int((diff & same).sum())}, among moved {int((diff & ~same).sum())} of {int((~same).sum())}")
print("first differing ranks", diff.nonzero().flatten()[:20].tolist())
print("diff magnitudes", d[diff][:10].tolist())
