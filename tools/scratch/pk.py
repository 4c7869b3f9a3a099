import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rift_amd import _ffi, synthetic as syn
from tests import helpers as H
torch.cuda.set_device(0)
eng = _ffi.Engine("cuda:0")
eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
batch = syn.collate_scenes([syn.make_scene(1000 + i) for i in range(256)])
data = batch["cur_pluto_feature_torch"]
eng.forward(data, train=True, no_drop=True, bn_update=False)
torch.cuda.synchronize()
print("pack hdr", eng.tap("pe_pack_hdr").view(torch.int32).tolist(), "live hdr", eng.tap("pe_live_hdr").view(torch.int32).tolist())
tab = eng.tap("pe_pack_tab").view(torch.int32).cpu().view(-1, 32)
nr = int(eng.tap("pe_pack_hdr").view(torch.int32)[0])
used = (tab[:nr, :16] >= 0).sum(1)
print("rounds", nr, "tiles used per round: mean", float(used.float().mean()), "min", int(used.min()), "total tiles", int(used.sum()))
v = data["reference_line"]["valid_mask"]
last = (v * torch.arange(1, 121)).amax(-1)
print("expected tiles", int(((last + 15) // 16).sum()), "lines", int((last > 0).sum()))
