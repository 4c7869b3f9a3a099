"""Per-step losses of the first epoch of RLFTPluto.train for consecutive fresh policies in one process (first-use effects)."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_host_rlft import _filled_buffer
from rift_amd.planning import CBV_POLICY_LIST
from rift_amd.planning.fine_tuner.rlft import trainer as T
torch.cuda.set_device(0)
orig = T.RLFTTrainer.training_step
log = []
def wrapped(self, fb, extras):
    r = orig(self, fb, extras)
    self.wait_update()
    if not log[-1]:
        torch.cuda.synchronize()
        names = ["nat_out", "x_ego", "poly_pe", "r_pe", "x_tokens", "enc_out", "r_emb", "dec3", "q_final"]
        print("  taps", {n: round(self.engine.tap(n).double().nan_to_num().sum().item(), 4) for n in names},
              "nan", {n: int(self.engine.tap(n).isnan().sum()) for n in names if self.engine.tap(n).isnan().any()})
    log[-1].append(float(self.loss.item()))
    return r
T.RLFTTrainer.training_step = wrapped
for rep in range(3):
    log.append([])
    root = pathlib.Path(tempfile.mkdtemp())
    cfg = {'num_scenario': 1, 'ROOT_DIR': str(root), 'model_path': 'ckpt', 'device': 'cuda:0',
           'rlft': {'epochs': 2, 'warmup_epochs': 1, 'train_batch_size': 8, 'val_batch_size': 8, 'lr': 1e-3}}
    torch.manual_seed(0)
    pol = CBV_POLICY_LIST["rift_pluto"](cfg, None)
    with torch.no_grad():
        for p in pol.pluto_model.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    pol.load_model(resume=True); pol.set_mode('train'); pol.set_buffer(_filled_buffer(48, with_ref=False))
    pol.train(1)
    torch.cuda.synchronize()
    print(rep, [round(x, 7) for x in log[-1]])
