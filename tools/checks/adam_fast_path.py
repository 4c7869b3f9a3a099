import torch, sys
sys.path.insert(0,'/root/repo')
# equivalence of the fast path with optimizer.step() on GPU
torch.manual_seed(0)
def make():
    ps=[torch.nn.Parameter(torch.randn(128,128,device='cuda')), torch.nn.Parameter(torch.randn(128,device='cuda'))]
    return ps
a=make(); b=[torch.nn.Parameter(p.detach().clone()) for p in a]
oa=torch.optim.AdamW([{"params":[a[0]],"weight_decay":1e-5},{"params":[a[1]],"weight_decay":0.0}],lr=1e-3,weight_decay=1e-5,fused=True)
ob=torch.optim.AdamW([{"params":[b[0]],"weight_decay":1e-5},{"params":[b[1]],"weight_decay":0.0}],lr=1e-3,weight_decay=1e-5,fused=True)
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer
class T: pass
t=T(); t.optimizer=ob; t._fast_groups=None
for it in range(5):
    for p,q in zip(a,b):
        g=torch.randn_like(p); p.grad=g.clone(); q.grad=g.clone() if q.grad is None else q.grad.copy_(g)
    oa.step(); RLFTTrainer._optimizer_step(t)
    for g in ob.param_groups: g["lr"]*=0.9
    for g in oa.param_groups: g["lr"]*=0.9
print("fast groups:", type(t._fast_groups), max(float((p-q).abs().max()) for p,q in zip(a,b)))
