import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R, os.path.join(R,"make-it-3d_amd"), os.path.join(R,"tests")]
import torch
from mi3d import sds_step
import test_sds_step_gpu as T
dev=torch.device('cuda:0')
def run(mode):
    opt, model, optimizer, scaler, (ro, rd, ds) = T._setup(dev, fp16=False)
    guidance = T._TinyGuidance(dev, deterministic=True)
    text_z = torch.randn(2, 77, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    cap = {}
    orig = torch.nn.utils.clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = lambda params, max_norm: cap.update({n: p.grad.detach().clone() for n, p in model.named_parameters()})
    try:
        torch.manual_seed(5)
        sds_step.sds_train_step(model, guidance, text_z, optimizer, scaler, ro, rd, ds, 32, 32, opt, sds_backward=mode, t=torch.tensor([400], device=dev))
    finally:
        torch.nn.utils.clip_grad_norm_ = orig
    return cap
runs={k:run(m) for k,m in [("s1","single"),("s2","single"),("r1","reference"),("r2","reference")]}
for a,b in [("s1","s2"),("r1","r2"),("s1","r1")]:
    for n in runs[a]:
        x,y=runs[a][n],runs[b][n]
        sc=float(y.abs().max())
        d=(x-y).abs()
        print(a,b,n,'scale %.3e maxdiff %.3e rel %.2e frac>1e-4scale %.4f'%(sc,float(d.max()),float(d.max())/sc,float((d>1e-4*sc).float().mean())))
