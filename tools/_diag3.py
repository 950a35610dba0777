import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R, os.path.join(R,"make-it-3d_amd")]
import torch
from mi3d import dp, grid_ops, rays as Rr, sd_standin, sds_step
dev=torch.device("cuda:0")
guidance=sd_standin.StableDiffusionStandIn(dev)
text_z=guidance.get_text_embeds()
t_fixed=torch.tensor([400],dtype=torch.long,device=dev)
def run(tag, **over):
    opt=sds_step.make_opt(max_steps=1024, **over)
    model,optimizer,scaler=sds_step.build_training_state(opt,dev,seed=0,bitfield="dense")
    scaler=torch.amp.GradScaler('cuda',init_scale=INIT)
    ro,rd,ds=Rr.view_rays(128,128,device=dev)
    orig=torch.nn.utils.clip_grad_norm_
    info={}
    def clip(params,max_norm):
        ps=list(params)
        info['nonfinite']={n:int((~torch.isfinite(p.grad)).sum()) for n,p in model.named_parameters()}
        info['gmax']={n:float(torch.nan_to_num(p.grad,nan=0,posinf=0,neginf=0).abs().max()) for n,p in model.named_parameters()}
        return orig(ps,max_norm)
    torch.nn.utils.clip_grad_norm_=clip
    try:
        for i in range(3):
            loss=sds_step.sds_train_step(model,guidance,text_z,optimizer,scaler,ro,rd,ds,128,128,opt,sds_backward="single",t=t_fixed)
            print(tag,i,"loss",float(loss),"scale",scaler.get_scale(),info['nonfinite'],{k:"%.2e"%v for k,v in info['gmax'].items()})
    finally:
        torch.nn.utils.clip_grad_norm_=orig

for INIT in (1.0, 2.0**-4, 2.0**-8, 2.0**-12):
    run("init%g"%INIT)
