import sys, os
sys.path[:0]=[os.environ.get("GRAFT_REPO_ROOT","/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"make-it-3d_amd")]
import torch
from mi3d import rays as R, sds_step
dev=torch.device('cuda:0')
opt=sds_step.make_opt(max_steps=64, fp16=False)
model,optimizer,scaler=sds_step.build_training_state(opt,dev,seed=0,bitfield=0.5)
with torch.no_grad(): model.encoder.params.uniform_(-0.1,0.1)
ro,rd,ds=R.view_rays(32,32,device=dev)
torch.manual_seed(5)
out=model.render(ro,rd,depth_scale=ds,bg_color=torch.rand(3,device=dev),perturb=True,force_all_rays=True,max_steps=64)
img=out['image']; ws=out['weights_sum']
params=[model.encoder.params]+list(model.sigma_net.parameters())
gi=torch.randn_like(img)
A=(img*gi).sum()
B=sds_step.regularisers(opt,out,ws.reshape(1,1,32,32))
def G(y): return [g.clone() for g in torch.autograd.grad(y,params,retain_graph=True)]
gA,gB,gAB=G(A),G(B),G(A+B)
gA2=G(A)
for i,(a,b,ab,a2) in enumerate(zip(gA,gB,gAB,gA2)):
    s=ab.abs().max().item()
    print(i, 'scale',s,'|A+B-(gA+gB)|',(ab-(a+b)).abs().max().item(),'repeat A diff',(a-a2).abs().max().item(), 'normA',a.abs().max().item(),'normB',b.abs().max().item())
# individual reg terms
for k in ('loss_orient','loss_smooth'):
    g1=G(out[k]); g2=G(out[k])
    print(k,'repeat diff',(g1[0]-g2[0]).abs().max().item(),'scale',g1[0].abs().max().item())
g1=G((ws**2).mean()); g2=G((ws**2).mean()); print('opacity repeat',(g1[0]-g2[0]).abs().max().item(), g1[0].abs().max().item())
