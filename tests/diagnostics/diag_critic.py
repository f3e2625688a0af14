"""Per-parameter gradient errors of the critic step (GPU vs CPU oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import nets, seeded, step
from learned_hologram_gan_amd.watermelon_hologram.discriminator import WGANGPDiscriminator192
DEV="cuda:0"
g=torch.Generator().manual_seed(21)
B,HW=2,int(os.environ.get("HW","32"))
real=torch.rand((B,3,HW,HW),generator=g); fake=torch.rand((B,3,HW,HW),generator=g)
alpha=torch.tensor([0.3,0.8]).view(2,1,1,1)
sd=nets.as_parameters(seeded.critic_state_dict())
def parts_ref():
    rv=nets.critic(sd,real,True); fv=nets.critic(sd,fake,True)
    gp=step.gradient_penalty(sd,real,fake,alpha)
    return rv,fv,gp
D=WGANGPDiscriminator192(None,32,True); D.load_state_dict(seeded.critic_state_dict()); D.train()
def parts_gpu():
    r,f=real.to(DEV),fake.to(DEV)
    rv=D(r); fv=D(f)
    xh=(alpha.to(DEV)*r+(1-alpha.to(DEV))*f).requires_grad_(True)
    s=D(xh)
    (gx,)=torch.autograd.grad(s,xh,torch.ones_like(s),create_graph=True,retain_graph=True)
    gp=((gx.view(B,-1).norm(2,dim=1)-1)**2).mean()
    return rv,fv,gp
for name,sel in (("wgan", lambda rv,fv,gp: -rv.mean()+fv.mean()), ("gp", lambda rv,fv,gp: gp)):
    for v in sd.values():
        if v.requires_grad: v.grad=None
    D.zero_grad()
    lr_=sel(*parts_ref()); lr_.backward()
    lg=sel(*parts_gpu()); lg.backward()
    print(f"== {name}: loss ref {lr_.item():.8f} gpu {lg.item():.8f}")
    for k,p in D.named_parameters():
        a=p.grad.cpu(); b=sd[k].grad
        print(f"  {k:18s} |g| {b.norm().item():.3e}  rel {((a-b).norm()/b.norm().clamp_min(1e-30)).item():.2e}  max-rel {((a-b).abs().max()/b.abs().max().clamp_min(1e-30)).item():.2e}")
