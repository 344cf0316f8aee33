import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import smsut_amd
from smsut_amd import ops
from smsut_amd.network.ugan import Discriminator
from oracle import recipe, smsut_oracle as O
import torch.nn.functional as F
H,nm,seed,B=64,4,61,4
dsd=recipe.fill(recipe.disc_shapes(H,nm,16,256),seed+1)
x=recipe.synth_images((B,1,H,H),seed+10); xf=recipe.synth_images((B,1,H,H),seed+11)*0.5
al=torch.from_numpy(np.random.RandomState(seed+30).standard_normal((B,1,1,1))).float()
xh=(al*x+(1-al)*xf)
d64={k:v.double() for k,v in dsd.items()}
xh64=xh.double().requires_grad_(True)
s64,_=O.discriminator_forward(d64,xh64)
dy64,=torch.autograd.grad(s64,xh64,torch.ones_like(s64),create_graph=True)
xh32=xh.clone().requires_grad_(True)
s32,_=O.discriminator_forward(dsd,xh32)
dy32,=torch.autograd.grad(s32,xh32,torch.ones_like(s32),create_graph=True)
print('oracle32 vs 64: src', (s32.double()-s64).abs().max().item()/s64.abs().max().item(), 'dydx', ((dy32.double()-dy64).norm()/dy64.norm()).item(), 'gp', O.gradient_penalty(s32,xh32).item(), O.gradient_penalty(s64,xh64).item())
for force in (False, True):
    ops.FORCE_GENERIC_CONV=force
    D=Discriminator(H,nm,16,256); D.load_state_dict(dsd); D.cuda().train()
    xg=xh.clone().cuda().requires_grad_(True)
    sg,_=D(xg)
    with ops.input_grads_only():
        dyg,=torch.autograd.grad(sg,xg,torch.ones_like(sg),create_graph=True)
    print('force_generic',force,'src', (sg.detach().cpu().double()-s64.detach()).abs().max().item()/s64.abs().max().item(), 'dydx', ((dyg.detach().cpu().double()-dy64.detach()).norm()/dy64.norm()).item(), 'gp', ops.grad_penalty(dyg).item())
    h=xg; hr=xh64
    mods=list(D.main)
    h=mods[1](mods[0](h)); hr=O._act(F.conv2d(hr,d64['main.0.weight'],d64['main.0.bias'],stride=2,padding=1))
    print('  stem', ((h.detach().cpu().double()-hr).norm()/hr.norm()).item())
    for i in range(2,len(mods)):
        h=mods[i](h); hr=O.bottle_block(d64,f'main.{i}.',hr)
        print('  block',i, ((h.detach().cpu().double()-hr.detach()).norm()/hr.norm()).item(), tuple(h.shape))
