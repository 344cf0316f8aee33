import sys, types; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import smsut_amd
from smsut_amd import ops, config as cfg
from smsut_amd.network.ugan import UGANnce, Discriminator
from oracle import recipe
g=np.load('tests/golden/iter_small.npz')
bs,H,nm,seed=int(g['bs']),int(g['H']),int(g['nm']),int(g['seed']); B=2*bs
gsd=recipe.fill(recipe.ugan_shapes(1,5,nm,16),seed); dsd=recipe.fill(recipe.disc_shapes(H,nm,16,256),seed+1)
x=recipe.synth_images((B,1,H,H),seed+10).cuda()
al=torch.from_numpy(np.random.RandomState(seed+30).standard_normal((B,1,1,1))).float().cuda()
ids=torch.from_numpy(np.random.RandomState(seed+40).permutation(16)[:64].astype(np.int64)).cuda()
mo=torch.tensor([1]*bs+[3]*bs); mj=int(g['mj0'])
oh=lambda idx: torch.nn.functional.one_hot(idx,nm).float()
vec_ot=(oh(torch.full_like(mo,mj))-oh(mo)).cuda()
res={}
for force in (True,False):
    ops.FORCE_GENERIC_CONV=force
    G=UGANnce(1,5,nm,16); G.load_state_dict(gsd); G.cuda().train()
    D=Discriminator(H,nm,16,256); D.load_state_dict(dsd); D.cuda().train()
    with torch.no_grad(): _,xf,_,_=G(x,vec_ot,sample_ids=[ids])
    src,_=D(x); d_real=ops.mean_all(src,-1.0)
    src,_=D(xf); d_fake=ops.mean_all(src,1.0)
    xh=ops.row_lerp(x,xf,al).requires_grad_(True)
    sh,_=D(xh)
    with ops.input_grads_only():
        dy,=torch.autograd.grad(sh,xh,torch.ones_like(sh),create_graph=True,retain_graph=True)
    gp=ops.grad_penalty(dy)
    res[force]=(xf.clone(),xh.detach().clone(),sh.detach().clone(),dy.detach().clone())
    print('force',force,'d_real',d_real.item(),'d_fake',d_fake.item(),'gp',gp.item(), 'golden', g['scalars'][0][:4])
a,b=res[True],res[False]
for n,(u,v) in zip(('xf','xh','sh','dy'),zip(a,b)):
    print(n, ((u-v).norm()/u.norm()).item(), (u-v).abs().max().item())
# per-sample norms
print((a[3].reshape(B,-1).norm(dim=1)).tolist(), (b[3].reshape(B,-1).norm(dim=1)).tolist())
# now feed the SAME xh to both
for force in (True,False):
    ops.FORCE_GENERIC_CONV=force
    D=Discriminator(H,nm,16,256); D.load_state_dict(dsd); D.cuda().train()
    xh=a[1].clone().requires_grad_(True)
    sh,_=D(xh)
    with ops.input_grads_only():
        dy,=torch.autograd.grad(sh,xh,torch.ones_like(sh),create_graph=True,retain_graph=True)
    print('same xh force',force, ops.grad_penalty(dy).item(), dy.reshape(B,-1).norm(dim=1).tolist())
print('---- flip hunt')
rec={}
orig_in, orig_add = ops.instnorm_act, ops.add_act
import smsut_amd.network.blocks as blk
for force in (True,False):
    ops.FORCE_GENERIC_CONV=force
    lst=[]
    def in_hook(x,g,b,s,_o=orig_in): y=_o(x,g,b,s); lst.append(('in',y.detach().clone(),x.detach().clone())); return y
    def add_hook(a,b,s,_o=orig_add): y=_o(a,b,s); lst.append(('add',y.detach().clone(),None)); return y
    ops.instnorm_act=in_hook; ops.add_act=add_hook
    D=Discriminator(H,nm,16,256); D.load_state_dict(dsd); D.cuda().train()
    xh=a[1].clone().requires_grad_(True)
    sh,_=D(xh)
    rec[force]=lst
ops.instnorm_act, ops.add_act = orig_in, orig_add
for i,(u,v) in enumerate(zip(rec[True],rec[False])):
    yu,yv=u[1],v[1]
    flips=((yu>0)!=(yv>0)).sum().item()
    msg=f'{i} {u[0]} shape {tuple(yu.shape)} flips {flips} maxdiff {(yu-yv).abs().max().item():.3e}'
    if flips:
        idx=((yu>0)!=(yv>0)).nonzero()
        msg+=f' at {idx[:3].tolist()} vals {yu[tuple(idx[0])].item():.3e} {yv[tuple(idx[0])].item():.3e}'
        if u[2] is not None: msg+=f' preIN {u[2][tuple(idx[0])].item():.6e} {v[2][tuple(idx[0])].item():.6e}'
    print(msg)
