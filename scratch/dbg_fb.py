import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, torch.nn.functional as F
import smsut_amd
from smsut_amd import ops
def rnd(*shape, seed=0): return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape)).float()
def hw(w):
    o=ops.new_weight(*w.shape, device='cuda'); o.copy_(w); return o
n,h,ci,co=16,32,32,64; slope=0.01
x=rnd(n,ci,h,h,seed=1).double().requires_grad_(True)
w1=(rnd(co,ci,3,3,seed=2)/np.sqrt(9*ci)).double().requires_grad_(True); w2=(rnd(co,co,3,3,seed=3)/np.sqrt(9*co)).double().requires_grad_(True)
aff=[((1+0.1*rnd(co,seed=4+k)) if k%2==0 else 0.1*rnd(co,seed=4+k)).double().requires_grad_(True) for k in range(6)]
g1,b1,g2,b2,gs,bs=aff
ws=(rnd(co,ci,1,1,seed=11)/np.sqrt(ci)).double().requires_grad_(True)
y=F.leaky_relu(F.instance_norm(F.conv2d(x,w1,padding=1),weight=g1,bias=b1),slope)
y=F.instance_norm(F.conv2d(y,w2,padding=1),weight=g2,bias=b2)
out=F.leaky_relu(y+F.instance_norm(F.conv2d(x,ws),weight=gs,bias=bs),slope)
gout=rnd(*out.shape,seed=12).double(); out.backward(gout)
res={}
for fused in (True,False):
    ops.FUSED_BLOCK=fused
    xd=x.detach().float().cuda().requires_grad_(True)
    ps=[hw(w1.detach().float()).requires_grad_(True), *[t.detach().float().cuda().requires_grad_(True) for t in (g1,b1)], hw(w2.detach().float()).requires_grad_(True), *[t.detach().float().cuda().requires_grad_(True) for t in (g2,b2)], hw(ws.detach().float()).requires_grad_(True), *[t.detach().float().cuda().requires_grad_(True) for t in (gs,bs)]]
    if fused: o=ops.basic_block(xd,*ps,slope)
    else:
        a=ops.instnorm_act(ops.conv2d(xd,ps[0],None,1,1,True),ps[1],ps[2],slope)
        a=ops.instnorm_act(ops.conv2d(a,ps[3],None,1,1,True),ps[4],ps[5],None)
        i=ops.instnorm_act(ops.conv2d(xd,ps[6],None,1,0,True),ps[7],ps[8],None)
        o=ops.add_act(a,i,slope)
    o.backward(gout.float().cuda())
    res[fused]=[o.detach().cpu()]+[t.grad.cpu() for t in [xd]+ps]
refs=[out.detach()]+[t.grad for t in (x,w1,g1,b1,w2,g2,b2,ws,gs,bs)]
names=['out','x','w1','g1','b1','w2','g2','b2','ws','gs','bs']
for i,nm in enumerate(names):
    r=refs[i]
    for fused in (True,False):
        d=(res[fused][i].double()-r)
        print(nm, 'fused' if fused else 'unfused', 'l2rel %.2e maxrel %.2e n>1e-3: %d'%((d.norm()/r.norm()).item(), (d.abs().max()/r.abs().max()).item(), (d.abs()>1e-3*r.abs().max()).sum().item()))
