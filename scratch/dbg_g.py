import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import smsut_amd
from smsut_amd import ops
from smsut_amd.network.ugan import UGANnce
from oracle import recipe, smsut_oracle as O
H,nm,seed,B=64,4,61,4
gsd=recipe.fill(recipe.ugan_shapes(1,5,nm,16),seed)
x=recipe.synth_images((B,1,H,H),seed+10)
m=torch.tensor([[0.,-1,1,0]]*2+[[0.,0,1,-1]]*2)
ids=torch.from_numpy(np.random.RandomState(seed+40).permutation(16)[:64].astype(np.int64))
g64={k:v.double() for k,v in gsd.items()}
with torch.no_grad():
    seg64,tsl64,f64,_=O.ugan_forward(g64,x.double(),m.double(),[ids],n_modal=nm)
    seg32,tsl32,f32,_=O.ugan_forward(gsd,x,m,[ids],n_modal=nm)
rel=lambda a,b:((a.double()-b).abs().max()/b.abs().max()).item()
print('oracle32 vs 64', rel(seg32,seg64), rel(tsl32,tsl64), rel(f32[0],f64[0]))
for force in (False,True):
    ops.FORCE_GENERIC_CONV=force
    G=UGANnce(1,5,nm,16); G.load_state_dict(gsd); G.cuda().train()
    with torch.no_grad():
        seg,tsl,f,_=G(x.cuda(),m.cuda(),[ids.cuda()])
    print('force',force, rel(seg.cpu(),seg64), rel(tsl.cpu(),tsl64), rel(f[0].cpu(),f64[0]))
