import sys, types; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import smsut_amd
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
from oracle import recipe
g=np.load('tests/golden/iter_small.npz')
bs,H,nm,seed=int(g['bs']),int(g['H']),int(g['nm']),int(g['seed']); B=2*bs
cfg.input_size, cfg.batch_size = H, bs
tr=UGANConsisTrainer('train', types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.load_state_dict(recipe.fill(recipe.ugan_shapes(1,5,nm,16),seed)); tr.D.load_state_dict(recipe.fill(recipe.disc_shapes(H,nm,16,256),seed+1))
tr.epoch, tr.iter = int(g['epoch']), int(g['it0'])
grads={}
def grab(prefix,module):
    for k,p in module.named_parameters():
        if p.grad is not None: grads[prefix+k]=p.grad.detach().clone()
d_step,g_step=tr.d_optimizer.step,tr.optimizer.step
def ds(): grab('D.',tr.D); print('D strides', tr.D.conv_cls.weight.stride(), tr.D.conv_cls.weight.grad.stride()); w0=tr.D.conv_cls.weight.detach().clone(); d_step(); print('max upd', (tr.D.conv_cls.weight.detach()-w0).abs().max().item(), (tr.D.conv_cls.weight.detach()-w0).abs().mean().item())
def gs(): grab('G.',tr.net); g_step()
tr.d_optimizer.step=ds; tr.optimizer.step=gs
x=recipe.synth_images((B,1,H,H),seed+10).cuda(); y=recipe.synth_labels(bs,H,H,5,seed+20,block=8).cuda()
al=torch.from_numpy(np.random.RandomState(seed+30).standard_normal((B,1,1,1))).float().cuda()
ids=torch.from_numpy(np.random.RandomState(seed+40).permutation(16)[:64].astype(np.int64)).cuda()
tr.train_iteration(x,y,torch.tensor([1]*bs+[3]*bs),mj=int(g['mj0']),alpha=al,sample_ids=[ids])
for pre,tag in (('D.','D0_'),('G.','G0_')):
    for n,ref in zip([str(n) for n in g[tag+'grad_names']], g[tag+'grad_l2']):
        got=float(grads[pre+n].double().norm())
        if abs(got-ref)/max(ref,1e-12)>2e-3: print(tag,n,got,ref)
    for k in g.files:
        if k.startswith(tag+'grad::'):
            a=grads[pre+k[len(tag)+6:]].cpu().numpy(); b=g[k]
            print(k, 'l2rel', np.linalg.norm(a-b)/np.linalg.norm(b), 'maxabs', np.abs(a-b).max(), np.abs(b).max())
d=tr.D.state_dict()['conv_cls.weight'].cpu().numpy(); r=g['post0_D_cls']
w0=recipe.fill(recipe.disc_shapes(H,nm,16,256),seed+1)['conv_cls.weight'].numpy()
print('upd got', np.unique(np.round((d-w0),4))[:10], 'upd ref', np.unique(np.round(r-w0,4))[:10])
