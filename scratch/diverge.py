import sys, types; sys.path.insert(0,'.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
from smsut_amd.misc.synthetic import SyntheticSliceLoader
torch.manual_seed(2020)
cfg.batch_size=8
tr=UGANConsisTrainer('train', types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb=SyntheticSliceLoader(8, device='cuda', labeled=True); ul=SyntheticSliceLoader(8, device='cuda', labeled=False)
fresh = len(sys.argv)>1 and sys.argv[1]=='fresh'
li, ui = iter(lb), iter(ul)
(x1,y1,m1,_),(x2,_,m2,_)=next(li),next(ui)
for it in range(20):
    if fresh and it>0: (x1,y1,m1,_),(x2,_,m2,_)=next(li),next(ui)
    s=tr.train_iteration(torch.cat([x1,x2],0), y1, torch.cat([m1,m2],0))
    print(it, ' '.join(f'{k}={v:.3g}' for k,v in zip(SCALARS,s.tolist())), flush=True)
