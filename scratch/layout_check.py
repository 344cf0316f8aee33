import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg, ops
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
from smsut_amd.trainer.unetTrainer import UnetTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False)
tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
c0 = ops.layout_copies(); tr.train_iteration(torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)); torch.cuda.synchronize()
print("ugan eager iteration: layout copies", ops.layout_copies() - c0)
cfg.batch_size = 32
ut = UnetTrainer("train", ns); ut.net.train()
img, msk = next(iter(SyntheticSliceLoader(32, device=dev)))[:2]
c0 = ops.layout_copies(); ut.train_step(img, msk); torch.cuda.synchronize()
print("unet eager step: layout copies", ops.layout_copies() - c0)
