"""Run only the D phase of the uganConsis iteration (eager), for rocprofv3 --kernel-trace --stats."""
import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False); dev = torch.device("cuda")
B = 16; cfg.batch_size = B // 2
tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
x = torch.randn(B, 1, 256, 256, device=dev).clamp(-1, 1); xf = torch.randn(B, 1, 256, 256, device=dev).clamp(-1, 1)
modal = torch.randint(0, 4, (B,), device=dev); alpha = torch.randn(B, 1, 1, 1, device=dev)
d_params = list(tr.D.parameters())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(N):
    for p in d_params: p.grad = None
    tr._d_phase(x, xf, modal, alpha)
torch.cuda.synchronize()
