"""How much of the uganConsis iteration is the generator alone?  The same iteration with the discriminator's work stubbed out
(D-step and D(x_fake) return zeros; results are WRONG by construction) -- a timing bound, nothing else.
python scratch/g_only.py [steps]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, smsut_amd  # noqa
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
B = 16; cfg.batch_size = B // 2


def run(stub):
    tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
    tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
    if stub:
        z4, z2 = torch.zeros(4, device=dev), torch.zeros(2, device=dev)
        tr._d_phase = lambda x_real, x_fake, modal_org, alpha: z4 + 0.0

        def g2d(modal_trg):
            tr._gx_d = torch.zeros_like(tr._g1[1])
            return z2 + 0.0
        tr._g2d_phase = g2d
    lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=steps + 12))
    ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=steps + 12))
    bs = []
    for _ in range(steps + 10):
        (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
        bs.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)))
    for b in bs[:10]:
        tr.train_iteration(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bs[10:]:
        tr.train_iteration(*b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full = run(False)
gen = run(True)
print(f"full iteration {full:.3f} ms, generator only (D stubbed) {gen:.3f} ms -> the discriminator's share of the wall clock {full - gen:.3f} ms")
