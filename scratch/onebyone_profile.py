"""1x1 / ConvT / thin-head calls of one uganConsis iteration: time, TFLOP/s and algorithmic GB/s per shape."""
import os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg, profiling, _hip as H
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=6)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=6))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
step = lambda: tr.train_iteration(x, y1, m)
for _ in range(2): step()
rows = profiling.replay(profiling.record_step(step))
sel = [r for r in rows if any(k in r.name for k in ("conv1x1", "convT", "thin"))]
tot = sum(r.total_us for r in sel)
print(f"1x1 / ConvT / thin calls: {tot/1e3:.2f} ms of {sum(r.total_us for r in rows)/1e3:.2f} ms")
for r in sorted(sel, key=lambda r: -r.total_us)[:24]:
    sig = H.SIGNATURES[r.name].replace(" ", "")
    a = list(r.args)
    # bytes: n * hw * (cin + cout) * 4 for the 1x1 forms (first ints are [ca,] n, hw, k, nd ...)
    off = 1 if ("_cat" in r.name or "_split" in r.name) else 0
    if "convT" in r.name:
        n, h, w, ci, co = a[0:5]; byts = 4.0 * n * h * w * (ci + 4 * co)
    else:
        n, hw, k, nd = a[off:off + 4]; byts = 4.0 * n * hw * (k + nd)
    print(f"{r.total_us/1e3:6.3f} ms {r.calls:3d} x {r.us:6.1f} us {r.flops/(r.us*1e-6)/1e12:5.1f} TF {byts/(r.us*1e-6)/1e12:5.2f} TB/s  {r.name.replace('smsut_','')} {r.args}")
