"""Per-call replay profile of one training step: every C-ABI call of the step is recorded (entry point + converted
arguments), then each distinct (entry point, integer arguments) is replayed alone 8x between HIP events on the pointers
it ran with (still owned by torch's caching allocator).  Gives time per SHAPE, which rocprofv3's per-kernel-name average
hides.   python scratch/replay_prof.py [ugan|unet] [min_share_pct]"""
import os, sys, types, collections
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H, config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
wl = sys.argv[1] if len(sys.argv) > 1 else "unet"
minpct = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False); dev = torch.device("cuda")
if wl == "ugan":
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    B = 16; cfg.batch_size = B // 2
    tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
    lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=4))
    (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
    bx, by, bm = torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0).cuda()
    step = lambda: tr.train_iteration(bx, by, bm)
else:
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    B = 32; cfg.batch_size = B
    tr = UnetTrainer("train", ns); tr.net.train()
    img, msk = next(iter(SyntheticSliceLoader(B, device=dev, n_batches=2)))[:2]
    step = lambda: tr.train_step(img, msk)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): step()
e1.record(); torch.cuda.synchronize()
step_ms = e0.elapsed_time(e1) / 4
lib = H.load()
rec = []
orig = H.call
def spy(name, *args):
    conv = [H.ptr(a) if isinstance(a, torch.Tensor) or a is None else a for a in args]
    if name not in H._NO_STATUS:
        rec.append((name, conv))
    return orig(name, *args)
H.call = spy; ops.H.call = spy
for m in list(sys.modules.values()):
    if getattr(m, "__name__", "").startswith("smsut") and getattr(m, "H", None) is H: pass
step(); torch.cuda.synchronize()
H.call = orig
print(f"{wl}: step {step_ms:.2f} ms, {len(rec)} C-ABI calls recorded", flush=True)
groups = collections.OrderedDict()
for name, conv in rec:
    # pointers are > 2^32; integer shape arguments are small
    key = (name, tuple(a for a in conv if isinstance(a, int) and 0 <= a < (1 << 31)) + tuple(a for a in conv if isinstance(a, float)))
    groups.setdefault(key, []).append(conv)
rows = []
for (name, shp), calls in groups.items():
    conv = calls[0]
    fn = getattr(lib, name)
    for _ in range(2): fn(*conv)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(8): fn(*conv)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 8 * 1e3
    rows.append((us * len(calls), name, shp, len(calls), us))
tot = sum(r[0] for r in rows)
print(f"replayed kernel time {tot/1e3:.2f} ms in {len(rows)} distinct calls")
byname = collections.Counter()
for t, name, *_ in rows: byname[name] += t
print("--- by entry point")
for name, t in byname.most_common(25): print(f"{t/1e3:7.3f} ms {100*t/tot:5.1f}%  {name}")
print("--- by shape")
for t, name, shp, n, us in sorted(rows, reverse=True):
    if 100 * t / tot < minpct: break
    print(f"{t/1e3:7.3f} ms {100*t/tot:5.1f}%  {n:3d} x {us:7.1f} us  {name.replace('smsut_', '')} {shp}")
