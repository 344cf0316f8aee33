"""Whole-step in-process A/B of two builds of libsmsut_hip.so (eager steps, the active library switched every few steps).
   python scratch/step_ab.py libA.so libB.so [ugan|unet]"""
import ctypes, os, sys, types
os.environ["SMSUT_GRAPH"] = "0"
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import _hip as H, config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader

def bind(path):
    lib = ctypes.CDLL(path)
    for name, sig in H.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = [H._CT[c] for c in sig.replace(" ", "")]
        fn.restype = ctypes.c_int64 if name in H._RET_I64 else ctypes.c_int
    return lib
libs = {"A": bind(sys.argv[1]), "B": bind(sys.argv[2])}
wl = sys.argv[3] if len(sys.argv) > 3 else "ugan"
H._lib = libs["A"]
ns = types.SimpleNamespace(fold=0, expr_name=None, write_env=False)
dev = torch.device("cuda")
if wl == "ugan":
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
    B = 16; cfg.batch_size = B // 2
    tr = UGANConsisTrainer("train", ns); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
    lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=8)); ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=8))
    batches = []
    for _ in range(8):
        (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
        batches.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0).cuda()))
    cnt = [0]
    def step():
        x, y, m = batches[cnt[0] % 8]; cnt[0] += 1
        tr.train_iteration(x, y, m)
else:
    from smsut_amd.trainer.unetTrainer import UnetTrainer
    B = 32; cfg.batch_size = B
    tr = UnetTrainer("train", ns); tr.net.train()
    ld = iter(SyntheticSliceLoader(B, device=dev, n_batches=8))
    batches = [next(ld)[:2] for _ in range(8)]
    cnt = [0]
    def step():
        img, msk = batches[cnt[0] % 8]; cnt[0] += 1
        tr.train_step(img, msk)
for _ in range(3): step()
res = {"A": [], "B": []}
for rnd in range(6):
    for k in ("A", "B"):
        H._lib = libs[k]
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): step()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 4)
print(wl, ' '.join(f'{k}: min {min(v):.2f} med {sorted(v)[len(v)//2]:.2f} ms' for k, v in res.items()), flush=True)
