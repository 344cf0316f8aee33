"""Spatial split of the chip between the generator's stream and the discriminator's side stream (hipExtStreamCreateWithCUMask):
the side stream's small kernels hold CUs that the generator's one-round persistent grids were sized to own -- a workgroup that waits
for such a CU holds its whole launch back (scratch/g_only.py: the discriminator costs 2.6 ms of wall clock although it is fully
overlapped).  Here the side stream gets K CUs (K/8 per XCD), the generator's stream the rest, and the generator's grids are sized
for the rest (SMSUT_CUS, SMSUT_RR_TARGET*).  python scratch/cu_mask_probe.py <K> [steps]"""
import ctypes, os, sys, time, types
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
TOTAL = 256
if K > 0:
    os.environ["SMSUT_CUS"] = str(TOTAL - K)
    os.environ["SMSUT_RR_TARGET4"] = str(TOTAL - K)
    os.environ["SMSUT_RR_TARGET"] = str(2 * (TOTAL - K))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, smsut_amd  # noqa
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer

dev = torch.device("cuda")
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


B = 16; cfg.batch_size = B // 2
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
main = torch.cuda.current_stream()
if K > 0:
    tr._side = masked_stream(range(0, K))
    main = masked_stream(range(K, TOTAL))
lb = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=True, n_batches=steps + 12))
ul = iter(SyntheticSliceLoader(B // 2, device=dev, labeled=False, n_batches=steps + 12))
bs = []
for _ in range(steps + 10):
    (x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
    bs.append((torch.cat([x1, x2], 0), y1, torch.cat([m1, m2], 0)))
torch.cuda.synchronize()
with torch.cuda.stream(main):
    for b in bs[:10]:
        tr.train_iteration(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bs[10:]:
        last = tr.train_iteration(*b)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"side stream on {K} CUs, generator on {TOTAL - K}: {ms:.3f} ms per iteration, graph {tr.graph_report()['mode']}, scalars finite "
      f"{bool(torch.isfinite(last).all())}")
