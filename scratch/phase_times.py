"""Stand-alone replay time of each captured phase of the uganConsis iteration (B = 8 + 8 @256^2)."""
import os, sys, types
sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import config as cfg
from smsut_amd.misc.synthetic import SyntheticSliceLoader
from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer
dev = torch.device("cuda"); cfg.batch_size = 8
tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False)); tr.net.train(); tr.D.train(); tr.iter, tr.epoch = 1000, 100
lb = iter(SyntheticSliceLoader(8, device=dev, labeled=True, n_batches=4)); ul = iter(SyntheticSliceLoader(8, device=dev, labeled=False, n_batches=4))
(x1, y1, m1, _), (x2, _, m2, _) = next(lb), next(ul)
x = torch.cat([x1, x2], 0); m = torch.cat([m1, m2], 0).cuda()
for _ in range(4): tr.train_iteration(x, y1, m)
torch.cuda.synchronize()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = 0
for k, g in tr._graphs.items():
    ms = t(g.graph.replay); tot += ms
    print(f"{k[0]:6s} {ms:7.3f} ms")
print("sum of phases", round(tot, 3))
print("D adam ", round(t(tr.d_optimizer.step), 3), " G sgd ", round(t(tr.optimizer.step), 3))
print("iteration", round(t(lambda: tr.train_iteration(x, y1, m)), 3))
# ---- D and G2gen concurrently on two streams (what train_iteration does)
gd = [g for k, g in tr._graphs.items() if k[0] == "D"][0].graph
gg = [g for k, g in tr._graphs.items() if k[0] == "G2gen"][0].graph
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2): gd.replay()
    with torch.cuda.stream(s1): gg.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("D alone %.3f  G2gen alone %.3f  both on two streams %.3f ms" % (t(gd.replay), t(gg.replay), t(both)))

s3 = torch.cuda.Stream()
def both0():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s3.wait_stream(cur)
    with torch.cuda.stream(s3): gd.replay()
    with torch.cuda.stream(s1): gg.replay()
    cur.wait_stream(s1); cur.wait_stream(s3)
def main_side():
    cur = torch.cuda.current_stream()
    s3.wait_stream(cur)
    with torch.cuda.stream(s3): gd.replay()
    gg.replay()
    cur.wait_stream(s3)
def serial2():
    gd.replay(); gg.replay()
print("both prio 0: %.3f   main+side: %.3f   serial same stream: %.3f ms" % (t(both0), t(main_side), t(serial2)))

# ---- the whole iteration issued from a pool stream instead of the default stream (does the D || G2gen overlap work there?)
sm = torch.cuda.Stream()
def it_on_pool():
    sm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sm):
        tr.train_iteration(x, y1, m)
    torch.cuda.current_stream().wait_stream(sm)
print("iteration on the default stream %.3f   on a pool stream %.3f ms" % (t(lambda: tr.train_iteration(x, y1, m)), t(it_on_pool)))

# ---- the four phases back to back: all on the current stream | D on a side stream, JOINED before the next phase (no overlap)
gs = {k[0]: g.graph for k, g in tr._graphs.items()}
sd = torch.cuda.Stream()
def all_main():
    gs["G1"].replay(); gs["D"].replay(); gs["G2gen"].replay(); gs["G2"].replay()
def d_side_joined():
    cur = torch.cuda.current_stream()
    gs["G1"].replay()
    sd.wait_stream(cur)
    with torch.cuda.stream(sd): gs["D"].replay()
    cur.wait_stream(sd)
    gs["G2gen"].replay(); gs["G2"].replay()
def d_g2_side_joined():
    cur = torch.cuda.current_stream()
    gs["G1"].replay()
    sd.wait_stream(cur)
    with torch.cuda.stream(sd): gs["D"].replay()
    cur.wait_stream(sd)
    gs["G2gen"].replay()
    sd.wait_stream(cur)
    with torch.cuda.stream(sd): gs["G2"].replay()
    cur.wait_stream(sd)
print("four phases: all on the current stream %.3f | D on a side stream, joined %.3f | D and G2 on a side stream, joined %.3f ms"
      % (t(all_main), t(d_side_joined), t(d_g2_side_joined)))

# ---- HOST time of the replays (is the iteration launch-bound?)
import time
torch.cuda.synchronize()
for k, g in tr._graphs.items():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.graph.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{k[0]:6s} host time of a replay call {1e3 * (t1 - t0) / 5:7.3f} ms   (5 replays drained after {1e3 * (t2 - t0):7.2f} ms)")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): tr.train_iteration(x, y1, m)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"train_iteration: host {1e3 * (t1 - t0) / 5:.3f} ms per call, wall {1e3 * (t2 - t0) / 5:.3f} ms per iteration")
