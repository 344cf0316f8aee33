"""Timeline of a graph-replayed bench run from a rocprofv3 --kernel-trace CSV: per queue busy time, gaps between consecutive
dispatches, overlap between queues -- where the wall time of an iteration goes besides kernel time.
usage: python scratch/timeline.py <kernel_trace.csv> [steps_from_end=3] [ms_per_step_hint]"""
import csv, sys, collections, re
path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
# the steady region: last N launches of the per-step marker kernel (multi_tensor SGD step = once per iteration on the main queue)
mark = [i for i, r in enumerate(rows) if "k_sgd_multi" in r[3]]          # r05: the one-launch SGD step, once per iteration
if not mark:
    mark = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[3] and "FusedSgd" in r[3]]
if not mark:
    mark = [i for i, r in enumerate(rows) if "FusedSgd" in r[3] or "fused_sgd" in r[3].lower()]
print("rows", len(rows), "sgd marks", len(mark))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# group marks: several multi_tensor launches per step may exist -> take distinct steps by gaps > 5 ms
steps = [mark[0]]
for i in mark[1:]:
    if rows[i][0] - rows[steps[-1]][0] > 5e6:
        steps.append(i)
print("steps found", len(steps))
a, b = steps[-n - 1], steps[-1]
seg = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
wall = (t1 - t0) / 1e6
print(f"region: {n} steps, wall {wall:.3f} ms = {wall / n:.3f} ms/step, {len(seg) / n:.0f} dispatches/step")
byq = collections.defaultdict(list)
for r in seg:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, *_ in rs) / 1e6
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 20000]
    big = [g for g in pos if g >= 20000]
    print(f"queue {q}: {len(rs) / n:.0f} dispatches/step, busy {busy / n:.3f} ms/step, gaps<20us: {len(small) / n:.0f}/step sum {sum(small) / 1e6 / n:.3f} ms "
          f"(median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us), gaps>=20us: {len(big) / n:.0f}/step sum {sum(big) / 1e6 / n:.3f} ms")
# union busy (any queue active) and both-active overlap
ev = []
for s, e, *_ in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
act = 0; last = ev[0][0]; any_busy = 0; multi = 0
for t, d in ev:
    if act >= 1: any_busy += t - last
    if act >= 2: multi += t - last
    act += d; last = t
print(f"any queue busy {any_busy / 1e6 / n:.3f} ms/step, >=2 kernels in flight {multi / 1e6 / n:.3f} ms/step, idle {(t1 - t0 - any_busy) / 1e6 / n:.3f} ms/step")
def key(name):
    m = re.search(r"(?:anonymous namespace\)::)?(\w+)(<[^(]*>)?\(", name.replace("void ", ""))
    base = name.replace("void ", "").replace("(anonymous namespace)::", "")
    base = base.split("(")[0]
    return base[:110]
# small launches on the busiest queue
q0 = max(byq.items(), key=lambda kv: sum(e - s for s, e, *_ in kv[1]))[0]
for q, rs in byq.items():
    cnt = collections.Counter(); tm = collections.Counter(); small = collections.Counter()
    for s_, e, q_, name, wgs in rs:
        k = key(name); cnt[k] += 1; tm[k] += e - s_
    print(f"\nqueue {q}: top kernels by time")
    for k, t in tm.most_common(45):
        print(f"  {cnt[k] / n:6.1f}/step  {t / 1e6 / n:7.3f} ms/step  avg {t / cnt[k] / 1e3:6.2f} us  {k}")
