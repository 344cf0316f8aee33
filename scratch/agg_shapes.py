"""Aggregate a per-shape dump (SMSUT_PROFILE_DUMP of bench.py) by entry point: total ms, calls."""
import re, sys, collections
tot = collections.Counter(); cnt = collections.Counter()
for l in open(sys.argv[1]):
    m = re.match(r'lost\s+(-?[\d.]+) ms\s+total\s+([\d.]+) ms\s+[\d.]+%\s+(\d+) x\s+([\d.]+) us\s+(?:([\d.]+) TF)?\s+(\w+) \((.*)\)', l)
    if m:
        tot[m.group(6)] += float(m.group(2)); cnt[m.group(6)] += int(m.group(3))
for k, v in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print(f"{v:7.3f} ms {cnt[k]:4d}  {k}")
print(f"{sum(tot.values()):7.3f} ms total, {sum(cnt.values())} calls")
