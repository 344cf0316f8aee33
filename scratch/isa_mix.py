"""Instruction mix per basic block of one kernel (from hipcc -S): how many VALU / SALU / LDS / VMEM instructions ride along
each MFMA.   python scratch/isa_mix.py <mangled-substring> [asm file]"""
import re, sys, collections
sub = sys.argv[1]; path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cm.s"
txt = open(path).read()
m = re.search(r'\n(_Z\w*' + re.escape(sub) + r'\w*):\s*;.*?\n(.*?)\n\.Lfunc_end', txt, re.S)
name, body = m[1], m[2]
def kind(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op in ('s_waitcnt', 's_barrier', 's_nop'): return op
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
blocks = re.split(r'\n(\.LBB\d+_\d+):', "\n" + body)
tot = collections.Counter()
print(name[:110])
rows = [("entry", blocks[0])] + [(blocks[i], blocks[i + 1]) for i in range(1, len(blocks), 2)]
for lab, b in rows:
    ls = [l.strip() for l in b.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(kind(l.split()[0]) for l in ls)
    tot.update(c)
    if c['mfma']:
        other = sum(v for k, v in c.items() if k != 'mfma')
        print(f"{lab:12s} {len(ls):5d} instrs  mfma {c['mfma']:4d}  valu {c['valu']:4d}  salu {c['salu']:4d}  lds {c['lds']:3d}  vmem {c['vmem']:3d}  "
              f"waitcnt {c['s_waitcnt']:3d}  barrier {c['s_barrier']:2d}  -> {other / c['mfma']:.1f} other per mfma, valu/mfma {c['valu'] / c['mfma']:.1f}")
print("whole kernel:", dict(tot))
