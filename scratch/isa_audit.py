"""ISA resource audit of every kernel (no GPU needed): compiles csrc/*.hip with --cuda-device-only -S and lists, per
kernel, fixed LDS bytes, scratch bytes, VGPR / AGPR counts and the waves per SIMD they allow.  Flags the two accidents
that cost r01 the most: accumulators promoted to LDS (fixed LDS larger than declared) and prefetch registers in scratch.
   python scratch/isa_audit.py [substring-of-kernel-name]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "smsut-medicalimgsegmentation_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
tmp = tempfile.mkdtemp(prefix="isa_")
procs = []
for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
                                         "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, f],
                                        stderr=subprocess.DEVNULL)))
rows = []
for out, p in procs:
    if p.wait() != 0:
        print("compile failed:", out); continue
    txt = open(out).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        body = m[2]
        g = lambda k: int(re.search(k + r" (\d+)", body)[1])
        lds, scr, nv = g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("next_free_vgpr")
        acc = re.search(r"accum_offset (\d+)", body)
        name = subprocess.run(["c++filt", m[1]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if flt in name:
            rows.append((os.path.basename(out), name, lds, scr, nv, int(acc[1]) if acc else nv))
print(f"{'file':14s} {'kernel':62s} {'LDS':>6s} {'scr':>5s} {'regs':>5s} {'waves/SIMD':>10s}")
for f, name, lds, scr, nv, acc in rows:
    waves = min(8, 512 // max(nv, 1))
    flag = "  <-- SCRATCH" if scr else ""
    print(f"{f:14s} {name[:62]:62s} {lds:6d} {scr:5d} {nv:5d} {waves:10d}{flag}")
