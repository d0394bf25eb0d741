"""Static instruction mix of a kernel's loops: python tools/isa_mix.py <file.hip> <kernel name substring>  (hipcc -S, device only)."""
import collections, subprocess, sys
src, name = sys.argv[1], sys.argv[2]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", "/tmp/isa_mix.s"] + sys.argv[3:],
                     capture_output=True, text=True); asm = open("/tmp/isa_mix.s").read().split("\n")
start = [i for i, l in enumerate(asm) if l.startswith("_Z") and name in l and l.split(";")[0].rstrip().endswith(":")][0]
end = [i for i, l in enumerate(asm) if i > start and l.startswith(".Lfunc_end")][0]
inloop, cnt, cat = False, collections.Counter(), collections.Counter()
tot = collections.Counter()
for l in asm[start:end]:
    s = l.strip()
    if "in Loop" in s or "Loop Header" in s: inloop = True
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    op = s.split()[0]
    c = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "nop" if op == "s_nop" else
         "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem")
    tot[c] += 1
    if inloop: cnt[op] += 1; cat[c] += 1
print("whole kernel:", dict(tot))
print("inside loops:", dict(cat))
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:40]: print("  %-28s %d" % (k, v))
