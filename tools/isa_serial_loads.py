"""Development aid: flags places where hipcc serialised global loads (a load followed within a few instructions by
s_waitcnt vmcnt(0), several times in a row) -- the pattern behind two 9-10K-cycle stalls found with the phase stamps.
    python tools/isa_serial_loads.py deepq-decoding_amd/csrc/fused.hip [...]"""
import re, subprocess, sys, tempfile, os

for src in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        kernel, lines = None, open(out).read().splitlines()
    runs = {}
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if kernel and re.search(r"\b(global|flat|scratch)_load", l):
            # is there a vmcnt(0) within the next 4 instructions, before another load?
            for k in range(1, 5):
                if i + k >= len(lines):
                    break
                n = lines[i + k]
                if re.search(r"_load", n):
                    break
                if "s_waitcnt vmcnt(0)" in n:
                    runs.setdefault(kernel, []).append(i + 1)
                    break
        i += 1
    for k, v in runs.items():
        # report clusters of >= 4 such loads within 60 lines
        cl, start = 1, v[0]
        for a, b in zip(v, v[1:]):
            if b - a < 20:
                cl += 1
            else:
                if cl >= 4:
                    print(f"{src}: {k[:50]}: {cl} serialised loads near asm line {start}")
                cl, start = 1, b
        if cl >= 4:
            print(f"{src}: {k[:50]}: {cl} serialised loads near asm line {start}")
