#!/bin/bash
# Usage: tools/kernel_resources.sh deepq-decoding_amd/csrc/fused.hip   -> one line per kernel: VGPR / AGPR / SGPR / spills / occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    print(k[:60].ljust(60), " ".join(f"{a}={b}" for a, b in v.items() if a in ("VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy", "LDS Size")))
'
rm -f /tmp/kr_$$.o
