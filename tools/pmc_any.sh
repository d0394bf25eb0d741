#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_any.sh "<counters>" <outname> <python script + args>   -> per-kernel averages of the counters
ctrs="$1"; out="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$PWD}"
rm -rf "$root/gpurun_out/$out"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctrs --kernel-trace -d "$root/gpurun_out/$out" -o pmc --output-format csv -- python "$root/$1" "${@:2}" > "$root/gpurun_out/$out.log" 2>&1)
python3 - "$root/gpurun_out/$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if "conv" in k or "dense" in k:
        print(k.ljust(42), "  ".join(f"{c}={x / max(1, cnt[(k, c)]):.4g}" for c, x in sorted(v.items())))
PY
