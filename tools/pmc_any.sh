#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_any.sh "<counters>" <outname> <python script + args>   -> per-kernel averages of the counters
# over the LAST $PMC_LAST launches of every kernel (default 40 = the timed steps of `bench.py --steps 40`, as tools/pmc_traffic.sh does: the set-up's launches of
# the same kernels -- 256 one-job ring-fill forwards, resets -- come first and would dilute a whole-run average 2.5x; PMC_LAST=0 averages everything)
ctrs="$1"; out="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$PWD}"
rm -rf "$root/gpurun_out/$out"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctrs --kernel-trace -d "$root/gpurun_out/$out" -o pmc --output-format csv -- python "$root/$1" "${@:2}" > "$root/gpurun_out/$out.log" 2>&1)
python3 - "$root/gpurun_out/$out" <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
last = int(os.environ.get("PMC_LAST", "40"))
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0][:40]
        rows[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
agg = collections.defaultdict(dict)
for (k, c), v in rows.items():
    v.sort()
    v = v[-last:] if last > 0 else v
    agg[k][c] = (sum(x for _, x in v) / len(v), len(v))
for k, v in sorted(agg.items()):
    if "conv" in k or "dense" in k or "reduce" in k or "pack" in k or "env" in k:
        print(k.ljust(42), "  ".join(f"{c}={x:.4g}" for c, (x, n) in sorted(v.items())), f"  (last {max(n for _, n in v.values())} launches)")
PY
