#!/bin/bash
# One-box A/B of library variants (tools/build_ab.sh): tools/ab_run.sh <tag> <variant> [<variant> ...]   ("base" = the regular build)
# Each variant: rocprofv3 kernel stats of the headline loop (top 5 kernels) + the un-profiled bench value, variants interleaved twice.
tag="$1"; shift; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset DQ_LIB_PATH; else export DQ_LIB_PATH="$root/tools/probe/ab/$v.so"; fi
  val=$(python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/${v}_$rep.csv"
  echo "$v rep $rep: $val |" $(python - "gpurun_out/$tag/${v}_$rep.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
short = {"conv_chain": "cf", "conv_bwd": "cb", "dense_chain": "df", "dense_bwd": "db", "dense_wgrad": "dw", "reduce_slices": "rs", "pack_weights": "pk"}
out = []
for r in rows[:8]:
    for k, s in short.items():
        if k in r["kernel"]: out.append("%s %.1f" % (s, float(r["avg_us"])))
print(" ".join(out))
PY
)
done; done
rm -rf gpurun_out/$tag/prof
