#!/bin/bash
# rocprofv3 kernel statistics of the headline loop (GPU box, repo root): tools/prof_loop.sh <tag> [bench args]  ->  gpurun_out/<tag>_kernel_stats.csv
tag="${1:-prof}"; shift; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out; rm -rf "gpurun_out/$tag.prof"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag.prof" -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$root/gpurun_out/$tag.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/$tag.prof/*/*.db | head -1) "gpurun_out/${tag}_kernel_stats.csv"
python tools/rocprof_shapes.py $(ls gpurun_out/$tag.prof/*/*.db | head -1) | head -16 > "gpurun_out/${tag}_shapes.txt"
rm -rf "gpurun_out/$tag.prof"
head -14 "gpurun_out/${tag}_kernel_stats.csv" | cut -c1-150
grep '^{"metric"' "gpurun_out/$tag.log" | tail -1 | cut -c1-400
