#!/bin/bash
# the c3 loop's kernels under rocprofv3 (+ the bench line without the profiler).  tools/probe/r5_loopstats.sh [extra bench args]
tag=r5ls; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/$tag/bench.json
python -c "import json; d=json.load(open('gpurun_out/$tag/bench.json')); print('loop', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$root/gpurun_out/$tag/prof.log" 2>&1)
python tools/rocprof_gaps.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) 3500 | head -10
rm -rf gpurun_out/$tag/prof
