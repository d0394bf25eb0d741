#!/bin/bash
# ratio 32 (32 updates of 4096 per vector step): where the time between the kernels goes.  tools/probe/r5_ratio.sh [tag]
tag="${1:-r5ratio}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/$tag/bench_ratio32.json
python -c "import json; d=json.load(open('gpurun_out/$tag/bench_ratio32.json')); print('ratio32', d['value'], d['ms_per_step'])"
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --updates-per-step 32 --steps 40 --warmup 3 --no-cpu-baseline > "$root/gpurun_out/$tag/prof.log" 2>&1)
python tools/rocprof_gaps.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) 8000 | tee gpurun_out/$tag/gaps_ratio32.txt
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --steps 600 --warmup 50 --no-cpu-baseline > "$root/gpurun_out/$tag/prof1.log" 2>&1)
python tools/rocprof_gaps.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) 3500 | tee gpurun_out/$tag/gaps_k1.txt
rm -rf gpurun_out/$tag/prof
