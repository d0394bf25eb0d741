#!/bin/bash
# the extra updates' target forwards on a side stream (DQNCore.target_ahead) against the one-stream order, same box.  tools/probe/r5_ahead.sh [tag]
tag="${1:-r5ahead}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_agent_gpu.py -x -q -m gpu -k "side_stream or three_updates" 2>&1 | tail -3
for rep in 1 2; do for ah in 1 0; do
DQ_TARGET_AHEAD=$ah python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/$tag/bench_ratio32_ahead$ah.json
python -c "import json; d=json.load(open('gpurun_out/$tag/bench_ratio32_ahead$ah.json')); print('ratio32 ahead=$ah', d['value'], d['ms_per_step'])"
done; done
for ah in 1 0; do
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && DQ_TARGET_AHEAD=$ah timeout 600 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --updates-per-step 32 --steps 40 --warmup 3 --no-cpu-baseline > "$root/gpurun_out/$tag/prof.log" 2>&1)
python tools/rocprof_gaps.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) 8000 | tee gpurun_out/$tag/gaps_ratio32_ahead$ah.txt
done
rm -rf gpurun_out/$tag/prof
