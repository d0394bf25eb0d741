#!/bin/bash
# where the configurations stand: ratio 32, c5, c2 bench lines + c5's kernels under rocprofv3
tag=r5st; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
line() { grep '^{"metric"' | tail -1; }
show() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], '%.4g' % d['value'], d['unit'], 'ms/step %.4f' % d['ms_per_step'], d['roofline']['kernel'], '%.1f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'])" $1; }
python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line > gpurun_out/$tag/ratio32.json; show gpurun_out/$tag/ratio32.json
python bench.py --config c5 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > gpurun_out/$tag/c5.json; show gpurun_out/$tag/c5.json
python bench.py --config c2 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > gpurun_out/$tag/c2.json; show gpurun_out/$tag/c2.json
python bench.py --mode learn --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > gpurun_out/$tag/learn.json; show gpurun_out/$tag/learn.json
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/bench.py" --config c5 --steps 500 --warmup 50 --no-cpu-baseline > "$root/gpurun_out/$tag/prof.log" 2>&1)
python tools/rocprof_gaps.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) 3500 | head -10
rm -rf gpurun_out/$tag/prof
