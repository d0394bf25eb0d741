#!/bin/bash
# one-box check of the 16-wave convolutional backward: parity tests with it forced, then both forms' kernel averages under rocprofv3 (tools/conv_ab.py), then the loop
tag="${1:-r5b16}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
DQ_CONV_BWD_FORM=16 timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -3
for form in 16 8; do
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && DQ_CONV_BWD_FORM=$form timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/tools/conv_ab.py" 4096 > "$root/gpurun_out/$tag/ab.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/ab_$form.csv"
grep -E "conv_bwd|conv_wave" gpurun_out/$tag/ab_$form.csv | cut -d, -f1,2,4
done
rm -rf gpurun_out/$tag/prof
for form in 16 8 16 8; do
DQ_CONV_BWD_FORM=$form python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/$tag/bench_c3_form$form.json
python -c "import json; d=json.load(open('gpurun_out/$tag/bench_c3_form$form.json')); print('c3 loop form $form', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
done
