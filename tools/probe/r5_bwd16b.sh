#!/bin/bash
# 16-wave convolutional backward: parity (forced), phase stamps, kernel averages of both forms, loop A/B.  tools/probe/r5_bwd16b.sh [quick]
tag=r5b16; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
DQ_CONV_BWD_FORM=16 timeout 900 python -m pytest tests/test_compact_gpu.py -x -q -m gpu -k "training_forward_backward or wave_private or loop_on" 2>&1 | tail -3
DQ_LIB_PATH=tools/probe/stamps/c16.so timeout 300 python tools/probe/c16_stamps.py 2>&1 | grep -v amdgpu.ids
for form in 16 8; do
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && DQ_CONV_BWD_FORM=$form timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/tools/conv_ab.py" 4096 > "$root/gpurun_out/$tag/ab.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/ab_$form.csv"
grep -E "conv_bwd16|conv_bwd_chain_kernel<2" gpurun_out/$tag/ab_$form.csv | cut -d, -f1,2,4
done
rm -rf gpurun_out/$tag/prof
[ "$1" = quick ] && exit 0
for form in 16 8 16 8; do
DQ_CONV_BWD_FORM=$form python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/$tag/bench_c3_form$form.json
python -c "import json; d=json.load(open('gpurun_out/$tag/bench_c3_form$form.json')); print('c3 loop form $form', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
done
