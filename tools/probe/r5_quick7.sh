#!/bin/bash
# d = 7 (c5) check of the wave-private conv forward: parity tests, then the kernels alone under rocprofv3 at c5's per-GPU minibatch (DQ_AB_CFG=c5 tools/conv_ab.py 1024)
tag="${1:-r5q7}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_compact_gpu.py -x -q -m gpu 2>&1 | tail -5
for form in wave group; do
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && DQ_CONV_FORM=$form DQ_AB_CFG=c5 timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/tools/conv_ab.py" ${CW_B:-1024} > "$root/gpurun_out/$tag/ab.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/ab_$form.csv"
echo "== $form"; grep -E "conv_wave|conv_chain_pkernel<0>|conv_chain_kernel<0>" gpurun_out/$tag/ab_$form.csv | cut -d, -f1,2,4
done
rm -rf gpurun_out/$tag/prof
