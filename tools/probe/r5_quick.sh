#!/bin/bash
# quick one-box check of the wave-private conv forward: parity tests of the patch-word forward, then the kernel's average under rocprofv3 (tools/conv_ab.py)
tag="${1:-r5q}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_compact_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
rm -rf gpurun_out/$tag/prof
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/tools/conv_ab.py" 4096 > "$root/gpurun_out/$tag/ab.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/ab.csv"
grep -E "conv_wave|conv_chain_pkernel<0>" gpurun_out/$tag/ab.csv | cut -d, -f1,2,4
done
rm -rf gpurun_out/$tag/prof
