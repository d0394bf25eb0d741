#!/bin/bash
# conv forward kernels alone at several minibatch sizes (fixed cost against per-sample cost): tools/probe/r5_scale.sh <tag>
tag="${1:-r5s}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
for b in 512 1024 2048 4096 8192 16384; do
  rm -rf gpurun_out/$tag/prof
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag/prof" -- python "$root/tools/conv_ab.py" $b > "$root/gpurun_out/$tag/ab_$b.log" 2>&1)
  python tools/rocprof_summary.py $(ls gpurun_out/$tag/prof/*/*.db | head -1) "gpurun_out/$tag/ab_$b.csv"
  echo "batch $b:" $(grep -E "conv_wave|conv_chain_pkernel<4>|conv_chain_kernel<4>" gpurun_out/$tag/ab_$b.csv | cut -d, -f1,4 | tr '\n' ' ')
done
rm -rf gpurun_out/$tag/prof
