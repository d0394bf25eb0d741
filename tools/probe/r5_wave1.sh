#!/bin/bash
# round 5, first GPU run of the wave-private conv forward: parity tests, then the kernels alone under rocprofv3 (both forms)
root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_compact_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5d/test_compact.txt
cat gpurun_out/r5d/test_compact.txt
for form in wave group; do
  rm -rf gpurun_out/r5d/prof
  (cd /tmp && export TMPDIR=/tmp && DQ_CONV_FORM=$form timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/r5d/prof" -- python "$root/tools/conv_ab.py" 4096 > "$root/gpurun_out/r5d/ab_$form.log" 2>&1)
  python tools/rocprof_summary.py $(ls gpurun_out/r5d/prof/*/*.db | head -1) "gpurun_out/r5d/ab_$form.csv"
  echo "== $form"; head -8 gpurun_out/r5d/ab_$form.csv | cut -c1-160; tail -4 gpurun_out/r5d/ab_$form.log
done
rm -rf gpurun_out/r5d/prof
