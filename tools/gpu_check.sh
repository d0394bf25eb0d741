#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_check.sh [tests|bench|prof|all] [extra bench args]
# Writes results under gpurun_out/.
mode=${1:-all}; shift
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
if [[ $mode == tests || $mode == all ]]; then
  python -m pytest tests -m gpu -x -q 2>&1 | tail -15
fi
if [[ $mode == bench || $mode == all ]]; then
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -2
fi
if [[ $mode == prof || $mode == all ]]; then
  rm -rf gpurun_out/prof_loop
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_loop" -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" > "$OLDPWD/gpurun_out/prof_loop.log" 2>&1)
  python tools/rocprof_summary.py $(ls gpurun_out/prof_loop/*/*.db | head -1) gpurun_out/prof_loop_stats.csv
  head -14 gpurun_out/prof_loop_stats.csv | cut -c1-150
  python tools/rocprof_shapes.py $(ls gpurun_out/prof_loop/*/*.db | head -1) | head -40
fi
