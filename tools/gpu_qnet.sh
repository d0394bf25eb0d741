#!/bin/bash
# Usage (GPU box): tools/gpu_qnet.sh [config] [batch]  -> qnet parity tests, micro-benchmark, per-kernel rocprof stats
cfg=${1:-c3}; batch=${2:-4096}
root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_qnet_gpu.py tests/test_agent_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/qnet_bench.py $cfg $batch 50 2>&1 | grep -E "fused\]|fused vs|Error|error" 
rm -rf gpurun_out/prof_qb
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/prof_qb" -- python "$root/tools/qnet_bench.py" $cfg $batch 30 > "$root/gpurun_out/prof_qb.log" 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/prof_qb/*/*.db | head -1) gpurun_out/prof_qb_stats.csv
grep -E "chain|wgrad_kernel\(|reduce_slices" gpurun_out/prof_qb_stats.csv | cut -c1-110
