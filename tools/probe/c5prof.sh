cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c5prof
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c5prof" -- python "$GRAFT_REPO_ROOT/bench.py" --config c5 --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/c5prof/*/*.db | head -1) gpurun_out/c5_kernel_stats.csv
head -12 gpurun_out/c5_kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/c5prof
