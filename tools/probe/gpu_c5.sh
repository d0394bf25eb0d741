root=$PWD; rm -rf gpurun_out/c5prof
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/c5prof" -- python "$root/bench.py" --config c5 --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/c5prof/*/*.db | head -1) gpurun_out/c5_kernel_stats.csv
rm -rf gpurun_out/c5prof
head -10 gpurun_out/c5_kernel_stats.csv | cut -c1-110
