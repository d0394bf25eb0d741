cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py -x -q -m gpu 2>&1 | tail -3
DQ_LIB_PATH=tools/probe/stamps/s15.so timeout 200 python tools/stamp_run.py 15 2>&1 | grep -v amdgpu.ids | tail -9
for rep in 1 2; do for v in 1 0; do
echo "DQ_WGRAD_PC=$v: $(if [ $v = nopf ]; then export DQ_LIB_PATH=$PWD/tools/probe/ab/nopf.so; else unset DQ_LIB_PATH; fi; DQ_WGRAD_PC=$v timeout 200 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"
done; done
for v in 1 0; do
rm -rf gpurun_out/pc_prof; (cd /tmp && export TMPDIR=/tmp && DQ_WGRAD_PC=$v timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pc_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(ls gpurun_out/pc_prof/*/*.db | head -1) gpurun_out/pc_$v.csv; echo "PC=$v"; grep -i "wgrad\|reduce" gpurun_out/pc_$v.csv | cut -c1-100
done; rm -rf gpurun_out/pc_prof
