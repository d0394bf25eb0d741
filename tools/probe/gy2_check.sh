cd $GRAFT_REPO_ROOT
python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py tests/test_env_gpu.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_agent_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== dense bwd in loop"; DQ_LIB_PATH=tools/probe/stamps/s3.so python tools/stamp_loop.py 3 2>&1 | tail -16
echo "== env rider phases"; DQ_LIB_PATH=tools/probe/stamps/s6.so python tools/stamp_loop.py 6 2>&1 | tail -10
echo "== dense bwd wg timeline"; DQ_LIB_PATH=tools/probe/stamps/s23.so python tools/stamp_loop.py 23 2>&1 | tail -4
bash tools/ab_run.sh gy2ab base
for m in act env; do python bench.py --mode $m --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], "%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))' $m; done
