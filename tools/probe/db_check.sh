cd $GRAFT_REPO_ROOT
echo "== dense bwd in loop"; DQ_LIB_PATH=tools/probe/stamps/s3.so python tools/stamp_loop.py 3 2>&1 | tail -16
echo "== dense bwd wg timeline"; DQ_LIB_PATH=tools/probe/stamps/s23.so python tools/stamp_loop.py 23 2>&1 | tail -4
python -m pytest tests/test_qnet_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))'; done
