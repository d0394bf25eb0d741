cd $GRAFT_REPO_ROOT
python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_agent_gpu.py -x -q -m gpu -k "loop or oracle or c3 or c1 or c2" 2>&1 | tail -2
echo "== dense fwd WG timeline (loop shape)"; DQ_LIB_PATH=tools/probe/stamps/s12.so python tools/probe/dense_fwd_timeline.py 2>&1 | tail -6
for v in 1 0 1 0; do export DQ_DROP_AHEAD=$v; python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], "%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))' "ahead=$v"; done
unset DQ_DROP_AHEAD
bash tools/ab_run.sh dropab base
