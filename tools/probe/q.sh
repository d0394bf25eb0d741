cd $GRAFT_REPO_ROOT
b() { echo "$1 S=$2: $(DQ_CONV_BWD_S=$2 timeout 200 python bench.py $3 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"; }
for rep in 1 2; do
for S in 8 4 2 1; do b mb32 $S "--minibatch 32"; done
for S in 8 4 2; do b c5 $S "--config c5"; done
for S in 8 4; do b mb1024 $S "--minibatch 1024"; done
done
DQ_CONV_BWD_S=0 timeout 900 python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | tail -2
DQ_CONV_BWD_S=1 timeout 900 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -2
