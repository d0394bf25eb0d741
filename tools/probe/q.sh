cd $GRAFT_REPO_ROOT
b() { echo "$1 split=$2: $(DQ_DENSE_BWD_SPLIT=$2 timeout 200 python bench.py $3 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"; }
for rep in 1 2; do
b mb32 0 "--minibatch 32"; b mb32 4 "--minibatch 32"
b mb256 0 "--minibatch 256"; b mb256 4 "--minibatch 256"
done
timeout 600 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -2
