cd $GRAFT_REPO_ROOT
b() { echo "$1 $2: $(env $2 timeout 200 python bench.py $3 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"; }
for rep in 1 2; do
b c5 A=1 "--config c5"; b c5 DQ_CONV_PERSIST=2 "--config c5"; b c5 DQ_CONV_PERSIST=0 "--config c5"
b c5 DQ_CONV_BWD_S=8 "--config c5"; b c5 DQ_DENSE_BWD_SPLIT=2 "--config c5"
done
DQ_BENCH_FAMILIES=1 timeout 300 python bench.py --config c5 --steps 300 --warmup 50 --no-cpu-baseline 2>&1 | grep -E "per-family" | cut -c1-400
