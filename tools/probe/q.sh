cd $GRAFT_REPO_ROOT
b() { echo "$1 slices=$2: $(DQ_WGRAD_SLICES=$2 timeout 200 python bench.py $3 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"; }
for rep in 1 2; do
for S in 10 8 4 3 2 0; do b c5 $S "--config c5"; done
for S in 10 0; do b c3 $S ""; done
done
