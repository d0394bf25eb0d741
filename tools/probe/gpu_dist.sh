run() { echo "$1: $(env $2 DQ_DIST_FORCE=1 python bench.py --gpus 1 --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | grep -E '^\{"metric"|Error|error' | tail -1 | python -c 'import json,sys; l=sys.stdin.readline(); 
try:
    d=json.loads(l); print("%.2f M/s %.1f us exposed %.1f identical %s" % (d["value"]/1e6, d["ms_per_step"]*1e3, d["allreduce"]["exposed_us_per_step"], d["replicas_identical"]))
except Exception as e: print("FAIL", l[:300])')"; }
for rep in 1 2 3; do
run "native single" "DQ_DIST_MODE=single"
run "torch single" "DQ_DIST_MODE=single DQ_DIST_NATIVE=0"
run "torch split" "DQ_DIST_MODE=split"
done
echo "plain: $(python bench.py --gpus 1 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.1f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"
