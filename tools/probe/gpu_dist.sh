python -m pytest tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -2
run() { echo "$1: $(env $2 DQ_DIST_FORCE=1 python bench.py --gpus 1 --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | grep -E '^\{"metric"' | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.1f us exposed %.1f identical %s" % (d["value"]/1e6, d["ms_per_step"]*1e3, d["allreduce"]["exposed_us_per_step"], d["replicas_identical"]))')"; }
for rep in 1 2; do run "native single" "DQ_DIST_MODE=single"; done
echo "plain: $(python bench.py --gpus 1 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.2f M/s %.1f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')"
