python -m pytest tests/test_qnet_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for v in 1 0; do
echo "persist=$v: $(DQ_CONV_PERSIST=$v python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us conv %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))')"
done; done
