mkdir -p gpurun_out/r06f; o=gpurun_out/r06f
python -m pytest tests/test_qnet_gpu.py tests/test_compact_gpu.py tests/test_agent_gpu.py -m gpu -x -q -s > $o/tests.log 2>&1; tail -5 $o/tests.log; grep "ReLU pre-activation" $o/tests.log > $o/fragile.txt; cat $o/fragile.txt | head -14
line() { grep '^{"metric"' | tail -1; }
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench.json
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench2.json
python - <<'PY'
import json
for f in ["bench","bench2"]:
    d=json.load(open(f"gpurun_out/r06f/{f}.json")); r=d["roofline"]; q=d.get("reference_replay_ratio") or {}
    print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], r["kernel"], "%.2f"%r["avg_launch_us"], "ratio32 %.4g"%q.get("value",0))
PY
