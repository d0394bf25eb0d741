mkdir -p gpurun_out/r06j; o=gpurun_out/r06j
python -m pytest tests/test_compact_gpu.py tests/test_qnet_gpu.py tests/test_agent_gpu.py -m gpu -x -q > $o/tests.log 2>&1; tail -3 $o/tests.log | cut -c1-250
line() { grep '^{"metric"' | tail -1; }
for i in 1 2; do
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_planes$i.json
DQ_X_PLANES=0 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_f32_$i.json
done
for v in 1 0; do
  rm -rf $o/prof
  (cd /tmp && export TMPDIR=/tmp && DQ_X_PLANES=$v rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --ratio-steps 0 > "$GRAFT_REPO_ROOT/$o/prof_$v.log" 2>&1)
  python tools/rocprof_shapes.py $(ls $o/prof/*/*.db | head -1) 500 > $o/shapes_xp$v.txt
done
rm -rf $o/prof
python - <<'PY'
import json
for f in ["planes1","f32_1","planes2","f32_2"]:
    d=json.load(open(f"gpurun_out/r06j/bench_{f}.json")); r=d["roofline"]; q=d.get("reference_replay_ratio") or {}
    print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], r["kernel"], "%.2f"%r["avg_launch_us"], "ratio32 %.4g"%q.get("value",0))
PY
for v in 1 0; do echo xplanes=$v; grep "last 500" -A8 $o/shapes_xp$v.txt | cut -c1-140; done
