mkdir -p gpurun_out/r06c; o=gpurun_out/r06c
python -m pytest tests/test_compact_gpu.py tests/test_qnet_gpu.py -m gpu -x -q > $o/tests_conv.log 2>&1; tail -3 $o/tests_conv.log
line() { grep '^{"metric"' | tail -1; }
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_recompute.json
DQ_CONV_BWD_A1=saved python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_saved.json
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_recompute2.json
DQ_CONV_BWD_A1=saved python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_saved2.json
for v in recompute saved; do
  rm -rf $o/prof
  (cd /tmp && export TMPDIR=/tmp && DQ_CONV_BWD_A1=$v rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --ratio-steps 0 > "$GRAFT_REPO_ROOT/$o/prof_$v.log" 2>&1)
  python tools/rocprof_shapes.py $(ls $o/prof/*/*.db | head -1) 500 > $o/shapes_$v.txt
done
rm -rf $o/prof
python - <<'PY'
import json
for f in ["recompute","saved","recompute2","saved2"]:
    d=json.load(open(f"gpurun_out/r06c/bench_{f}.json")); r=d["roofline"]; q=d.get("reference_replay_ratio") or {}
    print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], r["kernel"], "%.2f"%r["avg_launch_us"], "ratio32 %.4g"%q.get("value",0))
PY
grep -E "conv_wave|conv_bwd16|dense|reduce|pack" $o/shapes_recompute.txt | head -12; echo; grep -E "conv_wave|conv_bwd16" $o/shapes_saved.txt | head
