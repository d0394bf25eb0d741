mkdir -p gpurun_out/r06n; o=gpurun_out/r06n
python -m pytest tests/test_env_gpu.py tests/test_compact_gpu.py -m gpu -x -q -k "act_steps or full_size or policy or patch_words or trace" > $o/tests.log 2>&1; tail -3 $o/tests.log | cut -c1-250
line() { grep '^{"metric"' | tail -1; }
for T in 32 64; do python bench.py --mode env --env-steps-per-launch $T --steps 1920 --warmup 64 --no-cpu-baseline 2>/dev/null | line > $o/bench_env_T$T.json; done
python bench.py --mode act --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_act.json
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_loop.json
python - <<'PY'
import json
for f in ("env_T32","env_T64","act","loop"):
    d=json.load(open(f"gpurun_out/r06n/bench_{f}.json")); r=d["roofline"] or {}
    print(f, "%.4g"%d["value"], "us/step %.2f"%(1e3*d["ms_per_step"]), r.get("kernel"), "launch %.1f us"%r.get("avg_launch_us",0), "ratio32 %.4g" % ((d.get("reference_replay_ratio") or {}).get("value",0)))
PY
