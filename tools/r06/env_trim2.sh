mkdir -p gpurun_out/r06n; o=gpurun_out/r06n
python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "act_steps or full_size" > $o/tests.log 2>&1; tail -2 $o/tests.log | cut -c1-250
line() { grep '^{"metric"' | tail -1; }
for T in 32 64 128 256; do python bench.py --mode env --env-steps-per-launch $T --steps 2560 --warmup 256 --no-cpu-baseline 2>/dev/null | line > $o/bench_env_T$T.json; done
python - <<'PY'
import json
for f in ("env_T32","env_T64","env_T128","env_T256"):
    d=json.load(open(f"gpurun_out/r06n/bench_{f}.json")); r=d["roofline"] or {}
    print(f, "%.4g"%d["value"], "us/step %.2f"%(1e3*d["ms_per_step"]), r.get("kernel"), "launch %.1f us"%r.get("avg_launch_us",0))
PY
