mkdir -p gpurun_out/r06d; o=gpurun_out/r06d
python -m pytest tests/test_env_gpu.py -m gpu -x -q > $o/tests_env.log 2>&1; tail -3 $o/tests_env.log
line() { grep '^{"metric"' | tail -1; }
for T in 1 4 16 32; do python bench.py --mode env --env-steps-per-launch $T --steps 1920 --warmup 64 --no-cpu-baseline 2>$o/env_T$T.err | line > $o/bench_env_T$T.json; done
python - <<'PY'
import json
for T in (1,4,16,32):
    try:
        d=json.load(open(f"gpurun_out/r06d/bench_env_T{T}.json")); r=d["roofline"]
        print(T, "%.4g"%d["value"], "us/step %.2f"%(1e3*d["ms_per_step"]), r["kernel"], "launch %.1f us"%r["avg_launch_us"], "frac %.3f"%r["frac"], "in-kernel %.4g"%r["lattice_steps_per_s_in_kernel"])
    except Exception as e: print(T, "failed", e)
PY
tail -3 $o/env_T16.err
