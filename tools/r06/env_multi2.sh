mkdir -p gpurun_out/r06d; o=gpurun_out/r06d
line() { grep '^{"metric"' | tail -1; }
for f in patch uint8; do for T in 1 16 64; do python bench.py --mode env --env-obs $f --env-steps-per-launch $T --steps 1920 --warmup 64 --no-cpu-baseline 2>$o/env_${f}_T$T.err | line > $o/bench_env_${f}_T$T.json; done; done
python - <<'PY'
import json
for f in ("patch","uint8"):
  for T in (1,16,64):
    try:
        d=json.load(open(f"gpurun_out/r06d/bench_env_{f}_T{T}.json")); r=d["roofline"]
        print(f, T, "%.4g"%d["value"], "us/step %.2f"%(1e3*d["ms_per_step"]), r["kernel"], "launch %.1f us"%r["avg_launch_us"], "frac %.3f"%r["frac"], "in-kernel %.4g"%r["lattice_steps_per_s_in_kernel"])
    except Exception as e: print(f, T, "failed", e)
PY
