#!/bin/bash
# The round's measurement set (GPU box, repo root): tools/measure.sh <tag>  ->  gpurun_out/<tag>/
#   bench_<config>_<mode>.json   one JSON line each: c3 loop (headline, with cpu_baseline), c2 / c5 loop, c3 act / learn / env,
#                                c3 loop at minibatch 32 (the reference's replay ratio setting)
#   loop_c3_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the headline command
#   loop_c3_kernel_shapes.txt    the same trace per (kernel, grid)
#   pmc_traffic_<mode>_<cfg>.json  FETCH_SIZE / WRITE_SIZE passes (loop c3 / c5 / c2, env c3), stamped with the kernel sources' sha256
tag="${1:-meas}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; out="gpurun_out/$tag"; mkdir -p "$out"
line() { grep '^{"metric"' | tail -1; }
# the PMC passes first, copied into profiles/ on this box: the bench lines below then carry `roofline.traffic` of THESE sources
tools/pmc_traffic.sh loop c3 > "$out/pmc.log" 2>&1; cp gpurun_out/pmc_traffic_loop_c3.json "$out/" 2>/dev/null; cp gpurun_out/pmc_traffic_loop_c3.json profiles/ 2>/dev/null
for cfg in c5 c2; do tools/pmc_traffic.sh loop $cfg > "$out/pmc_$cfg.log" 2>&1; cp gpurun_out/pmc_traffic_loop_$cfg.json "$out/" 2>/dev/null; cp gpurun_out/pmc_traffic_loop_$cfg.json profiles/ 2>/dev/null; done
tools/pmc_traffic.sh env c3 > "$out/pmc_env.log" 2>&1; cp gpurun_out/pmc_traffic_env_c3.json "$out/" 2>/dev/null; cp gpurun_out/pmc_traffic_env_c3.json profiles/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
python bench.py 2>"$out/bench_c3_loop.err" | line > "$out/bench_c3_loop.json"
for cfg in c2 c5; do python bench.py --config $cfg --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_${cfg}_loop.json"; done
for mode in act learn; do python bench.py --mode $mode --steps 1000 --warmup 50 2>/dev/null | line > "$out/bench_c3_${mode}.json"; done
# environment only: 64 agent steps per launch (dq_env_act_steps; patch words into a ring), then one step per launch, then the reference's uint8 planes at 64 per launch
python bench.py --mode env --steps 2560 --warmup 256 2>/dev/null | line > "$out/bench_c3_env.json"
python bench.py --mode env --env-steps-per-launch 1 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_env_T1.json"
python bench.py --mode env --env-obs uint8 --steps 2560 --warmup 256 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_env_uint8.json"
python bench.py --minibatch 32 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_mb32.json"
# the reference's replay ratio, 32 trained samples per environment step (one 32-sample update per step, TRAIN:119-127), three ways:
# 32 updates of 4096 per vector step; 4 updates of 32768; and the reference's own schedule, 4096 updates of 32 (a few steps: 0.3 s each)
python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_ratio32.json"
python bench.py --minibatch 32768 --updates-per-step 4 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_ratio32_mb32768.json"
python bench.py --minibatch 32 --updates-per-step 4096 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_ratio32_mb32.json"
# the round-4 forward (one workgroup per group of samples, weights streamed from L2) beside the wave-private one (csrc/conv_wave.hip), same box
DQ_CONV_FORM=group python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_groupform.json"
DQ_CONV_FORM=group python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_ratio32_groupform.json"
# the 8-wave convolutional backward (csrc/fused_bwd.hip conv_bwd_chain_kernel) beside the 16-wave one (csrc/conv_bwd16.hip), same box; both round-4 forms together
DQ_CONV_BWD_FORM=8 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_bwd8form.json"
DQ_CONV_BWD_FORM=8 DQ_CONV_FORM=group python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_round4forms.json"
DQ_PAIR_TARGETS=0 python bench.py --updates-per-step 32 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_ratio32_unpaired.json"
# round 6: a1 through HBM (the round-5 form) beside the recomputing conv backward; the last convolution's output as piece planes (built, off by default)
DQ_CONV_BWD_A1=saved python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_a1saved.json"
DQ_X_PLANES=1 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_xplanes.json"
# the uint8 ring beside the patch-word ring (DQNCore.compact), same box
DQ_COMPACT_OBS=0 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c3_loop_uint8ring.json"
DQ_COMPACT_OBS=0 python bench.py --config c5 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > "$out/bench_c5_loop_uint8ring.json"
rm -rf "$out/prof"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$root/$out/prof" -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --ratio-steps 0 > "$root/$out/prof.log" 2>&1)
python tools/rocprof_summary.py $(ls $out/prof/*/*.db | head -1) "$out/loop_c3_kernel_stats.csv"
# per (kernel, grid): the step's launches apart from the set-up's (ring fill, resets) of the same kernels
python tools/rocprof_shapes.py $(ls $out/prof/*/*.db | head -1) 500 > "$out/loop_c3_kernel_shapes.txt"
rm -rf "$out/prof"
head -12 "$out/loop_c3_kernel_stats.csv" | cut -c1-120
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "%.3g %s" % (d["value"], d["unit"]), "ms/step %.4f" % d["ms_per_step"], r.get("kernel"), "frac %.3f" % r.get("frac", 0), "avg_us %.1f" % r.get("avg_launch_us", 0))
PY
done
