mkdir -p gpurun_out/r06h; o=gpurun_out/r06h
line() { grep '^{"metric"' | tail -1; }
python bench.py --config c5 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | line > $o/bench_c5.json
rm -rf $o/prof
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --config c5 --steps 500 --warmup 50 --no-cpu-baseline --ratio-steps 0 > "$GRAFT_REPO_ROOT/$o/prof.log" 2>&1)
python tools/rocprof_shapes.py $(ls $o/prof/*/*.db | head -1) 500 > $o/shapes_c5.txt
rm -rf $o/prof
python -c "
import json; d=json.load(open('$o/bench_c5.json')); print('c5', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], (d.get('reference_replay_ratio') or {}).get('value'))"
grep "last 500" -A14 $o/shapes_c5.txt
