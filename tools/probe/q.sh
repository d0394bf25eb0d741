cd $GRAFT_REPO_ROOT
run() { env $2 timeout 200 python bench.py --mode $3 --steps $4 --warmup 50 --no-cpu-baseline 2>gpurun_out/err.txt | grep "^{\"metric\"" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}; print(sys.argv[1], sys.argv[2], sys.argv[3], '%.2f M/s %.2f us/step' % (d['value']/1e6, d['ms_per_step']*1e3), r.get('kernel'), 'avg_launch_us', r.get('avg_launch_us'), 'n', r.get('launches_timed'), 'frac', r.get('frac'))" $1 $3 $4; grep "per-family" gpurun_out/err.txt; }
for rep in 1 2; do
run sampled A=1 loop 2000; run unarmed DQ_BENCH_NO_ARM=1 loop 2000
run sampled A=1 act 2000; run unarmed DQ_BENCH_NO_ARM=1 act 2000
run sampled A=1 env 2000; run unarmed DQ_BENCH_NO_ARM=1 env 2000
run sampled A=1 loop 20
done
run fam DQ_BENCH_FAMILIES=1 loop 1000
run fam DQ_BENCH_FAMILIES=1 learn 1000
