mkdir -p gpurun_out/r06p; o=gpurun_out/r06p
run() { name="$1"; shift; v=$(env "$@" python bench.py --config c5 --steps 1000 --warmup 50 --no-cpu-baseline --ratio-steps 0 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))'); echo "$name: $v"; }
run base A=1
run conv_bwd_S8 DQ_CONV_BWD_S=8
run conv_bwd_S2 DQ_CONV_BWD_S=2
run dense_rt2 DQ_DENSE_RT=2
run dbwd_split1 DQ_DENSE_BWD_SPLIT=1
run dbwd_split2 DQ_DENSE_BWD_SPLIT=2
run wgrad_sl4 DQ_WGRAD_SLICES=4
run wgrad_sl16 DQ_WGRAD_SLICES=16
run persist0 DQ_CONV_PERSIST=0
run pgrid256 DQ_CONV_PERSIST_GRID=256
run base2 A=1
