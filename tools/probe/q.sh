cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_env_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in base rb1; do if [ $v = base ]; then unset DQ_LIB_PATH; else export DQ_LIB_PATH=$PWD/tools/probe/ab/$v.so; fi
for m in env act loop; do timeout 100 python bench.py --mode $m --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | grep "^{\"metric\"" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], sys.argv[2], \"%.1f M/s %.2f us/step\" % (d[\"value\"]/1e6, d[\"ms_per_step\"]*1e3))" $v $m; done; done; done
