cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "HIP_FORCE_DEV_KERNARG=$v rep $rep:" $(python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us" % (d["value"]/1e6, d["ms_per_step"]*1e3))')
done; done
