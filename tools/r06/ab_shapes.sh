# usage: tools/r06/ab_shapes.sh <tag> <variant> ... ; variant = base | <name of tools/probe/ab/<name>.so> | ENV=VAL (environment switch on the regular build)
tag="$1"; shift; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; o=gpurun_out/$tag; mkdir -p $o
for rep in 1 2; do
for v in "$@"; do
  unset DQ_LIB_PATH; envs=""
  case "$v" in base) ;; *=*) envs="$v" ;; *) export DQ_LIB_PATH="$root/tools/probe/ab/$v.so" ;; esac
  val=$(env $envs python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f M/s %.2f us ratio32 %.4g" % (d["value"]/1e6, d["ms_per_step"]*1e3, (d.get("reference_replay_ratio") or {}).get("value", 0)))')
  rm -rf $o/prof
  (cd /tmp && export TMPDIR=/tmp && env $envs rocprofv3 --kernel-trace --stats -d "$root/$o/prof" -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --ratio-steps 0 > /dev/null 2>&1)
  python tools/rocprof_shapes.py $(ls $o/prof/*/*.db | head -1) 500 > $o/shapes_${v//[=\/]/_}_$rep.txt
  echo "$v rep $rep: $val |" $(grep "last 500:" $o/shapes_${v//[=\/]/_}_$rep.txt | head -7 | sed -E 's/^_Z[0-9]+([a-z_0-9]+kernel)[^ ]* +n= *[0-9]+ +last 500: avg= *([0-9.]+)us.*/\1 \2/' | tr '\n' ' ')
done; done
rm -rf $o/prof
