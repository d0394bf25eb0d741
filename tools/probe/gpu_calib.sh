root=$PWD; mkdir -p gpurun_out
for c in WRITE_SIZE; do
  rm -rf gpurun_out/calib_$c
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --kernel-trace -d $root/gpurun_out/calib_$c -o pmc --output-format csv -- $root/tools/probe/bin/pmc_calib > /dev/null 2>&1)
  python3 - $c <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
agg, cnt = collections.Counter(), collections.Counter()
for f in glob.glob(f"gpurun_out/calib_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"][:30]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"][:30]] += 1
for k in agg: print(c, k, "%.1f KB per launch = %.3f of 65536 KB" % (agg[k] / cnt[k], agg[k] / cnt[k] / 65536))
PY
  rm -rf gpurun_out/calib_$c
done
