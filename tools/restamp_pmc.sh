#!/bin/bash
# The FETCH_SIZE / WRITE_SIZE passes alone (tools/pmc_traffic.sh for loop c3 / c5 / c2 and env c3), copied into profiles/ on this box: run after the LAST edit of a
# kernel source -- bench.py prints `roofline.traffic` only while profiles/pmc_traffic_*.json carry the sources' sha256.  (tools/measure.sh does the same in front of
# the round's bench lines.)  GPU box, repo root; results also under gpurun_out/restamp/.
root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/restamp
tools/pmc_traffic.sh loop c3 > gpurun_out/restamp/c3.log 2>&1
for cfg in c5 c2; do tools/pmc_traffic.sh loop $cfg > gpurun_out/restamp/$cfg.log 2>&1; done
tools/pmc_traffic.sh env c3 > gpurun_out/restamp/env.log 2>&1
cp gpurun_out/pmc_traffic_*.json gpurun_out/restamp/; cp gpurun_out/pmc_traffic_*.json profiles/
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -4 gpurun_out/restamp/c3.log
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 | tee gpurun_out/restamp/bench_driver_shape.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver-shaped run:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])"
