run() { echo "== $*"; python bench.py "$@" --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); r=d.get('roofline') or {}; print('%.4g %s' % (d['value'], d['unit']), '%.2f us/step' % (1e3*d['ms_per_step']), r.get('kernel'), (d.get('reference_replay_ratio') or {}).get('value'))
except Exception as e: print('FAILED', l[:300])"; }
run --config c1 --steps 200 --warmup 20
run --config c5 --mode env --steps 640 --warmup 64
run --config c2 --mode env --steps 640 --warmup 64
run --config d9 --steps 50 --warmup 5 --ratio-steps 0
run --config c5 --mode act --steps 200 --warmup 20
run --config c3 --mode learn --steps 200 --warmup 20
run --config c3 --minibatch 1024 --steps 200 --warmup 20
run --config c3 --lattices 1000 --minibatch 1000 --steps 200 --warmup 20
