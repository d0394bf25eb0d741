for w in 5 100 400 1000; do for rep in 1 2; do python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('warmup $w:', '%.2f M' % (d['value']/1e6), '%.2f us/step' % (1e3*d['ms_per_step']), 'conv %.2f us' % d['roofline']['avg_launch_us'], 'ratio32 %.4g' % d['reference_replay_ratio']['value'])"; done; done
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('2000 steps:', '%.2f M' % (d['value']/1e6), '%.2f us/step' % (1e3*d['ms_per_step']))"
