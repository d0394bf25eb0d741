mkdir -p gpurun_out/r06q; o=gpurun_out/r06q
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -4 $o/gputests.log | cut -c1-300
tools/final_measure.sh r06 > gpurun_out/r06_final.log 2>&1; tail -3 gpurun_out/r06_final.log | cut -c1-200
python bench.py --steps 20 --warmup 5 > $o/bench_driver_shape.json 2>/dev/null
