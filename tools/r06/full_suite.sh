mkdir -p gpurun_out/r06g; o=gpurun_out/r06g
python -m pytest tests/test_distributed_gpu.py tests/test_shipped_weights.py -m gpu -x -q > $o/gputests2.log 2>&1; tail -4 $o/gputests2.log | cut -c1-300
bash tools/r06/c5_prof.sh
