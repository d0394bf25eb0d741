mkdir -p gpurun_out/r06f; o=gpurun_out/r06f
python -m pytest tests/test_qnet_gpu.py -m gpu -x -q -s -k "range_guard" > $o/tests2.log 2>&1; tail -5 $o/tests2.log | cut -c1-250; grep "forward range guard" $o/tests2.log | head -40
