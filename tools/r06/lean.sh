mkdir -p gpurun_out/r06s; o=gpurun_out/r06s
DQ_DENSE_LEAN=2 python -m pytest tests/test_qnet_gpu.py tests/test_compact_gpu.py -m gpu -x -q > $o/tests_lean2.log 2>&1; tail -3 $o/tests_lean2.log | cut -c1-250
tools/r06/ab_shapes.sh r06s DQ_DENSE_LEAN=0 DQ_DENSE_LEAN=1
