mkdir -p gpurun_out/r06l; o=gpurun_out/r06l
python -m pytest tests/test_qnet_gpu.py -m gpu -x -q -k "range_guard or forward_inference" > $o/tests.log 2>&1; tail -3 $o/tests.log | cut -c1-250
tools/r06/ab_shapes.sh r06l base noguard DQ_X_PLANES=0
