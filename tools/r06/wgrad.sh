mkdir -p gpurun_out/r06y; o=gpurun_out/r06y
DQ_LIB_PATH=$PWD/tools/probe/ab/wg4.so python -m pytest tests/test_qnet_gpu.py tests/test_agent_gpu.py -m gpu -x -q -k "backward or update or loop or gradient" > $o/tests_wg4.log 2>&1; tail -3 $o/tests_wg4.log | cut -c1-250
tools/r06/ab_shapes.sh r06y base wg4 wg3 wg4s10
