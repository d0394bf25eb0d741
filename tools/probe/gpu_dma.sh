echo "== parity pipe2"; DQ_LIB_PATH=$PWD/tools/probe/ab/pipe2.so python -m pytest tests/test_qnet_gpu.py -m gpu -x -q 2>&1 | tail -2
tools/pmc_any.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" pmc_base bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep conv_chain_p
DQ_LIB_PATH=$PWD/tools/probe/ab/pipe2.so tools/pmc_any.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" pmc_pipe2 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep conv_chain_p
rm -rf gpurun_out/pmc_base gpurun_out/pmc_pipe2
