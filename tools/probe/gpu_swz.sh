set -x
python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py -m gpu -x -q 2>&1 | tail -5
tools/ab_run.sh swz base noswz 2>&1 | grep rep
tools/pmc_any.sh "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" pmc_swz bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep conv_chain
DQ_LIB_PATH=$PWD/tools/probe/ab/noswz.so tools/pmc_any.sh "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" pmc_noswz bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep conv_chain
