python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/measure.sh r03a 2>&1 | tail -14
tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqA bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -8
tools/pmc_any.sh "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" sqB bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -8
rm -rf gpurun_out/sqA gpurun_out/sqB
