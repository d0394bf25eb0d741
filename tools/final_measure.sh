#!/bin/bash
# The round's final measurement set (GPU box, repo root): tools/final_measure.sh <tag>
tag="${1:-r03}"
tools/measure.sh "$tag" 2>&1 | tail -12
tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqA bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -8 | tee gpurun_out/$tag/sqA.txt
tools/pmc_any.sh "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" sqB bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -8 | tee gpurun_out/$tag/sqB.txt
rm -rf gpurun_out/sqA gpurun_out/sqB
