#!/bin/bash
# The round's final measurement set (GPU box, repo root): tools/final_measure.sh <tag>
tag="${1:-r06}"
tools/measure.sh "$tag" 2>&1 | tail -12
tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqA bench.py --steps 40 --warmup 5 --no-cpu-baseline --ratio-steps 0 2>&1 | tail -8 | tee gpurun_out/$tag/sqA.txt
tools/pmc_any.sh "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" sqB bench.py --steps 40 --warmup 5 --no-cpu-baseline --ratio-steps 0 2>&1 | tail -8 | tee gpurun_out/$tag/sqB.txt
rm -rf gpurun_out/sqA gpurun_out/sqB
# the same four + four SQ counters with the round-4 kernels (DQ_CONV_FORM=group DQ_CONV_BWD_FORM=8), for the deltas NOTEBOOK.md quotes
DQ_CONV_FORM=group DQ_CONV_BWD_FORM=8 tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqA bench.py --steps 40 --warmup 5 --no-cpu-baseline --ratio-steps 0 2>&1 | tail -8 | tee gpurun_out/$tag/sqA_groupform.txt
DQ_CONV_FORM=group DQ_CONV_BWD_FORM=8 tools/pmc_any.sh "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" sqB bench.py --steps 40 --warmup 5 --no-cpu-baseline --ratio-steps 0 2>&1 | tail -8 | tee gpurun_out/$tag/sqB_groupform.txt
rm -rf gpurun_out/sqA gpurun_out/sqB
# the printed error / fragile-fraction lines of the shipped-weight and replay tests (keep them in profiles/)
python -m pytest tests/test_shipped_weights.py tests/test_agent_gpu.py tests/test_qnet_gpu.py -q -s -m gpu -k "shipped or replayed or beyond or baseline_batch or range_guard" 2>&1 | grep -vE "^\s*$|warnings.warn|UserWarning" | tail -160 > gpurun_out/$tag/test_printed_lines.txt
