#!/bin/bash
# SQ counters of the convolutional backward's two forms (tools/conv_ab.py, kernels alone)
root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"
for form in 16 8; do
echo "== DQ_CONV_BWD_FORM=$form"
DQ_CONV_BWD_FORM=$form tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqA tools/conv_ab.py 4096 2>&1 | grep conv_bwd
DQ_CONV_BWD_FORM=$form tools/pmc_any.sh "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" sqB tools/conv_ab.py 4096 2>&1 | grep conv_bwd
DQ_CONV_BWD_FORM=$form tools/pmc_any.sh "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" sqC tools/conv_ab.py 4096 2>&1 | grep conv_bwd
done
rm -rf gpurun_out/sqA gpurun_out/sqB gpurun_out/sqC
