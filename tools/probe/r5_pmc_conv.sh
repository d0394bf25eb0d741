#!/bin/bash
# SQ counters of the conv forward kernels alone (tools/conv_ab.py), three passes: tools/probe/r5_pmc_conv.sh <tag> [form]
tag="${1:-r5pmc}"; form="${2:-wave}"; root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out/$tag
export DQ_CONV_FORM=$form
tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" $tag/p1 tools/conv_ab.py 4096 | grep -E "conv_wave|conv_chain_pkernel<0|conv_chain_pk" > gpurun_out/$tag/pmc_$form.txt
tools/pmc_any.sh "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT" $tag/p2 tools/conv_ab.py 4096 | grep -E "conv_wave|conv_chain_pk" >> gpurun_out/$tag/pmc_$form.txt
tools/pmc_any.sh "SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH" $tag/p3 tools/conv_ab.py 4096 | grep -E "conv_wave|conv_chain_pk" >> gpurun_out/$tag/pmc_$form.txt
rm -rf gpurun_out/$tag/p1 gpurun_out/$tag/p2 gpurun_out/$tag/p3
cat gpurun_out/$tag/pmc_$form.txt
