#!/bin/bash
# Usage (GPU box): tools/env_knobs.sh -- the c3 loop under one HIP-runtime environment knob at a time (ms per step; DESIGN.md section 4, launch boundaries)
run() { timeout 45 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(1e3*d['ms_per_step'],2), 'us/step', d['params_checksum'])"; }
run base
for kv in AMD_OPT_FLUSH=0 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 \
          ROC_SKIP_KERNEL_ARG_COPY=1 GPU_MAX_HW_QUEUES=1; do      # (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the run: every run under its own timeout)
  ( export $kv; run $kv )
done
run base
