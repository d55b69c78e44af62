#!/bin/bash
# The default bench under one HIP-runtime environment switch at a time (GPU box, repo root): bash tools/r4_env_ab.sh
run() { python bench.py --no-cpu-baseline --dist off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['ms_per_step_cold'], d['spinup']['last_batch_ms_per_step'])"; }
for r in 1 2; do
run default
HIP_FORCE_DEV_KERNARG=1 run dev_kernarg_1
HIP_FORCE_DEV_KERNARG=0 run dev_kernarg_0
AMD_OPT_FLUSH=0 run opt_flush_0
ROC_USE_FGS_KERNARG=0 run fgs_kernarg_0
DEBUG_HIP_KERNARG_COPY_OPT=0 run kernarg_copy_opt_0
done
