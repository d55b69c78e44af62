#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag>      e.g. r01_final
# 1. bench.py JSON line, 2. rocprofv3 --kernel-trace --stats, 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE:
# the TCC block cannot hold both at once, MI355X_MICROARCH.md §rocprofv3 PMC slots) -> gpurun_out/<tag>_*.
set -o pipefail
tag=${1:-run}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py --steps 32 --warmup 8 > $out/bench.log 2>&1 || { tail -20 $out/bench.log; exit 1; }
grep '^{' $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > $out/stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/write.log 2>&1 || exit 4
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/sq.log 2>&1 || exit 5
rm -f $out/stats_kernel_trace.csv
ls -la $out
