#!/bin/bash
# quick A/B: one bench line summary; usage: tools/r3_quick.sh <tag> <config> [VAR=val ...]
tag=$1; cfg=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 32 --warmup 6 --no-cpu-baseline $BENCH_ARGS > $out/bench.log 2>&1
python3 - <<PY
import json
d=json.loads([l for l in open("$out/bench.log") if l.startswith("{")][-1])
print("$tag $cfg $*", d["value"], d["ms_per_step"], d["ms_per_step_reference_lists"], [(k["kernel"],k["avg_ms"]) for k in d["roofline_kernels"]])
PY
