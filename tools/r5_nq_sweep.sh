#!/bin/bash
# usage (GPU box): tools/r5_nq_sweep.sh <tag>  -> backward work split (GSR_BWD_NQ) x blend log (GSR_BLEND_LOG) on the dense workloads
tag=$1; out=gpurun_out/$tag; mkdir -p $out
for cfg in dense1m dense1m5 cfg4 mid800; do
 for nq in 1 2; do
  for v in 2 0; do
    GSR_BLEND_LOG=$v GSR_BWD_NQ=$nq timeout -k 10 300 python bench.py --config $cfg --steps 48 --warmup 8 --no-cpu-baseline --dist off > $out/b_${cfg}_${nq}_${v}.log 2> $out/b_${cfg}_${nq}_${v}.err || { echo "bench failed $cfg $nq $v"; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_${cfg}_${nq}_${v}.log") if l.startswith("{")][-1])
print("$cfg NQ=$nq LOG=$v", d["ms_per_step"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"] if "render" in k["kernel"]}, "list mean", d["workload_stats"]["tile_list_mean"])
PY
  done
 done
done
