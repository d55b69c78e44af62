#!/bin/bash
# usage (GPU box): tools/r5_ab_configs.sh <tag>  -> blend log on / off over the named workloads (bench.py --config), two repeats
tag=$1; out=gpurun_out/$tag; mkdir -p $out
for cfg in cfg2 cfg1 cfg4 dense1m mid800 mid640; do
 for r in 1 2; do
  for v in 1 0; do
    GSR_BLEND_LOG=$v timeout -k 10 300 python bench.py --config $cfg --steps 64 --warmup 8 --no-cpu-baseline --dist off > $out/b_${cfg}_${v}_$r.log 2> $out/b_${cfg}_${v}_$r.err || { echo "bench failed $cfg $v"; tail -3 $out/b_${cfg}_${v}_$r.err; continue; }
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_${cfg}_${v}_$r.log") if l.startswith("{")][-1])
print("$cfg LOG=$v", d["ms_per_step"], "ref-lists", d["ms_per_step_reference_lists"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"] if "render" in k["kernel"]})
PY
  done
 done
done
