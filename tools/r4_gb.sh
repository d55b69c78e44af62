#!/bin/bash
# round 4: the per-Gaussian backward with the cost-balanced split (GSR_GB_BALANCED=1, default) against the plain one, on the
# workload's own (random) order and on Morton-sorted Gaussians; usage (GPU box): tools/r4_gb.sh <tag>
tag=${1:-r4gb}; out=gpurun_out/$tag; mkdir -p $out
for cfg in cfg2 cfg4; do for order in as-generated morton; do for bal in 1 0; do
  GSR_GB_BALANCED=$bal timeout -k 10 240 python3 bench.py --config $cfg --order $order --steps 128 --warmup 8 --no-cpu-baseline --dist off > $out/b_${cfg}_${order}_$bal.json 2> $out/b_${cfg}_${order}_$bal.err || { echo "FAILED $cfg $order $bal"; tail -3 $out/b_${cfg}_${order}_$bal.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_${cfg}_${order}_$bal.json") if l.startswith("{")][-1])
print("$cfg $order balanced=$bal", d["ms_per_step"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"]})
PY
done; done; done
