#!/bin/bash
mkdir -p gpurun_out/xv
for v in 0 1 2 4 8 15; do
  GSR_SKEW_XV=$v python bench.py --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/xv/b$v.log 2>&1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/xv/b$v.log").read().strip().splitlines()[-1])
print("XV=$v", [(k["kernel"],k["avg_ms"]) for k in d["roofline_kernels"]][:2])
PY
done
