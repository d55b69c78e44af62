#!/bin/bash
# usage (GPU box): tools/quick.sh <tag> [env assignments...]   -> parity tests of the rasterizer + one bench line summary
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
env "$@" python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $out/bench.log 2>&1
python3 - <<PY
import json
d=json.loads(open("$out/bench.log").read().strip().splitlines()[-1])
print("$tag", d["value"], d["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in d["roofline_kernels"]])
PY
