#!/bin/bash
# round 4: tile_sort_mid skipped by the frame-late hint, against its switch (the fused block scan it was measured beside is gone:
# profiles/README.md); usage (GPU box): tools/r4_launches.sh <tag> [config]
tag=${1:-r4l}; cfg=${2:-cfg2}; out=gpurun_out/$tag; mkdir -p $out
for v in "" "GSR_EMIT_SCAN=0" "" "GSR_EMIT_SCAN=0" "GSR_MID_HINT=0"; do
  n=$(echo "$v" | tr -c 'A-Z0-9=_' '_')
  env $v timeout -k 10 240 python3 bench.py --config $cfg --steps 256 --warmup 8 --no-cpu-baseline --dist off > $out/b_$n.json 2> $out/b_$n.err || { echo "FAILED [$v]"; tail -3 $out/b_$n.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_$n.json") if l.startswith("{")][-1])
print("[$v]", d["ms_per_step"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"]})
PY
done
