#!/bin/bash
# does the world-size-1 RCCL group cost the timed region anything?  usage (GPU box): tools/r4_dist_ab.sh <tag>
tag=${1:-r4dist}; out=gpurun_out/$tag; mkdir -p $out
i=0
for v in "--dist off" "--dist on" "--dist on --no-stage-timers" "--dist off --no-stage-timers" "--dist on" "--dist off"; do
  i=$((i+1))
  timeout -k 10 240 python3 bench.py --no-cpu-baseline $v > $out/b_$i.json 2> $out/b_$i.err || { echo "FAILED [$v]"; tail -3 $out/b_$i.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_$i.json") if l.startswith("{")][-1])
print("[$v]", d["ms_per_step"], "cold", d["ms_per_step_cold"], "spinup last", d["spinup"]["last_batch_ms_per_step"], "barriers", d["opening_barrier_ms"], d["closing_barrier_ms"])
PY
done
