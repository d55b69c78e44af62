#!/bin/bash
# the fork's training iteration with / without the side-stream weight gradients of the appearance head; usage (GPU box): tools/r4_overlap_ab.sh <tag>
tag=${1:-r4ov}; out=gpurun_out/$tag; mkdir -p $out
for v in 1 0 1 0; do
  GSR_APPEARANCE_OVERLAP=$v timeout -k 10 300 python3 bench.py --full-step --steps 40 --warmup 4 --no-cpu-baseline --dist off --spinup-ms 500 > $out/b_$v.json 2> $out/b_$v.err || { echo FAILED; tail -3 $out/b_$v.err; continue; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_$v.json") if l.startswith("{")][-1])["full_step"]
print("overlap=$v", {k.replace("train_iteration_with_appearance_","app:").replace("_ms",""): v for k,v in d.items() if "appearance" in k})
PY
done
