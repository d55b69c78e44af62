#!/bin/bash
# builds csrc/<file>.hip with different -D flags ON THE GPU BOX and benches each; usage: tools/r3_build_ab.sh <tag> <file> <config> "flags1" "flags2" ...
tag=$1; file=$2; cfg=$3; shift; shift; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd based_on_point_cloud_partitions_amd/csrc
n=0
for flags in "$@"; do
  n=$((n+1))
  rm -f $file.o
  make EXTRA="$flags" > $out/build_$n.log 2>&1 || { echo "build failed: $flags"; tail -3 $out/build_$n.log; continue; }
  ( cd ../.. && timeout -k 10 200 python bench.py --config $cfg --steps 256 --warmup 8 --no-cpu-baseline > $out/b_$n.log 2>&1
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_$n.log") if l.startswith("{")][-1])
print("[$flags]", d["value"], d["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in d["roofline_kernels"]][:4])
PY
  )
done
rm -f $file.o; make > /dev/null 2>&1
