#!/bin/bash
# Runs the variants built by tools/ab_build.sh on the GPU box, selected through GSR_LIBRARY (based_on_point_cloud_partitions_amd/_lib.py).
# usage: tools/ab_run.sh <tag> <config> bench|kstat [repeats] ["extra bench.py arguments"]
#   bench: bench.py per variant (value / ms_per_step / slowest stages);  kstat: rocprofv3 --kernel-trace --stats per variant
tag=$1; cfg=$2; mode=$3; rep=${4:-1}; extra=${5:-}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for r in $(seq 1 $rep); do
for d in $root/tools/scratch/ab_${tag}_*; do
  n=${d##*_}; flags=$(cat $d/flags.txt)
  export GSR_LIBRARY=$d/based_on_point_cloud_partitions_amd/csrc/libgsrast_mi355x.so
  [ -f $GSR_LIBRARY ] || { echo "no library for [$flags]"; continue; }
  if [ "$mode" = kstat ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k$n -- python3 bench.py --config $cfg --steps 64 --warmup 4 --no-cpu-baseline --no-stage-timers --spinup-ms 500 --dist off > $out/k$n.log 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("$out/**/k${n}_kernel_stats.csv",recursive=True)[0]
r={row["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]: float(row["AverageNs"])/1e3 for row in csv.DictReader(open(f))}
print("[$flags]", {k: round(v,1) for k,v in r.items() if any(s in k for s in ("tile_","emit_","scan_","render_","gaussian_","preprocess"))})
PY
    find $out -name "k${n}_kernel_trace.csv" -delete
  else
    timeout -k 10 300 python3 bench.py --config $cfg --steps 256 --warmup 8 --no-cpu-baseline --dist off $extra > $out/b_${n}_$r.log 2> $out/b_${n}_$r.err
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_${n}_$r.log") if l.startswith("{")][-1])
print("[$flags] $extra", d["ms_per_step"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"]})
PY
  fi
  unset GSR_LIBRARY
done
done
