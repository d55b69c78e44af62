#!/bin/bash
# A/B builds of the HIP library ON THE GPU BOX without touching the shipped .so: every flag set is compiled in its own copy of
# csrc/ under tools/scratch/ and selected through GSR_LIBRARY (based_on_point_cloud_partitions_amd/_lib.py).
# usage: tools/build_ab.sh <tag> <config> bench|kstat "flags1" "flags2" ...     ("" = the shipped flags)
#   bench: bench.py per variant, prints value / ms_per_step / the four slowest stages
#   kstat: rocprofv3 --kernel-trace --stats per variant, prints the average duration of every kernel of the step
tag=$1; cfg=$2; mode=$3; shift; shift; shift
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
n=0
for flags in "$@"; do
  n=$((n+1))
  d=$root/tools/scratch/ab_${tag}_$n
  rm -rf $d; mkdir -p $d/based_on_point_cloud_partitions_amd/csrc $d/include
  cp $root/based_on_point_cloud_partitions_amd/csrc/*.hip $root/based_on_point_cloud_partitions_amd/csrc/*.h $root/based_on_point_cloud_partitions_amd/csrc/Makefile $d/based_on_point_cloud_partitions_amd/csrc/
  cp $root/include/*.h $d/include/
  make -C $d/based_on_point_cloud_partitions_amd/csrc -j8 EXTRA="$flags" > $out/build_$n.log 2>&1 || { echo "build failed: [$flags]"; tail -3 $out/build_$n.log; continue; }
  export GSR_LIBRARY=$d/based_on_point_cloud_partitions_amd/csrc/libgsrast_mi355x.so
  if [ "$mode" = kstat ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k$n -- python3 bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-stage-timers --spinup-ms 0 --dist off > $out/k$n.log 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("$out/**/k${n}_kernel_stats.csv",recursive=True)[0]
r={row["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]: float(row["AverageNs"])/1e3 for row in csv.DictReader(open(f))}
print("[$flags]", {k: round(v,1) for k,v in r.items() if any(s in k for s in ("tile_","emit_","scan_","render_","gaussian_","preprocess"))})
PY
    find $out -name "k${n}_kernel_trace.csv" -delete
  else
    timeout -k 10 300 python3 bench.py --config $cfg --steps 256 --warmup 8 --no-cpu-baseline --dist off > $out/b_$n.log 2> $out/b_$n.err
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b_$n.log") if l.startswith("{")][-1])
print("[$flags]", d["build_info"], d["value"], d["ms_per_step"], [(k["kernel"],k["avg_ms"]) for k in d["roofline_kernels"]][:5])
PY
  fi
  unset GSR_LIBRARY
done
