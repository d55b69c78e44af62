#!/bin/bash
# builds csrc/<file>.hip with each flag set ON THE GPU BOX and prints rocprofv3 average durations of the binning kernels; usage: tools/r3_kstat.sh <tag> <file> "flags1" ...
tag=$1; file=$2; shift; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
n=0
for flags in "$@"; do
  n=$((n+1))
  ( cd based_on_point_cloud_partitions_amd/csrc && rm -f $file.o && make EXTRA="$flags" > $out/build_$n.log 2>&1 ) || { echo "build failed: $flags"; continue; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k$n -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-stage-timers --spinup-ms 0 > $out/k$n.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$out/**/k${n}_kernel_stats.csv",recursive=True)[0]
r={row["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]: float(row["AverageNs"])/1e3 for row in csv.DictReader(open(f))}
print("[$flags]", {k: round(v,1) for k,v in r.items() if any(s in k for s in ("tile_","emit_partition","scan_blocks"))})
PY
  find $out -name "k${n}_kernel_trace.csv" -delete
done
( cd based_on_point_cloud_partitions_amd/csrc && rm -f $file.o && make > /dev/null 2>&1 )
