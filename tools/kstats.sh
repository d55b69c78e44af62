#!/bin/bash
# usage (on the GPU box, from the repo root): tools/kstats.sh <tag> [bench args...]  -> gpurun_out/<tag>/kernel_stats.txt
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-stage-timers "$@" > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 2; }
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/stats_kernel_stats.csv",recursive=True)[0]
lines=[]
for r in list(csv.DictReader(open(f)))[:24]:
    lines.append("%s %6s %12s %10s %6s" % (r["Name"].replace("(anonymous namespace)::","")[:64].ljust(64), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
open("$out/kernel_stats.txt","w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
find $out -name "stats_kernel_trace.csv" -delete
