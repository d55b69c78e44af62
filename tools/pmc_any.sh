#!/bin/bash
# usage (GPU box, repo root): tools/pmc_any.sh <tag> "<counters>" [bench args] -> per-kernel averages of the given counters
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers "$@" > $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 2; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/**/pmc_counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur=collections.defaultdict(list)
kt=glob.glob("$out/**/pmc_kernel_trace.csv",recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        dur[name].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,v in per.items():
    if not any(s in k for s in ("render_", "gaussian_backward", "row_", "preprocess", "tile_", "emit_")): continue
    d={c:sum(x)/len(x) for c,x in v.items()}
    print(k[:44].ljust(44), "dur_us=%.1f" % (sum(dur[k])/max(1,len(dur[k]))/1e3), " ".join("%s=%.4g" % (c, d[c]) for c in sorted(d)))
PY
find $out -name "pmc_counter_collection.csv" -delete; find $out -name "pmc_kernel_trace.csv" -delete
