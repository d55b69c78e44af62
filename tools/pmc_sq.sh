#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <tag> [bench args]  -> gpurun_out/<tag>/sq.txt  (SQ counters per kernel, averaged over launches)
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers "$@" > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 2; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/**/sq_counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines=[]
for k,v in per.items():
    if not any(s in k for s in ("render_", "gaussian_backward", "row_", "preprocess", "tile_", "band_")): continue
    d={c:sum(x)/len(x) for c,x in v.items()}
    lines.append(k[:48].ljust(48)+" ".join("%s=%.4g" % (c.replace("SQ_",""), d[c]) for c in sorted(d)))
open("$out/sq.txt","w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
find $out -name "sq_counter_collection.csv" -delete
