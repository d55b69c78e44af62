#!/bin/bash
# kernel timeline of a few steps: per-step busy time, gaps between consecutive kernels; usage: tools/r3_gaps.sh <tag>
tag=${1:-gaps}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-stage-timers --dist off --spinup-ms 600 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/kt_kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]) for r in csv.DictReader(open(f))]
rows.sort()
# find steps: from preprocess_kernel to gaussian_backward_kernel
steps=[];cur=None
for s,e,n in rows:
    if n.startswith("preprocess_kernel"): cur=[]
    if cur is not None: cur.append((s,e,n))
    if n.startswith("gaussian_backward_kernel") and cur: steps.append(cur);cur=None
print(len(steps),"steps")
import collections
for st in steps[-6:-3]:
    t0=st[0][0]; busy=sum(e-s for s,e,n in st); span=st[-1][1]-t0
    print("span %.1f us busy %.1f us kernels %d"%(span/1e3,busy/1e3,len(st)))
st=steps[-4]
prev=None
for s,e,n in st:
    print("%-42s start %8.1f dur %7.1f gap %6.1f"%(n,(s-st[0][0])/1e3,(e-s)/1e3,((s-prev)/1e3 if prev else 0))); prev=e
PY
python3 tools/r3_gaps2.py $out; find $out -name "kt_kernel_trace.csv" -delete
