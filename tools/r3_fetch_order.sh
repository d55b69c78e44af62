#!/bin/bash
# FETCH_SIZE per launch of the blend kernels with the Gaussians as generated vs Morton-ordered; usage: tools/r3_fetch_order.sh <tag>
tag=${1:-fetch_order}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for ord in as-generated morton; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o f_$ord -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers --order $ord > $out/f_$ord.log 2>&1 || { tail -5 $out/f_$ord.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o w_$ord -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers --order $ord > $out/w_$ord.log 2>&1 || exit 2
done
python3 - <<PY
import csv,glob,collections
for ord in ("as-generated","morton"):
    for c in ("f","w"):
        f=glob.glob("$out/**/%s_%s_counter_collection.csv"%(c,ord),recursive=True)[0]
        per=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
            if any(k in n for k in ("render_","gaussian_backward","tile_scatter","tile_sort_small","preprocess_kernel")): per[n].append(float(r["Counter_Value"]))
        print(ord, "FETCH_SIZE" if c=="f" else "WRITE_SIZE", {k[:34]: round(sum(v)/len(v)/1024,1) for k,v in sorted(per.items())}, "MiB per launch (counter in KiB, not doubled)")
PY
find $out -name "*_counter_collection.csv" -delete
