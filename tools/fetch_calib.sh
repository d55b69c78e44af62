#!/bin/bash
# GPU box: FETCH_SIZE of the three calibration kernels next to their known byte counts -> gpurun_out/<tag>/fetch_calib.txt
tag=${1:-calib}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
# the probe is built from its source here (no binaries in the repository)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip || exit 2
./tools/fetch_calib > $out/fetch_calib.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fc -- ./tools/fetch_calib > $out/fc.log 2>&1 || { tail -5 $out/fc.log; exit 2; }
python3 - <<PY >> $out/fetch_calib.txt
import csv,glob,collections
f=glob.glob("$out/**/fc_counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE": per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in per.items(): print("FETCH_SIZE %-10s %s KiB -> %.1f MiB per launch" % (k, ["%.0f"%x for x in v], sum(v)/len(v)/1024))
PY
find $out -name "fc_counter_collection.csv" -delete
cat $out/fetch_calib.txt
