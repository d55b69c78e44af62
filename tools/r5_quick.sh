#!/bin/bash
# usage (GPU box, repo root): tools/r5_quick.sh <tag> [pytest -k expression]  -> a fast parity subset of the rasterizer, two bench lines, SQ counters
tag=$1; kexpr=${2:-"small_scene or config1 or snapshot or saturating or long_tile or very_long or fewer_than_16 or image_size or equal_depths_overflow"}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  python bench.py --steps 128 --warmup 8 --no-cpu-baseline --dist off > $out/bench_$r.log 2> $out/bench_$r.err || exit 3
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/bench_$r.log") if l.startswith("{")][-1])
print("$tag", d["ms_per_step"], {k["kernel"]: k["avg_ms"] for k in d["roofline_kernels"]})
PY
done
tools/pmc_sq.sh $tag/sq --dist off --spinup-ms 0 2>&1 | grep render_
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -o sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers --dist off --spinup-ms 0 > $out/sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$out/sq2/**/sq2_counter_collection.csv",recursive=True)
if fs:
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items():
        if "render_" in k: print(k[:48].ljust(48)," ".join("%s=%.4g"%(c.replace("SQ_",""),sum(x)/len(x)) for c,x in sorted(v.items())))
PY
find $out -name "*_counter_collection.csv" -delete
