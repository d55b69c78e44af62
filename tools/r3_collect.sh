#!/bin/bash
# round-3 evidence: profiles/collect.sh + the other configurations + the SQ counters the verdict asked for
tag=${1:-r03_v1}
profiles/collect.sh $tag || exit $?
out=gpurun_out/$tag
python3 profiles/pmc_to_json.py $out $out/pmc_traffic.json
for cfg in cfg1 cfg4; do python3 bench.py --config $cfg --no-cpu-baseline > $out/bench_$cfg.log 2>&1; grep '^{' $out/bench_$cfg.log > $out/bench_$cfg.json; done
python3 bench.py --full-step --steps 20 --warmup 4 --no-cpu-baseline > $out/bench_full.log 2>&1; grep '^{' $out/bench_full.log > $out/bench_full_step.json
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out -o sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS --output-format csv -d $out -o sq3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/sq3.log 2>&1
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for tagf in ("sq2","sq3"):
    fs=glob.glob("$out/**/%s_counter_collection.csv"%tagf,recursive=True)
    if not fs: continue
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items():
        if any(s in k for s in ("render_","gaussian_backward","preprocess","tile_","emit_","scan_blocks")):
            for c,x in v.items(): res[k][c]=sum(x)/len(x)
kt=glob.glob("$out/**/sq2_kernel_trace.csv",recursive=True)
if kt:
    dur=collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        dur[name].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    for k in res: 
        if k in dur: res[k]["duration_us_under_profiler"]=sum(dur[k])/len(dur[k])/1e3
json.dump(res,open("$out/sq_counters.json","w"),indent=1,sort_keys=True)
for k,v in res.items(): print(k[:50].ljust(50)," ".join("%s=%.4g"%(a.replace("SQ_",""),b) for a,b in sorted(v.items())))
PY
find $out -name "*_counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
ls $out
