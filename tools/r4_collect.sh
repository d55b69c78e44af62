#!/bin/bash
# round-4 evidence (GPU box, repo root): tools/r4_collect.sh <tag>  ->  gpurun_out/<tag>/
#   bench.json (default bench line), stats_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command), separate --pmc
#   passes (FETCH_SIZE / WRITE_SIZE / SQ counters; never combined with a trace domain other than the kernel trace), pmc_traffic.json,
#   sq_counters.json, bench lines of cfg1 / cfg4 / Morton order / the full training step, grad_error.json
set -o pipefail
tag=${1:-r04_v1}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
grep '^{' $out/bench.log > $out/bench.json
P="--steps 16 --warmup 4 --no-cpu-baseline --dist off --spinup-ms 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py $P > $out/stats.log 2>&1 || exit 2
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers --dist off --spinup-ms 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python3 bench.py $Q > $out/fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- python3 bench.py $Q > $out/write.log 2>&1 || exit 4
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out -o sq -- python3 bench.py $Q > $out/sq.log 2>&1 || exit 5
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT --output-format csv -d $out -o sq3 -- python3 bench.py $Q > $out/sq3.log 2>&1 || echo "sq3 pass failed (counter set not available)"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $out -o sq4 -- python3 bench.py $Q > $out/sq4.log 2>&1 || echo "sq4 pass failed (counter set not available)"
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for tagf in ("sq","sq3","sq4"):
    fs=glob.glob("$out/**/%s_counter_collection.csv"%tagf,recursive=True)
    if not fs: continue
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items():
        if any(s in k for s in ("render_","gaussian_backward","preprocess","tile_","emit_","scan_blocks")):
            for c,x in v.items(): res[k][c]=sum(x)/len(x)
json.dump(res,open("$out/sq_counters.json","w"),indent=1,sort_keys=True)
for k,v in res.items(): print(k[:50].ljust(50)," ".join("%s=%.4g"%(a.replace("SQ_",""),b) for a,b in sorted(v.items())))
PY
python3 profiles/pmc_to_json.py $out $out/pmc_traffic.json
for cfg in cfg1 cfg4; do python3 bench.py --config $cfg --no-cpu-baseline > $out/bench_$cfg.log 2>/dev/null; grep '^{' $out/bench_$cfg.log > $out/bench_$cfg.json; done
python3 bench.py --order morton --no-cpu-baseline > $out/bench_morton.log 2>/dev/null; grep '^{' $out/bench_morton.log > $out/bench_morton.json
python3 bench.py --full-step --steps 20 --warmup 4 --no-cpu-baseline > $out/bench_full.log 2>/dev/null; grep '^{' $out/bench_full.log > $out/bench_full_step.json
python3 tools/grad_error.py $out/grad_error.json > $out/grad_error.log 2>&1
find $out -name "*_counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.rocpd" -delete 2>/dev/null
rm -f $out/*.log.tmp
ls $out
