#!/bin/bash
# round-5 evidence (GPU box, repo root): tools/r5_collect.sh <tag>  ->  gpurun_out/<tag>/
#   bench.json (default bench line), stats_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command), separate --pmc
#   passes (FETCH_SIZE / WRITE_SIZE / SQ counters; never combined with a trace domain other than the kernel trace), pmc_traffic.json,
#   sq_counters.json, the VALU-active calibration on the probe loops, bench lines of cfg1 / cfg4 / Morton order / the full training
#   step / the blend log switched off, grad_error.json, the micro-benchmarks' own output
set -o pipefail
tag=${1:-r05_v1}
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
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --output-format csv -d $out -o sq3 -- python3 bench.py $Q > $out/sq3.log 2>&1 || echo "sq3 pass failed (counter set not available)"
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
# what does SQ_ACTIVE_INST_VALU count?  The probe's loops are known-busy: 8 waves per SIMD issuing one instruction class back to back
if [ -x tools/salu_probe ]; then
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o calib -- ./tools/salu_probe > $out/salu_probe.txt 2>&1
  python3 - <<PY
import csv,glob,collections,json
fs=glob.glob("$out/**/calib_counter_collection.csv",recursive=True)
kt=glob.glob("$out/**/calib_kernel_trace.csv",recursive=True)
dur=collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])): dur[r["Kernel_Name"]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
rows={}
if fs:
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])): per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items():
        d={c:sum(x)/len(x) for c,x in v.items()}
        ns=sum(dur[k])/max(len(dur[k]),1)
        d["duration_ns"]=ns
        if d.get("SQ_INSTS_VALU",0)>0: d["active_per_valu_inst"]=d["SQ_ACTIVE_INST_VALU"]/d["SQ_INSTS_VALU"]
        if ns>0 and "SQ_ACTIVE_INST_VALU" in d: d["active_quadcycles_x4_per_simd_cycle_at_2.4GHz"]=4*d["SQ_ACTIVE_INST_VALU"]/(1024*ns*2.4)
        rows[k]=d
json.dump(rows,open("$out/valu_active_calibration.json","w"),indent=1,sort_keys=True)
for k,d in rows.items(): print(k[:40].ljust(40), {a:round(b,3) for a,b in d.items() if a in ("active_per_valu_inst","active_quadcycles_x4_per_simd_cycle_at_2.4GHz","SQ_INSTS_VALU","SQ_INSTS_SALU","duration_ns")})
PY
fi
[ -x tools/smem_probe ] && ./tools/smem_probe > $out/smem_probe.txt 2>&1
for cfg in cfg1 cfg4; do python3 bench.py --config $cfg --no-cpu-baseline > $out/bench_$cfg.log 2>/dev/null; grep '^{' $out/bench_$cfg.log > $out/bench_$cfg.json; done
GSR_BLEND_LOG=0 python3 bench.py --no-cpu-baseline > $out/bench_nolog.log 2>/dev/null; grep '^{' $out/bench_nolog.log > $out/bench_blend_log_off.json
python3 bench.py --order morton --no-cpu-baseline > $out/bench_morton.log 2>/dev/null; grep '^{' $out/bench_morton.log > $out/bench_morton.json
python3 bench.py --full-step --steps 20 --warmup 4 --no-cpu-baseline > $out/bench_full.log 2>/dev/null; grep '^{' $out/bench_full.log > $out/bench_full_step.json
python3 tools/grad_error.py $out/grad_error.json > $out/grad_error.log 2>&1
prof=tools/scratch/ab_r5p_1/based_on_point_cloud_partitions_amd/csrc/libgsrast_mi355x.so
[ -f $prof ] && GSR_LIBRARY=$PWD/$prof python3 tools/r5_phase.py cfg2 > $out/phase_profile.json 2>/dev/null
find $out -name "*_counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.rocpd" -delete 2>/dev/null
rm -f $out/*.log.tmp
ls $out
