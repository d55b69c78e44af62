#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of profiles/collect.sh into profiles/pmc_traffic.json.

HBM traffic per launch and stage = 2 x FETCH_SIZE + WRITE_SIZE.  Units and correction follow
MI355X_MICROARCH.md §HBM: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests
of 16-byte-per-lane loads at 64 bytes, i.e. exactly half the bytes, so it is doubled (every global load of the hot
kernels is a 16-byte-per-lane load); WRITE_SIZE is exact for 16-byte stores.  Counters are averaged over the launches
of each kernel in the timed bench and the kernels of one stage are summed.
usage: pmc_to_json.py <dir with fetch_counter_collection.csv / write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

STAGE = {
    "preprocess_kernel": "preprocess", "scan_blocks_kernel": "scan", "emit_keys_kernel": "emit_keys",
    "tile_count_kernel": "radix_sort", "tile_colscan_kernel": "radix_sort", "tile_starts_kernel": "radix_sort",
    "emit_partition_kernel": "emit_keys", "band_partition_kernel": "radix_sort", "tile_count_banded_kernel": "radix_sort", "tile_colscan_banded_kernel": "radix_sort",
    "tile_scatter_banded_kernel": "radix_sort", "tile_sort_mid_kernel": "radix_sort",
    "tile_scatter_kernel": "radix_sort", "tile_sort_small_kernel": "radix_sort", "tile_sort_large_kernel": "radix_sort", "global_hist_kernel": "radix_sort",
    "block_hist_kernel": "radix_sort", "bin_scan_kernel": "radix_sort", "scatter_kernel": "radix_sort",
    "tile_ranges_kernel": "tile_ranges", "render_forward_kernel": "render_forward", "render_backward_kernel": "render_backward",
    "render_backward_log_kernel": "render_backward",
    "gaussian_backward_kernel": "gaussian_backward",
}


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        per[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}


def main(d, out):
    f = load(f"{d}/fetch_counter_collection.csv", "FETCH_SIZE")
    w = load(f"{d}/write_counter_collection.csv", "WRITE_SIZE")
    res = collections.defaultdict(lambda: {"fetch_size_kib": 0.0, "write_size_kib": 0.0})
    launches_per_step = {"block_hist_kernel": 6, "bin_scan_kernel": 6, "scatter_kernel": 6}
    # the LSD radix kernels also sort the Morton codes of the kNN initialisation (once per run): they belong to the step only
    # when the step sorts with them (GSR_BIN=0), i.e. when no tile-binning kernel was launched
    radix = ("global_hist_kernel", "block_hist_kernel", "bin_scan_kernel", "scatter_kernel")
    binned = "tile_scatter_banded_kernel" in f or "tile_scatter_kernel" in f
    for k, st in STAGE.items():
        if binned and k in radix:
            continue
        mult = launches_per_step.get(k, 1)
        if k in f:
            res[st]["fetch_size_kib"] += f[k] * mult
        if k in w:
            res[st]["write_size_kib"] += w[k] * mult
    import os
    sqp = f"{d}/sq_counter_collection.csv"
    sq = {c: load(sqp, c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")} if os.path.exists(sqp) else {}
    # instruction classes (separate passes sq3 / sq4 of tools/r4_collect.sh): what bench.py prices the VALU-bound kernels with
    for extra, names in (("sq3", ("SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")),
                         ("sq4", ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_INT32"))):
        ep = f"{d}/{extra}_counter_collection.csv"
        if os.path.exists(ep):
            sq.update({c: load(ep, c) for c in names})
    # ... or, when the raw CSVs are gone, the per-kernel summary tools/r4_collect.sh keeps (sq_counters.json)
    sj = f"{d}/sq_counters.json"
    if os.path.exists(sj):
        per_kernel = json.load(open(sj))
        for c in ("SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                  "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES",
                  "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if c not in sq or not sq[c]:
                sq[c] = {k.split("<")[0]: v[c] for k, v in per_kernel.items() if c in v}
    for c, per in sq.items():
        for k, st in STAGE.items():
            if binned and k in radix:
                continue
            if k in per:
                res[st][c.lower()] = res[st].get(c.lower(), 0.0) + per[k] * launches_per_step.get(k, 1)
    outd = {}
    for st, v in res.items():
        outd[st] = {"hbm_bytes": int((2 * v["fetch_size_kib"] + v["write_size_kib"]) * 1024), **{a: round(b, 1) for a, b in v.items()},
                    "formula": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes); FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM"}
    # which template instance ran the blend stages (bench.price_list_cycles looks up the instance's static instruction shares)
    if os.path.exists(sj):
        for key in json.load(open(sj)):
            st = STAGE.get(key.split("<")[0])
            if st in ("render_forward", "render_backward") and st in outd:
                outd[st]["kernel_name"] = key.strip()
    # provenance: bench.py attaches these figures to a live run only when the run's workload and kernel build match
    metap = f"{d}/bench.json"
    if os.path.exists(metap):
        try:
            b = json.loads(open(metap).read().strip().splitlines()[-1])
            outd["_meta"] = {"workload": b["config"]["workload"], "build": b.get("build"), "source": d}
        except Exception:
            pass
    json.dump(outd, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["hbm_bytes"] for k, v in outd.items() if k != "_meta"}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
