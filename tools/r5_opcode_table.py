#!/usr/bin/env python3
"""Static per-opcode table of the blend kernels: disassembles nothing -- it compiles render_backward.hip /
render_forward.hip to assembly exactly as the Makefile does and counts the instructions of each kernel by opcode and by issue class
(plain fp32 1.06 ns, other VALU 1.75, transcendental / permlane swap 3.6, SALU 1.75).  Static counts: every path of the
kernel once; the dynamic totals are the PMC class counters of profiles/r05_v1_sq_counters.json.
usage: python tools/r5_opcode_table.py [profiles/blend_kernels_issue_shares.json] > profiles/r05_blend_kernels_opcodes.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "based_on_point_cloud_partitions_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-pass-failed".split()
PLAIN = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mov_b32", "v_max_f32", "v_min_f32"}
TRANS = {"v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_permlane16_swap_b32", "v_permlane32_swap_b32"}


def klass(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if "row_" in line or "quad_perm" in line or base.endswith("_dpp") or op.endswith("_dpp"):
            return "valu dpp (1.75 ns)"
        if base in TRANS:
            return "valu trans / swap (3.6 ns)"
        if base in PLAIN and not re.search(r"\bs\d+|\bs\[|vcc|exec", line.split(base, 1)[1].split(";")[0]):
            # three DISTINCT vector registers read (v_fma_f32 a, b, c; v_fmac_f32 reads its destination too): 1.7 ns, not 0.95
            # (tools/salu_probe.hip, profiles/r05_issue_probe_v3.txt); two registers + a literal / inline constant: 0.95
            args = [a.strip().lstrip("-|").rstrip("|") for a in line.split(None, 1)[1].split(";")[0].split(",")]
            srcs = args if base in ("v_fmac_f32",) else args[1:]
            if base in ("v_fma_f32", "v_fmac_f32") and len({a for a in srcs if re.match(r"^v\d+$", a)}) >= 3:
                return "valu fp32 fma, 3 registers (1.7 ns)"
            return "valu plain fp32 (1.06 ns)"
        return "valu other (1.75 ns)"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_load") or op.startswith("s_store") or op.startswith("s_dcache") or op.startswith("s_memtime"):
        return "smem"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm"):
        return "wait / nop / barrier"
    if op.startswith("s_"):
        return "salu (1.75 ns)"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    return "other"


def main():
    shares, dpp_shares = {}, {}
    for src in ("render_backward.hip", "render_forward.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ['-DGSR_EXTRA_FLAGS=""', "-I" + CS, "-S", "--cuda-device-only", "-o", out, os.path.join(CS, src)],
                           check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
        kernels = re.split(r"\n(?=_ZN[^\n]*:\s*; @)", text)
        for k in kernels:
            m = re.match(r"(_ZN\S+):", k)
            if not m or "render_" not in m.group(1):
                continue
            full = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            mm = re.search(r"render_\w+<[^>]*>", full)
            name = mm.group(0) if mm else full[:80]
            body = k.split("s_endpgm")[0]
            ops, cls = collections.Counter(), collections.Counter()
            for line in body.splitlines():
                line = line.strip()
                if not line or line.startswith((";", ".", "_Z")) or line.endswith(":"):
                    continue
                op = line.split()[0]
                if not re.match(r"^[vs]_|^ds_|^global_|^flat_|^buffer_", op):
                    continue
                ops[op] += 1
                cls[klass(op, line)] += 1
            print(f"== {name}: {sum(ops.values())} instructions (static)")
            for c, n in cls.most_common():
                print(f"   {c:32s} {n:5d}")
            fma_all = sum(n for o, n in ops.items() if re.match(r"v_(fma|fmac|fmaak|fmamk)_f32", o))
            fma3 = cls.get("valu fp32 fma, 3 registers (1.7 ns)", 0)
            shares[name] = round(fma3 / max(fma_all, 1), 3)
            add_all = sum(n for o, n in ops.items() if re.match(r"v_(add|sub|subrev)_f32", o))
            dpp_shares[name] = round(ops.get("v_add_f32_dpp", 0) / max(add_all, 1), 3)
            print(f"   fp32 adds with a DPP modifier: {ops.get('v_add_f32_dpp', 0)} of {add_all}")
            print(f"   fp32 fma instructions reading three registers: {fma3} of {fma_all}")
            print("   top opcodes: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(24)))
            print()
    if len(sys.argv) > 1:  # static share of three-register fmas per kernel, for bench.price_list_cycles
        import json
        json.dump({"three_register_share_of_fp32_fma": shares, "dpp_share_of_fp32_add": dpp_shares,
                   "source": "static instruction listing (tools/r5_opcode_table.py); issue costs: three-register fma 1.7 ns against 0.95 ns, DPP add 1.76 ns "
                             "(profiles/r05_issue_probe_v3.txt)"},
                  open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
