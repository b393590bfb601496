#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection CSV (SQ counters) and the VALU micro-benchmark (tools/ubench/valu_roof.hip) into the JSON that
bench.py reads for `roofline.valu` — so the VALU roofline fraction is reproducible from profiles/ alone, no constants typed into bench.py.
usage: sq_summary.py OUT.json UBENCH.txt NOTE counter_collection.csv [more.csv ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import csv, json, re, sys, collections
out, ubench, note, files = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                      "sgpr": int(r["SGPR_Count"]), "scratch_bytes_per_lane": int(r["Scratch_Size"])}
kern = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    d.update(meta[k])
    if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d and d["SQ_WAVES"]:
        d["valu_instr_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        d["valu_lane_instr_per_launch"] = d["SQ_INSTS_VALU"] * 64           # SQ_INSTS_VALU counts wave-instructions
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        d["wait_any_frac_of_wave_cycles"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    kern[k] = d
# ubench: best chip-wide rate per instruction over the occupancies measured
peaks = collections.defaultdict(float); at2 = {}
for line in open(ubench):
    m = re.match(r"(.+?)\s+waves/SIMD=(\d+)\s+([\d.]+) ms\s+clock ([\d.]+) GHz.*?([\d.]+) T lane-ops/s", line)
    if m:
        nm, w, rate = m.group(1).strip(), int(m.group(2)), float(m.group(5))
        peaks[nm] = max(peaks[nm], rate)
        if w == 2: at2[nm] = rate
pick = lambda frag, d: next((v for k, v in d.items() if frag in k), None)
from src_hash import kernel_sources_sha16
try: _head = subprocess.check_output(["git", "-C", sys_path_root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception: _head = None
res = {"source": note, "kernel_sources_sha16": kernel_sources_sha16(), "git_head_when_folded": _head,
       "units": "SQ_INSTS_* count wave-instructions (x64 = lane-instructions); SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_VALU count quad-cycles (MI355X_MICROARCH.md)",
       "valu_peak": {"int_mad_lane_ops_per_s_T": pick("v_mad_u64_u32", peaks), "int_mad_at_2_waves_per_simd_T": pick("v_mad_u64_u32", at2),
                     "field_mix_lane_ops_per_s_T": pick("field mix", peaks), "field_mix_at_2_waves_per_simd_T": pick("field mix", at2),
                     "f32_fma_lane_ops_per_s_T": pick("v_fma_f32", peaks), "f32_pk_fma_lane_fma_per_s_T": pick("v_pk_fma_f32", peaks),
                     "add_u32_lane_ops_per_s_T": pick("v_add_u32", peaks),
                     "note": "chip-wide issue rates measured by tools/ubench/valu_roof.hip (best over 1/2/4/8 waves per SIMD, all 256 CUs busy, in-kernel clock 2.0-2.3 GHz); "
                             "v_fma_f32 reaches ~28 lanes/clk/SIMD (the harness sees SIMD-32), v_mad_u64_u32 / v_mul_lo_u32 / v_lshrrev_b64 about half of that"},
       "kernels": kern}
ka = next((v for k, v in kern.items() if k.startswith("k_accumulate<PrimeOps<FqC")), None)
if ka: res["k_accumulate_g1"] = ka
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({"valu_peak": res["valu_peak"], "k_accumulate_g1": ka}, indent=1))
