#!/usr/bin/env python3
"""Average every counter of rocprofv3 --pmc counter_collection CSVs per kernel: usage pmc_fold.py file.csv [...] -> one line per kernel"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = (int(r["Grid_Size"]), int(r["VGPR_Count"]), int(r["Scratch_Size"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    s = " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items()))
    extra = ""
    if "SQ_INSTS_VALU" in d and d.get("SQ_WAVES"): extra += f" valu/wave={d['SQ_INSTS_VALU']/d['SQ_WAVES']:.0f}"
    if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"): extra += f" wait_any/wave_cycles={d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES']:.3f}"
    print(f"{k:60s} grid={meta[k][0]} vgpr={meta[k][1]} scratch={meta[k][2]} n={max(len(v) for v in acc[k].values())} {s}{extra}")
