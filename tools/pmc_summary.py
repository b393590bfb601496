#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter) into the per-kernel JSON kept under profiles/.
usage: pmc_summary.py OUT.json NOTE FETCH_counter_collection.csv WRITE_counter_collection.csv"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import csv, json, sys, collections
out, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkt::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for k, cs in acc.items():
    kern[k] = {}
    for c, v in cs.items():
        kern[k][c + "_KB_avg_per_launch"] = sum(v) / len(v); kern[k]["launches_" + c] = len(v)
from src_hash import kernel_sources_sha16
try: _head = subprocess.check_output(["git", "-C", sys_path_root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception: _head = None
res = {"source": note, "kernel_sources_sha16": kernel_sources_sha16(), "git_head_when_folded": _head,
       "units": "KB as reported by rocprofv3; bytes = KB*1024.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide coalesced "
                "streaming read; k_accumulate is a 112-byte-per-lane gather (uncalibrated pattern), so the uncorrected figure is a lower bound and 2x an upper bound.",
       "kernels": kern}
ka = next((v for k, v in kern.items() if k.startswith("k_accumulate<PrimeOps<FqC")), None)
if ka and "FETCH_SIZE_KB_avg_per_launch" in ka and "WRITE_SIZE_KB_avg_per_launch" in ka:
    b = (ka["FETCH_SIZE_KB_avg_per_launch"] + ka["WRITE_SIZE_KB_avg_per_launch"]) * 1024
    res["k_accumulate_hbm_bytes_per_launch"] = {"uncorrected": b, "fetch_doubled_upper_bound": b + ka["FETCH_SIZE_KB_avg_per_launch"] * 1024}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("k_accumulate_hbm_bytes_per_launch")))
