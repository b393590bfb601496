#!/usr/bin/env python3
"""Timeline of the verification kernels of tools/bench_verify_latency.py from a rocprofv3 kernel_trace.csv: every kernel from the first k_fixed_table on, start/end in ms, queue."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkt::', ''), r.get('Queue_Id')) for r in rows)
want = ("k_fixed_table", "k_ate_key_prep", "k_stmt_wide", "dp::k_key_ab", "dp::k_dproduct", "k_short_loop_guards", "k_ate_guards", "k_shared", "k_product_resolve", "k_groth16_verify", "k_fixed_muls", "k_stmt_sums")
sel = [e for e in ev if e[2].startswith(want)]
base = None; n = 0
for s, e, name, q in sel:
    if base is None or s - last > 30e6: base = s; print("---")
    last = e
    print("%8.3f -> %8.3f  %-34s q=%s" % ((s - base) / 1e6, (e - base) / 1e6, name[:34], q))
    n += 1
    if n > int(sys.argv[2]) if len(sys.argv) > 2 else 120: break
