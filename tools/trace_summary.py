#!/usr/bin/env python3
"""Group a rocprofv3 kernel_trace.csv by (kernel, grid size): count, average and minimum duration in microseconds.
usage: trace_summary.py <kernel_trace.csv> [min_total_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkt::', '').replace('zkt_g2pair::', 'g2pair::')
    grid = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
    d[(name, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("# kernel, grid (work-items), launches, average and minimum duration in microseconds; sorted by total time")
for (name, grid), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-46s grid %9d  n %4d  avg %9.1f us  min %9.1f us" % (name[:46], grid, len(v), sum(v) / len(v), min(v)))
