#!/usr/bin/env python3
"""Print a per-queue timeline / overlap summary from a rocprofv3 kernel_trace.csv (pipelined MSM analysis)."""
import csv, glob, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkt::', '')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id')))
ev.sort()
acc = [e for e in ev if e[2].startswith('k_accumulate')]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = acc[-nlast][0]; t1 = max(e[1] for e in ev)
if len(sys.argv) > 4: t0 = [e for e in ev if e[2].startswith(sys.argv[4])][-1][0]      # window from the last launch of this kernel (e.g. k_to_mont = start of a proof)
sel = [e for e in ev if e[0] >= t0]
print("window ms %.3f for %d accumulates -> %.3f ms per MSM" % ((t1 - t0) / 1e6, nlast, (acc[-1][0] - acc[-nlast][0]) / 1e6 / (nlast - 1)))
d = collections.defaultdict(list)
for s, e, n, q in sel: d[n].append((e - s) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-34s n=%3d avg=%9.1f us total=%8.2f ms" % (n[:34], len(v), sum(v) / len(v), sum(v) / 1e3))
base = sel[0][0]
for s, e, n, q in sel[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    if n.startswith('__amd') or n.startswith('k_scan'): continue
    print("%8.3f -> %8.3f  %-24s q=%s" % ((s - base) / 1e6, (e - base) / 1e6, n[:24], q))
