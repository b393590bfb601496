#!/usr/bin/env python3
"""Host side of ONE proof beside its kernels: from a rocprofv3 --hip-trace --kernel-trace run, the HIP API calls (launches, graph launches, event waits) and the kernels
from the last k_to_mont launch on, on one clock.  usage: host_timeline.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv> [max lines] [marker kernel | gap]"""
import csv, glob, sys
d = sys.argv[1]; lim = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kt = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
at = list(csv.DictReader(open(glob.glob(d + "/*hip_api_trace.csv")[0])))
kern = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkt::', ''), r.get('Queue_Id')) for r in kt)
marker = sys.argv[3] if len(sys.argv) > 3 else 'k_to_mont'
if marker == 'gap':                                   # the last burst of kernels: from the kernel that follows the last pause of more than 1 ms
    t0 = kern[0][0]; end = kern[0][1]
    for s_, e_, n_, q_ in kern:
        if s_ - end > 1000000: t0 = s_
        end = max(end, e_)
else:
    t0 = [k for k in kern if k[2].startswith(marker)][-1][0]
api = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']) for r in at)
first_api = [a for a in api if a[0] <= t0][-40:]          # the launch of k_to_mont lies a little before its start
base = t0
ev = [("K", s, e, n + " q=" + str(q)) for s, e, n, q in kern if s >= t0] + [("H", s, e, n) for s, e, n in api if s >= t0 - 300000]
ev.sort(key=lambda x: x[1])
keep = ("hipGraphLaunch", "hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipStreamWaitEvent", "hipEventRecord", "hipStreamSynchronize", "hipMemcpyAsync", "hipEventSynchronize", "hipDeviceSynchronize")
n = 0
for kind, s, e, name in ev:
    if kind == "H" and not name.startswith(keep): continue
    if kind == "H" and (e - s) < 15000 and not name.startswith(("hipGraphLaunch", "hipStreamSynchronize")): continue      # only host calls that take time
    print("%s %8.3f -> %8.3f  %s" % (kind, (s - base) / 1e6, (e - base) / 1e6, name[:60]))
    n += 1
    if n >= lim: break
