#!/usr/bin/env python3
"""The twelve longest HIP API calls of the last 600 ms of a rocprofv3 --hip-trace run (what a slow call was waiting in).  usage: long_api_calls.py <dir with *_hip_api_trace.csv>"""
import csv,glob,sys
d=sys.argv[1]
at=list(csv.DictReader(open(glob.glob(d+"/*hip_api_trace.csv")[0])))
api=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Function']) for r in at)
t_end=api[-1][1]
tail=[a for a in api if a[0]>t_end-600e6]
for s,e,n in sorted(tail,key=lambda a:-(a[1]-a[0]))[:12]:
    print("%10.3f ms before end: %-28s %9.3f ms"%((t_end-s)/1e6,n,(e-s)/1e6))
