# where a Pinocchio verification spends its time (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/bench_pinocchio.py > gpurun_out/pin.log 2>&1 || { tail gpurun_out/pin.log; exit 1; }
tail -1 gpurun_out/pin.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pin -- python3 tools/bench_pinocchio.py --reps 2 > gpurun_out/prof_pin.log 2>&1 || { tail gpurun_out/prof_pin.log; exit 1; }
python3 - <<'PY' > gpurun_out/prof_pin_timeline.txt
import csv, glob
f = glob.glob('gpurun_out/prof_pin/*/*kernel_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkt::', ''), r['Queue_Id']) for r in csv.DictReader(open(f)))
t1 = ev[-1][1]
sel = [e for e in ev if e[0] > t1 - 80e6]
b = sel[0][0]
for s, e, n, q in sel: print("%8.3f -> %8.3f %7.3f  %-40s q=%s" % ((s - b) / 1e6, (e - b) / 1e6, (e - s) / 1e6, n[:40], q))
PY
tail -5 gpurun_out/prof_pin_timeline.txt
