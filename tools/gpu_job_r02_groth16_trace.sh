# where one Groth16 proof at 2^20 constraints spends its time: kernel statistics of tools/bench_groth16.py (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g16 -- python3 tools/bench_groth16.py --log-n 20 --proofs 8 > gpurun_out/prof_g16.log 2>&1 || { tail gpurun_out/prof_g16.log; exit 1; }
grep -v "^[WEI]2026" gpurun_out/prof_g16.log | tail -4
python3 tools/trace_timeline.py $(ls gpurun_out/prof_g16/*/*kernel_trace.csv | head -1) 3 140 > gpurun_out/prof_g16_timeline.txt 2>&1 || true
tail -5 gpurun_out/prof_g16_timeline.txt
