#!/bin/bash
# round 4, job c: the whole GPU suite, pairing A/B of the route pre-kernel, the Pinocchio soak, bench
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "=== pairing"; timeout -k 10 300 python3 tools/bench_pairing.py 65536 2>&1 | grep batch | tee $O/r04c_pairing.txt
echo "=== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > $O/r04c_gputests.log 2>&1; tail -25 $O/r04c_gputests.log
echo "=== soak"; ZKT_DTATE_MAX=0 ZKT_DPRODUCT_MAX=0 ZKT_MSM_GRAPH=0 ZKT_DEBUG_POISON=1 timeout -k 10 600 python3 tools/diag/pin_soak.py 200 > $O/r04c_pin_soak.txt 2>&1; tail -6 $O/r04c_pin_soak.txt
echo "=== bench"; timeout -k 10 600 python3 bench.py > $O/r04c_bench.json 2> $O/r04c_bench.err; tail -3 $O/r04c_bench.err; cut -c1-1500 $O/r04c_bench.json
echo done
