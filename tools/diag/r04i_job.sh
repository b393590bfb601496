#!/bin/bash
# round 4, job i: in-register cyclotomic squaring runs — pairing rate, verification rate, small-batch latency, then the pairing-family tests
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
{ timeout -k 10 200 python3 tools/bench_pairing.py 65536 2>&1 | grep batch
  timeout -k 10 200 python3 tools/bench_g16_batch_verify.py 65536 2>&1 | grep "proofs," 
  timeout -k 10 200 python3 tools/bench_pairing_small.py 2>&1 | grep "n=" | head -3; } | tee $O/r04i_cyc_runs.txt
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "pairing or tate or weil or miller or verify or signature or pinocchio_vs or dpairing or config3 or verification or smoke" > $O/r04i_tests.log 2>&1; tail -4 $O/r04i_tests.log
echo done
