#!/bin/bash
# round 4, job d: one-block scans for small MSMs — parity of the MSM / protocol tests, latency of the small protocols before / after is in the bench's bulletproofs leg
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "msm or range_proof or inner_product or groth16_r1cs or r1cs or config5 or pinocchio_resident or one_constraint" > $O/r04d_tests.log 2>&1; tail -6 $O/r04d_tests.log
echo "=== bp"; timeout -k 10 300 python3 tools/bench_bp.py > $O/r04d_bp.log 2>&1; tail -8 $O/r04d_bp.log
echo "=== small groth16"; for LN in 12 16; do timeout -k 10 300 python3 tools/bench_groth16.py --log-n $LN --proofs 20 2>&1 | grep "prove:" | sed "s/^/2^$LN /"; done | tee $O/r04d_g16_small.txt
echo "=== msm latency"; timeout -k 10 300 python3 tools/bench_msm_latency.py 2>&1 | grep -i latency | tee $O/r04d_msm_latency.txt
echo "=== verify latency"; timeout -k 10 300 python3 tools/bench_verify_latency.py 2>&1 | grep -v "^[WEI]2026" | tee $O/r04d_verify_latency.txt
echo done
