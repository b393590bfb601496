#!/bin/bash
# round 4, job e: coalesced one-block scans (parity + latencies), compiler-flag sweep of the pairing object
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "msm or range_proof or inner_product or groth16_r1cs or r1cs or config5 or pinocchio_resident or one_constraint" > $O/r04e_tests.log 2>&1; tail -4 $O/r04e_tests.log
echo "=== bp"; timeout -k 10 300 python3 tools/bench_bp.py > $O/r04e_bp.log 2>&1; tail -5 $O/r04e_bp.log
echo "=== small groth16"; for LN in 12 16; do timeout -k 10 300 python3 tools/bench_groth16.py --log-n $LN --proofs 20 2>&1 | grep "prove:" | sed "s/^/2^$LN /"; done | tee $O/r04e_g16_small.txt
echo "=== msm latency"; timeout -k 10 300 python3 tools/bench_msm_latency.py 2>&1 | grep -i latency | tee $O/r04e_msm_latency.txt
echo "=== pairing flag sweep"; for v in "" _tv_maxilp _tv_maxmem _tv_O2 _tv_Os; do ZKT_LIB_PATH=$PWD/zk-toolkit_amd/libzkt_hip$v.so timeout -k 10 300 python3 tools/bench_pairing.py 65536 2>&1 | grep batch; done | tee $O/r04e_pairing_flags.txt
echo done
