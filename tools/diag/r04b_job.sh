#!/bin/bash
# round 4, job b: new parity tests, G2 affine-rounds A/B, no-IPRA A/B, Groth16 A/B
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -k "outside_the_subgroup or verify_batch_mixed or verify_batch_matches or pinocchio_verify_g1 or pairing_product_check" > $O/r04b_tests.log 2>&1 || { tail -60 $O/r04b_tests.log; }
tail -5 $O/r04b_tests.log
echo "=== G2 MSM A/B"; for R in 0 1 2 3; do ZKT_G2_AFFINE_ROUNDS=$R timeout -k 10 300 python3 tools/bench_g2_msm.py 20 8 2>&1 | grep "G2 MSM" | sed "s/^/rounds=$R /" | tee -a $O/r04b_g2_ab.txt; done || exit 1
echo "=== pairing no-IPRA A/B"; timeout -k 10 300 python3 tools/bench_pairing.py 65536 2>&1 | grep batch | tee $O/r04b_ipra_ab.txt || exit 1
ZKT_LIB_PATH=$PWD/zk-toolkit_amd/libzkt_hip_noipra.so timeout -k 10 300 python3 tools/bench_pairing.py 65536 2>&1 | grep batch | tee -a $O/r04b_ipra_ab.txt || exit 1
echo "=== groth16 A/B"; for R in 0 2; do ZKT_G2_AFFINE_ROUNDS=$R timeout -k 10 400 python3 tools/bench_groth16.py --log-n 20 --proofs 6 2>&1 | grep "prove:" | sed "s/^/rounds=$R /" | tee -a $O/r04b_g16_ab.txt; done
echo done
