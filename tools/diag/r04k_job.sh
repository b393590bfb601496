#!/bin/bash
# round 4, job k: in-place line multiplications — small forced batches first, then the rates, then the deciding entry points' tests (stops at the first failure)
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
ZKT_DTATE_MAX=0 ZKT_DPRODUCT_MAX=0 timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "verify_batch_mixed or verify_batch_matches or pairing_product_check or test_tate_batch_vs_oracle" -p no:cacheprovider > $O/r04k_forced.log 2>&1 || { tail -20 $O/r04k_forced.log; exit 1; }
tail -2 $O/r04k_forced.log
{ timeout -k 10 200 python3 tools/bench_g16_batch_verify.py 65536 2>&1 | grep "proofs," || exit 1
  timeout -k 10 200 python3 tools/bench_pairing.py 65536 2>&1 | grep batch || exit 1; } | tee $O/r04k_inplace_lines.txt
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "verify or signature or pinocchio_vs or pairing_product or verification or outside or tate or weil or miller or dpairing or config3" > $O/r04k_tests.log 2>&1; tail -4 $O/r04k_tests.log
echo done
