# k_merge_partials one bucket per lane: MSM parity (skewed inputs split buckets) and single-MSM latencies (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "msm or fullsize or range or ipa or groth16_r1cs or sharded or comm" > gpurun_out/exp11_tests.log 2>&1 || { tail -30 gpurun_out/exp11_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp11_tests.log
timeout -k 10 300 python3 tools/bench_msm_latency.py > gpurun_out/exp11_lat.log 2>&1 || { tail gpurun_out/exp11_lat.log; exit 1; }
grep -v "^/opt" gpurun_out/exp11_lat.log
timeout -k 10 300 python3 tools/bench_bp.py > gpurun_out/exp11_bp.log 2>&1 || exit 1
grep "resident" gpurun_out/exp11_bp.log
