# fixed-base tables for a verifying key's statement points: parity of the Groth16 consumers, then the single-verification latency (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "groth16 or verify or pinocchio or large_batch" > gpurun_out/exp12_tests.log 2>&1 || { tail -30 gpurun_out/exp12_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp12_tests.log
timeout -k 10 300 python3 tools/bench_verify_latency.py > gpurun_out/exp12_verify.log 2>&1 || exit 1
grep -v "^/opt" gpurun_out/exp12_verify.log
timeout -k 10 600 python3 tools/bench_groth16.py --log-n 20 --proofs 6 > gpurun_out/exp12_g16.log 2>&1 || exit 1
tail -2 gpurun_out/exp12_g16.log
