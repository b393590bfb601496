# fixed-base tables for the single-point products of the Bulletproofs range proof: parity, then timings (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_protocols.py tests/test_gpu_fullsize.py -m gpu -x -q -k "range or bulletproof or ipa or bp" > gpurun_out/exp10_tests.log 2>&1 || { tail -30 gpurun_out/exp10_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp10_tests.log
timeout -k 10 300 python3 tools/bench_bp.py > gpurun_out/exp10_bp.log 2>&1 || { tail gpurun_out/exp10_bp.log; exit 1; }
cat gpurun_out/exp10_bp.log
