# the range proof's vector stage in one launch: parity, then timings with and without it (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_protocols.py tests/test_gpu_fullsize.py -m gpu -x -q -k "range or bulletproof or ipa or bp" > gpurun_out/exp14_tests.log 2>&1 || { tail -30 gpurun_out/exp14_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp14_tests.log
ZKT_RP_FUSED=0 timeout -k 10 600 python -m pytest tests/test_gpu_protocols.py -m gpu -x -q -k "range" > gpurun_out/exp14_tests_legacy.log 2>&1 || { tail -30 gpurun_out/exp14_tests_legacy.log; echo "legacy tests FAILED"; exit 1; }
tail -1 gpurun_out/exp14_tests_legacy.log
timeout -k 10 300 python3 tools/bench_bp.py > gpurun_out/exp14_bp.log 2>&1 || { tail gpurun_out/exp14_bp.log; exit 1; }
grep "range\|IPA" gpurun_out/exp14_bp.log
ZKT_RP_FUSED=0 timeout -k 10 300 python3 tools/bench_bp.py > gpurun_out/exp14_bp_legacy.log 2>&1 || exit 1
echo "step by step:"; grep "resident" gpurun_out/exp14_bp_legacy.log
