# Pinocchio verification with its five equalities in one launch of the lane-distributed product kernel (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_pinocchio.py tests/test_gpu_signature.py tests/test_gpu_dpairing.py -m gpu -x -q > gpurun_out/exp19_tests.log 2>&1 || { tail -30 gpurun_out/exp19_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp19_tests.log
timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 6 2>&1 | tail -1
