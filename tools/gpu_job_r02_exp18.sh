# Pinocchio verification with fixed-base tables of the key's io points and the two product checks side by side (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_pinocchio.py tests/test_gpu_parity.py tests/test_gpu_protocols.py -m gpu -x -q > gpurun_out/exp18_tests.log 2>&1 || { tail -30 gpurun_out/exp18_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp18_tests.log
timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 6 > gpurun_out/exp18_pin.log 2>&1 || { tail gpurun_out/exp18_pin.log; exit 1; }
tail -1 gpurun_out/exp18_pin.log
ZKT_PINOCCHIO_FAST_VERIFY=0 timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 3 2>&1 | tail -1
