# the GPU suite with one guard stream and the single-stream Pinocchio verifier, free-memory log per test (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
rm -f gpurun_out/diag5_meminfo.txt
ZKT_TEST_MEMINFO=gpurun_out/diag5_meminfo.txt timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/diag5_tests.log 2>&1 || { head -5 gpurun_out/diag5_tests.log | cut -c1-300; tail -3 gpurun_out/diag5_meminfo.txt; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/diag5_tests.log
tail -2 gpurun_out/diag5_meminfo.txt
timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 6 2>&1 | tail -1
