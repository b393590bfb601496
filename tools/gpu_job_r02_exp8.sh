# 127-step loop in the lane-distributed (small-batch) kernels with the guard kernel on a side stream: smoke, full suite, latencies (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_dpairing.py tests/test_gpu_parity.py -m gpu -x -q -k "small_batch_tate or outside_the_subgroup or distributed" > gpurun_out/exp8_smoke.log 2>&1 || { tail -30 gpurun_out/exp8_smoke.log; echo "smoke FAILED"; exit 1; }
tail -2 gpurun_out/exp8_smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/exp8_tests.log 2>&1 || { tail -30 gpurun_out/exp8_tests.log; echo "tests FAILED"; exit 1; }
tail -3 gpurun_out/exp8_tests.log
timeout -k 10 300 python3 tools/bench_pairing_small.py > gpurun_out/exp8_small.log 2>&1 || { tail gpurun_out/exp8_small.log; exit 1; }
cat gpurun_out/exp8_small.log
timeout -k 10 300 python3 tools/bench_verify_latency.py > gpurun_out/exp8_verify.log 2>&1 || { tail gpurun_out/exp8_verify.log; exit 1; }
cat gpurun_out/exp8_verify.log
