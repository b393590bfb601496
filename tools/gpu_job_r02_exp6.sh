# 127-step Miller loop: smoke run at a small batch, parity of every pairing consumer, then throughput (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
ZKT_DTATE_MAX=0 timeout -k 10 120 python3 tools/bench_pairing.py 256 > gpurun_out/exp6_smoke.log 2>&1 || { tail -5 gpurun_out/exp6_smoke.log; echo "smoke FAILED"; exit 1; }
cat gpurun_out/exp6_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pairing or tate or outside or bilinear or miller or weil or fullsize or large_batch" > gpurun_out/exp6_tests.log 2>&1 || { tail -30 gpurun_out/exp6_tests.log; echo "tests FAILED"; exit 1; }
tail -3 gpurun_out/exp6_tests.log
for m in 65536 262144; do timeout -k 10 300 python3 tools/bench_pairing.py $m >> gpurun_out/exp6.log 2>> gpurun_out/exp6.err || exit 1; done
cat gpurun_out/exp6.log
