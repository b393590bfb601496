# idle lanes keep executing in the latency-bound kernels (trees, joins, one-lane programs): parity, then latencies (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_protocols.py tests/test_gpu_pinocchio.py tests/test_gpu_signature.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/exp23_tests.log 2>&1 || { tail -30 gpurun_out/exp23_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp23_tests.log
timeout -k 10 300 python3 tools/bench_msm_latency.py 2>&1 | tail -4
timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 5 2>&1 | tail -1
timeout -k 10 300 python3 tools/bench_verify_latency.py 2>&1 | tail -4
timeout -k 10 600 python3 bench.py --no-cpu --pairings 0 --groth16-log2n 0 --steps 10 > gpurun_out/exp23_bench.json 2> gpurun_out/exp23_bench.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/exp23_bench.json').read().strip().splitlines()[-1]); print('msm', d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], d['config']['one_shot_msm_ms']); print('bp', {k:v for k,v in d['bulletproofs'].items() if not isinstance(v,str)})"
