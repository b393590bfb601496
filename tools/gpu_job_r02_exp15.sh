# Groth16 prover A/B job: tests, sequential proofs (tools/bench_groth16.py) and the pipelined bench leg (round 2; used for the submit-order and stream-priority trials, both neutral)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "groth16" > gpurun_out/exp15_tests.log 2>&1 || { tail -30 gpurun_out/exp15_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp15_tests.log
timeout -k 10 600 python3 tools/bench_groth16.py --log-n 20 --proofs 8 > gpurun_out/exp15_g16.log 2>&1 || exit 1
tail -2 gpurun_out/exp15_g16.log
timeout -k 10 600 python3 bench.py --no-cpu --pairings 0 --no-bulletproofs --steps 10 > gpurun_out/exp15_bench.json 2> gpurun_out/exp15_bench.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/exp15_bench.json').read().strip().splitlines()[-1]); g=d['groth16']; print('bench groth16', g['value'], g['ms_per_proof'], g['verifies'])"
