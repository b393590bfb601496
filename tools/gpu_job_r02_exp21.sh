# re-measure the single Groth16 verification after the comb-table change (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/bench_groth16.py --log-n 20 --proofs 4 2>&1 | grep "setup\|prove" || exit 1
timeout -k 10 600 python3 tools/bench_groth16.py --log-n 16 --proofs 4 2>&1 | grep "setup\|prove" || exit 1
timeout -k 10 600 python3 bench.py --no-cpu --pairings 0 --no-bulletproofs --steps 10 > gpurun_out/exp21_bench.json 2> gpurun_out/exp21_bench.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/exp21_bench.json').read().strip().splitlines()[-1]); g=d['groth16']; print('bench groth16', g['value'], g['ms_per_proof'], g['setup_s'], g['verify_first_call_ms'], g['verify_ms'], g['verifies'])"
