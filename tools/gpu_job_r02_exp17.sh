# does the number of hardware queues the HIP runtime multiplexes streams onto matter for the pipelined prover? (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python3 bench.py --no-cpu --pairings 0 --no-bulletproofs --steps 10 > gpurun_out/exp17_bench_$q.json 2> gpurun_out/exp17_bench_$q.err || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/exp17_bench_$q.json').read().strip().splitlines()[-1]); g=d['groth16']; print('queues $q: msm', d['value'], d['ms_per_step'], 'groth16', g['value'], g['ms_per_proof'], g['verifies'])"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python3 tools/bench_groth16.py --log-n 20 --proofs 8 2>&1 | grep "prove:" || exit 1
done
