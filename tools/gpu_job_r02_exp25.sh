# batch public keys through the G1 generator's comb table (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_signature.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/exp25_tests.log 2>&1 || { tail -30 gpurun_out/exp25_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp25_tests.log
timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp25_protocols.json 2> gpurun_out/exp25_protocols.err || { tail gpurun_out/exp25_protocols.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/exp25_protocols.json'))
print('bls', {k:v for k,v in d['bls'].items() if k!='cpu_baseline'})"
