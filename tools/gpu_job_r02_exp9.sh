# shared G2 points tested once per launch in the Groth16 batch verification: parity (both kernel families), then throughput (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_protocols.py tests/test_gpu_dpairing.py -m gpu -x -q -k "verify or large_batch or groth16" > gpurun_out/exp9_tests.log 2>&1 || { tail -30 gpurun_out/exp9_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp9_tests.log
timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp9_protocols.json 2> gpurun_out/exp9_protocols.err || exit 1
python3 -c "
import json; q=json.load(open('gpurun_out/exp9_protocols.json'))
print('bls verify', round(q['bls']['verify_per_s']), 'g16 verify batch', round(q['groth16_verify_batch']['verifications_per_s']), 'pinocchio verify', q['pinocchio']['verify_s'])"
