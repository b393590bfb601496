# 127-step loop in the fused verification kernels: smoke, parity of every pairing consumer, protocol throughput (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
ZKT_DTATE_MAX=0 ZKT_DPRODUCT_MAX=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "outside_the_subgroup" > gpurun_out/exp7_smoke.log 2>&1 || { tail -20 gpurun_out/exp7_smoke.log; echo "smoke FAILED"; exit 1; }
tail -2 gpurun_out/exp7_smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/exp7_tests.log 2>&1 || { tail -30 gpurun_out/exp7_tests.log; echo "tests FAILED"; exit 1; }
tail -3 gpurun_out/exp7_tests.log
timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp7_protocols.json 2> gpurun_out/exp7_protocols.err || exit 1
python3 -c "
import json; q=json.load(open('gpurun_out/exp7_protocols.json'))
print('bls verify', round(q['bls']['verify_per_s']), 'g16 verify batch', round(q['groth16_verify_batch']['verifications_per_s']), 'pinocchio verify', q['pinocchio']['verify_s'])"
