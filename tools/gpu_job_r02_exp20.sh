# generator multiplications through a comb table (hash_to_g2, CRS::new of Groth16 and Pinocchio): parity tests, then the protocol figures (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_signature.py tests/test_gpu_pinocchio.py tests/test_gpu_groth16_r1cs.py tests/test_gpu_protocols.py -m gpu -x -q > gpurun_out/exp20_tests.log 2>&1 || { tail -30 gpurun_out/exp20_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp20_tests.log
timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp20_protocols.json 2> gpurun_out/exp20_protocols.err || { tail gpurun_out/exp20_protocols.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/exp20_protocols.json'))
print('bls', d['bls']['sign_per_s'], d['bls']['verify_per_s']); print('pinocchio', {k:v for k,v in d['pinocchio'].items() if k!='note'})"
timeout -k 10 600 python3 tools/bench_groth16.py --log-n 20 --proofs 4 2>&1 | grep "setup\|prove"
