# small MSMs replayed as one graph launch per submit (round 2): parity on every path that submits them, then the latencies
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_protocols.py tests/test_gpu_fullsize.py tests/test_gpu_groth16_r1cs.py tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/exp27_tests.log 2>&1 || { tail -30 gpurun_out/exp27_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/exp27_tests.log
for g in 0 1; do
  echo "== ZKT_MSM_GRAPHS=$g"
  ZKT_MSM_GRAPHS=$g timeout -k 10 300 python3 tools/bench_bp_rp_only.py 0 2>&1 | grep "range proof" | tail -2
  ZKT_MSM_GRAPHS=$g timeout -k 10 300 python3 tools/bench_bp_rp_only.py 1 2>&1 | grep "range proof" | tail -2
  ZKT_MSM_GRAPHS=$g timeout -k 10 300 python3 tools/bench_msm_latency.py 2>&1 | grep "2^17"
done
ZKT_MSM_GRAPHS=1 timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp27_protocols.json 2> gpurun_out/exp27_protocols.err || { tail gpurun_out/exp27_protocols.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/exp27_protocols.json'))
print('ipa', d['bulletproofs_ipa']['seconds'], d['bulletproofs_ipa']['seconds_resident_generators'], 'rp', d['bulletproofs_range_proof']['seconds_without_ipa'], d['bulletproofs_range_proof']['seconds_with_ipa'])"
