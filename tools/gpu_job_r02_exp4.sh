# three-scan (Karatsuba) Fq2 product in the pairing / group objects: A/B against the four-scan default (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for v in base karaP; do
  if [ $v = base ]; then unset ZKT_LIB_PATH; else export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_$v.so; fi
  timeout -k 10 300 python3 tools/bench_pairing.py 65536 >> gpurun_out/exp4.log 2>> gpurun_out/exp4.err || exit 1
  timeout -k 10 300 python3 tools/bench_pairing.py 262144 >> gpurun_out/exp4.log 2>> gpurun_out/exp4.err || exit 1
  timeout -k 10 600 python3 tools/bench_protocols.py > gpurun_out/exp4_protocols_$v.json 2>> gpurun_out/exp4.err || exit 1
done
for v in base karaPG; do
  if [ $v = base ]; then unset ZKT_LIB_PATH; else export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_$v.so; fi
  timeout -k 10 300 python3 tools/bench_g2_mul.py 262144 >> gpurun_out/exp4.log 2>> gpurun_out/exp4.err || exit 1
done
echo done
