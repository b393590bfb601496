# G2 bucket accumulation (two lanes per task): gather not held in registers, at one and at two waves per SIMD (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for v in base g2np g2w2np; do
  if [ $v = base ]; then unset ZKT_LIB_PATH; else export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_$v.so; fi
  timeout -k 10 300 python3 tools/bench_g2_msm.py 20 4 >> gpurun_out/exp5.log 2>> gpurun_out/exp5.err || exit 1
done
echo done
