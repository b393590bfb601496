# where the small-batch pairing kernels stop paying after the 127-step loop: both kernel families at 8 k ... 48 k pairings (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for m in 8192 16384 24576 32768 49152; do
  ZKT_DTATE_MAX=1000000 timeout -k 10 300 python3 tools/bench_pairing.py $m >> gpurun_out/exp13_d.log 2>> gpurun_out/exp13.err || exit 1
  ZKT_DTATE_MAX=0 timeout -k 10 300 python3 tools/bench_pairing.py $m >> gpurun_out/exp13_k.log 2>> gpurun_out/exp13.err || exit 1
done
echo "lane-distributed:"; cat gpurun_out/exp13_d.log; echo "one pairing per lane:"; cat gpurun_out/exp13_k.log
