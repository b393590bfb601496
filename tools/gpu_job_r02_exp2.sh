# pairing kernel at one and two waves per SIMD over batch sizes; counters of the two-wave build (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
export ZKT_DTATE_MAX=0
for v in base tatew2; do
  if [ $v = base ]; then unset ZKT_LIB_PATH; else export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_$v.so; fi
  for m in 32768 65536 131072 262144; do
    timeout -k 10 300 python3 tools/bench_pairing.py $m >> gpurun_out/exp2_sizes.log 2>> gpurun_out/exp2_sizes.err || exit 1
  done
done
export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_tatew2.so
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/exp2_w2_sq --output-format csv -- python3 tools/bench_pairing.py 131072 > gpurun_out/exp2_w2_sq.log 2>&1
echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/exp2_w2_fetch --output-format csv -- python3 tools/bench_pairing.py 131072 > gpurun_out/exp2_w2_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/exp2_w2_write --output-format csv -- python3 tools/bench_pairing.py 131072 > gpurun_out/exp2_w2_write.log 2>&1
echo "done"
