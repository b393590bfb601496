# A/B of bucket-accumulation occupancy variants + memory-side counters of the pairing kernel (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for v in base acc3 acc3np; do
  if [ $v = base ]; then unset ZKT_LIB_PATH; else export ZKT_LIB_PATH=/root/repo/build/exp/libzkt_$v.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu --pairings 0 --groth16-log2n 0 --no-bulletproofs > gpurun_out/exp1_msm_$v.json 2> gpurun_out/exp1_msm_$v.err || exit 1
  echo "$v rc=$?"
done
unset ZKT_LIB_PATH
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/exp1_tate_fetch --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/exp1_tate_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/exp1_tate_write --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/exp1_tate_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d gpurun_out/exp1_tate_icache --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/exp1_tate_icache.log 2>&1
echo "icache rc=$?"
