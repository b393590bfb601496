set -x
cd /root/repo
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof_sq_tate --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/prof_sq_tate.log 2>&1
echo tate rc=$?
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof_sq_g2 --output-format csv -- python3 tools/bench_g2_msm.py 20 4 > gpurun_out/prof_sq_g2.log 2>&1
echo g2 rc=$?
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch --output-format csv -- python3 bench.py --no-cpu --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0 > gpurun_out/prof_fetch.log 2>&1
echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write --output-format csv -- python3 bench.py --no-cpu --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0 > gpurun_out/prof_write.log 2>&1
echo write rc=$?
