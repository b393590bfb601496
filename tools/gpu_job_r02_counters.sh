set -x
cd /root/repo
export TMPDIR=/tmp
./build/exp/valu_roof > gpurun_out/r02_valu_ubench.txt 2>&1
rocprofv3 -L > gpurun_out/r02_counters_list.txt 2>&1 || true
python3 bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
echo bench rc=$?
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/prof_sq_g1 --output-format csv -- python3 bench.py --no-cpu --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0 > gpurun_out/prof_sq_g1.log 2>&1
echo sq rc=$?
