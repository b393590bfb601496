# what SQ_WAIT_ANY counts: the VALU micro-benchmark (no memory instructions in its loops) under the SQ counters (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/exp3_ubench_sq --output-format csv -- ./build/exp/valu_roof > gpurun_out/exp3_ubench.txt 2>&1
echo "rc=$?"
