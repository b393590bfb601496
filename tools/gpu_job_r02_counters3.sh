# counters of the pairing kernel after the 127-step loop (round 2): SQ set, then memory-side bytes, one group per pass
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof_sq_tate3 --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/prof_sq_tate3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch_tate3 --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/prof_fetch_tate3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write_tate3 --output-format csv -- python3 tools/bench_pairing.py 65536 > gpurun_out/prof_write_tate3.log 2>&1 || exit 1
echo counters done
