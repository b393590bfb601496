#!/bin/bash
# round 4, job f: scheduler-strategy sweep, one object at a time (max-memory-clause won 2.6 % on k_tate in job e)
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O; Z=$PWD/zk-toolkit_amd
run() { ZKT_LIB_PATH=$Z/libzkt_hip$1.so timeout -k 10 300 python3 $2 2>&1 | grep -v "^[WEI]2026\|amdgpu.ids" | tail -${3:-1} | sed "s/^/[$1] /"; }
{
echo "--- k_tate"; run "" "tools/bench_pairing.py 65536"; run _tv_maxmem "tools/bench_pairing.py 65536"; run _tv2_tate_itminreg "tools/bench_pairing.py 65536"
echo "--- verification kernels (zkt_pairing.o)"; run "" "tools/bench_g16_batch_verify.py 65536" 2; run _tv2_pairing_maxmem "tools/bench_g16_batch_verify.py 65536" 2
echo "--- G2 accumulate (zkt_msm_g2pair.o)"; run "" "tools/bench_g2_msm.py 20 8"; run _tv2_g2pair_maxmem "tools/bench_g2_msm.py 20 8"
echo "--- G1 MSM (zkt_msm.o)"; for v in "" _tv2_msm_maxmem; do ZKT_LIB_PATH=$Z/libzkt_hip$v.so timeout -k 10 300 python3 bench.py --no-cpu --no-bulletproofs --pairings 0 --groth16-log2n 0 --g2-log2n 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['full_size_check'])"; done
echo "--- small-batch kernels (zkt_dpairing.o)"; run "" "tools/bench_pairing_small.py" 6; run _tv2_dpairing_maxmem "tools/bench_pairing_small.py" 6
} 2>&1 | tee $O/r04f_sched_sweep.txt
echo done
